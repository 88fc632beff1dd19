# round 6, GPU call 20: cross attention - workgroups per CU (query tiles per wave): 1 (4 tiles per wave), 2 (product), 3, 4
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6r; mkdir -p $O
for r in 1 2 3; do
  for w in 2 1 3 4; do
    echo -n "VQ_CROSS_WGS=$w  " >> $O/cross_wgs.txt
    VQ_CROSS_WGS=$w timeout 120 python tools/attn_ab.py cross 2>/dev/null | grep -v amdgpu >> $O/cross_wgs.txt
  done
done
cat $O/cross_wgs.txt
