# round 6, GPU call 21: cross attention (the LDS-resident kernel this time) - scalar v_fma vs v_pk_fma in the softmax
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6s; mkdir -p $O
for r in 1 2 3 4; do
  for d in vidit-q_amd/csrc _ab_cross_pk; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 120 python tools/attn_ab.py cross 2>/dev/null | grep -v amdgpu >> $O/cross_fma.txt
  done
done
cat $O/cross_fma.txt
