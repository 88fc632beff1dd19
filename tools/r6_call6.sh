# round 6, GPU call 6: full GPU suite on the tree with 16-byte attention stores + LDS-staged multi-output codes; W4A8 store-form A/B
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6f; mkdir -p $O
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_smm_narrow _ab_smm_wide_all; do
    echo "== $d" >> $O/rq_time.txt
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/rq_time.py 2>> $O/rq_time.err | grep -i "smooth\|multi" >> $O/rq_time.txt
  done
done
cat $O/rq_time.txt
AB_FLAGS="--plan w4a8" bash tools/ab_env.sh $O/ab_w4 2 "wide:" "narrow:VIDITQ_LIB=$R/_ab_smm_narrow/libviditq_hip.so" "wideall:VIDITQ_LIB=$R/_ab_smm_wide_all/libviditq_hip.so" > $O/ab_w4.txt 2>&1
bash tools/ab_env.sh $O/ab 1 "w8:" >> $O/ab_w4.txt 2>&1
cat $O/ab_w4.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1
tail -5 $O/gpu_suite.txt
