# round 6, GPU call 2: lean GEMM header + persistent form - tests, back-to-back, in-step A/B; cross attention fma arm; telemetry A/B
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -k "gemm or fp_edge" -x -q > $O/gemm_tests.txt 2>&1
tail -3 $O/gemm_tests.txt
timeout 600 python tools/gemm_persist6.py > $O/gemm_persist6.txt 2>&1
cat $O/gemm_persist6.txt
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_cross_pk; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/attn_ab.py cross >> $O/cross_ab.txt 2>> $O/cross_ab.err
  done
done
cat $O/cross_ab.txt
bash tools/ab_env.sh $O/ab 2 "base:" "persist:VIDITQ_LIB=$R/_ab_persist/libviditq_hip.so" > $O/ab.txt 2>&1
AB_FLAGS="--no-telemetry" bash tools/ab_env.sh $O/ab_notel 2 "notel:" > $O/ab_notel.txt 2>&1
AB_FLAGS="--plan w4a8" bash tools/ab_env.sh $O/ab_w4 1 "base:" "persist:VIDITQ_LIB=$R/_ab_persist/libviditq_hip.so" > $O/ab_w4.txt 2>&1
cat $O/ab.txt $O/ab_notel.txt $O/ab_w4.txt
