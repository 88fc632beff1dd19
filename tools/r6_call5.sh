# round 6, GPU call 5: attention tests on the fixed kernels, temporal v2 timing, 4-wave 64-row arm, cross timing after the row-max fix, bench A/B
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "attn or temporal" -q > $O/attn_tests.txt 2>&1
tail -6 $O/attn_tests.txt
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_tq_v1; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/attn_ab.py temporal cross >> $O/temporal_ab.txt 2>> $O/temporal_ab.err
  done
done
cat $O/temporal_ab.txt
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_attn64 _ab_attn64w4; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/attn_ab.py spatial image >> $O/attn64_ab.txt 2>> $O/attn64_ab.err
  done
done
cat $O/attn64_ab.txt
bash tools/ab_env.sh $O/ab 2 "new:" "tqv1:VIDITQ_LIB=$R/_ab_tq_v1/libviditq_hip.so" > $O/ab.txt 2>&1
cat $O/ab.txt
