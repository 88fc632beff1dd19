# round 6, GPU call 4: trimmed temporal attention + quantizer (v2) vs v1; 64-queries-per-wave spatial attention arms
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "attn or temporal" -x -q > $O/attn_tests.txt 2>&1
tail -4 $O/attn_tests.txt
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_tq_v1; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/attn_ab.py temporal >> $O/temporal_ab.txt 2>> $O/temporal_ab.err
  done
done
cat $O/temporal_ab.txt
VIDITQ_LIB=$R/vidit-q_amd/csrc/libviditq_hip.so timeout 300 python tools/attn_ab.py --dump=/tmp/attn_ref.pt > $O/attn64_cmp.txt 2>&1
for d in _ab_attn64 _ab_attn64k128; do
  VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/attn_ab.py --cmp=/tmp/attn_ref.pt >> $O/attn64_cmp.txt 2>&1
done
cat $O/attn64_cmp.txt
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_attn64 _ab_attn64k128; do
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/attn_ab.py spatial image >> $O/attn64_ab.txt 2>> $O/attn64_ab.err
  done
done
cat $O/attn64_ab.txt
