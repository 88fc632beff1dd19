# round 6, GPU call 11: the stream form of spatial attention (two query tiles of a pair as one tile stream): tests, timing A/B, bench A/B
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6k; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "attn_fwd" -x -q > $O/attn_tests.txt 2>&1
tail -15 $O/attn_tests.txt
for r in 1 2 3; do
  VQ_ATTN_STREAM=0 timeout 120 python tools/attn_ab.py spatial >> $O/stream_ab.txt 2>> $O/stream_ab.err
  timeout 120 python tools/attn_ab.py spatial >> $O/stream_ab.txt 2>> $O/stream_ab.err
done
cat $O/stream_ab.txt
bash tools/ab_env.sh $O/ab 2 "stream:" "plain:VQ_ATTN_STREAM=0" > $O/ab.txt 2>&1
cat $O/ab.txt
