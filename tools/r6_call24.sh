# round 6, GPU call 24: LDS-resident cross attention for prompts of up to 320 tokens (PixArt-Sigma): tests, PixArt leg A/B
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6v; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "attn_cross" -x -q > $O/cross_tests.txt 2>&1
tail -4 $O/cross_tests.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_config_gpu.py -k "pixart or sigma" -x -q > $O/pixart_tests.txt 2>&1
tail -4 $O/pixart_tests.txt
for r in 1 2 3; do
  for v in 1 0; do
    echo -n "VQ_ATTN_CROSS_LONG=$v  " >> $O/pixart_ab.txt
    VQ_ATTN_CROSS_LONG=$v timeout 300 python tools/bench_pixart.py --steps 12 2>/dev/null | tail -1 >> $O/pixart_ab.txt
  done
done
cat $O/pixart_ab.txt
