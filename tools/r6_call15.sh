# round 6, GPU call 15: PixArt kv compression + qk norm tests; PixArt / model tests unchanged
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6n; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -k "pixart" > $O/pixart_tests.txt 2>&1
tail -40 $O/pixart_tests.txt
