# round 6, GPU call 26: tie-test branch of the quantizers marked unlikely (fast path falls through) vs product
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6w; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "rowquant or ln_modulate or quant or gelu" -x -q > $O/rq_tests.txt 2>&1
tail -3 $O/rq_tests.txt
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_expect0; do
    echo "== $d" >> $O/rq_time.txt
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/rq_time.py 2>> $O/rq_time.err | grep -i "LN+mod+quant\|rowquant C=1152  \|GELU + rowquant C=4608\|rowquant C=4608" >> $O/rq_time.txt
  done
done
cat $O/rq_time.txt
bash tools/ab_env.sh $O/ab 3 "expect:" "base:VIDITQ_LIB=$R/_ab_expect0/libviditq_hip.so" > $O/ab.txt 2>&1
cat $O/ab.txt
