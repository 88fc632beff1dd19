# round 6, GPU call 14: register-lean LayerNorm + modulate + quantizer (8 / 7 waves per SIMD) vs the round-5 form (5)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6m; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -k "ln_modulate or rowquant" -x -q > $O/ln_tests.txt 2>&1
tail -3 $O/ln_tests.txt
for r in 1 2 3; do
  for d in vidit-q_amd/csrc _ab_ln0 _ab_ln2; do
    echo "== $d" >> $O/rq_time.txt
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/rq_time.py 2>> $O/rq_time.err | grep -i "LN+mod+quant\|rowquant C=1152" >> $O/rq_time.txt
  done
done
cat $O/rq_time.txt
bash tools/ab_env.sh $O/ab 2 "lean8:" "ln0:VIDITQ_LIB=$R/_ab_ln0/libviditq_hip.so" "lean7:VIDITQ_LIB=$R/_ab_ln2/libviditq_hip.so" > $O/ab.txt 2>&1
cat $O/ab.txt
