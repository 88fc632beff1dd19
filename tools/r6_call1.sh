# round 6, GPU call 1: attention instruction-mix arms, GEMM LDS-conflict attribution, host launch cost, traffic INT 0 vs 1
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6a; mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/dev.txt 2>&1
# 1. attention arms, alternating
for r in 1 2 3; do
  for arm in base:vidit-q_amd/csrc _ab_attn_max1:_ab_attn_max1 _ab_attn_sfma:_ab_attn_sfma _ab_attn_both:_ab_attn_both; do
    d=${arm#*:}
    VIDITQ_LIB=$R/$d/libviditq_hip.so timeout 300 python tools/attn_ab.py spatial image >> $O/attn_ab.txt 2>> $O/attn_ab.err
  done
done
VIDITQ_LIB=$R/_ab_attn_both/libviditq_hip.so timeout 600 python -m pytest tests/test_kernels_gpu.py -k "attn" -x -q > $O/attn_tests_both.txt 2>&1
# 2. GEMM LDS conflicts by K
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/conf -o p -- python $R/tools/gemm_conflicts.py > $O/conf.log 2>&1)
python tools/gemm_conflicts.py summarize $O/conf > $O/gemm_conflicts.md 2>&1
rm -rf $O/conf
# 3. bench headline with the graph census + idle-queue launch cost
VQ_GRAPH_DUMP=/tmp/vqdot timeout 900 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err
# 4. GEMM traffic: general form (INT 0) vs interior form (INT 1), depth 8
for m in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && VQ_GEMM_INT=$m timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/int$m/$c -o p -- python $R/bench.py --steps 1 --warmup 1 --depth 8 --no-cpu-baseline --no-extras --no-graph --no-roofline-events --no-telemetry > $O/int${m}_$c.log 2>&1)
  done
  python tools/traffic_ab.py $O/int$m > $O/traffic_int$m.txt 2>&1
  rm -rf $O/int$m
done
tail -n 20 $O/attn_ab.txt; cat $O/gemm_conflicts.md; cat $O/traffic_int*.txt; tail -c 600 $O/bench_line.json; tail -3 $O/attn_tests_both.txt
