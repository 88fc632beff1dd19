export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/final
timeout 500 python -m pytest tests -m gpu -q --no-header --timeout 300 -p no:cacheprovider 2>&1 | grep -vE "Extension modules" | tail -3 > gpurun_out/final/pytest.log
timeout 400 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
(cd /tmp && timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/stats -o b -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/final/stats.log 2>&1)
cat gpurun_out/final/pytest.log; tail -c 600 gpurun_out/final/bench.json; find gpurun_out/final/stats -name "*kernel_stats.csv"
