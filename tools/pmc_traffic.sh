# HBM traffic of the bench kernels from the TCC fabric counters (rocprofv3 --pmc, one counter set per pass,
# --kernel-trace only), plus the kernel-trace stats of the default bench command.  GPU box only.
mkdir -p gpurun_out/traffic; export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
# 1. kernel trace + stats of the default (graph, two-stream) bench command
(cd /tmp && timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/traffic/stats -o b -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/traffic/stats.log 2>&1)
# 2. PMC passes (eager launches so every dispatch carries its own counters; depth 4 keeps it short: the
#    per-launch traffic of a kernel does not depend on the depth)
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic/$c -o p -- python $R/bench.py --steps 2 --warmup 1 --depth 4 --no-cpu-baseline --no-graph --no-roofline-events > $R/gpurun_out/traffic/$c.log 2>&1)
done
# calibration: known-size fill (write) and copy (read + write) at the fc1 output size
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic/cal_$c -o p -- python $R/tools/traffic_cal.py > $R/gpurun_out/traffic/cal_$c.log 2>&1)
done
find gpurun_out/traffic -name "*.csv" | head -30
