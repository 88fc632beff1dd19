# Per-round evidence for profiles/ (usage: bash tools/measure_round.sh r03 [extra bench flags]): everything on the SHIPPING
# kernels, inside bench.py at depth 28 (headline W8A8 leg only: --no-extras).
#   1. rocprofv3 --kernel-trace --stats of the default bench command (HIP graph, two streams)
#   2. PMC passes of the same bench launched eagerly (every dispatch carries its counters), ONE counter set per pass,
#      --kernel-trace only (no other trace domains): FETCH_SIZE, WRITE_SIZE, two SQ sets, GRBM
#   3. FETCH/WRITE calibration on known byte counts
# then: python tools/summarize_round.py <tag>  ->  gpurun_out/<tag>_summary/ (copy into profiles/).  GPU box only.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
TAG=${1:-r04}
shift || true
O=$R/gpurun_out/$TAG
mkdir -p $O
EXTRA="$@"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras $EXTRA > $O/stats.log 2>&1)
pass() {  # name, counters...
  n=$1; shift
  (cd /tmp && timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-graph --no-roofline-events --no-telemetry $EXTRA > $O/$n.log 2>&1)
}
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
if [ -z "$VQ_MEASURE_SKIP_SQ" ]; then   # the SQ / GRBM sets only change when the GEMM kernel does (summary keeps the previous file otherwise)
pass SQ1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS
pass SQ2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
pass GRBM GRBM_GUI_ACTIVE GRBM_COUNT
fi
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/cal_$c -o p -- python $R/tools/traffic_cal.py > $O/cal_$c.log 2>&1)
done
python tools/summarize_round.py $TAG > $O/summary.log 2>&1
# the traffic figure just measured becomes the one bench.py reports (same sources: the hash matches)
cp $R/gpurun_out/${TAG}_summary/${TAG}_gemm_traffic.json $R/profiles/ 2>/dev/null
python bench.py $EXTRA > $O/bench_line.json 2> $O/bench.err
cp $O/bench_line.json $R/gpurun_out/${TAG}_summary/${TAG}_bench_line.json
# the raw per-dispatch counter tables are tens of MB: only the summaries (and the kernel stats) travel back
for d in FETCH_SIZE WRITE_SIZE SQ1 SQ2 GRBM cal_FETCH_SIZE cal_WRITE_SIZE; do rm -rf $O/$d; done
find $O/stats -type f ! -name "*kernel_stats.csv" -delete
tail -c 300 $O/bench_line.json; cat $O/summary.log | tail -30
