# Per-kernel time of the TIMED steps of one bench plan (usage: bash tools/step_kernels.sh <tag> <bench flags...>): two
# rocprofv3 --kernel-trace --stats runs with 2 and 6 steps, differenced by tools/stats_diff.py (set-up, calibration,
# warm-up and capture cancel).  --no-roofline-events keeps the eager event pass out.  GPU box only.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
TAG=$1; shift
O=$R/gpurun_out/sk_$TAG
mkdir -p $O
for n in 2 6; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$n -o b -- python $R/bench.py --steps $n --warmup 2 --no-cpu-baseline --no-extras --no-roofline-events "$@" > $O/s$n.log 2>&1)
done
python tools/stats_diff.py $(find $O/s2 -name "b_kernel_stats.csv") $(find $O/s6 -name "b_kernel_stats.csv") 4 > $O/${TAG}_step_kernels.txt
find $O -type f ! -name "*step_kernels.txt" -delete
cat $O/${TAG}_step_kernels.txt | head -34
