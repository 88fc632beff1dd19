# Per-kernel time of the sampling steps of the PixArt-Sigma 1024^2 W4A8 leg (usage: bash tools/pixart_step_kernels.sh <tag>): two
# rocprofv3 --kernel-trace --stats runs of tools/bench_pixart.py with 4 and 12 steps, differenced by tools/stats_diff.py.  GPU box only.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
TAG=${1:-r04}
O=$R/gpurun_out/skp_$TAG
mkdir -p $O
for n in 4 12; do
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$n -o b -- python $R/tools/bench_pixart.py --steps $n > $O/s$n.log 2>&1)
done
python tools/stats_diff.py $(find $O/s4 -name "b_kernel_stats.csv") $(find $O/s12 -name "b_kernel_stats.csv") 8 > $O/${TAG}_pixart_step_kernels.txt
find $O -type f ! -name "*step_kernels.txt" ! -name "*.log" -delete
head -30 $O/${TAG}_pixart_step_kernels.txt; tail -2 $O/s12.log
