# Counter evidence for the HBM-bound kernels of a step (usage: bash tools/pmc_hbm.sh r04 [extra bench flags]): the bench
# launched eagerly at depth 4 (per-kernel counters do not depend on depth; every dispatch carries its counters), ONE
# counter set per pass, --kernel-trace only (no other trace domains).  GPU box only.
#   python tools/pmc_hbm_summary.py <tag>  ->  gpurun_out/<tag>_hbm/<tag>_hbm_kernels_pmc.md
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
TAG=${1:-r04}
shift || true
O=$R/gpurun_out/${TAG}_hbm
mkdir -p $O
EXTRA="$@"
pass() {  # name, counters...
  n=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -o p -- python $R/bench.py --depth 4 --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-graph --no-roofline-events $EXTRA > $O/$n.log 2>&1)
}
pass SQ1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass SQ2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass GRBM GRBM_GUI_ACTIVE GRBM_COUNT
python tools/pmc_hbm_summary.py $TAG > $O/summary.log 2>&1
for d in SQ1 SQ2 FETCH_SIZE WRITE_SIZE GRBM; do rm -rf $O/$d; done
cat $O/summary.log | tail -60
