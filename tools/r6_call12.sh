# round 6, GPU call 12: graph launch cost per node count; full GPU suite (floor legs at every full-size checkpoint)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6l; mkdir -p $O
timeout 300 python tools/graph_launch_cost.py > $O/graph_launch_cost.txt 2>&1
cat $O/graph_launch_cost.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1
tail -5 $O/gpu_suite.txt
