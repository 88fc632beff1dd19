# round 6, GPU call 17: full GPU suite on the end-of-round tree
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6p; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1
tail -5 $O/gpu_suite.txt
