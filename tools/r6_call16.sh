# round 6, GPU call 16: the full bench line (all legs, 20 timed headline steps) on the end-of-round library + smoke()
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6o; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_line_steps20.json 2> $O/bench.err
tail -c 1500 $O/bench_line_steps20.json
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -8 $O/smoke.txt
