# round 6, GPU call 27: one stream vs two streams today (round 1: +13 % for two)
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; O=$R/gpurun_out/r6x; mkdir -p $O
for r in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/two_$r.json 2> $O/two_$r.err
  python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --one-stream > $O/one_$r.json 2> $O/one_$r.err
done
python - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    t = d.get("telemetry") or {}
    print(os.path.basename(f), "%.2f steps/s" % d["value"], "gemm avg %.1f us" % d["roofline"]["avg_launch_us"], "gemm clock", (t.get("gemm_shader_clock") or {}).get("ghz"),
          "power", round((t.get("during_timed_region") or {}).get("power_w", {}).get("mean", 0)))
PY
