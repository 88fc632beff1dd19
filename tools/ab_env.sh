#!/bin/bash
# One box, alternating: the bench's headline leg under different library switches (environment variables read once per process).
# usage: bash tools/ab_env.sh <outdir> <rounds> "NAME1:ENV1=V ENV2=V" "NAME2:" ...
set -u
OUT=$1; R=$2; shift 2
mkdir -p $OUT
FLAGS="--steps 20 --warmup 3 --no-extras --no-cpu-baseline ${AB_FLAGS:-}"
for r in $(seq 1 $R); do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py $FLAGS > $OUT/${name}_$r.json 2> $OUT/${name}_$r.err
  done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    t = d.get("telemetry") or {}
    du = t.get("during_timed_region") or {}
    clk = (t.get("gemm_shader_clock") or {})
    mem = t.get("memory") or {}
    print("%-14s %.2f steps/s  %.2f ms/step  gemm avg %.1f us frac %.3f  gemm clock %s (%s..%s) GHz  power %s W  sclk %s MHz  temp %s C  fclk %s  copy %s / %s TB/s" % (
        os.path.basename(f)[:-5], d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"],
        clk.get("ghz"), clk.get("ghz_min"), clk.get("ghz_max"),
        ("%.0f" % du["power_w"]["mean"]) if isinstance(du.get("power_w"), dict) else du.get("power_w", "-"),
        ("%.0f" % du["sclk_mhz"]["mean"]) if isinstance(du.get("sclk_mhz"), dict) else du.get("sclk_mhz", "-"),
        ("%.0f" % du["temp_c"]["max"]) if isinstance(du.get("temp_c"), dict) else du.get("temp_c", "-"),
        ("%.0f" % du["fclk_mhz"]["mean"]) if isinstance(du.get("fclk_mhz"), dict) else "-",
        mem.get("hbm_copy_1GiB_TBps", "-"), mem.get("mall_copy_48MiB_TBps", "-")))
PY
