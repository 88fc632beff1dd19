#!/bin/bash
# Round 5: one box, back to back, alternating - is a cross-box spread the box or the code?
#   A = HEAD                       (python bench.py)
#   B = round-3 tree               (_ab_r03/: `git archive c728cb4`, its own package + library built from its sources)
#   C = HEAD, integer dequant      (_ab_int/libviditq_hip.so: HEAD built with -DVQ_GEMM_FP_DEQUANT=0, bound via VIDITQ_LIB)
# each: --steps 20 --warmup 3, headline leg only; power / clock / temperature sampled around every run.
# usage: bash tools/ab_box_variance.sh <rounds> > gpurun_out/ab_box/...   (writes gpurun_out/ab_box/*.json + smi.txt)
set -u
R=${1:-2}
OUT=gpurun_out/ab_box
mkdir -p $OUT
smi() { echo "== $1 $(date +%s.%N)" >> $OUT/smi.txt; rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" >> $OUT/smi.txt; }
FLAGS="--steps 20 --warmup 3 --no-extras --no-cpu-baseline"
for r in $(seq 1 $R); do
  smi "before A$r"; python bench.py $FLAGS > $OUT/A_$r.json 2> $OUT/A_$r.err; smi "after A$r"
  if [ -d _ab_r03 ]; then (cd _ab_r03 && python bench.py $FLAGS > ../$OUT/B_$r.json 2> ../$OUT/B_$r.err); smi "after B$r"; fi
  if [ -f _ab_int/libviditq_hip.so ]; then VIDITQ_LIB=$PWD/_ab_int/libviditq_hip.so python bench.py $FLAGS > $OUT/C_$r.json 2> $OUT/C_$r.err; smi "after C$r"; fi
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/ab_box/[ABC]_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    t = d.get("telemetry") or {}
    clk = (t.get("gemm_shader_clock") or {}).get("ghz")
    du = t.get("during_timed_region") or {}
    print("%-10s %.2f steps/s  %.2f ms/step  gemm avg %.1f us frac %.3f  gemm clock %s GHz  power %s W  sclk %s MHz  temp %s C" % (
        os.path.basename(f), d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], clk,
        ("%.0f" % du["power_w"]["mean"]) if "power_w" in du else "-", ("%.0f" % du["sclk_mhz"]["mean"]) if "sclk_mhz" in du else "-",
        ("%.0f" % du["temp_c"]["max"]) if "temp_c" in du else "-"))
PY
