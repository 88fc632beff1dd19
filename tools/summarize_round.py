"""gpurun_out/<tag> (tools/measure_round.sh <tag>) -> profiles/<tag>_bench_kernel_stats.csv, <tag>_kernel_table.md (per-kernel achieved
GB/s and TOP/s next to the gfx950 peaks), r02_hbm_traffic.md, r02_gemm_pmc.md, r02_gemm_traffic.json, r02_bench_line.json.
Pure CSV processing: runs anywhere."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
T = os.path.join(ROOT, "gpurun_out", TAG)
# written next to the raw data (gpurun_out/ is what travels back from the GPU box); copy into profiles/ afterwards:
#   cp gpurun_out/<tag>_summary/* profiles/
P = os.path.join(ROOT, "gpurun_out", TAG + "_summary")
RND = TAG[1:].lstrip("0") or "?"


def gemm_sources_sha256():
    """hash of every GEMM source the traffic figure was measured on: bench.py refuses the figure when it differs"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in sorted(glob.glob(os.path.join(ROOT, "vidit-q_amd", "csrc", "gemm_*"))):
        if fn.endswith((".h", ".hip")):
            with open(fn, "rb") as f:
                h.update(f.read())
    return h.hexdigest()
os.makedirs(P, exist_ok=True)
PEAK_I8, PEAK_F16, PEAK_HBM = 5.03e15, 2.5e15, 8.0e12
M = 16384


def counters(name):
    """{kernel: {counter: [values per dispatch]}}"""
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(T, name, "p_counter_collection.csv")
    if not os.path.exists(path):
        return d
    with open(path) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def mean(v):
    return sum(v) / len(v) if v else float("nan")


stats = list(csv.DictReader(open(os.path.join(T, "stats", "b_kernel_stats.csv"))))
shutil.copy(os.path.join(T, "stats", "b_kernel_stats.csv"), os.path.join(P, TAG + "_bench_kernel_stats.csv"))
if os.path.exists(os.path.join(T, "bench_line.json")):
    shutil.copy(os.path.join(T, "bench_line.json"), os.path.join(P, TAG + "_bench_line.json"))
F, W = counters("FETCH_SIZE"), counters("WRITE_SIZE")
CF, CW = counters("cal_FETCH_SIZE"), counters("cal_WRITE_SIZE")
cal_fill_w = [mean(v["WRITE_SIZE"]) for k, v in CW.items() if "FillFunc" in k]
cal_copy_f = [mean(v["FETCH_SIZE"]) for k, v in CF.items() if "copyBuffer" in k or "copy" in k.lower()]
cal_fill_w = cal_fill_w[0] if cal_fill_w else float("nan")
cal_copy_f = cal_copy_f[0] if cal_copy_f else float("nan")


def short(k):
    for a, b in (("gemm_i8_wide_kernel<256, 288, 4, 2, 0", "GEMM epi none (qkv x2, cross-q, fc1)"), ("gemm_i8_wide_kernel<256, 288, 4, 2, 1", "GEMM fc1 + GELU epilogue"),
                 ("gemm_i8_wide_kernel<256, 288, 4, 2, 2", "GEMM + gate*y + resid (proj x2, fc2)"), ("gemm_i8_wide_kernel<256, 288, 4, 2, 3", "GEMM + resid (cross proj)"),
                 ("gemm_i8_wide_kernel<128, 288, 4, 2, 0", "GEMM 128-row tiles (prompt K/V of all 28 blocks, one launch)"),
                 ("attn_fwd32d_kernel", "spatial attention (flash, 1024 keys)"), ("attn_fwd8_kernel", "spatial attention, previous generation"), ("attn_temporal_quant", "temporal attention + proj quantizer"),
                 ("attn_cross_reg", "cross attention (K/V^T in registers)"), ("attn_cross32", "cross attention (K/V resident in LDS)"),
                 ("fp_linear_kernel", "FP edge Linears (embedders, t_block, final layer, patch embedding)"), ("ln_modulate_rowquant_half", "LN + modulate + quantizer C=1152"),
                 ("rowquant_half", "per-token quantizer C=1152"), ("rowquant_split", "per-token quantizer C=4608"),
                 ("rowquant_fast_kernelILi9", "per-token quantizer C=4608")):
        if a in k:
            return b
    return None


# algorithmic work per launch at 16 x 512 x 512: bytes moved once, ops
ALG = {
    "GEMM fc1 + GELU epilogue": (M * 1152 + 4608 * 1152 + 2 * M * 4608, 2.0 * M * 4608 * 1152, "i8"),
    # launch-weighted means over the launches of one block-sample:
    # qkv x 2 (N 3456) + cross-q (N 1152) + fc1 (N 4608, GELU in the next quantizer since round 3), all K = 1152
    "GEMM epi none (qkv x2, cross-q, fc1)": ((2 * (M * 1152 + 3456 * 1152 + 2 * M * 3456) + (M * 1152 + 1152 * 1152 + 2 * M * 1152)
                                                   + (M * 1152 + 4608 * 1152 + 2 * M * 4608)) / 4,
                                                  2.0 * M * 1152 * (2 * 3456 + 1152 + 4608) / 4, "i8"),
    # proj x 2 (N = K = 1152) + fc2 (N 1152, K 4608), each reading the residual as well
    "GEMM + gate*y + resid (proj x2, fc2)": ((2 * (M * 1152 + 1152 * 1152 + 4 * M * 1152) + (M * 4608 + 1152 * 4608 + 4 * M * 1152)) / 3,
                                             2.0 * M * 1152 * (2 * 1152 + 4608) / 3, "i8"),
    "GEMM + resid (cross proj)": (M * 1152 + 1152 * 1152 + 4 * M * 1152, 2.0 * M * 1152 * 1152, "i8"),
    "spatial attention (flash, 1024 keys)": (4 * 2 * M * 1152, 4.0 * 16 * 16 * 1024 * 1024 * 72, "f16"),
    "temporal attention + proj quantizer": (3 * 2 * M * 1152 + M * 1152, 4.0 * 1024 * 16 * 16 * 16 * 72, "hbm"),
    "cross attention (K/V^T in registers)": (2 * 2 * M * 1152, None, "hbm"),
    "cross attention (K/V resident in LDS)": (2 * 2 * M * 1152, None, "hbm"),
    "LN + modulate + quantizer C=1152": (2 * M * 1152 + M * 1152, None, "hbm"),
    "per-token quantizer C=1152": (2 * M * 1152 + M * 1152, None, "hbm"),
    "per-token quantizer C=4608": (2 * M * 4608 + M * 4608, None, "hbm"),
}
rows = []
tot = sum(float(r["TotalDurationNs"]) for r in stats if "spin_kernel" not in r["Name"])
lines = ["# Round %s - " % RND + "per-kernel time and achieved rates inside the bench step (MI355X, gfx950)", "",
         "Source: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 2` (tools/measure_round.sh; raw:",
         "`profiles/%s_bench_kernel_stats.csv`).  Peaks (MI355X_MICROARCH.md): int8 MFMA 5.03 POP/s, fp16 MFMA 2.5 PFLOP/s, HBM3E 8 TB/s" % TAG,
         "(6.3 TB/s achievable).  Algorithmic bytes = operands read once + result written once.", "",
         "| kernel | launches | avg us | share of GPU time | achieved | of peak |", "|---|---|---|---|---|---|"]
for r in stats:
    name = short(r["Name"])
    if name is None:
        continue
    avg = float(r["AverageNs"]) / 1e9
    share = float(r["TotalDurationNs"]) / tot
    ach, frac = "", ""
    if name in ALG:
        b, ops, kind = ALG[name]
        if kind == "i8":
            ach, frac = "%.2f POP/s" % (ops / avg / 1e15), "%.1f %% of int8 MFMA" % (100 * ops / avg / PEAK_I8)
        elif kind == "f16":
            ach, frac = "%.0f TFLOP/s, %.2f TB/s" % (ops / avg / 1e12, b / avg / 1e12), "%.1f %% of fp16 MFMA" % (100 * ops / avg / PEAK_F16)
        else:
            ach, frac = "%.2f TB/s" % (b / avg / 1e12), "%.1f %% of HBM" % (100 * b / avg / PEAK_HBM)
    lines.append("| %s | %s | %.1f | %.1f %% | %s | %s |" % (name, r["Calls"], avg * 1e6, 100 * share, ach, frac))
open(os.path.join(P, TAG + "_kernel_table.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[6:]))

# ---- HBM / fabric traffic (raw counter tables present only on the GPU box: nothing is rewritten without them)
if not F or not W:
    sys.exit(0)
lines = ["# Round %s - " % RND + "HBM / fabric traffic per launch (rocprofv3 --pmc, MI355X gfx950)", "",
         "`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 1 --warmup 1 --no-graph",
         "--no-roofline-events --no-cpu-baseline` (depth 28, eager launches: every dispatch carries its counters; one counter per pass).",
         "Units / corrections as MI355X_MICROARCH.md prescribes: KB; FETCH_SIZE x 2 on gfx950 (calibrated here: `copy_` of 151.0 MB reads",
         "%.1f MB by the raw counter; `fill_` writes %.1f MB: exact)." % (cal_copy_f * 1024 / 1e6, cal_fill_w * 1024 / 1e6), "",
         "| kernel | launches | read MB | write MB |", "|---|---|---|---|"]
tot_c = tot_b = 0
for k in sorted(F, key=lambda k: -sum(F[k]["FETCH_SIZE"])):
    if k not in W:
        continue
    f, w = F[k]["FETCH_SIZE"], W[k]["WRITE_SIZE"]
    rb, wb = 2 * mean(f) * 1024 / 1e6, mean(w) * 1024 / 1e6
    if rb + wb < 5:
        continue
    lines.append("| `%s` | %d | %.1f | %.1f |" % ((short(k) or k[:70]).replace("|", "/"), len(f), rb, wb))
    if "gemm_i8" in k:
        tot_c += len(f)
        tot_b += len(f) * (rb + wb)
if tot_c:
    lines += ["", "GEMM launches: %d, launch-weighted mean traffic %.1f MB per launch (`roofline.traffic` of the bench line)." % (tot_c, tot_b / tot_c)]
    json.dump({"gemm_launches": tot_c, "hbm_bytes_per_launch": tot_b / tot_c * 1e6, "gemm_sources_sha256": gemm_sources_sha256(),
               "source": "profiles/%s_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py at depth 28 on the shipping kernels, 2 x FETCH_SIZE correction)" % TAG},
              open(os.path.join(P, TAG + "_gemm_traffic.json"), "w"))
open(os.path.join(P, TAG + "_hbm_traffic.md"), "w").write("\n".join(lines) + "\n")

# ---- SQ / GRBM counters of the GEMM kernels
S1, S2, G = counters("SQ1"), counters("SQ2"), counters("GRBM")
if not S1:
    sys.exit(0)
lines = ["# Round %s - " % RND + "PMC counters of the SHIPPING GEMM kernels inside the bench (gemm_i8_wide_kernel<256,288,4,2,EPI>)", "",
         "`rocprofv3 --pmc <one set per pass> --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-graph ...` (depth 28).  SQ counters",
         "are summed over all waves of a dispatch; SQ_*_CYCLES in quad-cycles except SQ_VALU_MFMA_BUSY_CYCLES (cycles; 16 per",
         "`mfma_i32_16x16x64_i8`, summed over SIMDs).  GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (8 x the kernel's cycles:",
         "it equals 8 x duration x clock), so MFMA utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); cross-check:",
         "SQ_INSTS_MFMA x 16 cycles / the same denominator.", "",
         "| kernel | n | GUI_ACTIVE cyc | MFMA busy / SIMD-cycle | WAIT_ANY / WAVE_CYCLES | WAIT_INST_ANY / WAVE | ACTIVE_INST_ANY / WAVE | LDS bank conflict / LDS active | INSTS VALU : MFMA : LDS |",
         "|---|---|---|---|---|---|---|---|---|"]
for k in sorted(S1):
    if "gemm_i8" not in k:
        continue
    a, b, g = S1[k], S2.get(k, {}), G.get(k, {})
    gui = mean(g.get("GRBM_GUI_ACTIVE", []))
    wc = mean(a["SQ_WAVE_CYCLES"])
    lines.append("| %s | %d | %.0f | %.3f | %.3f | %.3f | %.3f | %.4f | %.0f : %.0f : %.0f |" % (
        short(k) or k[:60], len(a["SQ_WAVE_CYCLES"]), gui, mean(a["SQ_VALU_MFMA_BUSY_CYCLES"]) / (gui / 8 * 1024) if gui == gui else float("nan"),
        mean(a["SQ_WAIT_ANY"]) / wc, mean(a["SQ_WAIT_INST_ANY"]) / wc, mean(a["SQ_ACTIVE_INST_ANY"]) / wc,
        mean(b.get("SQ_LDS_BANK_CONFLICT", [])) / max(mean(b.get("SQ_LDS_IDX_ACTIVE", [])), 1),
        mean(b.get("SQ_INSTS_VALU", [])), mean(b.get("SQ_INSTS_MFMA", [])), mean(a["SQ_INSTS_LDS"])))
open(os.path.join(P, TAG + "_gemm_pmc.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[6:]))
