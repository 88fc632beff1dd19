"""gpurun_out/traffic (tools/pmc_traffic.sh) -> profiles/r01_hbm_traffic.md, profiles/r01_gemm_traffic.json,
profiles/r01_bench_kernel_stats.csv.  Runs anywhere (pure CSV processing)."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = os.path.join(ROOT, "gpurun_out", "traffic")


def load(path):
    d = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


F = load(os.path.join(T, "FETCH_SIZE", "p_counter_collection.csv"))
W = load(os.path.join(T, "WRITE_SIZE", "p_counter_collection.csv"))
CF = load(os.path.join(T, "cal_FETCH_SIZE", "p_counter_collection.csv"))
CW = load(os.path.join(T, "cal_WRITE_SIZE", "p_counter_collection.csv"))
cal_fill_w = [sum(v) / len(v) for k, v in CW.items() if "FillFunc" in k][0]
cal_copy_f = [sum(v) / len(v) for k, v in CF.items() if "copyBuffer" in k][0]
alg = {
    "gemm_i8_wide_kernel<256, 288, 4, 2, 2,": "proj / fc2 (+gate*resid): N=K=1152: 18.9 X + 1.3 W + 37.7 resid = 57.9 MB read, 37.7 MB write; fc2 (K=4608): 118.5 MB read",
    "gemm_i8_wide_kernel<256, 288, 4, 2, 1,": "fc1 (N=4608, K=1152, GELU): 18.9 X + 5.3 W = 24.2 MB read, 151.0 MB write",
    "gemm_i8_wide_kernel<256, 288, 4, 2, 3,": "cross-attn proj (N=K=1152, +resid): 57.9 MB read, 37.7 MB write",
    "gemm_i8_wide_kernel<256, 288, 4, 2, 0,": "qkv x2 (N=3456), cross-q (N=1152), kv (M<=120): launch-weighted mean 23.2 MB read, 66.1 MB write",
    "attn_fwd8_kernel<72": "spatial attention: 113 MB of q/k/v read once, 37.7 MB written",
    "attn_temporal_kernel<72": "temporal attention: 113 MB read, 37.7 MB written",
    "rowquant_half_kernel": "per-token quantizer C=1152: 37.7 MB read, 18.9 MB written",
    "ln_modulate_rowquant_half_kernel": "LN + modulate + quantizer C=1152: 37.7 MB read, 18.9 MB written",
    "rowquant_fast_kernelILi9": "per-token quantizer C=4608: 151.0 MB read, 75.5 MB written",
}
lines = ["# Round 1 - HBM / fabric traffic per launch (rocprofv3 --pmc, MI355X gfx950)", "",
         "Command (tools/pmc_traffic.sh; one counter per pass, `--kernel-trace` only):",
         "`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --depth 4 --no-cpu-baseline --no-graph --no-roofline-events`",
         "(eager launches so that every dispatch carries its own counters; per-launch traffic does not depend on depth).", "",
         "Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KB; on gfx950 FETCH_SIZE reports half",
         "of the bytes of a wide coalesced read.  Calibrated here on known byte counts (tools/traffic_cal.py, 151.0 MB tensors):",
         "`fill_` WRITE_SIZE = %.1f KB = %.1f MB (exact); `copy_` FETCH_SIZE = %.0f KB = %.1f MB for 151.0 MB read -> x2." % (
             cal_fill_w, cal_fill_w * 1024 / 1e6, cal_copy_f, cal_copy_f * 1024 / 1e6),
         "So read bytes = 2 x FETCH_SIZE x 1024, write bytes = WRITE_SIZE x 1024.  Infinity-Cache hits are counted (fabric-side counters).", "",
         "| kernel | launches | read MB | write MB | algorithmic |", "|---|---|---|---|---|"]
tot_c = tot_b = 0
for k in sorted(F, key=lambda k: -sum(F[k])):
    if k not in W:
        continue
    f, w = F[k], W[k]
    rb = 2 * sum(f) / len(f) * 1024 / 1e6
    wb = sum(w) / len(w) * 1024 / 1e6
    if rb + wb < 5:
        continue
    note = ""
    for kk, v in alg.items():
        if kk in k:
            note = v
    lines.append("| `%s` | %d | %.1f | %.1f | %s |" % (k[:78].replace("|", "/"), len(f), rb, wb, note))
    if "gemm_i8" in k:
        tot_c += len(f)
        tot_b += len(f) * (rb + wb)
lines += ["", "GEMM launches: %d, launch-weighted mean traffic %.1f MB per launch (bench.py reports this as `roofline.traffic`)." % (
    tot_c, tot_b / tot_c), "",
    "Reading: writes are exactly algorithmic everywhere.  History of the read side: with the first GEMM tile order an XCD ran",
    "2 token panels x ALL weight panels at a time, so the 5.3 MB fc1 weight matrix streamed through each 4 MB L2 once per 2",
    "token panels (fc1 read 192 MB for 24 MB of operands); the 2-D order (`xcd_tile`: 8 token x 4 channel panels per round)",
    "and the XCD-aware (sequence, head) placement of the attention workgroups (first kernel: 341 MB read for 113 MB of q/k/v,",
    "the 4 query tiles of a head ran on 4 different XCDs) are what the table above shows.  Kernel time of the GEMM did not",
    "move with its traffic (+-3 %): its main loop is not fabric-bound; the attention kernel gained 10 % from the placement."]
open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.md"), "w").write("\n".join(lines) + "\n")
json.dump({"gemm_launches": tot_c, "hbm_bytes_per_launch": tot_b / tot_c * 1e6,
           "source": "profiles/r01_hbm_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, 2 x FETCH_SIZE correction)"},
          open(os.path.join(ROOT, "profiles", "r01_gemm_traffic.json"), "w"))
shutil.copy(os.path.join(T, "stats", "b_kernel_stats.csv"), os.path.join(ROOT, "profiles", "r01_bench_kernel_stats.csv"))
print("\n".join(lines[12:30]))
