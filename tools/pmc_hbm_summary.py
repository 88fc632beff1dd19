"""gpurun_out/<tag>_hbm (tools/pmc_hbm.sh) -> <tag>_hbm_kernels_pmc.md: per-kernel means of the SQ / TCC / GRBM counters for
the HBM-bound kernels of a step, with the derived figures the guide names (MI355X_MICROARCH.md, rocprofv3 PMC slots):
WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY shares of wave cycles, mean resident waves per SIMD, VALU and VMEM issue
shares, instructions per wave, fabric bytes (FETCH_SIZE x 2 for 16-byte streaming reads, WRITE_SIZE as counted).
Pure CSV processing: runs anywhere."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
T = os.path.join(ROOT, "gpurun_out", TAG + "_hbm")
KERNELS = (("attn_cross", "cross attention"), ("attn_temporal_quant", "temporal attention + proj quantizer"),
           ("rowquant_split_kernel", "per-token quantizer C=4608 + GELU, row split over two partner waves (round 5)"),
           ("rowquant_fast_kernel", "per-token quantizer, one row per wave (C=4608 +GELU; C<=1536 generic)"),
           ("ln_modulate_rowquant_half", "LN + modulate + quantizer C=1152"), ("ln_modulate_rowquant_fast", "LN + modulate + quantizer (final layer)"),
           ("Z20rowquant_half", "per-token quantizer C=1152"), ("rowquant_smooth", "smoothed quantizer"),
           ("smooth_rowquant", "smoothed quantizer (half-wave)"), ("attn_fwd32d", "spatial attention (reference point: MFMA-bound)"))


def load(name):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(T, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(T, name, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
    return d, dur


def mean(v):
    return sum(v) / len(v) if v else float("nan")


sets = {n: load(n) for n in ("SQ1", "SQ2", "FETCH_SIZE", "WRITE_SIZE", "GRBM")}
names = set()
for d, _ in sets.values():
    names |= set(d)
out = ["# %s - counters of the HBM-bound kernels inside the step (MI355X, gfx950)" % TAG, "",
       "Source: `tools/pmc_hbm.sh %s` = `rocprofv3 --pmc <one set per pass> --kernel-trace -- python bench.py --depth 4 --steps 1 "
       "--warmup 1 --no-graph` (eager launches: kernels run ALONE, serialised by the profiler; W8A8, 16384 tokens).  SQ_* cycle "
       "counters are in quad-cycles summed over waves; shares are of SQ_WAVE_CYCLES.  `occ` = SQ_WAVE_CYCLES x 4 / (us x 2100 x "
       "1024 SIMDs) = mean resident waves per SIMD over the kernel's traced duration at an ASSUMED 2.1 GHz (`GUI/8/us` = "
       "GRBM_GUI_ACTIVE / 8 XCDs / us is shown beside it: 2.1 for the 105 us attention kernel, above the 2.4 GHz maximum for the "
       "short kernels - the counter keeps running around a dispatch, so it is not used as the clock).  Fabric bytes: FETCH_SIZE x 2 (gfx950 streaming-read calibration, "
       "MI355X_MICROARCH.md HBM section; KiB units -> bytes), WRITE_SIZE as counted." % TAG, ""]
hdr = ("| kernel | launches | us (profiled) | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | ACTIVE VALU | GUI/8/us | ACTIVE LDS | occ (waves/SIMD) | "
       "VALU / VMEM_RD / VMEM_WR / LDS / SALU insts per wave | fabric read MB | write MB | TB/s (fabric bytes / us) |")
out += [hdr, "|" + "---|" * (hdr.count("|") - 1)]
for pat, label in KERNELS:
    for k in sorted(n for n in names if pat in n):
        s1, d1 = sets["SQ1"][0].get(k, {}), sets["SQ1"][1].get(k, [])
        s2 = sets["SQ2"][0].get(k, {})
        fs, ws = sets["FETCH_SIZE"][0].get(k, {}), sets["WRITE_SIZE"][0].get(k, {})
        gr = sets["GRBM"][0].get(k, {})
        wc = mean(s1.get("SQ_WAVE_CYCLES", []))
        waves = mean(s1.get("SQ_WAVES", []))
        gui = mean(gr.get("GRBM_GUI_ACTIVE", []))
        us = mean(sets["GRBM"][1].get(k, []) or d1)
        rd = mean(fs.get("FETCH_SIZE", [])) * 1024 * 2 / 1e6
        wr = mean(ws.get("WRITE_SIZE", [])) * 1024 / 1e6
        sh = lambda c, s=s1: ("%.2f" % (mean(s.get(c, [])) / wc)) if wc == wc and wc else "-"
        per = lambda c: ("%.0f" % (mean(s2.get(c, [])) / waves)) if waves == waves and waves else "-"
        tmpl = k[k.find("<"):k.find(">") + 1] if "<" in k else ""
        out.append("| %s `%s%s` | %d | %.1f | %s | %s | %s | %s | %s | %s | %.2f | %s / %s / %s / %s / %s | %.1f | %.1f | %.2f |" % (
            label, pat, tmpl[:40], len(d1), us, sh("SQ_WAIT_ANY"), sh("SQ_WAIT_INST_ANY"), sh("SQ_ACTIVE_INST_ANY"),
            sh("SQ_ACTIVE_INST_VALU"), ("%.2f" % (gui / 8 / us / 1e3)) if gui == gui and us == us and us else "-", sh("SQ_ACTIVE_INST_LDS", s2) if False else
            (("%.2f" % (mean(s2.get("SQ_ACTIVE_INST_LDS", [])) / wc)) if wc == wc and wc else "-"),
            (wc * 4 / (us * 2100.0 * 1024)) if us == us and us else float("nan"),
            per("SQ_INSTS_VALU"), per("SQ_INSTS_VMEM_RD"), per("SQ_INSTS_VMEM_WR"), per("SQ_INSTS_LDS"), per("SQ_INSTS_SALU"),
            rd, wr, (rd + wr) / us if us == us and us else float("nan")))
txt = "\n".join(out) + "\n"
open(os.path.join(T, TAG + "_hbm_kernels_pmc.md"), "w").write(txt)
print(txt)
