"""Launch-weighted fabric bytes per GEMM launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the eager bench:
python tools/traffic_ab.py <dir with FETCH_SIZE/ and WRITE_SIZE/ subdirectories>.  FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import sys

tot = collections.defaultdict(lambda: [0.0, 0])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (sys.argv[1], c), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_i8_wide_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                k = (r["Kernel_Name"].split("(")[0][-40:], c)
                tot[k][0] += float(r["Counter_Value"]) * (2.0 if c == "FETCH_SIZE" else 1.0) / 1024.0   # KB -> MB
                tot[k][1] += 1
names = sorted({k[0] for k in tot})
allb, alln = 0.0, 0
for n in names:
    f, w = tot.get((n, "FETCH_SIZE"), [0, 1]), tot.get((n, "WRITE_SIZE"), [0, 1])
    print("%-42s n %5d  read %7.1f MB  write %7.1f MB per launch" % (n, f[1], f[0] / max(f[1], 1), w[0] / max(w[1], 1)))
    allb += f[0] + w[0]
    alln += f[1]
print("launch-weighted mean traffic: %.1f MB per GEMM launch over %d launches" % (allb / max(alln, 1), alln))
