"""Print the mean of every PMC counter per kernel from the rocprofv3 counter CSVs under a directory (tools/pmc_*.sh)."""
import collections
import csv
import glob
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in d.items():
    if pat in k:
        print(k[:100])
        wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) / max(len(cs.get("SQ_WAVE_CYCLES", [1])), 1)
        for c, v in sorted(cs.items()):
            m = sum(v) / len(v)
            print("   %-28s %14.0f %s" % (c, m, ("(%.3f of WAVE_CYCLES)" % (m / wc)) if wc and c.startswith("SQ_") and "CYCLES" not in c[:3] else ""))
