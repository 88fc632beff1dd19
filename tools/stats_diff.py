"""Per-kernel time of the TIMED steps only: difference of two `rocprofv3 --kernel-trace --stats` kernel_stats.csv files
taken from the same bench command with different --steps (set-up work such as calibration, warm-up and graph capture
is identical in both and cancels).  usage: stats_diff.py short.csv long.csv n_extra_steps"""
import csv
import sys


def load(p):
    out = {}
    with open(p) as f:
        for r in csv.DictReader(f):
            out[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return out


a, b, n = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
for k, (cb, tb) in b.items():
    ca, ta = a.get(k, (0, 0.0))
    if cb - ca > 0:
        rows.append((tb - ta, cb - ca, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("GPU time per step (sum over kernels, both streams): %.2f ms" % (tot / n / 1e6))
print("%9s %8s %9s %6s  kernel" % ("us/step", "calls/st", "avg us", "share"))
for t, c, k in rows[:40]:
    print("%9.1f %8.1f %9.1f %5.1f%%  %s" % (t / n / 1e3, c / n, t / c / 1e3, 100 * t / tot, k[:150]))
