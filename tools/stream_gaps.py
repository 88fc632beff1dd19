"""Gaps between consecutive kernels of ONE hardware queue inside the timed region, from a `rocprofv3 --kernel-trace
--output-format csv` trace: what a dependent launch costs behind its predecessor (HIP-graph replay vs eager launches).
Per queue over the last `window_ms` of the trace: kernels, busy time, and the distribution of end -> next-start gaps
(gaps above 30 us are counted separately: host stalls / joins, not launch overhead).
usage: stream_gaps.py kernel_trace.csv [window_ms]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6
t_end = max(int(r["End_Timestamp"]) for r in rows)
t0 = t_end - win
by_q = defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > t0:
        by_q[r.get("Queue_Id", "?")].append((s, e))
print("window %.1f ms" % (win / 1e6))
for q, iv in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    iv.sort()
    if len(iv) < 20:
        continue
    gaps = [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
    small = sorted(g for g in gaps if 0 <= g < 30e3)
    neg = sum(1 for g in gaps if g < 0)
    big = [g for g in gaps if g >= 30e3]
    busy = sum(e - s for s, e in iv)
    n = len(small)
    print("queue %s: %d kernels, kernel time %.2f ms; back-to-back gaps: %d, median %.2f us, mean %.2f us, p90 %.2f us, "
          "total %.2f ms; overlapping successors %d; gaps >= 30 us: %d (%.2f ms)" % (
              q, len(iv), busy / 1e6, n, small[n // 2] / 1e3 if n else 0, sum(small) / max(n, 1) / 1e3,
              small[int(n * 0.9)] / 1e3 if n else 0, sum(small) / 1e6, neg, len(big), sum(big) / 1e6))
