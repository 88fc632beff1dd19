"""GPU busy fraction and concurrency inside the bench's timed region from a `rocprofv3 --kernel-trace` CSV:
union of kernel intervals / wall, time with >= 2 kernels in flight, and the distribution of idle gaps, over the last
`window_ms` milliseconds of the trace (the timed steps run last).  usage: trace_busy.py kernel_trace.csv [window_ms]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t_end = max(e for _, e in iv)
t0 = t_end - win
iv = [(max(s, t0), e) for s, e in iv if e > t0]
ev = []
for s, e in iv:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = two = 0
depth = 0
last = t0
gaps = []
gap_start = t0
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        two += t - last
    if depth == 0 and d == 1 and t > last:
        gaps.append(t - last)
    depth += d
    last = t
wall = t_end - t0
print("window %.1f ms: %d kernels, busy %.1f %% of wall, >= 2 kernels in flight %.1f %%, sum of kernel time %.1f %% of wall" % (
    wall / 1e6, len(iv), 100 * busy / wall, 100 * two / wall, 100 * sum(e - s for s, e in iv) / wall))
gaps.sort()
if gaps:
    n = len(gaps)
    print("idle gaps: %d, total %.2f ms (%.1f %%), median %.2f us, p90 %.2f us, max %.1f us" % (
        n, sum(gaps) / 1e6, 100 * sum(gaps) / wall, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3, gaps[-1] / 1e3))
