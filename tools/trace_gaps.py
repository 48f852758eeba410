#!/usr/bin/env python3
"""Busy / idle accounting of a rocprofv3 kernel trace: tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]
Merges the kernel intervals of the second half of the trace (the timed region of bench.py) and prints the wall time, the time with at
least one kernel running, the idle time, and the summed kernel time (sum > busy means kernels ran beside each other)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
cut = t0 + (t1 - t0) * skip
iv = [x for x in iv if x[0] >= cut]
wall = max(e for _, e, _ in iv) - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e, _ in iv)
gaps.sort(reverse=True)
print("kernels %d  wall %.3f ms  busy %.3f ms (%.1f %%)  idle %.3f ms in %d gaps (largest %s us)  summed kernel time %.3f ms (%.2f x busy)" % (
    len(iv), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6, len(gaps), [round(g / 1e3, 1) for g in gaps[:5]], ksum / 1e6, ksum / busy))
