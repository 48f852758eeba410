#!/usr/bin/env python3
"""Timeline of the TIMED REGION of a short bench run (rocprofv3 kernel trace of `bench.py --steps K ... --no-cpu-baseline --no-pcie
--no-single-call`): the region is the K steps before the last long idle gap... found as the block of kernels between the two largest gaps
that holds K launches of k_bf_mfma.  Prints its wall time, idle time at the head / tail, and per 0.25 ms slice how many kernels ran and the
share of the slice with fewer than two kernels in flight.  usage: trace_timed_region.py <kernel_trace.csv> K"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2])
name = lambda k: (re.search(r"(k_\w+)", k) or re.search(r"(\w+)", k)).group(1)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name(r["Kernel_Name"])) for r in rows)
# split at gaps > 150 us (fences, host work between passes)
blocks, cur = [], [iv[0]]
end = iv[0][1]
for x in iv[1:]:
    if x[0] - end > 150000:
        blocks.append(cur); cur = []
    cur.append(x); end = max(end, x[1])
blocks.append(cur)
cands = [b for b in blocks if sum("k_bf_mfma" in n for _, _, n in b) == K]
if not cands:
    print("no block with %d matcher launches; blocks: %s" % (K, [sum("k_bf_mfma" in n for _, _, n in b) for b in blocks])); sys.exit(1)
b = cands[0]
t0, t1 = b[0][0], max(e for _, e, _ in b)
print("timed region: %d kernels, wall %.3f ms (%.4f ms per step)" % (len(b), (t1 - t0) / 1e6, (t1 - t0) / 1e6 / K))
step = 250000
t = t0
while t < t1:
    a, z = t, min(t + step, t1)
    evs = []
    for s, e, n in b:
        if e > a and s < z:
            evs.append((max(s, a), 1)); evs.append((min(e, z), -1))
    evs.sort()
    depth, last, low, idle = 0, a, 0, 0
    for tt, d in evs:
        if depth < 2: low += tt - last
        if depth == 0: idle += tt - last
        depth += d; last = tt
    if depth < 2: low += z - last
    if depth == 0: idle += z - last
    names = sorted({n.replace("orbx::", "")[:14] for s, e, n in b if e > a and s < z})
    print("  +%6.2f ms: idle %3.0f %%  <2 kernels in flight %3.0f %%  %s" % ((a - t0) / 1e6, 100.0 * idle / (z - a), 100.0 * low / (z - a), " ".join(names)[:150]))
    t += step
