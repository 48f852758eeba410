#!/usr/bin/env python3
"""Reduce a tools/collect_profiles.sh output directory to one JSON summary (per kernel: mean duration, HBM traffic)."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]


def one(pat):
    g = glob.glob(pat)
    return g[0] if g else None


def short(n):
    import re

    m = re.search(r"\b(k_[a-z0-9_]+)", n)
    return m.group(1) if m else n


out = {"kernels": {}, "calibration": {}}
f = one(d + "/trace/*/*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k.startswith("k_"):
            out["kernels"].setdefault(k, {}).update(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, pct=float(r["Percentage"]))
f = one(d + "/trace_default/*/*kernel_stats.csv")
if f:  # the default command (several chunks in flight)
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k.startswith("k_"):
            out["kernels"].setdefault(k, {}).update(avg_ms_default_command=float(r["AverageNs"]) / 1e6, calls_default_command=int(r["Calls"]))
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = one(d + "/%s/*/*counter_collection.csv" % sub)
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k.startswith("k_"):
            out["kernels"].setdefault(k, {})[name + "_KB_raw"] = sum(v) / len(v)
f = one(d + "/calib/*/*counter_collection.csv")
factor = None
if f:
    for r in csv.DictReader(open(f)):
        if "k_calib_read" in r["Kernel_Name"]:
            w = r["Kernel_Name"].split("<")[1].split(">")[0]
            frac = float(r["Counter_Value"]) * 1024 / float(1 << 30)
            out["calibration"]["bytes_per_lane_%s" % w] = {"FETCH_SIZE_KB": float(r["Counter_Value"]), "fraction_of_true_bytes": frac}
    fr = [v["fraction_of_true_bytes"] for v in out["calibration"].values()]
    if fr:
        factor = 1.0 / (sum(fr) / len(fr))
out["fetch_correction_factor"] = factor if factor else 2.0  # gfx950: FETCH_SIZE counts 128-B requests as 64 B
for k, v in out["kernels"].items():
    if "FETCH_SIZE_KB_raw" in v:
        v["hbm_read_MB"] = v["FETCH_SIZE_KB_raw"] * 1024 * out["fetch_correction_factor"] / 1e6
    if "WRITE_SIZE_KB_raw" in v:
        v["hbm_write_MB"] = v["WRITE_SIZE_KB_raw"] * 1024 / 1e6
    if "hbm_read_MB" in v and "hbm_write_MB" in v:
        v["hbm_traffic_MB_per_launch"] = v["hbm_read_MB"] + v["hbm_write_MB"]
for line in open(d + "/bench_trace.log"):
    if line.startswith('{"metric"'):
        out["bench"] = json.loads(line)
try:
    for line in open(d + "/bench_trace_default.log"):
        if line.startswith('{"metric"'):
            out["bench_default_command"] = json.loads(line)
except OSError:
    pass
print(json.dumps(out, indent=1))
