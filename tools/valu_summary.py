#!/usr/bin/env python3
"""tools/collect_profiles.sh output directory -> per kernel: VALU wave-instructions per launch (SQ_INSTS_VALU summed over the waves of
a dispatch), achieved wave-instructions per second (/ the rocprof average duration of the SAME command's kernel-trace pass) and the
fraction of the MEASURED issue ceiling (tools/valu_calibrate.py; no residency constants, no clamps).  The ceiling depends on the
rate class of the instructions (tools/valu_mix.py: full rate ~1.07 per ns per SIMD for plain VGPR / inline-constant VOP2 forms of the
opcodes that measure so, half rate ~0.59 for everything else); the static share of the two classes in each kernel's code object gives
the kernel's own ceiling 1 / (f_full / R_full + f_half / R_half).  Also the L2 hit rate.  Usage: valu_summary.py <dir> [mix.json]"""
import collections
import csv
import glob
import json
import re
import sys

d = sys.argv[1]


def short(n):
    m = re.search(r"\b(k_[a-z0-9_]+)", n)
    return m.group(1) if m else n


def one(pat):
    g = glob.glob(pat)
    return g[0] if g else None


cal = json.load(open(d + "/valu_calibration.json"))
r2 = max(v["ginst_per_s"] for v in cal["ops"]["v_add_u32"].values())
r3 = max(v["ginst_per_s"] for v in cal["ops"]["v_pk_min_i16"].values())
summ = json.load(open(d + "/summary.json"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
f = one(d + "/pmc_sq/*/*counter_collection.csv")
for r in csv.DictReader(open(f)):
    agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
tcc = collections.defaultdict(lambda: collections.defaultdict(list))
f = one(d + "/pmc_tcc/*/*counter_collection.csv")
if f:
    for r in csv.DictReader(open(f)):
        tcc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
import os

mix = {}
for cand in ([sys.argv[2]] if len(sys.argv) > 2 else []) + [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_valu_mix.json")]:
    try:
        mix = json.load(open(cand))
        break
    except Exception:
        pass
out = {"calibration": {"full_rate_ginst_per_s": r2, "half_rate_ginst_per_s": r3, "peak_ginst_per_s": r2, "source": "tools/valu_calibrate.py (csrc/orbx_calib.hip), same box, same run"},
       "kernels": {}}
for k, c in agg.items():
    if not k.startswith("k_") or k not in summ["kernels"] or "avg_ms" not in summ["kernels"][k]:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    ms = summ["kernels"][k]["avg_ms"]
    g = m["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9
    f3 = mix.get(k, {}).get("half_rate_share")
    ceil_k = 1.0 / ((1 - f3) / r2 + f3 / r3) if f3 is not None else r3
    e = {"avg_ms": round(ms, 5), "valu_insts_per_launch": m["SQ_INSTS_VALU"], "waves": m.get("SQ_WAVES"), "valu_ginst_per_s": round(g, 2),
         "half_rate_share_static": f3, "kernel_ceiling_ginst_per_s": round(ceil_k, 1), "valu_frac": round(g / ceil_k, 4),
         "wave_cycles_per_wave": round(m.get("SQ_WAVE_CYCLES", 0) / max(m.get("SQ_WAVES", 1), 1), 1),
         "wait_any_share": round(m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3),
         "wait_inst_any_share": round(m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 3)}
    if k in tcc and "TCC_HIT_sum" in tcc[k]:
        h, mi = sum(tcc[k]["TCC_HIT_sum"]), sum(tcc[k]["TCC_MISS_sum"])
        e["l2_hit_rate"] = round(h / max(h + mi, 1), 4)
    out["kernels"][k] = e
print(json.dumps(out, indent=1))
