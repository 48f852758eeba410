#!/usr/bin/env python3
"""profiles/r01_f_sq_counters.txt -> profiles/r01_sq_summary.json: per kernel, the share of SIMD issue slots taken by VALU
instructions = SQ_INSTS_VALU per wave / SQ_WAVE_CYCLES per wave (both in 4-clock issue units: a wave64 VALU op holds the SIMD for
one unit) x waves resident per SIMD.  Residency = min(register / LDS occupancy of the kernel, waves the grid puts on a SIMD)."""
import json, re, sys
RES = {  # waves per SIMD while the kernel runs (compiler occupancy report; grid-limited where noted)
    "k_resize_level": 8, "k_fast_cells": 7,  # FAST: 15 two-wave workgroups per CU (10 KB of LDS each)
    "k_blur_levels": 8, "k_orient_describe": 6,
    "k_bf_topk": 4,      # 1020 workgroups of 4 waves on 256 CUs: 16 waves per CU
    "k_octree": 2,       # 2048 one-wave problems on 1024 SIMDs
    "k_group_resolve": 1, "k_blur_edges": 4, "k_layout": 1,
}
out = {}
for line in open(sys.argv[1]):
    m = re.match(r"(k_\w+)\s+waves=\s*(\d+)\s+(.*)", line)
    if not m or "INSTS_VALU" not in line:
        continue
    kv = dict(p.split("=") for p in m.group(3).split())
    valu, cyc = float(kv["INSTS_VALU"]), float(kv["WAVE_CYCLES"])
    r = RES.get(m.group(1), 1)
    out[m.group(1)] = {"valu_insts_per_wave": valu, "wave_cycles_per_wave": cyc, "waves_per_simd": r, "valu_issue_frac": round(min(valu / cyc * r, 1.0), 3)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
