#!/usr/bin/env python3
"""Measured VALU issue ceilings of the GPU (csrc/orbx_calib.hip) -> JSON on stdout (committed as profiles/r02_valu_calibration.json).

Per instruction class and residency: wave-instructions per nanosecond over the whole chip, and shader cycles per instruction per SIMD
(= cycles per wave / waves per SIMD / instructions per wave).  `peak_ginst_per_s` is the best rate seen for the packed-16-bit min
(the instruction the FAST kernel is made of) and is what roofline.valu in bench.py prices against."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslam3_cpp_fork_amd import _lib  # noqa: E402

L = _lib.lib()
res = {"ops": {}, "simds": 1024}
op = 0
while True:
    name = L.orbx_debug_valu_calib_name(op)
    if not name:
        break
    name = name.decode()
    rows = {}
    for w in (1, 2, 4, 8):
        out = (C.c_double * 3)()
        rc = L.orbx_debug_valu_calib(op, w, 2000, out)
        if rc:
            raise SystemExit("orbx_debug_valu_calib(%s, %d) -> %d: %s" % (name, w, rc, _lib.last_error()))
        ns, cyc, per_wave = out[0], out[1], out[2]
        waves = 256 * 4 * w
        rows[str(w)] = {
            "ginst_per_s": round(per_wave * waves / ns, 2),                       # wave-instructions per ns, whole chip
            "cycles_per_inst_per_simd": round(cyc / per_wave / w, 3),              # in-kernel shader clock
            "ns_per_inst_per_simd": round(ns / per_wave / w, 4),
            "implied_clock_GHz": round(cyc / ns, 3),
        }
    res["ops"][name] = rows
    op += 1
pk = res["ops"]["v_pk_min_i16"]
res["peak_ginst_per_s"] = max(r["ginst_per_s"] for r in pk.values())
print(json.dumps(res, indent=1))
