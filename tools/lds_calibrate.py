#!/usr/bin/env python3
"""LDS read issue rate, aligned and unaligned (csrc/orbx_calib.hip::k_lds_calib) -> JSON on stdout."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslam3_cpp_fork_amd import _lib  # noqa: E402

L = _lib.lib()
res = {}
for width in (4, 8):
    for off in (0, 1, 2, 3, 4):
        for w in (1, 2, 4):
            out = (C.c_double * 4)()
            rc = L.orbx_debug_lds_calib(width, off, w, 2000, out)
            if rc:
                raise SystemExit("orbx_debug_lds_calib -> %d: %s" % (rc, _lib.last_error()))
            ns, cyc, per_wave, bad = out
            res["b%d_off%d_w%d" % (width * 8, off, w)] = {
                "cycles_per_read_per_cu": round(cyc / per_wave / w / 4 * 4, 3),  # 4 waves of a workgroup share the CU's LDS: per wave / (waves per SIMD)
                "ginst_per_s": round(per_wave * 256 * 4 * w / ns, 2),
                "bytes_per_clk_per_cu": round(per_wave * 4 * w * 64 * width / cyc, 1),
                "wrong_lanes": int(bad),
            }
print(json.dumps(res, indent=1))
