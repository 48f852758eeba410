#!/usr/bin/env python3
"""Static share of 64-bit-encoded (VOP3 / VOP3P / SDWA / DPP) instructions among the VALU instructions of every kernel.

tools/valu_mix.py > profiles/r02_valu_mix.json.  Compiles the HIP translation units for gfx950 with -save-temps in a scratch directory and
reads the device assembly: a `v_*` mnemonic with the `_e32` suffix is a 32-bit-encoded VOP1 / VOP2 / VOPC instruction, everything else
(`_e64`, `_sdwa`, `_dpp`, and the VOP3-only / VOP3P opcodes, which carry no suffix) is 64-bit encoded.  tools/valu_calibrate.py measures
the issue rate of the two classes; tools/valu_summary.py combines them with this mix into a per-kernel ceiling."""
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orbslam3_cpp_fork_amd import build  # noqa: E402

out, allv = {}, {}
for src in build.HIP_SOURCES:
    d = tempfile.mkdtemp(prefix="mix_")
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + build.HIPFLAGS + ["-c", os.path.join(build.CSRC, src), "-o", os.path.join(d, "x.o"), "-save-temps"],
                   cwd=d, stderr=subprocess.DEVNULL, check=True)
    s = open(glob.glob(os.path.join(d, "*gfx950*.s"))[0]).read()
    for m in re.finditer(r"^([A-Za-z_][\w$.]*):\s*;\s*@\1\s*$", s, re.M):
        sym = m.group(1)
        a = m.end()
        b = s.find(".Lfunc_end", a)
        if b < 0 or ".amdhsa_kernel " + sym not in s:
            continue
        valu = vop3 = 0
        for ln in s[a:b].split("\n"):
            t = ln.strip()
            if not t.startswith("v_"):
                continue
            op = t.split()[0]
            if op in ("v_nop",):
                continue
            valu += 1
            vop3 += 0 if op.endswith("_e32") else 1
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        m2 = re.search(r"(k_\w+)(<[^(]*>)?", dem)
        if not m2:
            continue
        short, variant = m2.group(1), m2.group(1) + (m2.group(2) or "")
        e = {"valu_static": valu, "vop3_share": round(vop3 / max(valu, 1), 3), "symbol": sym, "variant": variant}
        allv[variant] = e
        # the short name stands for the instantiation the default bench runs
        prefer = {"k_fast_tiles": "<true, false>", "k_octree": "WaveCtx", "k_bf_topk": "<false>", "k_fast_cells": "<52>"}
        if short not in out or (short in prefer and prefer[short] in variant):
            out[short] = e
out["_all_instantiations"] = allv
json.dump(out, sys.stdout, indent=1)
