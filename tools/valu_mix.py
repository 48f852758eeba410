#!/usr/bin/env python3
"""Static share of FULL-RATE instructions among the VALU instructions of every kernel.

tools/valu_mix.py [valu_calibration.json] > profiles/r03_valu_mix.json.  gfx950 issues a wave64 VALU instruction at one of two rates
(csrc/orbx_calib.hip, tools/valu_calibrate.py): ~1.07 per ns per SIMD ("full") or ~0.59 ("half").  Round 2 read the split off the
encoding width (32-bit VOP1 / VOP2 / VOPC = full, 64-bit = half); the longer table of round 3 shows that is only half of it:
  * every VOP3 / VOP3P / SDWA / DPP form is half rate;
  * a 32-bit-encoded form is half rate as well when one of its sources is an SGPR or a literal (v_and_b32 with an SGPR: 0.60;
    inline constants are free), and some opcodes are half rate in any form (v_lshlrev_b32, v_min / max_{u,i}32, v_mul_u32_u24, ...).
So an instruction counts as full rate iff its mnemonic measured full rate in the calibration (>= 80 % of v_add_u32) AND it is _e32 AND
none of its sources is an SGPR / vcc / exec / literal.  Mnemonics the calibration does not cover count as full rate (the ceiling derived
from the mix is then an upper bound, the reported fraction of it a lower bound); their share is listed."""
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

cal_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_valu_calibration.json")
cal = json.load(open(cal_path))
ref = max(v["ginst_per_s"] for v in cal["ops"]["v_add_u32"].values())
rate_class = {}
for name, rows in cal["ops"].items():
    base = name.split("(")[0].split("+")[0]
    plain = "(" not in name and "+" not in name and not base.endswith("_sdwa") and not base.endswith("_dpp")
    if plain:
        rate_class[base] = max(v["ginst_per_s"] for v in rows.values()) >= 0.8 * ref

INLINE = re.compile(r"^(-?(?:[0-9]|1[0-6])|[1-5][0-9]|6[0-4]|0\.5|-0\.5|1\.0|-1\.0|2\.0|-2\.0|4\.0|-4\.0|0)$")


def full_rate(line):
    t = line.split(";")[0].strip()
    op = t.split()[0]
    if not op.endswith("_e32"):
        return False, None
    base = op[:-4]
    known = rate_class.get(base)
    srcs = [x.strip() for x in t[len(op):].split(",")][1:]
    for x in srcs:
        if re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|ttmp\d+)$", x):
            return False, known
        if re.match(r"^(0x[0-9a-fA-F]+|-?\d+(\.\d+)?(e[-+]?\d+)?)$", x) and not INLINE.match(x):
            return False, known
    return (True if known is None else known), known


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
        valu = full = unknown = 0
        for ln in s[a:b].split("\n"):
            t = ln.strip()
            if not t.startswith("v_") or t.split()[0] in ("v_nop",):
                continue
            valu += 1
            f, known = full_rate(t)
            full += 1 if f else 0
            unknown += 1 if (f and known is None) else 0
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        m2 = re.search(r"(k_\w+)(<[^(]*>)?", dem)
        if not m2:
            continue
        short, variant = m2.group(1), m2.group(1) + (m2.group(2) or "")
        e = {"valu_static": valu, "full_rate_share": round(full / max(valu, 1), 3), "uncalibrated_share": round(unknown / max(valu, 1), 3),
             "half_rate_share": round(1 - full / max(valu, 1), 3), "symbol": sym, "variant": variant}
        allv[variant] = e
        # the short name stands for the instantiation the default bench runs
        prefer = {"k_fast_tiles": "<true>", "k_octree": "WaveCtx", "k_bf_topk": "<false>", "k_pyramid_bands": "<true>"}
        if short not in out or (short in prefer and prefer[short] in variant):
            out[short] = e
out["_all_instantiations"] = allv
out["_calibration"] = os.path.relpath(cal_path, ROOT)
json.dump(out, sys.stdout, indent=1)
