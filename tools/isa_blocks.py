#!/usr/bin/env python3
"""Static instruction mix of one kernel, per basic block: tools/isa_blocks.py <file.hip> <kernel name substring> [min instrs].
Compiles the translation unit for gfx950 with -save-temps into a scratch directory and reads the device assembly."""
import glob
import os
import re
import subprocess
import sys
import tempfile

src, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 10
d = tempfile.mkdtemp(prefix="isa_")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-c", os.path.abspath(src), "-o", os.path.join(d, "x.o"), "-save-temps"],
               cwd=d, stderr=subprocess.DEVNULL, check=True)
s = open(glob.glob(os.path.join(d, "*gfx950*.s"))[0]).read()
names = [l.split(":")[0] for l in s.split("\n") if pat in l.split(":")[0] and re.match(r"^[A-Za-z_][\w$.]*:", l)]
for name in names:
    a = s.index(name + ":")
    b = s.index(".Lfunc_end", a)
    body = s[a:b].split("\n")
    out = os.path.join(d, name[:60] + ".s")
    open(out, "w").write("\n".join(body))
    print("==", name, "->", out)
    blocks = []
    cur = ["entry", 0, 0, 0, 0, 0]
    blocks.append(cur)
    for ln in body:
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            cur = [t.split(":")[0], 0, 0, 0, 0, 0]
            blocks.append(cur)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            cur[1] += 1
        elif op.startswith("s_"):
            cur[2] += 1
        elif op.startswith("ds_"):
            cur[3] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
            cur[4] += 1
        else:
            cur[5] += 1
    print("%-14s %5s %5s %5s %5s %5s" % ("block", "VALU", "SALU", "LDS", "VMEM", "other"))
    for bl in blocks:
        if bl[1] + bl[2] + bl[3] + bl[4] >= minn:
            print("%-14s %5d %5d %5d %5d %5d" % tuple(bl))
    tot = [sum(b[i] for b in blocks) for i in range(1, 6)]
    print("%-14s %5d %5d %5d %5d %5d" % ("TOTAL", *tot))
