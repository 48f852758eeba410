import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import oracle_py as O
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ORBextractor
img = synth.make_frame(480, 640, 0, 0)
ex = ORBextractor(1000); oex = O.OracleExtractor(1000, 1.2, 8, 20, 7)
ex(img, None, (0, 0)); oex(img, (0, 0))
for l in range(8):
    g = set(zip(*[a.tolist() for a in ex.debug_level_candidates(l)])) if False else None
    gx = ex.debug_level_candidates(l)
    ox = oex.level_candidates(l)
    import itertools
    def unpack(t):
        if isinstance(t, tuple): return set(zip(t[0].tolist(), t[1].tolist(), t[2].tolist()))
        return set(((int(v) & 0xfff), (int(v) >> 12) & 0xfff, int(v) >> 24) for v in t)
    G, R = unpack(gx), unpack(ox)
    miss, extra = sorted(R - G), sorted(G - R)
    print("level", l, "ref", len(R), "gpu", len(G), "missing", len(miss), "extra", len(extra))
    print("  missing x%4:", np.bincount([m[0] % 4 for m in miss], minlength=4).tolist(), " first:", miss[:8])
    print("  extra first:", extra[:8])
