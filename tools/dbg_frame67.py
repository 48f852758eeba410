"""Frame 67 of the stereo bench's left batch comes out differently from step to step (tools/check_determinism.py).  This fetches its level
candidates (FAST) and level keypoints (octree) after each of a number of repeated batch extractions and compares them with the oracle's:
which stage varies, and what the varying records are."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ORBextractor
from oracle import oracle_py as oracle  # diagnosis tool: the oracle is the checker here, as in tests/
rows, cols, nf, B, F = 376, 1241, 2000, 128, 67
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
imgs = np.stack([synth.make_stereo_pair(rows, cols, camera_idx=i % 8, frame_idx=i // 8)[0] for i in range(B)])
oex = oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
oex(imgs[F], (0, 0))
pk = lambda t: set(zip(*[a.tolist() for a in t]))
ocand = [pk(oex.level_candidates(l)) for l in range(8)]
okp = [oex.level_keypoints(l) for l in range(8)]
okp = [set(zip((k["x"].astype(int) - 16).tolist(), (k["y"].astype(int) - 16).tolist(), k["response"].astype(int).tolist())) for k in okp]
d = torch.from_numpy(imgs).cuda()
for mode, (dd, b, f) in (("batch of 128", (d, B, F)), ("frame alone", (d[F:F + 1].contiguous(), 1, 0))):
    ex = ORBextractor(nf)
    cap = ex.max_keypoints(rows, cols)
    k, ds, c = torch.zeros((b, cap, 7), dtype=torch.float32, device="cuda"), torch.zeros((b, cap, 32), dtype=torch.uint8, device="cuda"), torch.zeros((b, 2), dtype=torch.int32, device="cuda")
    for r in range(reps):
        ex.extract_batch_device(dd, b, rows, cols, cols, rows * cols, k, ds, c, cap, (0, 0))
        torch.cuda.synchronize()
        msgs = []
        for l in range(8):
            gc, gk = pk(ex.debug_level_candidates(l, frame=f)), pk(ex.debug_level_keypoints(l, frame=f))
            if gc != ocand[l]:
                msgs.append("L%d candidates: %d missing %s, %d extra %s" % (l, len(ocand[l] - gc), sorted(ocand[l] - gc)[:6], len(gc - ocand[l]), sorted(gc - ocand[l])[:6]))
            if gk != okp[l]:
                msgs.append("L%d keypoints: gpu %d oracle %d, %d missing, %d extra" % (l, len(gk), len(okp[l]), len(okp[l] - gk), len(gk - okp[l])))
        print("%s rep %d: count %d; %s" % (mode, r, int(c[f, 0]), "; ".join(msgs) if msgs else "every level = oracle"), flush=True)
    ex.close()
