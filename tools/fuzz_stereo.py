"""Randomised parity campaign for Frame::ComputeStereoMatches (N1): left images of different statistics (tools/fuzz_parity.py's generators),
right images = the left shifted by a disparity that varies over the image, plus noise; sizes, feature counts, pyramid depth and the
baseline / focal pair vary.  Device (orbx_stereo_matches on the two handles' device pyramids) against the oracle: number of matches, uRight
and depth as float bits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from fuzz_parity import fuzz_image


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from orbslam3_cpp_fork_amd.extractor import ComputeStereoMatches, ORBextractor
    from oracle import oracle_py as oracle  # the checker, as in tests/
    rng = np.random.default_rng(seed)
    total = 0
    for case in range(cases):
        h, w = int(rng.integers(150, 500)), int(rng.integers(300, 1300))
        nf = int(rng.choice([300, 1000, 2000, 4000]))
        nl = int(rng.integers(3, 9))
        sfac = float(rng.choice([1.2, 1.2, 1.3]))
        left = fuzz_image(rng, h, w + 80)
        # disparity field: piecewise constant over row blocks (depth layers), 0 .. 70 px
        right = np.empty((h, w), np.uint8)
        y = 0
        while y < h:
            bh = int(rng.integers(8, 120))
            d = int(rng.integers(0, 70))
            right[y:y + bh] = left[y:y + bh, 80 - d:80 - d + w] if rng.random() < 0.85 else left[y:y + bh, 80:80 + w][:, ::-1]
            y += bh
        left = np.ascontiguousarray(left[:, 80:80 + w])
        right = np.clip(right.astype(np.int32) + rng.integers(-int(rng.integers(0, 6)) - 1, int(rng.integers(0, 6)) + 2, right.shape), 0, 255).astype(np.uint8)
        mbf = float(rng.uniform(40, 500)); mb = mbf / float(rng.uniform(300, 900))
        tag = "case %d seed %d: %dx%d nf %d sf %.1f nl %d mb %.4f mbf %.3f" % (case, seed, w, h, nf, sfac, nl, mb, mbf)
        gl, gr = ORBextractor(nf, sfac, nl, 20, 7), ORBextractor(nf, sfac, nl, 20, 7)
        _, kl, dl = gl(left, None, (0, 0))
        _, kr, dr = gr(right, None, (0, 0))
        ol, orr = oracle.OracleExtractor(nf, sfac, nl, 20, 7), oracle.OracleExtractor(nf, sfac, nl, 20, 7)
        _, kl0, dl0 = ol(left, (0, 0))
        _, kr0, dr0 = orr(right, (0, 0))
        if len(kl) != len(kl0) or len(kr) != len(kr0) or not (dl == dl0).all() or not (dr == dr0).all():
            print("MISMATCH (extraction) " + tag, flush=True)
            return 1
        n0, u0, d0 = oracle.stereo_matches(ol, orr, kl, dl, kr, dr, mb, mbf)
        n1, u1, d1 = ComputeStereoMatches(gl, gr, kl, dl, kr, dr, mb, mbf)
        gl.close(), gr.close()
        if n0 != n1 or not (u0.view(np.uint32) == u1.view(np.uint32)).all() or not (d0.view(np.uint32) == d1.view(np.uint32)).all():
            print("MISMATCH " + tag + ": matches %d vs %d, uRight differing %d, depth differing %d" % (n1, n0, int((u0.view(np.uint32) != u1.view(np.uint32)).sum()), int((d0.view(np.uint32) != d1.view(np.uint32)).sum())), flush=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_stereo_mismatch_%d_%d.npz" % (seed, case)), left=left, right=right)
            return 1
        total += n0
        if case % 20 == 19:
            print("%d cases, %d stereo matches so far, all equal" % (case + 1, total), flush=True)
    print("fuzz stereo: %d cases, %d stereo matches, all equal" % (cases, total), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
