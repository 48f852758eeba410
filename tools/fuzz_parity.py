"""Randomised parity campaign: images of very different statistics (flat with a few dots, noise, smoothed noise, gradients, checkerboards,
the synthetic worlds, crops with large flat areas) x sizes x constructor parameters, device against oracle, every field of every keypoint and
every descriptor bit; single-call form and, every few cases, the same frames as a batch.  Data-dependent corner cases (a level with
barely more candidates than wanted corners, tiles without a candidate, cells at the thresholds) are what this goes looking for.

usage: fuzz_parity.py [cases] [seed]   -> exit code 1 on the first mismatch (the case is printed so that it can be replayed)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np


def fuzz_image(rng, h, w):
    kind = int(rng.integers(0, 9))
    if kind == 0:  # flat with sparse dots and short strokes
        img = np.full((h, w), int(rng.integers(20, 230)), np.int32)
        for _ in range(int(rng.integers(5, 400))):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            img[y:y + int(rng.integers(1, 4)), x:x + int(rng.integers(1, 6))] += int(rng.integers(-120, 120))
    elif kind == 1:  # white noise of random amplitude
        img = 128 + rng.normal(0, rng.uniform(2, 60), (h, w))
    elif kind == 2:  # smoothed noise (box filter), contrast stretched
        n = rng.normal(0, 1, (h + 8, w + 8))
        k = int(rng.integers(2, 8))
        c = np.cumsum(np.cumsum(n, 0), 1)
        s = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
        s = s[:h, :w]
        img = 128 + s / (s.std() + 1e-9) * rng.uniform(10, 70)
    elif kind == 3:  # gradient + weak noise
        yy, xx = np.mgrid[0:h, 0:w]
        img = (xx * rng.uniform(-0.3, 0.3) + yy * rng.uniform(-0.3, 0.3)) + 128 + rng.normal(0, rng.uniform(0, 6), (h, w))
    elif kind == 4:  # checkerboard of random pitch and contrast
        p = int(rng.integers(3, 40))
        yy, xx = np.mgrid[0:h, 0:w]
        img = 128 + (((yy // p + xx // p) % 2) * 2 - 1) * rng.uniform(3, 100) + rng.normal(0, rng.uniform(0, 3), (h, w))
    elif kind in (5, 6):  # the synthetic world of the benches
        from orbslam3_cpp_fork_amd import synth
        H, W = max(h, 240), max(w, 320)
        img = synth.make_frame(H, W, int(rng.integers(0, 8)), int(rng.integers(0, 50)))[:h, :w].astype(np.int32)
        if kind == 6:  # with a large flat area
            y0, x0 = int(rng.integers(0, h // 2)), int(rng.integers(0, w // 2))
            img[y0:y0 + h // 2, x0:x0 + w // 2] = int(rng.integers(0, 255))
    elif kind == 7:  # saturated blobs
        img = rng.choice([0, 255], (h // 4 + 1, w // 4 + 1)).repeat(4, 0).repeat(4, 1)[:h, :w] + rng.normal(0, 1, (h, w))
    else:  # low-contrast texture around the thresholds
        img = 128 + rng.integers(-int(rng.integers(4, 12)), int(rng.integers(4, 12)) + 1, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    from orbslam3_cpp_fork_amd._lib import KP_DTYPE
    from orbslam3_cpp_fork_amd.extractor import ORBextractor, OrbxError
    from oracle import oracle_py as oracle  # the checker, as in tests/
    rng = np.random.default_rng(seed)
    done = 0
    for case in range(cases):
        h, w = int(rng.integers(70, 700)), int(rng.integers(90, 1300))
        if case % 10 == 0:
            h, w = [(480, 640), (376, 1241), (480, 752), (240, 320)][int(rng.integers(0, 4))]
        nf = int(rng.choice([20, 100, 500, 1000, 2000, 5000]))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5]))
        nl = int(rng.integers(1, 10))
        ini, mn = [(20, 7), (20, 7), (10, 5), (40, 20), (7, 7), (5, 20), (30, 3)][int(rng.integers(0, 7))]
        lap = [(0, 0), (0, 1000), (100, 300)][int(rng.integers(0, 3))]
        img = fuzz_image(rng, h, w)
        tag = "case %d seed %d: %dx%d nf %d sf %.1f nl %d th (%d, %d) lap %s" % (case, seed, w, h, nf, sf, nl, ini, mn, lap)
        oex = oracle.OracleExtractor(nf, sf, nl, ini, mn)
        ref = oex(img, lap)
        try:
            ex = ORBextractor(nf, sf, nl, ini, mn)
            got = ex(img, None, lap)
        except OrbxError as e:
            if ref[0] < 0:  # both refuse (a level without room for a cell, ...)
                continue
            print("MISMATCH " + tag + ": device refuses (%s), oracle returns %d keypoints" % (e, len(ref[1])), flush=True)
            return 1
        pyr_ok = all((ex.pyramid_level(l) == oex.pyramid_level(l)).all() for l in range(nl))
        if not pyr_ok:
            print("MISMATCH " + tag + ": pyramid levels %s differ" % [l for l in range(nl) if not (ex.pyramid_level(l) == oex.pyramid_level(l)).all()], flush=True)
            return 1
        ok = got[0] == ref[0] and len(got[1]) == len(ref[1]) and all((got[1][f].view(np.uint32) == ref[1][f].view(np.uint32)).all() for f in ref[1].dtype.names) and (got[2] == ref[2]).all()
        if ok and case % 4 == 0:  # the same frame three times in a batch (throughput form of every kernel)
            cap = ex.max_keypoints(h, w)
            B = 9
            d = torch.from_numpy(np.stack([img] * B)).cuda()
            k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); ds = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); c = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
            ex.extract_batch_device(d, B, h, w, w, h * w, k, ds, c, cap, lap)
            torch.cuda.synchronize()
            for f in (0, B - 1):
                n = int(c[f, 0])
                kb = k[f, :n].cpu().numpy().view(KP_DTYPE).reshape(-1)
                ok = ok and n == len(ref[1]) and int(c[f, 1]) == ref[0] and all((kb[x].view(np.uint32) == ref[1][x].view(np.uint32)).all() for x in ref[1].dtype.names) and (ds[f, :n].cpu().numpy() == ref[2]).all()
            ok = ok and ex.debug_fault() is None
        ex.close()
        if not ok:
            print("MISMATCH " + tag + ": device %d keypoints (rc %d), oracle %d (rc %d)" % (len(got[1]), got[0], len(ref[1]), ref[0]), flush=True)
            np.save(os.path.join(ROOT, "gpurun_out", "fuzz_mismatch_%d_%d.npy" % (seed, case)), img)
            return 1
        done += 1
        if case % 25 == 24:
            print("%d cases, %d compared, all equal" % (case + 1, done), flush=True)
    print("fuzz: %d cases compared, all equal" % done, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
