"""Randomised parity campaign for the brute-force SearchByBoW pass (the matrix-core Hamming kernel + claim resolution): descriptor sets
drawn from a SMALL pool with a few flipped bits, so that equal distances, equal best / second-best pairs and claim conflicts are the rule
(what decides them is the order of the reference's loops); sizes from 0 to 2500 on either side, thresholds, ratios, both comparison
forms, with and without the rotation histogram.  Device (orbm_match_groups, one group) against the oracle: count and both match arrays."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from orbslam3_cpp_fork_amd.matcher import ORBmatcher
    from oracle import oracle_py as oracle  # the checker, as in tests/
    rng = np.random.default_rng(seed)
    total = 0
    for case in range(cases):
        nq = int(rng.choice([0, 1, 7, 63, 64, 65, 255, 256, 257, 1000, int(rng.integers(0, 2500))]))
        nt = int(rng.choice([0, 1, 5, 63, 64, 65, 500, 1000, int(rng.integers(0, 2500))]))
        pool = rng.integers(0, 256, (int(rng.integers(1, 60)), 32), dtype=np.uint8)
        def draw(n):
            d = pool[rng.integers(0, len(pool), n)].copy()
            flips = int(rng.integers(0, 5))
            for _ in range(flips):
                rows = rng.random(n) < 0.5
                byte, bit = rng.integers(0, 32, n), rng.integers(0, 8, n)
                d[np.arange(n)[rows], byte[rows]] ^= (1 << bit[rows]).astype(np.uint8)
            if rng.random() < 0.3 and n:  # a share of unrelated descriptors
                k = rng.random(n) < 0.4
                d[k] = rng.integers(0, 256, (int(k.sum()), 32), dtype=np.uint8)
            return d
        q, t = draw(nq), draw(nt)
        bins = rng.choice([0.0, 11.9, 12.0, 100.0, 348.0, 359.9], 3)
        qa, ta = rng.choice(bins, nq).astype(np.float32), rng.choice(bins, nt).astype(np.float32)
        ratio = float(rng.choice([0.6, 0.75, 0.9, 1.0]))
        th = int(rng.choice([0, 3, 30, 50, 100, 256]))
        ori = bool(rng.integers(0, 2))
        strict = bool(rng.integers(0, 2))
        tag = "case %d seed %d: nq %d nt %d pool %d ratio %.2f th %d strict %s ori %s" % (case, seed, nq, nt, len(pool), ratio, th, strict, ori)
        m = ORBmatcher(ratio, ori)
        n0, mq0, mt0 = oracle.match_groups(q, t, qa, ta, th=th, ratio=ratio, strict_lt=strict, check_ori=ori)
        n1, mq1, mt1 = m._groups(q, t, qa, ta, None, None, None, th, strict)
        m.close()
        if n0 != n1 or not (mq0 == mq1).all() or not (mt0 == mt1).all():
            print("MISMATCH " + tag + ": %d vs %d matches, %d query entries differ" % (n1, n0, int((mq0 != mq1).sum())), flush=True)
            np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_matcher_mismatch_%d_%d.npz" % (seed, case)), q=q, t=t, qa=qa, ta=ta)
            return 1
        total += n0
        if case % 50 == 49:
            print("%d cases, %d matches so far, all equal" % (case + 1, total), flush=True)
    print("fuzz matcher: %d cases, %d matches, all equal" % (cases, total), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
