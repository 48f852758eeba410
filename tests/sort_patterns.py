"""Inputs for the std::sort re-enactment tests (host: test_abi_and_hostmodel.py, device: test_gpu_extractor.py)."""
import os

import numpy as np


def sort_patterns(rng):
    """Key sequences that stress introsort: ties, descending runs (what a sparse level's node list looks like: tests/golden/
    octree_depth_limit_case.npz), saw teeth, organ pipes, the median-of-3 killer; as records (key << 16 | position)."""
    out = []
    for n in (2, 3, 15, 16, 17, 18, 31, 32, 33, 57, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 600):
        ks = [rng.integers(0, 4, n), rng.integers(0, 1 << 20, n), np.arange(n)[::-1], np.arange(n), np.arange(n)[::-1] // 3, np.arange(n) % 7, (np.arange(n) * 37) % 11,
              np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]), np.concatenate([np.arange(n // 2)[::-1], np.arange(n - n // 2)[::-1]])]
        k, a = n // 2, np.zeros(n, np.int64)
        for i in range(1, k + 1):  # Musser's sequence against median-of-3
            if i % 2 == 1:
                a[i - 1] = i
                if i < n:
                    a[i] = k + i
            if k + i - 1 < n:
                a[k + i - 1] = 2 * i
        ks.append(a)
        for key in ks:
            key = np.asarray(key).astype(np.uint64)
            out.append(((key << np.uint64(16)) | np.arange(n, dtype=np.uint64)).astype(np.uint64))
    out.append(np.load(os.path.join(os.path.dirname(__file__), "golden", "octree_depth_limit_case.npz"))["sort_records"].astype(np.uint64))
    return out
