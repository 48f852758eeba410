#!/usr/bin/env python3
"""pack.py <frames.npy> <generator outdir> <tests/golden/opencv_NAME.npz>: bundle the generator's .npy files, the frames and meta.txt
into the fixture tests/test_golden.py::test_*_opencv_golden consume."""
import glob
import os
import sys

import numpy as np

frames, d, out = sys.argv[1:4]
meta = dict(line.split(None, 1) for line in open(os.path.join(d, "meta.txt")).read().splitlines() if line.strip())
nf, sf, nl, ini, mn = meta["params"].split()
arrs = {os.path.splitext(os.path.basename(p))[0]: np.load(p) for p in glob.glob(os.path.join(d, "*.npy"))}
arrs["frames"] = np.load(frames)
arrs["params"] = np.array([int(nf), int(nl), int(ini), int(mn)], np.int32)
arrs["scale"] = np.float32(sf)
arrs["built_with_fma"] = np.int32(meta["built_with_fma"])
arrs["opencv_version"] = np.array(meta["opencv_version"].strip())
np.savez_compressed(out, **arrs)
print("wrote %s (%d arrays, OpenCV %s, fma %s)" % (out, len(arrs), meta["opencv_version"].strip(), meta["built_with_fma"].strip()))
