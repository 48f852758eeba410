#!/usr/bin/env python3
"""Export the frames the OpenCV-side generator runs on: the two committed 320x240 golden frames (tests/golden/extract_320x240_500.npz,
500 features) and two seeded 640x480 frames (1000 features) -> frames_320x240.npy, frames_640x480.npy in the given directory."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orbslam3_cpp_fork_amd import synth  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "."
os.makedirs(out, exist_ok=True)
z = np.load(os.path.join(ROOT, "tests", "golden", "extract_320x240_500.npz"))
np.save(os.path.join(out, "frames_320x240.npy"), np.ascontiguousarray(z["frames"]))
np.save(os.path.join(out, "frames_640x480.npy"), np.stack([synth.make_frame(480, 640, camera_idx=c) for c in (0, 1)]))
print("wrote", out)
