"""The path is deterministic by contract: the same batch gives the same bytes every time, whatever else is in flight.  Three lanes
(two extractor handles and the stereo kernels each, timing events and stage profiling as bench.py has them) run the same stereo
batch over and over; every step's keypoints, descriptors, counts and stereo results are compared on the device with the first step's
(tools/check_determinism.py).  This is the test that found the quadtree's depth-limit sort case (tests/golden/octree_depth_limit_case.npz):
a race or an uninitialised read shows up here long before it shows up as a wrong answer in a single-shot parity test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shape", [(376, 1241, 2000, 72), (480, 752, 1000, 48)])
def test_repeated_steps_give_identical_bytes(shape):
    rows, cols, nf, B = shape
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_determinism.py"), "36", "12", str(rows), str(cols), str(nf), str(B)], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    assert "differing" not in p.stdout and "count fault" not in p.stdout and "done" in p.stdout, p.stdout[-3000:]


def test_repeated_mono_steps_with_the_matcher_give_identical_bytes():
    """bench.py's default step (extract 256 VGA frames, brute-force match of consecutive frames with greedy claims in speculative rounds)
    on three lanes: keypoints, descriptors, counts, match lists and match counts of every step equal the first step's."""
    env = dict(os.environ, MODE="mono")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_determinism.py"), "24", "12", "480", "640", "1000", "256"], capture_output=True, text=True, timeout=600,
                       env=env)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-2000:])
    assert "differing" not in p.stdout and "count fault" not in p.stdout and "done" in p.stdout, p.stdout[-3000:]
