"""Exit-path probe: how the process ends when handles are leaked / closed / destroyed during interpreter finalization (run plain and under rocprofv3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from orbslam3_cpp_fork_amd import synth, _lib
from orbslam3_cpp_fork_amd.extractor import ORBextractor

mode = sys.argv[1]
img = synth.make_frame(240, 320, 2, 3)
ex = ORBextractor(500)
if "nograph" in mode:
    ex.set_graph_replay(False)
for _ in range(5):
    ex(img, None, (0, 1000))


class Late:
    def __init__(self, e):
        self.e = e

    def __del__(self):
        h, self.e._h = self.e._h, None
        _lib.lib().orbx_destroy(h)


if mode.startswith("close"):
    ex.close()
elif mode.startswith("late"):
    keep = Late(ex)  # module global: destroyed while the interpreter finalizes
print("done", mode, flush=True)
