"""A few dozen synchronous orbx_extract calls at one size (for a kernel trace of the single-call path): tools/trace_single.sh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ORBextractor

h, w, nf = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (480, 640, 1000)))
img = synth.make_frame(h, w, 2, 3)
ex = ORBextractor(nf)
if len(sys.argv) > 4 and sys.argv[4] == "eager":
    ex.set_graph_replay(False)
for _ in range(40):
    ex(img, None, (0, 1000))
