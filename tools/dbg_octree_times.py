import os, sys, ctypes as C, numpy as np, torch
os.environ["ORBX_DBG_OCT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslam3_cpp_fork_amd import synth, _lib
from orbslam3_cpp_fork_amd.extractor import ORBextractor
B, rows, cols = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 480, 640
fr = synth.make_stream(rows, cols, B)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts)
d = torch.from_numpy(fr).cuda()
ex = ORBextractor(1000); cap = ex.max_keypoints(rows, cols)
kps = torch.zeros((B, cap, 7), device="cuda"); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
for _ in range(2):
    ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=ts.cuda_stream)
torch.cuda.synchronize()
out = np.zeros(B * 8 * 8, np.int64)
n = _lib.lib().orbx_debug_octree_times(ex._h, out.ctypes.data_as(C.c_void_p), len(out))
t = out[:n].reshape(B, 8, 8)
for l in range(8):
    m = t[:, l].mean(0)
    print("level %d: total %7.0f cyc = sort %6.0f + points %6.0f + scans %6.0f + rewrite %6.0f + roots %6.0f + rest %6.0f;  steps %.1f sorts %.1f" % (
        l, m[1], m[0], m[4], m[5], m[6], m[7], m[1] - m[0] - m[4] - m[5] - m[6] - m[7], m[2], m[3]))
