"""FAST stage time of a 256-frame VGA batch for (iniThFAST, minThFAST) = (20, 7) and (20, 20): what the minThFAST retry costs."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ORBextractor
B, rows, cols = 256, 480, 640
d = torch.from_numpy(synth.make_stream(rows, cols, B)).cuda()
for ini, mn in ((20, 7), (20, 20), (7, 7)):
    ex = ORBextractor(1000, 1.2, 8, ini, mn); cap = ex.max_keypoints(rows, cols)
    kps = torch.zeros((B, cap, 7), device="cuda"); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=st)
    torch.cuda.synchronize(); ex.set_profiling(True)
    for _ in range(5): ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=st)
    torch.cuda.synchronize()
    print((ini, mn), {k: round(v, 4) for k, v in ex.stage_ms().items()}, "kp/frame", float(cnt[:, 0].float().mean()))
    ex.close()
