import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ORBextractor
from orbslam3_cpp_fork_amd.vocabulary import ORBVocabulary
B, rows, cols = 256, 480, 640
fr = synth.make_stream(rows, cols, B)
dev = torch.device('cuda', 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
d = torch.from_numpy(fr).to(dev)
ex = ORBextractor(1000); cap = ex.max_keypoints(rows, cols)
kps = torch.zeros((B, cap, 7), device=dev); desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev); cnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=st.cuda_stream)
voc = synth.make_vocabulary(10, 6, seed=11)
gv = ORBVocabulary.from_arrays(10, 6, *voc)
out = {"bow_words": torch.zeros((B, cap), dtype=torch.int32, device=dev), "bow_vals": torch.zeros((B, cap), dtype=torch.float64, device=dev),
       "n_words": torch.zeros(B, dtype=torch.int32, device=dev), "fv_nodes": torch.zeros((B, cap), dtype=torch.int32, device=dev),
       "fv_off": torch.zeros((B, cap + 1), dtype=torch.int32, device=dev), "fv_feat": torch.zeros((B, cap), dtype=torch.int32, device=dev),
       "n_nodes": torch.zeros(B, dtype=torch.int32, device=dev)}
for _ in range(3):
    gv.transform_batch_device(desc, cnt, B, cap, 4, out, stream=st.cuda_stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(10):
    gv.transform_batch_device(desc, cnt, B, cap, 4, out, stream=st.cuda_stream)
e1.record(st); torch.cuda.synchronize()
print("transform of %d frames: %.3f ms  (words/frame %.1f, nodes/frame %.1f)" % (B, e0.elapsed_time(e1) / 10, out["n_words"].float().mean().item(), out["n_nodes"].float().mean().item()))
