import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ORBextractor
for (h, w, nf) in ((480, 640, 1000), (480, 752, 1000), (376, 1241, 2000)):
    img = synth.make_frame(h, w, 2, 3)
    ex = ORBextractor(nf)
    for _ in range(5): ex(img, None, (0, 1000))
    t = time.perf_counter(); n = 50
    for _ in range(n): rc, k, d = ex(img, None, (0, 1000))
    dt = (time.perf_counter() - t) / n * 1e3
    ex.set_profiling(True)
    ex(img, None, (0, 1000))
    st = ex.stage_ms()
    ex.set_profiling(False)
    from orbslam3_cpp_fork_amd.extractor import pinned_array
    pimg = pinned_array(img.shape, np.uint8); pimg[...] = img
    for _ in range(5): ex(pimg, None, (0, 1000))
    t = time.perf_counter()
    for _ in range(n): ex(pimg, None, (0, 1000))
    dtp = (time.perf_counter() - t) / n * 1e3
    print("%dx%d/%d: %.3f ms with the image in pinned memory" % (w, h, nf, dtp))
    print("%dx%d/%d: %.3f ms per orbx_extract call (%d keypoints); device stages: %s" % (w, h, nf, dt, len(k), {a: round(b, 3) for a, b in st.items()}))
