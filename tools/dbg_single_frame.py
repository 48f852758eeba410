"""Wall time of one synchronous orbx_extract call (upload, kernels, download, one synchronisation) at the three single-camera sizes:
the default form (the call replays a hipGraph recorded at its second use), the eager form, and the default form fed from pinned memory."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from orbslam3_cpp_fork_amd import synth
from orbslam3_cpp_fork_amd.extractor import ORBextractor, pinned_array


def timed(ex, img, n=50):
    for _ in range(5):
        ex(img, None, (0, 1000))
    t = time.perf_counter()
    for _ in range(n):
        r = ex(img, None, (0, 1000))
    return (time.perf_counter() - t) / n * 1e3, r


for (h, w, nf) in ((480, 640, 1000), (480, 752, 1000), (376, 1241, 2000)):
    img = synth.make_frame(h, w, 2, 3)
    ex = ORBextractor(nf)
    dt, (rc, k, d) = timed(ex, img)
    ex.set_graph_replay(False)
    dte, (rce, ke, de) = timed(ex, img)
    same = rce == rc and ke.tobytes() == k.tobytes() and de.tobytes() == d.tobytes()
    ex.set_profiling(True)
    ex(img, None, (0, 1000))
    st = ex.stage_ms()
    ex.set_profiling(False)
    ex.set_graph_replay(True)
    pimg = pinned_array(img.shape, np.uint8); pimg[...] = img
    dtp, _ = timed(ex, pimg)
    print("%dx%d/%d: %.3f ms per orbx_extract call (%d keypoints; hipGraph replay, the default); %.3f ms eager (results identical: %s); "
          "%.3f ms with the image in pinned memory; device stages of an eager call: %s" % (w, h, nf, dt, len(k), dte, same, dtp, {a: round(b, 3) for a, b in st.items()}))
    del pimg, ex
