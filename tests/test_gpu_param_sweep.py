"""Parity beyond the canonical (scaleFactor 1.2, nlevels 8): the reference's constructor takes its five parameters freely
(src/ORBextractor.cc:403-465) and orbx_create accepts them, so the HIP path must equal the oracle for them too -- float bits of every
keypoint field, every descriptor byte, the return value -- through BOTH forms of the extraction: the single synchronous call (latency form:
band-fused pyramid, workgroup-per-problem quadtree) and the device batch (one resize launch per level, wave-per-problem quadtree)."""
import numpy as np
import pytest

from tests.sweep_cases import CASES, sweep_image


def _id(case):
    (h, w), nf, sf, nl, ini, mn, lap, strided = case
    return "%dx%d-nf%d-sf%g-L%d-th%d_%d-lap%d_%d%s" % (w, h, nf, sf, nl, ini, mn, lap[0], lap[1], "-strided" if strided else "")


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=_id)
def test_extractor_parameter_sweep(oracle, case):
    import torch

    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    (h, w), nf, sf, nl, ini, mn, lap, strided = case
    img = sweep_image(h, w, strided)
    rc0, k0, d0 = oracle.OracleExtractor(nf, sf, nl, ini, mn)(img, lap)
    gex = ORBextractor(nf, sf, nl, ini, mn)
    rc, k, d = gex(img, None, lap)
    assert rc == rc0 and len(k) == len(k0), (rc, rc0, len(k), len(k0))
    assert (k.view(np.uint32) == k0.view(np.uint32)).all() and (d == d0).all()
    oex = oracle.OracleExtractor(nf, sf, nl, ini, mn)
    oex(img, lap)
    for l in range(nl):  # the public pyramid (mvImagePyramid), levels of a few pixels included
        assert (gex.pyramid_level(l) == oex.pyramid_level(l)).all(), "pyramid level %d" % l
    if h * w > 1300 * 1700:
        return
    B, cap, n = 9, gex.max_keypoints(h, w), len(k0)
    dimg = torch.from_numpy(np.ascontiguousarray(img)).cuda().unsqueeze(0).repeat(B, 1, 1).contiguous()
    kps = torch.zeros((B, cap, 7), device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    gex.extract_batch_device(dimg, B, h, w, w, h * w, kps, desc, cnt, cap, lap, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    kk, dd, cc = kps.cpu().numpy().view(np.uint32), desc.cpu().numpy(), cnt.cpu().numpy()
    for f in (0, 4, 8):
        assert cc[f, 0] == n and cc[f, 1] == rc0, (f, cc[f], n, rc0)
        assert (kk[f, :n] == k0.view(np.uint32).reshape(n, 7)).all() and (dd[f, :n] == d0).all(), "batch form, frame %d" % f
    assert gex.debug_fault() is None  # no kernel met a count outside the capacity it indexes
    gex.close()


@pytest.mark.gpu
def test_fuzz_parity_campaign():
    """tools/fuzz_parity.py, 120 cases: images of very different statistics (flat with dots, noise, smoothed noise, gradients,
    checkerboards, saturated blobs, texture at the thresholds, the bench worlds) x sizes x constructor parameters, single call and batch."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "120", "7"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "all equal" in p.stdout and "MISMATCH" not in p.stdout, (p.stdout[-2000:], p.stderr[-1500:])


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args", [("fuzz_stereo.py", ["40", "5"]), ("fuzz_matcher.py", ["150", "5"])])
def test_fuzz_campaigns_of_the_next_rows(tool, args):
    """tools/fuzz_stereo.py (ComputeStereoMatches on random image pairs with layered disparities) and tools/fuzz_matcher.py (the
    brute-force SearchByBoW pass on tie-heavy descriptor sets), short runs; the long runs are on record in DESIGN.md."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", tool)] + args, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "all equal" in p.stdout and "MISMATCH" not in p.stdout, (p.stdout[-2000:], p.stderr[-1500:])
