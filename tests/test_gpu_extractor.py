"""GPU parity tests (run with -m gpu on an MI355X): HIP path through the C ABI vs the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_ex():
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    yield ex
    ex.close()


def _unpack(xs, ys, rs):
    return list(zip(xs.tolist(), ys.tolist(), rs.tolist()))


def assert_same_output(ref, got, what=""):
    rc0, k0, d0 = ref
    rc1, k1, d1 = got
    assert rc0 == rc1, (what, rc0, rc1)
    assert len(k0) == len(k1), (what, len(k0), len(k1))
    for f in k0.dtype.names:
        a, b = k0[f], k1[f]
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), (what, f, int((a != b).sum()))
    assert (d0 == d1).all(), (what, "descriptor bits differing", int(np.unpackbits(d0 ^ d1).sum()))


def test_stagewise_vga(oracle, gpu_ex, vga_frame):
    """Config 2 (640x480, 8 levels, 1000 features): every stage bit-exact vs the oracle."""
    oex = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    ref = oex(vga_frame, (0, 0))
    got = gpu_ex(vga_frame, None, (0, 0))
    for l in range(8):
        assert (gpu_ex.pyramid_level(l) == oex.pyramid_level(l)).all(), "pyramid level %d" % l
    for l in range(8):
        assert (gpu_ex.debug_blurred_level(l) == oex.blurred_level(l)).all(), "blur level %d" % l
    for l in range(8):
        assert _unpack(*gpu_ex.debug_level_candidates(l)) == _unpack(*oex.level_candidates(l)), "FAST candidates level %d" % l
    for l in range(8):
        kx, ky, kr = gpu_ex.debug_level_keypoints(l)
        ok = oex.level_keypoints(l)
        assert (kx + 16).tolist() == ok["x"].astype(int).tolist() and (ky + 16).tolist() == ok["y"].astype(int).tolist(), "octree level %d" % l
        assert kr.tolist() == ok["response"].astype(int).tolist()
    assert_same_output(ref, got, "vga natural order")
    assert len(got[1]) > 900


@pytest.mark.parametrize("lap", [(0, 0), (0, 1000), (200, 400)])
def test_lapping_order(oracle, gpu_ex, vga_frame, lap):
    oex = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    assert_same_output(oex(vga_frame, lap), gpu_ex(vga_frame, None, lap), "lap %s" % (lap,))


@pytest.mark.parametrize(
    "h,w,nf",
    [(480, 752, 1000), (350, 600, 1000), (376, 1241, 2000), (480, 752, 5000), (240, 320, 500), (1080, 1920, 1000)],
)
def test_other_configs(oracle, h, w, nf):
    from orbslam3_cpp_fork_amd import synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    img = synth.make_frame(h, w, camera_idx=3, frame_idx=w % 17)
    oex = oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
    gex = ORBextractor(nf, 1.2, 8, 20, 7)
    lap = (0, 1000) if nf != 2000 else (0, 0)
    assert_same_output(oex(img, lap), gex(img, None, lap), "%dx%d/%d" % (w, h, nf))
    gex.close()


def test_edge_inputs(oracle, gpu_ex):
    from orbslam3_cpp_fork_amd import synth

    # empty image -> -1 (src/ORBextractor.cc:1060-1062)
    assert gpu_ex(np.zeros((0, 0), np.uint8))[0] == -1
    # flat image: no corners at either threshold
    rc, k, d = gpu_ex(np.full((480, 640), 128, np.uint8), None, (0, 0))
    assert rc == 0 and len(k) == 0
    # low-contrast image: exercises the minThFAST retry in most cells
    img = (synth.make_frame(480, 640, 9, 9).astype(np.int32) // 6 + 100).astype(np.uint8)
    oex = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    assert_same_output(oex(img, (0, 0)), gpu_ex(img, None, (0, 0)), "low contrast")
    # noise image: thousands of candidates per level, quadtree under pressure
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    assert_same_output(oex(img, (0, 0)), gpu_ex(img, None, (0, 0)), "noise")
    # non-contiguous rows (step > cols)
    big = synth.make_frame(500, 700, 1, 2)
    view = big[10:490, 30:670]
    assert_same_output(oex(view, (0, 0)), gpu_ex(view, None, (0, 0)), "strided view")


def test_octree_device_random(oracle, gpu_ex):
    rng = np.random.default_rng(123)
    for trial in range(40):
        W, H = int(rng.integers(100, 1300)), int(rng.integers(100, 700))
        if round(np.float32(W) / np.float32(H)) < 1:
            continue
        n = int(rng.integers(0, 4000))
        N = int(rng.choice([1, 17, 60, 217, 434, 1086]))
        px, py = rng.integers(3, W - 3, n), rng.integers(3, H - 3, n)
        key = rng.permutation(np.unique(py.astype(np.int64) * 4096 + px))
        px, py = (key % 4096).astype(np.int32), (key // 4096).astype(np.int32)
        rs = rng.integers(7, 40, len(px)).astype(np.int32)
        n1, idx = oracle.distribute_octtree(px, py, rs, 16, 16 + W, 16, 16 + H, N)
        out = gpu_ex.debug_octree_device(px, py, rs, W, H, N)
        exp = px[idx].astype(np.uint32) | (py[idx].astype(np.uint32) << 12) | (rs[idx].astype(np.uint32) << 24)
        assert len(out) == n1 and (out == exp).all(), trial


def test_octree_device_sparse_levels(oracle, gpu_ex):
    """Levels with barely more candidates than wanted corners (n in [N / 2, 1.3 N]): many ordered passes, long runs of equal and
    descending sort keys -- the regime of tests/golden/octree_depth_limit_case.npz, random this time."""
    rng = np.random.default_rng(321)
    checked = 0
    for trial in range(60):
        W, H = int(rng.integers(300, 1300)), int(rng.integers(90, 400))
        if round(np.float32(W) / np.float32(H)) < 1:
            continue
        N = int(rng.choice([60, 123, 217, 434, 900]))
        n = int(N * rng.uniform(0.5, 1.3))
        if trial % 3 == 0:  # clustered: several points per node for a long time
            cx, cy = rng.integers(20, W - 20, 12), rng.integers(10, H - 10, 12)
            k = rng.integers(0, 12, n)
            px = np.clip(cx[k] + rng.normal(0, W / 40, n), 3, W - 4).astype(int)
            py = np.clip(cy[k] + rng.normal(0, H / 20, n), 3, H - 4).astype(int)
        else:
            px, py = rng.integers(3, W - 3, n), rng.integers(3, H - 3, n)
        key = np.unique(py.astype(np.int64) * 4096 + px)  # feed order = row-major, as the extractor feeds it
        px, py = (key % 4096).astype(np.int32), (key // 4096).astype(np.int32)
        rs = rng.integers(7, 60, len(px)).astype(np.int32)
        n1, idx = oracle.distribute_octtree(px, py, rs, 16, 16 + W, 16, 16 + H, N)
        out = gpu_ex.debug_octree_device(px, py, rs, W, H, N)
        exp = px[idx].astype(np.uint32) | (py[idx].astype(np.uint32) << 12) | (rs[idx].astype(np.uint32) << 24)
        assert len(out) == n1 and (out == exp).all(), (trial, W, H, len(px), N)
        checked += 1
    assert checked > 40


def test_wave_sort_is_std_sort(oracle, gpu_ex):
    """The wave-parallel std::sort re-enactment (ot_wave_sort) against the host build of the library algorithm (pinned against the real
    std::sort in tests/test_abi_and_hostmodel.py): every record in the same place, for tie-heavy, descending and median-of-3-hostile
    inputs, and with depth limits 0..3 that force the heap-sort path on every range."""
    import ctypes as C

    from orbslam3_cpp_fork_amd import _lib
    from tests.sort_patterns import sort_patterns

    L = _lib.lib()
    checked = 0
    for rec in sort_patterns(np.random.default_rng(5)):
        for depth in ((-1, 0, 1, 2, 3) if len(rec) <= 257 else (-1,)):
            ref, got = rec.copy(), rec.copy()
            L.orbx_debug_std_sort(ref.ctypes.data_as(C.c_void_p), len(ref), depth)
            assert L.orbx_debug_wave_sort(gpu_ex._h, got.ctypes.data_as(C.c_void_p), len(got), depth) == 0
            assert (ref == got).all(), (len(rec), depth, np.nonzero(ref != got)[0][:8].tolist())
            if depth == -1:
                assert (oracle.std_sort(rec) == got).all()
            checked += 1
    assert checked > 500


def test_octree_depth_limit_case_device(oracle, gpu_ex):
    """tests/golden/octree_depth_limit_case.npz on the device (stand-alone quadtree), twice: = the oracle, and the same both times."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "octree_depth_limit_case.npz"))
    xs, ys, rs = (d["cand"][i].astype(np.int32) for i in range(3))
    W, H, N = int(d["W"]), int(d["H"]), int(d["N"])
    n1, idx = oracle.distribute_octtree(xs, ys, rs, 16, 16 + W, 16, 16 + H, N)
    exp = xs[idx].astype(np.uint32) | (ys[idx].astype(np.uint32) << 12) | (rs[idx].astype(np.uint32) << 24)
    for _ in range(2):
        out = gpu_ex.debug_octree_device(xs, ys, rs, W, H, N)
        assert len(out) == n1 and (out == exp).all()


def test_sparse_level_frame(oracle):
    """The whole frame that fixture comes from (left eye of pair 67 of the stereo bench: level 0 has barely more candidates than wanted
    corners), single and inside a batch, against the oracle."""
    import torch

    from orbslam3_cpp_fork_amd import synth
    from orbslam3_cpp_fork_amd._lib import KP_DTYPE
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    rows, cols = 376, 1241
    frames = np.stack([synth.make_stereo_pair(rows, cols, camera_idx=i % 8, frame_idx=i // 8)[0] for i in (66, 67, 68)])
    oex = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    ex = ORBextractor(2000)
    ref = oex(frames[1], (0, 0))
    assert_same_output(ref, ex(frames[1], None, (0, 0)), "sparse level, single frame")
    cap = ex.max_keypoints(rows, cols)
    k = torch.zeros((3, cap, 7), dtype=torch.float32, device="cuda")
    ds = torch.zeros((3, cap, 32), dtype=torch.uint8, device="cuda")
    c = torch.zeros((3, 2), dtype=torch.int32, device="cuda")
    ex.extract_batch_device(torch.from_numpy(frames).cuda(), 3, rows, cols, cols, rows * cols, k, ds, c, cap, (0, 0))
    torch.cuda.synchronize()
    n = int(c[1, 0])
    got = (int(c[1, 1]), k[1, :n].cpu().numpy().view(KP_DTYPE).reshape(-1), ds[1, :n].cpu().numpy())
    assert_same_output(ref, got, "sparse level, inside a batch")
    assert ex.debug_fault() is None
    ex.close()


def test_batch_device_matches_single(oracle, gpu_ex):
    """The device-resident batch entry point yields, per frame, exactly the single-frame result."""
    import torch

    from orbslam3_cpp_fork_amd import synth
    from orbslam3_cpp_fork_amd._lib import KP_DTYPE

    B, h, w = 5, 480, 640
    frames = np.stack([synth.make_frame(h, w, camera_idx=c) for c in range(B)])
    d_imgs = torch.from_numpy(frames).cuda()
    cap = gpu_ex.max_keypoints(h, w)
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        gpu_ex.extract_batch_device(d_imgs, B, h, w, w, h * w, d_kps, d_desc, d_cnt, cap, (0, 1000), stream=ts.cuda_stream)
    torch.cuda.synchronize()
    cnt = d_cnt.cpu().numpy()
    oex = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    for f in range(B):
        n, mono = int(cnt[f, 0]), int(cnt[f, 1])
        k = d_kps[f, :n].cpu().numpy().view(KP_DTYPE).reshape(-1)
        d = d_desc[f, :n].cpu().numpy()
        assert_same_output(oex(frames[f], (0, 1000)), (mono, k, d), "batch frame %d" % f)


@pytest.mark.parametrize("repitch", [True, False])
def test_unaligned_batch(oracle, repitch, monkeypatch):
    """Frames with an odd width and a tight step (KITTI's 1241): the batch entry repitches them on the device; with
    ORBX_NO_REPITCH=1 the byte-load variants of FAST / blur / resize take level 0 directly.  Same results either way."""
    import torch

    from orbslam3_cpp_fork_amd import synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    if not repitch:
        monkeypatch.setenv("ORBX_NO_REPITCH", "1")
    B, rows, cols = 2, 241, 397
    fr = np.stack([synth.make_frame(rows, cols, 3, i) for i in range(B)])
    ex = ORBextractor(600)
    cap = ex.max_keypoints(rows, cols)
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(fr).to(dev)
    kps = torch.zeros((B, cap, 7), device=dev)
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    cnt = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    oex = oracle.OracleExtractor(600, 1.2, 8, 20, 7)
    for f in range(B):
        rc, k, dd = oex(fr[f], (0, 1000))
        n = int(cnt[f, 0])
        assert n == len(k) and int(cnt[f, 1]) == rc
        got = np.frombuffer(kps[f, :n].cpu().numpy().tobytes(), oracle.KP_DTYPE)
        assert got.tobytes() == k.tobytes() and (desc[f, :n].cpu().numpy() == dd).all()
    ex.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [True, False])
def test_streaming_abi_chunks_in_flight(oracle, pinned):
    """orbx_extract_async / orbx_wait: three chunks in flight, pinned (DMA direct) and pageable (staged) host memory, results equal
    the oracle's frame by frame; the device-side records of a ticket equal what lands on the host."""
    import torch

    from orbslam3_cpp_fork_amd import synth
    from orbslam3_cpp_fork_amd._lib import KP_DTYPE, OrbxError
    from orbslam3_cpp_fork_amd.extractor import ORBextractor, pinned_array

    h, w, nf, B, nchunks = 240, 320, 500, 3, 4
    frames = synth.make_stream(h, w, B * nchunks, camera_idx=3)
    ex = ORBextractor(nf, 1.2, 8, 20, 7)
    oex = oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(h, w)
    alloc = pinned_array if pinned else (lambda shape, dt: np.zeros(shape, dt))
    imgs = [alloc((B, h, w), np.uint8) for _ in range(nchunks)]
    for c in range(nchunks):
        imgs[c][:] = frames[c * B: (c + 1) * B]
    outs = [(alloc((B, cap), KP_DTYPE), alloc((B, cap, 32), np.uint8), alloc((B, 2), np.int32)) for _ in range(nchunks)]
    ex.set_async_depth(3)
    tickets = [ex.extract_async(imgs[c], *outs[c], cap, (0, 1000)) for c in range(3)]
    with pytest.raises(OrbxError, match="all slots in flight"):
        ex.extract_async(imgs[3], *outs[3], cap, (0, 1000))
    # device-side view of ticket 1 (what a chained matcher would read)
    dk, dd, dc, st = ex.async_device_buffers(tickets[1])
    ex.wait(tickets[1])
    dev_cnt = np.zeros((B, 2), np.int32)
    torch.cuda.synchronize()
    import ctypes as C

    from orbslam3_cpp_fork_amd import _lib

    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(dev_cnt.ctypes.data_as(C.c_void_p), C.c_void_p(dc), dev_cnt.nbytes, 2) == 0
    assert (dev_cnt == outs[1][2]).all()
    ex.wait(tickets[0])
    tickets.append(ex.extract_async(imgs[3], *outs[3], cap, (0, 1000)))  # slot of ticket 0 is free again
    ex.wait(tickets[2])
    ex.wait(tickets[3])
    with pytest.raises(OrbxError, match="unknown ticket"):
        ex.wait(tickets[3])
    for c in range(nchunks):
        k, d, cnt = outs[c]
        for f in range(B):
            rc0, k0, d0 = oex(frames[c * B + f], (0, 1000))
            n = int(cnt[f, 0])
            assert n == len(k0) and int(cnt[f, 1]) == rc0
            assert (k[f, :n].view(np.uint32) == k0.view(np.uint32)).all() and (d[f, :n] == d0).all()


@pytest.mark.gpu
def test_min_threshold_above_ini_threshold(oracle, vga_frame):
    """The reference never checks minThFAST <= iniThFAST (src/ORBextractor.cc:403-410): the retry then uses the HIGHER threshold."""
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    ex = ORBextractor(1000, 1.2, 8, 12, 30)
    oex = oracle.OracleExtractor(1000, 1.2, 8, 12, 30)
    rc, k, d = ex(vga_frame, None, (0, 0))
    rc0, k0, d0 = oex(vga_frame, (0, 0))
    assert rc == rc0 and len(k) == len(k0) and (k.view(np.uint32) == k0.view(np.uint32)).all() and (d == d0).all()


@pytest.mark.gpu
def test_gather_through_the_c_abi_single_rank():
    """orbx_comm_create / orbx_gather (csrc/orbx_comm.hip) with one rank: RCCL is opened, the communicator is built, and the root's
    slice of the gathered arrays is the rank's own records.  (More ranks need more GPUs than a test box has: RCCL refuses two
    ranks on one device.  The N-rank exchange is covered on CPU by tests/test_shard_gloo.py through the torch.distributed form.)"""
    import torch

    from orbslam3_cpp_fork_amd import shard, synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    B, rows, cols = 3, 240, 320
    fr = synth.make_stream(rows, cols, B, camera_idx=2)
    ex = ORBextractor(500)
    cap = ex.max_keypoints(rows, cols)
    d = torch.from_numpy(fr).cuda()
    kps = torch.zeros((B, cap, 7), device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=st)
    g = shard.CommGather(B, cap, 0, 1, 0)
    ak, ad, ac = g(kps, desc, cnt, stream=st)
    torch.cuda.synchronize()
    assert int(cnt[:, 0].min()) > 100
    assert torch.equal(ac[0], cnt) and torch.equal(ad[0], desc) and torch.equal(ak[0].view(torch.int32), kps.view(torch.int32))
    g.close()


@pytest.mark.gpu
def test_graph_replay_of_the_single_call(oracle):
    """orbx_set_graph_replay: the third and later calls of one shape replay a captured hipGraph; results stay those of the eager call for
    changing image contents, and a change of shape or lapping area falls back to eager and records again."""
    from orbslam3_cpp_fork_amd import synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    ex = ORBextractor(500)
    ref = ORBextractor(500)
    ex.set_graph_replay(True)
    ref.set_graph_replay(False)  # the eager sequence
    frames = synth.make_stream(240, 320, 6, camera_idx=1)
    for i, fr in enumerate(frames):  # call 0 eager, call 1 records, 2.. replay
        a, b = ex(fr, None, (0, 1000)), ref(fr, None, (0, 1000))
        assert a[0] == b[0] and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes(), i
    big = synth.make_frame(376, 620, 2, 3)
    for lap in ((0, 1000), (0, 1000), (0, 1000), (100, 300), (100, 300), (100, 300)):
        a, b = ex(big, None, lap), ref(big, None, lap)
        assert a[0] == b[0] and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes(), lap
    for fr in frames[:3]:
        a, b = ex(fr, None, (0, 1000)), ref(fr, None, (0, 1000))
        assert a[1].tobytes() == b[1].tobytes()
    # a batch call of another size on the same handle in between (workspace re-laid out, buffers reallocated): the recorded graph of the
    # small shape must not be replayed on freed memory
    import torch

    B, rows, cols = 4, 480, 640
    d = torch.from_numpy(synth.make_stream(rows, cols, B, camera_idx=3)).cuda()
    cap = ex.max_keypoints(rows, cols)
    kps = torch.zeros((B, cap, 7), device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    for rep in range(2):
        for fr in frames[:3]:
            a, b = ex(fr, None, (0, 1000)), ref(fr, None, (0, 1000))
            assert a[0] == b[0] and a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes()
        ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 1000), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    ref.set_graph_replay(False)
    a, b = ex(frames[0], None, (0, 1000)), ref(frames[0], None, (0, 1000))
    assert a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes()


@pytest.mark.gpu
def test_replayed_call_owns_the_pyramid_accessors(oracle):
    """A replayed orbx_extract must leave the handle's host-side state as an eager call does: after a batch call on the SAME handle and the
    same workspace (no reallocation, so the recorded graph stays valid) level 0 of orbx_pyramid_level -- the adaptor's mvImagePyramid[0], and
    the level-0 SAD source of orbx_stereo_matches -- is the replayed call's own image again, not the batch's caller-owned frames."""
    import torch

    from orbslam3_cpp_fork_amd import synth
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    rows, cols, B = 240, 320, 4
    frames = synth.make_stream(rows, cols, 6, camera_idx=4)
    ex = ORBextractor(500)
    oex = oracle.OracleExtractor(500, 1.2, 8, 20, 7)
    d = torch.from_numpy(synth.make_stream(rows, cols, B, camera_idx=9)).cuda()
    cap = ex.max_keypoints(rows, cols)
    kps = torch.zeros((B, cap, 7), device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 0), stream=st)  # workspace for B frames exists from here on
    torch.cuda.synchronize()
    for download in (False, True):
        ex.set_pyramid_download(download)
        for i, fr in enumerate(frames):  # eager, record, replay, ... with a batch call between every two
            rc, k, dd = ex(fr, None, (0, 0))
            rc0, k0, d0 = oex(fr, (0, 0))
            assert rc == rc0 and (k.view(np.uint32) == k0.view(np.uint32)).all() and (dd == d0).all()
            assert (ex.pyramid_level(0) == fr).all(), ("level 0 after call %d is not the call's image" % i)
            for l in (1, 4, 7):
                assert (ex.pyramid_level(l) == oex.pyramid_level(l)).all(), (download, i, l)
            ex.extract_batch_device(d, B, rows, cols, cols, rows * cols, kps, desc, cnt, cap, (0, 0), stream=st)
            torch.cuda.synchronize()
            assert (ex.pyramid_level(0, frame=2) == d[2].cpu().numpy()).all()  # and the batch call owns them after itself


@pytest.mark.gpu
def test_async_tickets_waited_in_any_order(oracle):
    """include/orbx.h promises that tickets may be waited for in any order: waiting for the NEWEST first must free a slot for the next
    submission (the old ring took slot ticket % depth and refused while a slot was free)."""
    from orbslam3_cpp_fork_amd import synth
    from orbslam3_cpp_fork_amd._lib import KP_DTYPE
    from orbslam3_cpp_fork_amd.extractor import ORBextractor, pinned_array

    h, w, nf, B, nchunks = 240, 320, 500, 2, 6
    frames = synth.make_stream(h, w, B * nchunks, camera_idx=8)
    ex = ORBextractor(nf)
    oex = oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
    cap = ex.max_keypoints(h, w)
    imgs = [pinned_array((B, h, w), np.uint8) for _ in range(nchunks)]
    for c in range(nchunks):
        imgs[c][:] = frames[c * B: (c + 1) * B]
    outs = [(pinned_array((B, cap), KP_DTYPE), pinned_array((B, cap, 32), np.uint8), pinned_array((B, 2), np.int32)) for _ in range(nchunks)]
    ex.set_async_depth(3)
    t = [ex.extract_async(imgs[c], *outs[c], cap, (0, 0)) for c in range(3)]
    ex.wait(t[2])                                              # newest first
    t.append(ex.extract_async(imgs[3], *outs[3], cap, (0, 0)))  # must find the freed slot
    ex.wait(t[1])
    t.append(ex.extract_async(imgs[4], *outs[4], cap, (0, 0)))
    ex.wait(t[4])
    ex.wait(t[0])
    t.append(ex.extract_async(imgs[5], *outs[5], cap, (0, 0)))
    ex.wait(t[3])
    ex.wait(t[5])
    for c in range(nchunks):
        k, d, cnt = outs[c]
        for f in range(B):
            rc0, k0, d0 = oex(frames[c * B + f], (0, 0))
            n = int(cnt[f, 0])
            assert n == len(k0) and (k[f, :n].view(np.uint32) == k0.view(np.uint32)).all() and (d[f, :n] == d0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cap_entries", ["2", "64", "300"])
def test_fast_candidate_list_overflow(oracle, vga_frame, cap_entries, monkeypatch):
    """ORBX_FAST_CAND_CAP shrinks the per-tile candidate list of k_fast_tiles: lanes whose pre-test survivors do not fit keep them and score
    them themselves.  Results must not depend on the list size (the default 2048 never overflows on natural images, so this path would
    otherwise only be reached by adversarial inputs)."""
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    monkeypatch.setenv("ORBX_FAST_CAND_CAP", cap_entries)
    ex = ORBextractor(1000)
    oex = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    noise = np.random.default_rng(5).integers(0, 256, vga_frame.shape, dtype=np.uint8)  # every other pixel survives the pre-test
    for img in (vga_frame, noise):
        rc, k, d = ex(img, None, (0, 0))
        rc0, k0, d0 = oex(img, (0, 0))
        assert rc == rc0 and len(k) == len(k0) and (k.view(np.uint32) == k0.view(np.uint32)).all() and (d == d0).all()


@pytest.mark.gpu
def test_graph_env_switch(oracle, vga_frame, monkeypatch):
    """ORBX_GRAPH=0 keeps every orbx_extract eager (the operator's switch beside orbx_set_graph_replay); results are the same."""
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    monkeypatch.setenv("ORBX_GRAPH", "0")
    ex = ORBextractor(1000)
    rc0, k0, d0 = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)(vga_frame, (0, 1000))
    for _ in range(4):
        rc, k, d = ex(vga_frame, None, (0, 1000))
        assert rc == rc0 and (k.view(np.uint32) == k0.view(np.uint32)).all() and (d == d0).all()
