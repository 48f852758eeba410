"""CPU tests: the C-ABI library loads and exports every symbol include/orbx.h declares (no compute calls without
a GPU), and the host builds of the shared host/device cores agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    from orbslam3_cpp_fork_amd import _lib

    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "orbx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(orb[xmv]_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(L, n), "liborbx.so does not export %s" % n
    # the ctypes table covers the header exactly
    assert names == set(_lib.SIGNATURES), names.symmetric_difference(_lib.SIGNATURES)


def test_keypoint_layout():
    from orbslam3_cpp_fork_amd._lib import KP_DTYPE

    assert KP_DTYPE.itemsize == 28  # cv::KeyPoint
    assert [KP_DTYPE.fields[n][1] for n in KP_DTYPE.names] == [0, 4, 8, 12, 16, 20, 24]


def test_library_shares_one_hip_runtime():
    """liborbx.so must resolve the HIP runtime by the plain name so that it shares torch's copy in-process."""
    import subprocess

    from orbslam3_cpp_fork_amd import build

    out = subprocess.check_output(["readelf", "-d", build.OUT], text=True)
    assert "libamdhip64.so]" in out and "libamdhip64.so.7" not in out
    assert "RPATH" in out and "RUNPATH" not in out


def test_host_paths_that_need_no_gpu():
    from orbslam3_cpp_fork_amd.matcher import ORBmatcher

    z, o = np.zeros(32, np.uint8), np.full(32, 255, np.uint8)
    assert ORBmatcher.DescriptorDistance(z, o) == 256
    assert ORBmatcher.ComputeThreeMaxima(np.array([5, 100, 9, 50, 20] + [0] * 25)) == (1, 3, 4)
    assert (ORBmatcher.TH_HIGH, ORBmatcher.TH_LOW, ORBmatcher.HISTO_LENGTH) == (100, 50, 30)


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly (-3 / exception), never fall back to the oracle."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from orbslam3_cpp_fork_amd._lib import OrbxError
    from orbslam3_cpp_fork_amd.extractor import ORBextractor

    with pytest.raises(OrbxError):
        ORBextractor(1000)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "orbslam3_cpp_fork_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in txt and "orb_oracle" not in txt and "liborb_oracle" not in txt, f


def test_octree_host_model_matches_oracle(oracle):
    """octree_core.h (the source the kernel compiles) run single-threaded on the host vs the std::list/std::sort oracle."""
    from orbslam3_cpp_fork_amd.extractor import octree_host_model

    rng = np.random.default_rng(0)
    checked = 0
    for trial in range(600):
        W, H = int(rng.integers(40, 1300)), int(rng.integers(40, 700))
        if round(np.float32(W) / np.float32(H)) < 1:
            continue
        n = int(rng.integers(0, 3000)) if trial % 3 else int(rng.integers(0, 40))
        N = int(rng.choice([0, 1, 3, 17, 60, 87, 217, 434, 1086]))
        if trial % 4 == 0:
            px = np.clip(rng.normal(W / 2, W / 10, n), 3, W - 4).astype(int)
            py = np.clip(rng.normal(H / 2, H / 10, n), 3, H - 4).astype(int)
        else:
            px, py = rng.integers(3, max(4, W - 3), n), rng.integers(3, max(4, H - 3), n)
        key = rng.permutation(np.unique(py.astype(np.int64) * 4096 + px))
        px, py = (key % 4096).astype(np.int32), (key // 4096).astype(np.int32)
        rs = (rng.integers(7, 40, len(px)) if trial % 2 else rng.integers(7, 255, len(px))).astype(np.int32)
        n1, idx = oracle.distribute_octtree(px, py, rs, 16, 16 + W, 16, 16 + H, N)
        rc, out = octree_host_model(px, py, rs, W, H, N)
        exp = (px[idx].astype(np.uint32) | (py[idx].astype(np.uint32) << 12) | (rs[idx].astype(np.uint32) << 24)) if n1 > 0 else np.zeros(0, np.uint32)
        assert rc == 0 and len(out) == n1 and (out == exp).all(), (trial, W, H, len(px), N)
        checked += 1
    assert checked > 400


def test_octree_on_real_candidates(oracle, vga_frame):
    from orbslam3_cpp_fork_amd.extractor import octree_host_model

    oex = oracle.OracleExtractor(1000)
    oex(vga_frame, (0, 0))
    quota = oex.features_per_level()
    for l in range(8):
        xs, ys, rs = oex.level_candidates(l)
        lv = oex.pyramid_level(l)
        W, H = lv.shape[1] - 32, lv.shape[0] - 32
        rc, out = octree_host_model(xs, ys, rs, W, H, int(quota[l]))
        kp = oex.level_keypoints(l)
        assert rc == 0 and ((out & 0xFFF) + 16).tolist() == kp["x"].astype(int).tolist()
        assert (((out >> 12) & 0xFFF) + 16).tolist() == kp["y"].astype(int).tolist()


def test_std_sort_reenactment():
    """ot_std_sort vs the C++ library's std::sort on tie-heavy records, via numpy's knowledge of nothing:
    the oracle's DistributeOctTree calls the real std::sort, so the tie order is pinned by test_octree_*;
    here the basic contract (sorted by key, payload permutation) and the heap path are checked."""
    from orbslam3_cpp_fork_amd import _lib

    L = _lib.lib()
    rng = np.random.default_rng(3)
    for depth in (-1, 0, 1):
        for n in (0, 1, 5, 16, 17, 100, 1000):
            keys = rng.integers(0, 6, n).astype(np.uint64)
            rec = ((keys << np.uint64(16)) | np.arange(n, dtype=np.uint64)).astype(np.uint64)
            buf = rec.copy()
            L.orbx_debug_std_sort(buf.ctypes.data_as(C.c_void_p), n, depth)
            assert (np.diff((buf >> np.uint64(16)).astype(np.int64)) >= 0).all()
            assert sorted(buf.tolist()) == sorted(rec.tolist())


def test_std_sort_reenactment_is_libstdcxx(oracle):
    """ot_std_sort (host build of the re-enactment the kernel follows) against the real std::sort (oracle.std_sort) record for record --
    equal keys included, and on inputs that drive introsort into its depth limit (the heap-sort path)."""
    from orbslam3_cpp_fork_amd import _lib

    L = _lib.lib()
    from tests.sort_patterns import sort_patterns

    for rec in sort_patterns(np.random.default_rng(11)):
        mine = rec.copy()
        L.orbx_debug_std_sort(mine.ctypes.data_as(C.c_void_p), len(mine), -1)
        assert (mine == oracle.std_sort(rec)).all(), (len(rec), rec[:8].tolist())


def test_octree_host_model_sparse_levels(oracle):
    """The same comparison where a level has barely more candidates than wanted corners (row-major feed, as the extractor feeds it)."""
    from orbslam3_cpp_fork_amd.extractor import octree_host_model

    rng = np.random.default_rng(4)
    for trial in range(300):
        W, H = int(rng.integers(300, 1300)), int(rng.integers(90, 400))
        if round(np.float32(W) / np.float32(H)) < 1:
            continue
        N = int(rng.choice([17, 60, 123, 217, 434, 900]))
        n = int(N * rng.uniform(0.5, 1.3))
        px, py = rng.integers(3, W - 3, n), rng.integers(3, H - 3, n)
        key = np.unique(py.astype(np.int64) * 4096 + px)
        px, py = (key % 4096).astype(np.int32), (key // 4096).astype(np.int32)
        rs = rng.integers(7, 60, len(px)).astype(np.int32)
        n1, idx = oracle.distribute_octtree(px, py, rs, 16, 16 + W, 16, 16 + H, N)
        rc, out = octree_host_model(px, py, rs, W, H, N)
        exp = px[idx].astype(np.uint32) | (py[idx].astype(np.uint32) << 12) | (rs[idx].astype(np.uint32) << 24)
        assert rc == 0 and len(out) == n1 and (out == exp).all(), (trial, W, H, len(px), N)


def test_octree_depth_limit_case(oracle):
    """Level 0 of a KITTI-sized frame with 456 candidates for 434 wanted corners (golden fixture: the frame of the stereo bench the
    device quadtree once got wrong): its third ordered pass sorts 57 nodes whose keys run mostly downhill, and libstdc++'s introsort
    hits its depth limit on them.  Host model = oracle here; the device form is checked in tests/test_gpu_extractor.py."""
    from orbslam3_cpp_fork_amd.extractor import octree_host_model

    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "octree_depth_limit_case.npz"))
    xs, ys, rs = (d["cand"][i].astype(np.int32) for i in range(3))
    W, H, N = int(d["W"]), int(d["H"]), int(d["N"])
    n1, idx = oracle.distribute_octtree(xs, ys, rs, 16, 16 + W, 16, 16 + H, N)
    rc, out = octree_host_model(xs, ys, rs, W, H, N)
    exp = xs[idx].astype(np.uint32) | (ys[idx].astype(np.uint32) << 12) | (rs[idx].astype(np.uint32) << 24)
    assert rc == 0 and n1 == 434 and (out == exp).all()


def test_sincos_port_exhaustive(oracle):
    """The glibc sinf/cosf port the kernel uses equals this host's libm for EVERY float angle in [0, 360] degrees
    (1,135,869,953 inputs), in both evaluation orders (plain and FMA-contracted)."""
    import struct

    hi = struct.unpack("<I", struct.pack("<f", 360.0))[0] + 1
    for variant in (0, 1):
        bad, first = oracle.sincos_port_mismatches(0, hi, variant, nthreads=os.cpu_count() or 4)
        assert bad == 0, (variant, bad, first[:4])
    # and the exported host build of the device function agrees on a sample
    from orbslam3_cpp_fork_amd import _lib

    deg = np.random.default_rng(1).uniform(0, 360, 100000).astype(np.float32)
    c0, s0 = oracle.libm_sincos_deg(deg)
    c1, s1 = np.zeros_like(deg), np.zeros_like(deg)
    _lib.lib().orbx_debug_sincos_deg(deg.ctypes.data_as(C.c_void_p), len(deg), c1.ctypes.data_as(C.c_void_p), s1.ctypes.data_as(C.c_void_p))
    assert (c0.view(np.uint32) == c1.view(np.uint32)).all() and (s0.view(np.uint32) == s1.view(np.uint32)).all()


def test_image_bounds_host(oracle):
    """orbx_image_bounds is host arithmetic over the core shared with the device kernel (undistort_core.h): checked against the
    oracle's separate restatement of cv::undistortPoints without a GPU."""
    from orbslam3_cpp_fork_amd.extractor import ComputeImageBounds

    for K, dist, (w, h) in (
        ([458.654, 457.296, 367.215, 248.375], [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05], (752, 480)),
        ([517.306408, 516.469215, 318.643040, 255.313989], [0.262383, -0.953104, -0.005358, 0.002628, 1.163314], (640, 480)),
        ([718.856, 718.856, 607.1928, 185.2157], [0.0, 0.0, 0.0, 0.0], (1241, 376)),
    ):
        assert list(ComputeImageBounds(w, h, K, dist)) == oracle.image_bounds(w, h, K, dist).tolist()


def test_bad_arguments_are_rejected_before_the_device():
    """Argument errors come back as -2 without touching the GPU (same convention as the extractor: -1 empty, -2 bad args, -3 HIP)."""
    import ctypes as C

    from orbslam3_cpp_fork_amd import _lib

    L = _lib.lib()
    h = C.c_void_p()
    one = np.zeros(4, np.float32)
    assert L.orbv_create(None, 10, 6, 0, None, None, None, None, 0) == -2
    assert L.orbv_create(C.byref(h), 10, 6, 3, None, None, None, None, 0) == -2  # arrays missing
    assert L.orbv_load_text(C.byref(h), None, 0) == -2
    assert L.orbv_transform(None, None, 0, 4, None, None, None, None, None, None, None) == -2
    assert L.orbx_image_bounds(640, 480, None, None, 0, _lib.ptr(one)) == -2
    assert L.orbx_image_bounds(640, 480, _lib.ptr(one), None, 3, _lib.ptr(one)) == -2  # coefficients promised but missing
    assert L.orbx_set_input_resize(None, 600, 350) == -2
    # the gather of SURVEY 8(e): argument errors are caught before RCCL is even opened
    assert L.orbx_comm_unique_id(None) == -2
    assert L.orbx_comm_create(None, 0, 0, 1, None) == -2
    assert L.orbx_comm_create(C.byref(h), 0, 2, 2, _lib.ptr(one)) == -2  # rank outside [0, nranks)
    assert L.orbx_gather(None, None, None, None, 1, 16, None, None, None, 0, None) == -2
    assert L.orbx_comm_rank(None) == -1 and L.orbx_comm_size(None) == -1
    assert L.orbx_undistort_keypoints(None, None, 0, None, None, 0, None) == -2
    assert L.orbm_search_by_projection(None, None, None, None, None, None, None, 0, None, None, None, None, 0, None, 8, 15.0, 0, 0.0, 0.0, 1.0, 0.0, 1.0, 100, 1, None, None) == -2
    assert L.orbm_search_by_projection_map(None, None, None, None, None, None, None, 0, None, None, None, None, 0, None, 8, 0.8, 0.0, 1.0, 0.0, 1.0, 100, None, None) == -2
    assert L.orbm_match_groups_two_cameras(None, None, 0, None, 0, 0, 1, None, None, None, None, None, None, None, 50, 0.7, 0, None) == -2
    assert L.orbm_search_by_projection_map_two_cameras(None, None, 0, *([None] * 10), 0, 0, None, None, None, None, 8, 0.8, 0.0, 1.0, 0.0, 1.0, 100, None) == -2
    assert L.orbm_assign_features_to_grid_device(None, None, None, 1, 16, 0.0, 1.0, 0.0, 1.0, None, None, None, None, None) == -2
    assert L.orbm_search_in_windows(None, None, None, None, None, None, 0, None, None, None, None, 0, None, None, 8, 4.0, 0, 0.0, 1.0, 0.0, 1.0, 50, None) == -2
    assert L.orbm_search_in_windows(None, None, None, None, None, None, 0, None, None, None, None, 0, None, None, 8, 4.0, 2, 0.0, 1.0, 0.0, 1.0, 50, None) == -2  # claim is 0 / 1
    assert L.orbm_search_for_triangulation(None, None, None, None, None, 0, None, None, None, None, 0, 0, None, None, None, None, None, None, None, None, 8, 0, 0, 50, 1,
                                           None) == -2
    assert L.orbx_stereo_matches_batch_device(None, None, None, None, None, None, None, None, 1, 16, 0.5, 300.0, None, None, None, None) == -2
    # the message of a failed orbm_* / orbv_* call is readable through its own getter AND through orbx_last_error() (one slot per thread)
    assert L.orbm_search_in_windows(None, None, None, None, None, None, 0, None, None, None, None, 0, None, None, 8, 4.0, 2, 0.0, 1.0, 0.0, 1.0, 50, None) == -2
    msg = L.orbm_last_error().decode()
    assert "orbm_search_in_windows" in msg and L.orbx_last_error().decode() == msg
    assert L.orbv_create(C.byref(h), 10, 6, 3, None, None, None, None, 0) == -2
    assert L.orbv_last_error().decode() and L.orbx_last_error().decode() == L.orbv_last_error().decode()
    # streaming entry points: argument errors before any device work
    t = C.c_int(7)
    assert L.orbx_extract_async(None, None, 1, 480, 640, 640, 640 * 480, 0, 0, None, None, None, 0, C.byref(t)) == -2 and t.value == -1
    assert L.orbx_wait(None, 0) == -2 and L.orbx_set_async_depth(None, 2) == -2 and L.orbx_set_tap_contraction(None, 1) == -2
    # vocabulary text header checks happen on the host, before any allocation
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        f.write("30 6 0 0\n")  # k > 20 (TemplatedVocabulary.h:1358-1362)
    assert L.orbv_load_text(C.byref(h), f.name.encode(), 0) == -2
    os.unlink(f.name)


def test_rccl_that_cannot_be_opened_is_an_error_not_a_crash():
    """orbx_comm_unique_id / orbx_comm_create on a host where no RCCL can be opened return -3 with the loader's message
    (ORBX_RCCL_LIBRARY names the one library to try).  Own process: the library caches a successful open."""
    import subprocess
    import sys

    code = (
        "import ctypes as C, sys\n"
        "from orbslam3_cpp_fork_amd import _lib\n"
        "L = _lib.lib()\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = L.orbx_comm_unique_id(buf)\n"
        "msg = _lib.last_error()\n"
        "h = C.c_void_p()\n"
        "rc2 = L.orbx_comm_create(C.byref(h), 0, 0, 1, buf)\n"
        "print(rc, rc2, msg)\n"
        "sys.exit(0 if (rc == -3 and rc2 == -3 and 'cannot open' in msg and 'no_such_rccl' in msg) else 1)\n"
    )
    env = dict(os.environ, ORBX_RCCL_LIBRARY="/nonexistent/libno_such_rccl.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)


def test_env_switches_of_the_product_library_are_few_and_tested():
    """Every getenv("ORBX_...") the default build of liborbx.so reads is named in a test; tuning switches only exist under -DORBX_TUNING."""
    csrc = os.path.join(ROOT, "orbslam3_cpp_fork_amd", "csrc")
    live, tuning = set(), set()
    for f in os.listdir(csrc):
        if not f.endswith((".hip", ".h", ".cpp")):
            continue
        depth = 0
        for line in open(os.path.join(csrc, f)):
            t = line.strip()
            if t.startswith("#ifdef ORBX_TUNING"):
                depth += 1
            elif t.startswith("#endif") and depth:
                depth -= 1
            for name in re.findall(r'getenv\("(ORBX_[A-Z0-9_]+)"\)', line):
                (tuning if depth else live).add(name)
    assert live == {"ORBX_FAST_CAND_CAP", "ORBX_NO_REPITCH", "ORBX_GRAPH", "ORBX_RCCL_LIBRARY"}, live
    tests = "".join(open(os.path.join(ROOT, "tests", f)).read() for f in os.listdir(os.path.join(ROOT, "tests")) if f.endswith(".py"))
    for name in live:
        assert ('setenv("%s"' % name) in tests or ('%s=' % name) in tests, name
