"""ctypes binding of the CPU oracle (oracle/orb_oracle.h).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (orbslam3_cpp_fork_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborb_oracle.so")

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
assert KP_DTYPE.itemsize == 28


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    import hashlib

    # stale by content, not by mtime (the .so travels with the snapshot, and a copy may reorder timestamps)
    digest = hashlib.sha256(b"".join(open(os.path.join(_HERE, f), "rb").read() for f in ("orb_oracle.cpp", "orb_oracle.h", "Makefile"))).hexdigest()
    stamp = os.path.join(_HERE, "_build", "liborb_oracle.stamp")
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if force or not os.path.exists(_SO) or have != digest:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
        with open(stamp, "w") as f:
            f.write(digest)
    return _SO


_lib = None


def select_native():
    """Switch to a -march=native build of the same source, compiled on THIS host (bench.py's cpu_baseline leg).
    Must be called before the library is first used; returns False (and keeps the portable build) if that is no longer possible or
    the compile fails."""
    global _SO
    if _lib is not None:
        return False
    nat = os.path.join(_HERE, "_build", "liborb_oracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return False
    _SO = nat
    return True


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, i32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_float)
        L.orao_create.restype = C.c_void_p
        L.orao_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orao_destroy.argtypes = [C.c_void_p]
        L.orao_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, i32p]
        L.orao_get_levels.argtypes = [C.c_void_p]
        L.orao_get_scale_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orao_get_features_per_level.argtypes = [C.c_void_p, C.c_void_p]
        L.orao_get_umax.argtypes = [C.c_void_p, C.c_void_p]
        for f in (L.orao_pyramid_level, L.orao_blurred_level):
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, i32p, i32p]
        L.orao_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orao_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orao_resize_linear_8u.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        L.orao_gaussian_blur7_8u.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        L.orao_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orao_fast_atan2.restype = C.c_float
        L.orao_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orao_cv_round_f.argtypes = [C.c_float]
        L.orao_ic_angle.restype = C.c_float
        L.orao_ic_angle.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int]
        L.orao_orb_descriptor.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orao_orb_descriptor_c.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int]
        L.orao_set_tap_contraction.argtypes = [C.c_void_p, C.c_int]
        L.orao_distribute_octtree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 5 + [C.c_void_p, C.c_int]
        L.orao_libm_sincos_deg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orao_sincos_port_mismatches.restype = C.c_long
        L.orao_sincos_port_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, i32p]
        L.orao_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orao_hamming.argtypes = [C.c_void_p, C.c_void_p]
        L.orao_three_maxima.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p]
        L.orao_match_groups.argtypes = (
            [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        )
        L.orao_match_window.argtypes = (
            [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
            + [C.c_float] * 4
            + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        )
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8img(img):
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    if img.strides[1] != 1:
        img = np.ascontiguousarray(img)
    return img


class OracleExtractor:
    """Mirror of ORB_SLAM3::ORBextractor (include/ORBextractor.h:41-119) over the CPU oracle."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7, tap_contraction=1):
        self._h = lib().orao_create(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast)
        if not self._h:
            raise ValueError("bad extractor parameters")
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.set_tap_contraction(tap_contraction)

    def set_tap_contraction(self, mode):
        """1 = the reference as its CMake builds it on an FMA host (default), 0 = no contraction, 2 = the other fused form."""
        lib().orao_set_tap_contraction(self._h, int(mode))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orao_destroy(self._h)
            self._h = None

    def __call__(self, image, lapping=(0, 0), cap=None):
        """Returns (mono_index, keypoints[KP_DTYPE], descriptors[n,32] u8)."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        img = _u8img(image)
        # -2 is also what a too small record capacity returns: a very elongated image with a small quota starts every level with
        # round(width / height) root nodes and divides each of them once (up to 4 x roots keypoints per level), far more than nfeatures
        for c in ([cap] if cap else [(self.nfeatures + 8 * self.nlevels + 64) * 2, 65536]):
            kps = np.zeros(c, KP_DTYPE)
            desc = np.zeros((c, 32), np.uint8)
            n = C.c_int(0)
            rc = lib().orao_extract(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0], lapping[0], lapping[1], _p(kps), _p(desc), c, C.byref(n))
            if rc != -2:
                break
        return rc, kps[: n.value].copy(), desc[: n.value].copy()

    def scale_tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        lib().orao_get_scale_tables(self._h, *[_p(a) for a in t])
        return t

    def features_per_level(self):
        a = np.zeros(self.nlevels, np.int32)
        lib().orao_get_features_per_level(self._h, _p(a))
        return a

    def umax(self):
        a = np.zeros(16, np.int32)
        lib().orao_get_umax(self._h, _p(a))
        return a

    def _level(self, fn, level):
        r, c = C.c_int(0), C.c_int(0)
        if fn(self._h, level, None, 0, C.byref(r), C.byref(c)) != 0:
            raise IndexError(level)
        out = np.zeros((r.value, c.value), np.uint8)
        fn(self._h, level, _p(out), out.strides[0], C.byref(r), C.byref(c))
        return out

    def pyramid_level(self, level):
        return self._level(lib().orao_pyramid_level, level)

    def blurred_level(self, level):
        return self._level(lib().orao_blurred_level, level)

    def level_candidates(self, level, cap=1 << 20):
        xs, ys, rs = (np.zeros(cap, np.int32) for _ in range(3))
        n = lib().orao_level_candidates(self._h, level, _p(xs), _p(ys), _p(rs), cap)
        return xs[:n].copy(), ys[:n].copy(), rs[:n].copy()

    def level_keypoints(self, level, cap=1 << 16):
        out = np.zeros(cap, KP_DTYPE)
        n = lib().orao_level_keypoints(self._h, level, _p(out), cap)
        return out[:n].copy()


def resize_linear(src, dw, dh):
    src = _u8img(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orao_resize_linear_8u(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dst.strides[0])
    return dst


def gaussian_blur7(src):
    src = _u8img(src)
    dst = np.zeros_like(src)
    lib().orao_gaussian_blur7_8u(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast9_16(img, threshold, cap=1 << 20):
    img = _u8img(img)
    xs, ys, sc = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orao_fast9_16(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, _p(xs), _p(ys), _p(sc), cap)
    return xs[:n].copy(), ys[:n].copy(), sc[:n].copy()


def fast_atan2(y, x):
    return lib().orao_fast_atan2(float(y), float(x))


def cv_round(v):
    return lib().orao_cv_round_f(float(v))


def ic_angle(img, x, y):
    img = _u8img(img)
    return lib().orao_ic_angle(_p(img), img.strides[0], x, y)


def orb_descriptor(blurred, x, y, angle_deg, tap_contraction=1):
    img = _u8img(blurred)
    d = np.zeros(32, np.uint8)
    lib().orao_orb_descriptor_c(_p(img), img.strides[0], x, y, float(angle_deg), _p(d), int(tap_contraction))
    return d


def distribute_octtree(xs, ys, resp, min_x, max_x, min_y, max_y, n_target):
    xs = np.ascontiguousarray(xs, np.float32)
    ys = np.ascontiguousarray(ys, np.float32)
    resp = np.ascontiguousarray(resp, np.float32)
    cap = len(xs) + 8
    out = np.zeros(cap, np.int32)
    n = lib().orao_distribute_octtree(_p(xs), _p(ys), _p(resp), len(xs), min_x, max_x, min_y, max_y, n_target, _p(out), cap)
    if n < 0:
        return n, None
    return n, out[:n].copy()


def std_sort(records):
    """libstdc++'s std::sort on uint64 records ordered by (record >> 16); returns the sorted copy."""
    r = np.ascontiguousarray(records, np.uint64).copy()
    L = lib()
    L.orao_std_sort.argtypes = [C.c_void_p, C.c_int]
    L.orao_std_sort.restype = None
    L.orao_std_sort(_p(r), len(r))
    return r


def libm_sincos_deg(deg):
    deg = np.ascontiguousarray(deg, np.float32)
    c, s = np.zeros_like(deg), np.zeros_like(deg)
    lib().orao_libm_sincos_deg(_p(deg), len(deg), _p(c), _p(s))
    return c, s


def sincos_port_mismatches(bits_lo, bits_hi, fma_variant=0, nthreads=8):
    bad = np.zeros(16, np.uint32)
    nb = C.c_int(0)
    n = lib().orao_sincos_port_mismatches(bits_lo, bits_hi, fma_variant, nthreads, _p(bad), C.byref(nb))
    return n, bad[: nb.value]


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orao_hamming(_p(a), _p(b))


def three_maxima(counts):
    counts = np.ascontiguousarray(counts, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    lib().orao_three_maxima(_p(counts), len(counts), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def _opt(a, dt):
    return None if a is None else np.ascontiguousarray(a, dt)


def match_groups(q, t, q_angle=None, t_angle=None, th=50, ratio=0.7, strict_lt=False, check_ori=True, groups=None, q_valid=None, t_valid=None):
    """groups = None (brute force) or (q_off, q_idx, t_off, t_idx)."""
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    nq, nt = len(q), len(t)
    if groups is None:
        ng, qo, qi, to, ti = 1, None, None, None, None
    else:
        qo, qi, to, ti = (np.ascontiguousarray(a, np.int32) for a in groups)
        ng = len(qo) - 1
    qa = _opt(q_angle if q_angle is not None else np.zeros(nq), np.float32)
    ta = _opt(t_angle if t_angle is not None else np.zeros(nt), np.float32)
    qv, tv = _opt(q_valid, np.uint8), _opt(t_valid, np.uint8)
    mq, mt = np.zeros(nq, np.int32), np.zeros(nt, np.int32)
    n = lib().orao_match_groups(
        _p(q), nq, _p(t), nt, ng, _p(qo), _p(qi), _p(to), _p(ti), _p(qv), _p(tv), _p(qa), _p(ta), th, int(strict_lt), ratio, int(check_ori), _p(mq), _p(mt)
    )
    return n, mq, mt


def match_groups_two_cameras(q, t, t_nleft, q_angle=None, t_angle=None, th=50, ratio=0.7, check_ori=True, groups=None, q_valid=None):
    """SearchByBoW(KeyFrame*, Frame&) with F.Nleft = t_nleft (see orb_oracle.h).  Returns (nmatches, match_q_of_t)."""
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    nq, nt = len(q), len(t)
    if groups is None:
        ng, qo, qi, to, ti = 1, None, None, None, None
    else:
        qo, qi, to, ti = (np.ascontiguousarray(a, np.int32) for a in groups)
        ng = len(qo) - 1
    qa = _opt(q_angle if q_angle is not None else np.zeros(nq), np.float32)
    ta = _opt(t_angle if t_angle is not None else np.zeros(nt), np.float32)
    qv = _opt(q_valid, np.uint8)
    mt = np.zeros(nt, np.int32)
    L = lib()
    L.orao_match_groups_two_cameras.restype = C.c_int
    L.orao_match_groups_two_cameras.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_float, C.c_int, C.c_void_p]
    n = L.orao_match_groups_two_cameras(_p(q), nq, _p(t), nt, int(t_nleft), ng, _p(qo), _p(qi), _p(to), _p(ti), _p(qv), _p(qa), _p(ta), int(th), float(ratio), int(check_ori), _p(mt))
    return n, mt


def match_window(d1, kp1, d2, kp2, prev_matched, bounds, window=100, th=50, ratio=0.9, check_ori=True):
    d1 = np.ascontiguousarray(d1, np.uint8)
    d2 = np.ascontiguousarray(d2, np.uint8)
    kp1 = np.ascontiguousarray(kp1, KP_DTYPE)
    kp2 = np.ascontiguousarray(kp2, KP_DTYPE)
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(len(kp1), np.int32)
    n = lib().orao_match_window(_p(d1), _p(kp1), len(kp1), _p(d2), _p(kp2), len(kp2), _p(pm), *[float(b) for b in bounds], window, th, ratio, int(check_ori), _p(m12))
    return n, m12, pm


def stereo_matches(ex_left, ex_right, kl, dl, kr, dr, mb, mbf):
    """Frame::ComputeStereoMatches on the pyramids of the last extract of the two OracleExtractor objects."""
    kl = np.ascontiguousarray(kl, KP_DTYPE)
    kr = np.ascontiguousarray(kr, KP_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8)
    dr = np.ascontiguousarray(dr, np.uint8)
    u, d = np.zeros(len(kl), np.float32), np.zeros(len(kl), np.float32)
    n = lib().orao_stereo_matches(ex_left._h, ex_right._h, _p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), mb, mbf, _p(u), _p(d))
    return n, u, d


def search_by_projection_map_two_cameras(mp_desc, left, right, cur_desc, cur_kp, nleft, scale_factors, ratio, bounds, th_dist=100, cur_occupied=None,
                                         left_to_right=None, right_to_left=None):
    """SearchByProjection(F, vpMapPoints, ...) with F.Nleft = nleft (orb_oracle.h).  left / right = (uv, level, radius, valid).  -> (n, mp_of_cur)."""
    d1 = np.ascontiguousarray(mp_desc, np.uint8)
    d2 = np.ascontiguousarray(cur_desc, np.uint8)
    kp = np.ascontiguousarray(cur_kp, KP_DTYPE)
    cams = []
    for uv, lv, rad, va in (left, right):
        cams.append((np.ascontiguousarray(uv, np.float32).reshape(-1, 2), np.ascontiguousarray(lv, np.int32), np.ascontiguousarray(rad, np.float32),
                     np.ascontiguousarray(va, np.uint8)))
    occ, l2r, r2l = _opt(cur_occupied, np.uint8), _opt(left_to_right, np.int32), _opt(right_to_left, np.int32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    out = np.zeros(len(d2), np.int32)
    L = lib()
    L.orao_search_by_projection_map_two_cameras.restype = C.c_int
    L.orao_search_by_projection_map_two_cameras.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 10 + [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_float] * 5 + [C.c_int, C.c_void_p]
    (u0, l0, r0, v0), (u1, l1, r1, v1) = cams
    n = L.orao_search_by_projection_map_two_cameras(_p(d1), len(d1), _p(u0), _p(l0), _p(r0), _p(v0), _p(u1), _p(l1), _p(r1), _p(v1), _p(d2), _p(kp), len(d2), int(nleft),
                                                    _p(occ), _p(l2r), _p(r2l), _p(sf), float(ratio), *[float(b) for b in bounds], int(th_dist), _p(out))
    return n, out


def search_by_projection(last_desc, last_uv, last_invz, last_octave, last_angle, last_valid, cur_desc, cur_kp, cur_u_right, cur_occupied,
                         scale_factors, th, direction, mbf, bounds, th_dist=100, check_ori=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) at array level (see orb_oracle.h)."""
    last_desc = np.ascontiguousarray(last_desc, np.uint8)
    cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
    cur_kp = np.ascontiguousarray(cur_kp, KP_DTYPE)
    n1, n2 = len(last_desc), len(cur_desc)
    uv = np.ascontiguousarray(last_uv, np.float32).reshape(-1, 2)
    iz = _opt(last_invz, np.float32)
    oc = np.ascontiguousarray(last_octave, np.int32)
    an = np.ascontiguousarray(last_angle, np.float32)
    lv, ur, occ = _opt(last_valid, np.uint8), _opt(cur_u_right, np.float32), _opt(cur_occupied, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    col, loc = np.zeros(n1, np.int32), np.zeros(n2, np.int32)
    L = lib()
    L.orao_search_by_projection.argtypes = [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_float, C.c_int] + [C.c_float] * 5 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    n = L.orao_search_by_projection(_p(last_desc), _p(uv), _p(iz), _p(oc), _p(an), _p(lv), n1, _p(cur_desc), _p(cur_kp), _p(ur), _p(occ), n2, _p(sf),
                                    float(th), int(direction), float(mbf), *[float(b) for b in bounds], int(th_dist), int(check_ori), _p(col), _p(loc))
    return n, col, loc


def search_by_projection_map(mp_desc, mp_uv, mp_proj_xr, mp_level, mp_radius, mp_valid, cur_desc, cur_kp, cur_u_right, cur_occupied, scale_factors, ratio, bounds, th_dist=100):
    """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) at array level (see orb_oracle.h)."""
    mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    cur_desc = np.ascontiguousarray(cur_desc, np.uint8)
    cur_kp = np.ascontiguousarray(cur_kp, KP_DTYPE)
    n1, n2 = len(mp_desc), len(cur_desc)
    uv = np.ascontiguousarray(mp_uv, np.float32).reshape(-1, 2)
    xr = _opt(mp_proj_xr, np.float32)
    lv = np.ascontiguousarray(mp_level, np.int32)
    rad = np.ascontiguousarray(mp_radius, np.float32)
    va, ur, occ = _opt(mp_valid, np.uint8), _opt(cur_u_right, np.float32), _opt(cur_occupied, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    com, moc = np.zeros(n1, np.int32), np.zeros(n2, np.int32)
    L = lib()
    L.orao_search_by_projection_map.argtypes = [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p] + [C.c_float] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
    n = L.orao_search_by_projection_map(_p(mp_desc), _p(uv), _p(xr), _p(lv), _p(rad), _p(va), n1, _p(cur_desc), _p(cur_kp), _p(ur), _p(occ), n2, _p(sf), float(ratio),
                                        *[float(b) for b in bounds], int(th_dist), _p(com), _p(moc))
    return n, com, moc


def search_in_windows(q_desc, q_uv, q_ur, q_level, q_valid, t_desc, t_kp, t_u_right, t_occupied, scale_factors, inv_level_sigma2, th, claim, bounds, th_dist):
    """Windowed best-candidate search of SearchByProjection(KeyFrame, Sim3, ...) / Fuse / SearchBySim3 at array level (see orb_oracle.h)."""
    q_desc = np.ascontiguousarray(q_desc, np.uint8)
    t_desc = np.ascontiguousarray(t_desc, np.uint8)
    t_kp = np.ascontiguousarray(t_kp, KP_DTYPE)
    n1, n2 = len(q_desc), len(t_desc)
    uv = np.ascontiguousarray(q_uv, np.float32).reshape(-1, 2)
    qr, lv = _opt(q_ur, np.float32), np.ascontiguousarray(q_level, np.int32)
    va, ur, occ = _opt(q_valid, np.uint8), _opt(t_u_right, np.float32), _opt(t_occupied, np.uint8)
    sf, s2 = np.ascontiguousarray(scale_factors, np.float32), _opt(inv_level_sigma2, np.float32)
    toq = np.zeros(n1, np.int32)
    L = lib()
    L.orao_search_in_windows.argtypes = [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_void_p]
    n = L.orao_search_in_windows(_p(q_desc), _p(uv), _p(qr), _p(lv), _p(va), n1, _p(t_desc), _p(t_kp), _p(ur), _p(occ), n2, _p(sf), _p(s2), float(th), int(claim),
                                 *[float(b) for b in bounds], int(th_dist), _p(toq))
    return n, toq


def search_for_triangulation(d1, kp1, mp1, stereo1, d2, kp2, mp2, stereo2, groups, F12, ep, scale_factors, level_sigma2, only_stereo=False, coarse=False,
                             th_low=50, check_ori=True):
    """ORBmatcher::SearchForTriangulation at array level (see orb_oracle.h); groups = (q_off, q_idx, t_off, t_idx)."""
    d1, d2 = np.ascontiguousarray(d1, np.uint8), np.ascontiguousarray(d2, np.uint8)
    kp1, kp2 = np.ascontiguousarray(kp1, KP_DTYPE), np.ascontiguousarray(kp2, KP_DTYPE)
    qo, qi, to, ti = (np.ascontiguousarray(a, np.int32) for a in groups)
    m1, s1, m2, s2 = _opt(mp1, np.uint8), _opt(stereo1, np.uint8), _opt(mp2, np.uint8), _opt(stereo2, np.uint8)
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    e = np.ascontiguousarray(ep, np.float32).reshape(2)
    sf, sg = np.ascontiguousarray(scale_factors, np.float32), np.ascontiguousarray(level_sigma2, np.float32)
    m12 = np.zeros(len(kp1), np.int32)
    L = lib()
    L.orao_search_for_triangulation.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_void_p]
    n = L.orao_search_for_triangulation(_p(d1), _p(kp1), _p(m1), _p(s1), len(kp1), _p(d2), _p(kp2), _p(m2), _p(s2), len(kp2), len(qo) - 1, _p(qo), _p(qi), _p(to),
                                        _p(ti), _p(F), _p(e), _p(sf), _p(sg), int(only_stereo), int(coarse), int(th_low), int(check_ori), _p(m12))
    return n, m12


def undistort_keypoints(kps, K, dist):
    """Frame::UndistortKeyPoints: K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2[, k3])."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    K = np.ascontiguousarray(K, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(len(kps), KP_DTYPE)
    L = lib()
    L.orao_undistort_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orao_undistort_keypoints.restype = None
    L.orao_undistort_keypoints(_p(kps), len(kps), _p(K), _p(dist), len(dist), _p(out))
    return out


def image_bounds(cols, rows, K, dist):
    K = np.ascontiguousarray(K, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(4, np.float32)
    L = lib()
    L.orao_image_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orao_image_bounds.restype = None
    L.orao_image_bounds(cols, rows, _p(K), _p(dist), len(dist), _p(out))
    return out


class OracleVocabulary:
    """ORBVocabulary (DBoW2 TemplatedVocabulary, TF_IDF / L1_NORM) restated on the CPU: transform()."""

    def __init__(self, k, L, parent, is_leaf, desc, weight):
        L_ = lib()
        L_.orao_vocab_create.restype = C.c_void_p
        L_.orao_vocab_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L_.orao_vocab_destroy.argtypes = [C.c_void_p]
        L_.orao_vocab_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 9
        parent = np.ascontiguousarray(parent, np.int32)
        is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8)
        weight = np.ascontiguousarray(weight, np.float64)
        self._h = L_.orao_vocab_create(k, L, len(parent), _p(parent), _p(is_leaf), _p(desc), _p(weight))
        if not self._h:
            raise ValueError("bad vocabulary arrays")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orao_vocab_destroy(self._h)
            self._h = None

    def transform(self, desc, levelsup=4):
        """-> (word_ids, word_vals, node_ids, node_off, feat_idx, word_of_feat, node_of_feat)"""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        wi, wv = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.float64)
        ni, no, fi = np.zeros(n + 1, np.uint32), np.zeros(n + 2, np.int32), np.zeros(n + 1, np.uint32)
        wf, nf = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint32)
        nw, nn = C.c_int(0), C.c_int(0)
        lib().orao_vocab_transform(self._h, _p(desc), n, levelsup, _p(wi), _p(wv), C.byref(nw), _p(ni), _p(no), _p(fi), C.byref(nn), _p(wf), _p(nf))
        return wi[: nw.value], wv[: nw.value], ni[: nn.value], no[: nn.value + 1], fi[: no[nn.value]], wf[:n], nf[:n]
