"""Host-side mirror of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:41-119) over the C ABI.

Same constructor arguments, same call convention (`operator()(image, mask, keypoints, descriptors,
vLappingArea)` -> returns monoIndex, fills keypoints/descriptors), same getters, same error behaviour
(-1 on an empty image).  The compute runs in the HIP kernels of liborbx.so; nothing here computes.
"""
import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, OrbxError, check, ptr


class ORBextractor:
    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device=0):
        self._h = C.c_void_p()
        L = _lib.lib()
        check(L.orbx_create(C.byref(self._h), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device), "orbx_create")
        self.nfeatures, self.nlevels, self.device = nfeatures, nlevels, device

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().orbx_destroy(self._h)
            self._h = None

    def __del__(self):
        # Also during interpreter shutdown: the destroy entry points are safe at any point of process teardown (the library releases the
        # device side of live handles from its own exit handler and only frees host state afterwards), and a handle that is destroyed is
        # one the runtime does not find alive in its own teardown.
        try:
            self.close()
        except Exception:
            pass

    # ---- include/ORBextractor.h:63-85
    def GetLevels(self):
        return _lib.lib().orbx_get_levels(self._h)

    def _tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        _lib.lib().orbx_get_scale_tables(self._h, *[ptr(a) for a in t])
        return t

    def GetScaleFactor(self):
        return float(self._tables()[0][1]) if self.nlevels > 1 else 1.0

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        a = np.zeros(self.nlevels, np.int32)
        _lib.lib().orbx_get_features_per_level(self._h, ptr(a))
        return a

    def set_input_resize(self, new_cols, new_rows):
        """cv::resize(im, im, Size(new_cols, new_rows)) of System::Track* (src/System.cc:472-478) fused in front of the extractor; 0, 0 = off."""
        check(_lib.lib().orbx_set_input_resize(self._h, int(new_cols), int(new_rows)), "orbx_set_input_resize")

    def set_graph_replay(self, enable):
        """orbx_set_graph_replay: record the single-image call once (second call of a shape) and replay it as a hipGraph."""
        check(_lib.lib().orbx_set_graph_replay(self._h, int(bool(enable))), "orbx_set_graph_replay")

    def set_pyramid_download(self, enable):
        """orbx_set_pyramid_download: levels >= 1 come home inside every single-frame call (what the C++ adaptor's mbDownloadPyramid turns on)."""
        check(_lib.lib().orbx_set_pyramid_download(self._h, 1 if enable else 0), "orbx_set_pyramid_download")

    def set_tap_contraction(self, mode):
        """Tap-rotation evaluation of computeOrbDescriptor (include/orbx.h): 1 = reference as built on an FMA host (default), 0 = none, 2 = other."""
        check(_lib.lib().orbx_set_tap_contraction(self._h, int(mode)), "orbx_set_tap_contraction")

    def max_keypoints(self, rows, cols):
        return check(_lib.lib().orbx_max_keypoints(self._h, rows, cols), "orbx_max_keypoints")

    # ---- operator() (include/ORBextractor.h:55-61)
    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """Returns (monoIndex, keypoints[KP_DTYPE], descriptors[n, 32] uint8).  `mask` is ignored, as in the reference."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        img = np.asarray(image)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise OrbxError("image must be 8-bit single channel (assert(image.type() == CV_8UC1))")
        if img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        rows, cols = img.shape
        cap = self.max_keypoints(rows, cols)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rc = check(
            _lib.lib().orbx_extract(self._h, ptr(img), rows, cols, img.strides[0], int(vLappingArea[0]), int(vLappingArea[1]), ptr(kps), ptr(desc), cap, C.byref(n)),
            "orbx_extract",
        )
        return rc, kps[: n.value].copy(), desc[: n.value].copy()

    def extract_batch_device(self, d_imgs, nframes, rows, cols, step, frame_stride, d_kps, d_desc, d_counts, cap, vLappingArea=(0, 0), stream=None):
        """Enqueue extraction of `nframes` device-resident frames; outputs stay on the device.  Pointers may be
        torch tensors (data_ptr is taken) or raw ints; `stream` is a hipStream_t value or None."""
        return check(
            _lib.lib().orbx_extract_batch_device(
                self._h, ptr(d_imgs), nframes, rows, cols, step, frame_stride, int(vLappingArea[0]), int(vLappingArea[1]), ptr(d_kps), ptr(d_desc), ptr(d_counts), cap,
                C.c_void_p(stream) if stream else None,
            ),
            "orbx_extract_batch_device",
        )

    # ---- streaming form (host frames in, host records out, several chunks in flight)
    def set_async_depth(self, depth):
        check(_lib.lib().orbx_set_async_depth(self._h, int(depth)), "orbx_set_async_depth")

    def extract_async(self, imgs, kps, desc, counts, cap, vLappingArea=(0, 0)):
        """imgs: uint8 [nframes, rows, cols] host array (pinned_array() memory is read by DMA directly); kps / desc / counts: host
        arrays [nframes, cap] KP_DTYPE, [nframes, cap, 32] uint8, [nframes, 2] int32 that hold the results after wait(ticket)."""
        nframes, rows, cols = imgs.shape
        t = C.c_int(-1)
        check(
            _lib.lib().orbx_extract_async(self._h, ptr(imgs), nframes, rows, cols, imgs.strides[1], imgs.strides[0], int(vLappingArea[0]), int(vLappingArea[1]),
                                          ptr(kps), ptr(desc), ptr(counts), cap, C.byref(t)),
            "orbx_extract_async",
        )
        return t.value

    def async_device_buffers(self, ticket):
        """(d_kps, d_desc, d_counts, stream) raw device pointers / hipStream_t of a ticket in flight."""
        a, b, c, s = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(_lib.lib().orbx_async_device_buffers(self._h, ticket, C.byref(a), C.byref(b), C.byref(c), C.byref(s)), "orbx_async_device_buffers")
        return a.value, b.value, c.value, s.value

    def async_download(self, ticket, host_dst, device_src, nbytes=None):
        """One more D2H copy for a ticket in flight, behind what has been chained on its stream so far (host_dst: pinned_array)."""
        n = host_dst.nbytes if nbytes is None else nbytes
        check(_lib.lib().orbx_async_download(self._h, ticket, ptr(host_dst), ptr(device_src), n), "orbx_async_download")

    def wait(self, ticket):
        check(_lib.lib().orbx_wait(self._h, ticket), "orbx_wait")

    # ---- mvImagePyramid (include/ORBextractor.h:87)
    def _level(self, fn, frame, level):
        r, c = C.c_int(0), C.c_int(0)
        check(fn(self._h, frame, level, None, 0, C.byref(r), C.byref(c)), "level query")
        out = np.zeros((r.value, c.value), np.uint8)
        check(fn(self._h, frame, level, ptr(out), out.strides[0], C.byref(r), C.byref(c)), "level copy")
        return out

    def pyramid_level(self, level, frame=0):
        return self._level(_lib.lib().orbx_pyramid_level, frame, level)

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    # ---- parity hooks
    def debug_blurred_level(self, level, frame=0):
        return self._level(_lib.lib().orbx_debug_blurred_level, frame, level)

    def _packed(self, fn, frame, level, cap):
        out = np.zeros(cap, np.uint32)
        n = check(fn(self._h, frame, level, ptr(out), cap), "debug list")
        out = out[:n]
        return (out & 0xFFF).astype(np.int32), ((out >> 12) & 0xFFF).astype(np.int32), (out >> 24).astype(np.int32)

    def debug_level_candidates(self, level, frame=0, cap=1 << 20):
        return self._packed(_lib.lib().orbx_debug_level_candidates, frame, level, cap)

    def debug_level_keypoints(self, level, frame=0, cap=1 << 16):
        return self._packed(_lib.lib().orbx_debug_level_keypoints, frame, level, cap)

    def debug_octree_device(self, xs, ys, resp, W, H, N):
        cand = (np.asarray(xs, np.uint32) | (np.asarray(ys, np.uint32) << 12) | (np.asarray(resp, np.uint32) << 24)).astype(np.uint32)
        out = np.zeros(max(N + 3, 4 * max(1, int(round(W / H)))) + 64, np.uint32)
        cnt = C.c_int(0)
        check(_lib.lib().orbx_debug_octree_device(self._h, ptr(cand), len(cand), W, H, N, ptr(out), C.byref(cnt)), "octree_device")
        return out[: cnt.value].copy()

    def debug_fault(self):
        """orbx_debug_fault: None, or (code, frame, value, where) of the first inconsistent count a kernel met since the last query."""
        out = (C.c_int * 4)()
        rc = check(_lib.lib().orbx_debug_fault(self._h, out), "orbx_debug_fault")
        return tuple(out) if rc == 1 else None

    def set_profiling(self, enable=True):
        _lib.lib().orbx_set_profiling(self._h, int(enable))

    def stage_ms(self):
        ms = np.zeros(_lib.NUM_STAGES, np.float32)
        check(_lib.lib().orbx_get_stage_ms(self._h, ptr(ms)), "orbx_get_stage_ms")
        return dict(zip(_lib.STAGE_NAMES, ms.tolist()))


def pinned_array(shape, dtype):
    """numpy array over orbx_host_alloc memory (page-locked: the streaming entry points DMA straight from / into it).  The
    memory is released when the array (and every view of it) is gone."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = _lib.lib().orbx_host_alloc(max(n, 1))
    if not p:
        raise OrbxError("orbx_host_alloc(%d) failed" % n)

    class _Owner:
        def __init__(self, p):
            self.p = p

        def __del__(self, _finalizing=sys.is_finalizing):
            if _finalizing():  # at interpreter shutdown the HIP runtime may already be gone: the OS reclaims the pages
                return
            try:
                _lib.lib().orbx_host_free(C.c_void_p(self.p))
            except Exception:
                pass

    buf = (C.c_uint8 * max(n, 1)).from_address(p)
    buf._owner = _Owner(p)
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def UndistortKeyPoints(ex, keys, K, dist):
    """Frame::UndistortKeyPoints (src/Frame.cc:886-913): keys (KP_DTYPE) -> mvKeysUn; K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2[, k3])."""
    k = np.ascontiguousarray(keys, KP_DTYPE)
    Kf = np.ascontiguousarray(K, np.float32)
    d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(len(k), KP_DTYPE)
    check(_lib.lib().orbx_undistort_keypoints(ex._h, ptr(k), len(k), ptr(Kf), ptr(d), len(d), ptr(out)), "orbx_undistort_keypoints")
    return out


def ComputeImageBounds(cols, rows, K, dist):
    """Frame::ComputeImageBounds (src/Frame.cc:915-943) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    Kf = np.ascontiguousarray(K, np.float32)
    d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(4, np.float32)
    check(_lib.lib().orbx_image_bounds(int(cols), int(rows), ptr(Kf), ptr(d), len(d), ptr(out)), "orbx_image_bounds")
    return tuple(float(v) for v in out)


def ComputeStereoMatches(ex_left, ex_right, keys_left, desc_left, keys_right, desc_right, mb, mbf):
    """Frame::ComputeStereoMatches (reference src/Frame.cc:944-1113) on the pyramids the two extractors hold on the device
    after their last single-image call.  Returns (n_kept, mvuRight, mvDepth)."""
    kl = np.ascontiguousarray(keys_left, KP_DTYPE)
    kr = np.ascontiguousarray(keys_right, KP_DTYPE)
    dl = np.ascontiguousarray(desc_left, np.uint8)
    dr = np.ascontiguousarray(desc_right, np.uint8)
    u, d = np.zeros(len(kl), np.float32), np.zeros(len(kl), np.float32)
    n = check(
        _lib.lib().orbx_stereo_matches(ex_left._h, ex_right._h, ptr(kl), ptr(dl), len(kl), ptr(kr), ptr(dr), len(kr), float(mb), float(mbf), ptr(u), ptr(d)),
        "orbx_stereo_matches",
    )
    return n, u, d


def octree_host_model(xs, ys, resp, W, H, N):
    """Host build of the kernel's quadtree core (octree_core.h) -- CPU tests only."""
    cand = (np.asarray(xs, np.uint32) | (np.asarray(ys, np.uint32) << 12) | (np.asarray(resp, np.uint32) << 24)).astype(np.uint32)
    out = np.zeros(max(N + 3, 4 * max(1, int(round(W / H)))) + 64, np.uint32)
    cnt = C.c_int(0)
    rc = _lib.lib().orbx_debug_octree_host(ptr(cand), len(cand), W, H, N, ptr(out), C.byref(cnt))
    return rc, out[: cnt.value].copy()
