"""ORBVocabulary (DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>) on the GPU: the transform() that Frame::ComputeBoW calls
(reference src/Frame.cc:879-884, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259).  ctypes mirror of the orbv_* C ABI
(include/orbx.h); there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _lib


class VocabError(RuntimeError):
    pass


def _check(rc, what):
    if rc < 0:
        raise VocabError("%s failed (%d): %s" % (what, rc, _lib.lib().orbv_last_error().decode()))
    return rc


class ORBVocabulary:
    """Vocabulary tree resident on one GPU.

    ORBVocabulary(path)                                    -- loadFromTextFile (ORBvoc.txt format)
    ORBVocabulary.from_arrays(k, L, parent, is_leaf, desc, weight)
    """

    def __init__(self, path=None, device=0, _handle=None):
        self._h = _handle
        if _handle is None:
            h = C.c_void_p()
            _check(_lib.lib().orbv_load_text(C.byref(h), str(path).encode(), device), "orbv_load_text")
            self._h = h

    @classmethod
    def from_arrays(cls, k, L, parent, is_leaf, desc, weight, device=0):
        parent = np.ascontiguousarray(parent, np.int32)
        is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8)
        weight = np.ascontiguousarray(weight, np.float64)
        assert desc.shape == (len(parent), 32) and len(is_leaf) == len(parent) == len(weight)
        h = C.c_void_p()
        _check(_lib.lib().orbv_create(C.byref(h), k, L, len(parent), _lib.ptr(parent), _lib.ptr(is_leaf), _lib.ptr(desc), _lib.ptr(weight), device), "orbv_create")
        return cls(_handle=h)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().orbv_destroy(self._h)
            self._h = None

    def __del__(self):
        # Also during interpreter shutdown: the destroy entry points are safe at any point of process teardown (the library releases the
        # device side of live handles from its own exit handler and only frees host state afterwards), and a handle that is destroyed is
        # one the runtime does not find alive in its own teardown.
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        k, L, n, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _check(_lib.lib().orbv_info(self._h, C.byref(k), C.byref(L), C.byref(n), C.byref(w)), "orbv_info")
        return {"k": k.value, "L": L.value, "nodes": n.value, "words": w.value}

    def transform(self, descriptors, levelsup=4):
        """-> (bow, fv): bow = (word_ids[uint32] ascending, values[float64]); fv = (node_ids ascending, node_off, feat_idx),
        i.e. the flattened DBoW2::BowVector and DBoW2::FeatureVector of transform(features, v, fv, levelsup)."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        wi, wv = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.float64)
        ni, no, fi = np.zeros(n + 1, np.uint32), np.zeros(n + 2, np.int32), np.zeros(n + 1, np.uint32)
        nw, nn = C.c_int(0), C.c_int(0)
        _check(
            _lib.lib().orbv_transform(self._h, _lib.ptr(d), n, levelsup, _lib.ptr(wi), _lib.ptr(wv), C.byref(nw), _lib.ptr(ni), _lib.ptr(no), _lib.ptr(fi), C.byref(nn)),
            "orbv_transform",
        )
        return (wi[: nw.value], wv[: nw.value]), (ni[: nn.value], no[: nn.value + 1], fi[: no[nn.value]])

    def transform_batch_device(self, d_desc, d_counts, nframes, cap, levelsup, out, stream=None):
        """Device-resident batch form; `out` = dict of torch tensors bow_words[n,cap] (int32/uint32), bow_vals[n,cap] (float64),
        n_words[n], fv_nodes[n,cap], fv_off[n,cap+1], fv_feat[n,cap], n_nodes[n].  Enqueue only."""
        _check(
            _lib.lib().orbv_transform_batch_device(
                self._h, _lib.ptr(d_desc), _lib.ptr(d_counts), nframes, cap, levelsup, _lib.ptr(out["bow_words"]), _lib.ptr(out["bow_vals"]),
                _lib.ptr(out["n_words"]), _lib.ptr(out["fv_nodes"]), _lib.ptr(out["fv_off"]), _lib.ptr(out["fv_feat"]), _lib.ptr(out["n_nodes"]),
                C.c_void_p(stream) if stream else None,
            ),
            "orbv_transform_batch_device",
        )


def feature_vector_groups(fv_a, fv_b):
    """Merge-walk of two FeatureVectors as SearchByBoW does (reference src/ORBmatcher.cc:256-268, 368-375): for every node
    present in both, the feature lists of the two frames.  Returns (q_off, q_idx, t_off, t_idx) for ORBmatcher.SearchByBoW /
    orbm_match_groups (a = queries, b = trains)."""
    na, oa, fa = fv_a
    nb, ob, fb = fv_b
    q_off, q_idx, t_off, t_idx = [0], [], [0], []
    i = j = 0
    while i < len(na) and j < len(nb):
        if na[i] == nb[j]:
            q_idx.extend(fa[oa[i] : oa[i + 1]].tolist())
            t_idx.extend(fb[ob[j] : ob[j + 1]].tolist())
            q_off.append(len(q_idx))
            t_off.append(len(t_idx))
            i += 1
            j += 1
        elif na[i] < nb[j]:
            i += 1
        else:
            j += 1
    return (np.asarray(q_off, np.int32), np.asarray(q_idx, np.int32), np.asarray(t_off, np.int32), np.asarray(t_idx, np.int32))
