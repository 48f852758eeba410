"""Host-side mirror of ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:35-148) over the C ABI.

The reference's methods take Frame / KeyFrame objects; here they take the arrays those objects hold
(descriptors N x 32 u8, keypoint records, DBoW2 FeatureVectors as {node_id: [feature indices]} dicts, MapPoint
validity masks).  Constants, thresholds, the greedy claim order and the rotation check are the reference's.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr


def flatten_feature_vectors(fv_q, fv_t):
    """Merge-walk of two DBoW2::FeatureVector (std::map<NodeId, vector<uint>>) as SearchByBoW does
    (src/ORBmatcher.cc:259-399): only nodes present in both, ascending node id."""
    q_off, q_idx, t_off, t_idx = [0], [], [0], []
    for node in sorted(set(fv_q).intersection(fv_t)):
        q_idx.extend(fv_q[node])
        t_idx.extend(fv_t[node])
        q_off.append(len(q_idx))
        t_off.append(len(t_idx))
    return (np.asarray(q_off, np.int32), np.asarray(q_idx, np.int32), np.asarray(t_off, np.int32), np.asarray(t_idx, np.int32))


class ORBmatcher:
    TH_HIGH = 100  # src/ORBmatcher.cc:30
    TH_LOW = 50  # src/ORBmatcher.cc:31
    HISTO_LENGTH = 30  # src/ORBmatcher.cc:32

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._h = C.c_void_p()
        check(_lib.lib().orbm_create(C.byref(self._h), device), "orbm_create")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().orbm_destroy(self._h)
            self._h = None

    def __del__(self):
        # Also during interpreter shutdown: the destroy entry points are safe at any point of process teardown (the library releases the
        # device side of live handles from its own exit handler and only frees host state afterwards), and a handle that is destroyed is
        # one the runtime does not find alive in its own teardown.
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def DescriptorDistance(a, b):
        """src/ORBmatcher.cc:2075-2089"""
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return _lib.lib().orbm_hamming(ptr(a), ptr(b))

    @staticmethod
    def ComputeThreeMaxima(counts):
        """src/ORBmatcher.cc:2038-2071 on bin sizes."""
        counts = np.ascontiguousarray(counts, np.int32)
        i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        _lib.lib().orbm_three_maxima(ptr(counts), len(counts), C.byref(i1), C.byref(i2), C.byref(i3))
        return i1.value, i2.value, i3.value

    def _groups(self, q, t, q_angle, t_angle, groups, q_valid, t_valid, th, strict_lt):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq, nt = len(q), len(t)
        qa = np.ascontiguousarray(q_angle if q_angle is not None else np.zeros(nq), np.float32)
        ta = np.ascontiguousarray(t_angle if t_angle is not None else np.zeros(nt), np.float32)
        qv = None if q_valid is None else np.ascontiguousarray(q_valid, np.uint8)
        tv = None if t_valid is None else np.ascontiguousarray(t_valid, np.uint8)
        if groups is None:
            ng, qo, qi, to, ti = 1, None, None, None, None
        else:
            qo, qi, to, ti = (np.ascontiguousarray(a, np.int32) for a in groups)
            ng = len(qo) - 1
            if ng == 0:
                return 0, np.full(nq, -1, np.int32), np.full(nt, -1, np.int32)
        mq, mt = np.zeros(nq, np.int32), np.zeros(nt, np.int32)
        n = check(
            _lib.lib().orbm_match_groups(
                self._h, ptr(q), nq, ptr(t), nt, ng, ptr(qo), ptr(qi), ptr(to), ptr(ti), ptr(qv), ptr(tv), ptr(qa), ptr(ta), th, int(strict_lt),
                self.mfNNratio, int(self.mbCheckOrientation), ptr(mq), ptr(mt),
            ),
            "orbm_match_groups",
        )
        return n, mq, mt

    def SearchByBoW(self, kf_desc, kf_angle, kf_featvec, kf_has_mappoint, f_desc, f_angle, f_featvec, f_nleft=-1):
        """SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:237-421).  f_nleft = F.Nleft: -1 for a one-camera
        frame; else f_desc / f_angle hold the left keys [0, Nleft) followed by the right keys and the two-camera branch
        (:305-333, 362-389) runs, where one KeyFrame feature may hand its MapPoint to a left AND a right Frame feature.
        Returns (nmatches, match_kf_of_f[F.N]) -- the KeyFrame feature whose MapPoint each Frame feature received, -1 = NULL."""
        groups = None if kf_featvec is None else flatten_feature_vectors(kf_featvec, f_featvec)
        if f_nleft != -1:
            q = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32)
            t = np.ascontiguousarray(f_desc, np.uint8).reshape(-1, 32)
            nq, nt = len(q), len(t)
            qa = np.ascontiguousarray(kf_angle if kf_angle is not None else np.zeros(nq), np.float32)
            ta = np.ascontiguousarray(f_angle if f_angle is not None else np.zeros(nt), np.float32)
            qv = None if kf_has_mappoint is None else np.ascontiguousarray(kf_has_mappoint, np.uint8)
            if groups is None:
                ng, qo, qi, to, ti = 1, None, None, None, None
            else:
                qo, qi, to, ti = (np.ascontiguousarray(a, np.int32) for a in groups)
                ng = len(qo) - 1
                if ng == 0:
                    return 0, np.full(nt, -1, np.int32)
            mt = np.zeros(nt, np.int32)
            n = check(_lib.lib().orbm_match_groups_two_cameras(self._h, ptr(q), nq, ptr(t), nt, int(f_nleft), ng, ptr(qo), ptr(qi), ptr(to), ptr(ti), ptr(qv),
                                                               ptr(qa), ptr(ta), self.TH_LOW, self.mfNNratio, int(self.mbCheckOrientation), ptr(mt)),
                      "orbm_match_groups_two_cameras")
            return n, mt
        n, mq, mt = self._groups(kf_desc, f_desc, kf_angle, f_angle, groups, kf_has_mappoint, None, self.TH_LOW, False)
        return n, mt

    def SearchByBoW_KF(self, desc1, angle1, featvec1, has_mp1, desc2, angle2, featvec2, has_mp2):
        """SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (src/ORBmatcher.cc:784-914): `< TH_LOW`, both sides need MapPoints.
        Returns (nmatches, match_2_of_1[N1])."""
        groups = None if featvec1 is None else flatten_feature_vectors(featvec1, featvec2)
        n, mq, mt = self._groups(desc1, desc2, angle1, angle2, groups, has_mp1, has_mp2, self.TH_LOW, True)
        return n, mq

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, vbPrevMatched, bounds, windowSize=10):
        """src/ORBmatcher.cc:666-782.  kps are undistorted keypoints (KP_DTYPE); bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY).
        Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        kps1 = np.ascontiguousarray(kps1, KP_DTYPE)
        kps2 = np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        pm = np.ascontiguousarray(vbPrevMatched, np.float32).reshape(-1, 2).copy()
        m12 = np.zeros(len(kps1), np.int32)
        n = check(
            _lib.lib().orbm_match_window(
                self._h, ptr(d1), ptr(kps1), len(kps1), ptr(d2), ptr(kps2), len(kps2), ptr(pm), float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]),
                int(windowSize), self.TH_LOW, self.mfNNratio, int(self.mbCheckOrientation), ptr(m12),
            ),
            "orbm_match_window",
        )
        return n, m12, pm

    def SearchByProjection(self, last_desc, last_uv, last_octave, last_angle, cur_kps, cur_desc, scale_factors, th, bounds, last_valid=None,
                           last_invz=None, cur_u_right=None, cur_occupied=None, direction=0, mbf=0.0, check_ori=None, th_dist=None):
        """SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1684-1908, Nleft == -1) on
        arrays: last_* describe the last frame's MapPoints already projected into the current frame by the caller, cur_kps are the
        current frame's undistorted keypoints (KP_DTYPE), bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY), direction 0 / 1 (bForward) /
        2 (bBackward).  Returns (nmatches, cur_of_last[n1], last_of_cur[n2])."""
        d1 = np.ascontiguousarray(last_desc, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(cur_desc, np.uint8).reshape(-1, 32)
        kp2 = np.ascontiguousarray(cur_kps, KP_DTYPE)
        n1, n2 = len(d1), len(d2)
        uv = np.ascontiguousarray(last_uv, np.float32).reshape(-1, 2)
        oc = np.ascontiguousarray(last_octave, np.int32)
        an = np.ascontiguousarray(last_angle, np.float32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        opt = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        lv, iz, ur, occ = opt(last_valid, np.uint8), opt(last_invz, np.float32), opt(cur_u_right, np.float32), opt(cur_occupied, np.uint8)
        col, loc = np.zeros(n1, np.int32), np.zeros(n2, np.int32)
        n = check(
            _lib.lib().orbm_search_by_projection(
                self._h, ptr(d1), ptr(uv), ptr(iz), ptr(oc), ptr(an), ptr(lv), n1, ptr(d2), ptr(kp2), ptr(ur), ptr(occ), n2, ptr(sf), len(sf), float(th),
                int(direction), float(mbf), float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]), self.TH_HIGH if th_dist is None else int(th_dist),
                int(self.mbCheckOrientation if check_ori is None else check_ori), ptr(col), ptr(loc),
            ),
            "orbm_search_by_projection",
        )
        return n, col, loc

    def SearchByProjectionMapPoints(self, mp_desc, mp_uv, mp_level, mp_radius, cur_kps, cur_desc, scale_factors, bounds, mp_valid=None, mp_proj_xr=None,
                                    cur_u_right=None, cur_occupied=None):
        """SearchByProjection(Frame& F, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) (src/ORBmatcher.cc:38-227, Nleft == -1) on arrays;
        mp_radius = RadiusByViewingCos(viewCos) * th.  Returns (nmatches, cur_of_mp[n1], mp_of_cur[n2])."""
        d1 = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(cur_desc, np.uint8).reshape(-1, 32)
        kp2 = np.ascontiguousarray(cur_kps, KP_DTYPE)
        n1, n2 = len(d1), len(d2)
        uv = np.ascontiguousarray(mp_uv, np.float32).reshape(-1, 2)
        lv = np.ascontiguousarray(mp_level, np.int32)
        rad = np.ascontiguousarray(mp_radius, np.float32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        opt = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        va, xr, ur, occ = opt(mp_valid, np.uint8), opt(mp_proj_xr, np.float32), opt(cur_u_right, np.float32), opt(cur_occupied, np.uint8)
        com, moc = np.zeros(n1, np.int32), np.zeros(n2, np.int32)
        n = check(
            _lib.lib().orbm_search_by_projection_map(
                self._h, ptr(d1), ptr(uv), ptr(xr), ptr(lv), ptr(rad), ptr(va), n1, ptr(d2), ptr(kp2), ptr(ur), ptr(occ), n2, ptr(sf), len(sf), self.mfNNratio,
                float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]), self.TH_HIGH, ptr(com), ptr(moc),
            ),
            "orbm_search_by_projection_map",
        )
        return n, com, moc

    def SearchByProjectionMapPointsTwoCameras(self, mp_desc, left, right, cur_kps, cur_desc, nleft, scale_factors, bounds, cur_occupied=None,
                                              left_to_right=None, right_to_left=None):
        """SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) for a two-camera frame (src/ORBmatcher.cc:139-146, 152-226).
        left / right = (uv[n1,2], level[n1], radius[n1], valid[n1]) per camera (see include/orbx.h); cur_kps / cur_desc hold the left keys
        [0, nleft) followed by the right keys.  Returns (nmatches, mp_of_cur[n])."""
        d1 = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(cur_desc, np.uint8).reshape(-1, 32)
        kp = np.ascontiguousarray(cur_kps, _lib.KP_DTYPE)
        n1, n = len(d1), len(d2)
        cams = []
        for uv, lv, rad, va in (left, right):
            cams += [np.ascontiguousarray(uv, np.float32).reshape(-1, 2), np.ascontiguousarray(lv, np.int32), np.ascontiguousarray(rad, np.float32),
                     np.ascontiguousarray(va, np.uint8)]
        occ = None if cur_occupied is None else np.ascontiguousarray(cur_occupied, np.uint8)
        l2r = None if left_to_right is None else np.ascontiguousarray(left_to_right, np.int32)
        r2l = None if right_to_left is None else np.ascontiguousarray(right_to_left, np.int32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        out = np.zeros(n, np.int32)
        nm = check(_lib.lib().orbm_search_by_projection_map_two_cameras(self._h, ptr(d1), n1, *[ptr(c) for c in cams], ptr(d2), ptr(kp), n, int(nleft), ptr(occ),
                                                                        ptr(l2r), ptr(r2l), ptr(sf), len(sf), self.mfNNratio, *[float(b) for b in bounds], self.TH_HIGH,
                                                                        ptr(out)), "orbm_search_by_projection_map_two_cameras")
        return nm, out

    def SearchByProjectionKeyFrame(self, kf_mp_desc, mp_uv, mp_level, kf_angle, cur_kps, cur_desc, scale_factors, th, ORBdist, bounds, mp_valid=None,
                                   cur_occupied=None):
        """SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1910-2036), the relocalisation
        search: mp_level = PredictScale(dist3D, &CurrentFrame), kf_angle = pKF->mvKeysUn[i].angle, mp_valid = MapPoint present, not bad, not
        in sAlreadyFound, inside the bounds and the distance range; cur_occupied = CurrentFrame.mvpMapPoints[i2] != NULL.  Same window, level
        range [level - 1, level + 1], closing rule and rotation histogram as the Frame-to-Frame search, with threshold ORBdist."""
        return self.SearchByProjection(kf_mp_desc, mp_uv, mp_level, kf_angle, cur_kps, cur_desc, scale_factors, th, bounds, last_valid=mp_valid,
                                       cur_occupied=cur_occupied, direction=0, th_dist=int(ORBdist))

    def _search_in_windows(self, q_desc, q_uv, q_level, t_kps, t_desc, scale_factors, th, bounds, th_dist, claim, q_valid=None, q_ur=None,
                           t_u_right=None, t_occupied=None, inv_level_sigma2=None):
        d1 = np.ascontiguousarray(q_desc, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(t_desc, np.uint8).reshape(-1, 32)
        kp2 = np.ascontiguousarray(t_kps, KP_DTYPE)
        n1, n2 = len(d1), len(d2)
        uv = np.ascontiguousarray(q_uv, np.float32).reshape(-1, 2)
        lv = np.ascontiguousarray(q_level, np.int32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        opt = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        va, qr, ur, occ, s2 = opt(q_valid, np.uint8), opt(q_ur, np.float32), opt(t_u_right, np.float32), opt(t_occupied, np.uint8), opt(inv_level_sigma2, np.float32)
        toq = np.zeros(n1, np.int32)
        n = check(
            _lib.lib().orbm_search_in_windows(
                self._h, ptr(d1), ptr(uv), ptr(qr), ptr(lv), ptr(va), n1, ptr(d2), ptr(kp2), ptr(ur), ptr(occ), n2, ptr(sf), ptr(s2), len(sf), float(th),
                int(claim), float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]), int(th_dist), ptr(toq),
            ),
            "orbm_search_in_windows",
        )
        return n, toq

    def SearchByProjectionSim3(self, mp_desc, mp_uv, mp_level, kf_kps, kf_desc, scale_factors, th, ratioHamming, bounds, mp_valid=None, kf_matched=None):
        """SearchByProjection(KeyFrame* pKF, Sim3f& Scw, vpPoints, vpMatched, th, ratioHamming) (src/ORBmatcher.cc:423-538; :540-664 is the same
        search): kf_matched[idx] = (vpMatched[idx] != NULL) before the call.  Returns (nmatches, kf_of_point[n1])."""
        th_dist = int(np.floor(np.float32(self.TH_LOW) * np.float32(ratioHamming)))  # bestDist <= TH_LOW * ratioHamming (:516)
        return self._search_in_windows(mp_desc, mp_uv, mp_level, kf_kps, kf_desc, scale_factors, th, bounds, th_dist, 1, q_valid=mp_valid, t_occupied=kf_matched)

    def Fuse(self, mp_desc, mp_uv, mp_level, kf_kps, kf_desc, scale_factors, th, bounds, mp_valid=None, inv_level_sigma2=None, mp_ur=None, kf_u_right=None):
        """The search of Fuse(KeyFrame*, vpMapPoints, th, bRight) (src/ORBmatcher.cc:1149-1328; pass inv_level_sigma2 = mvInvLevelSigma2, and for
        stereo keyframes mp_ur / kf_u_right) and of Fuse(KeyFrame*, Sim3f&, vpPoints, th, vpReplacePoint) (:1330-1452; no gate).  Returns
        (nFused, kf_of_point[n1]); Replace / AddObservation on the result stay with the caller."""
        return self._search_in_windows(mp_desc, mp_uv, mp_level, kf_kps, kf_desc, scale_factors, th, bounds, self.TH_LOW, 0, q_valid=mp_valid, q_ur=mp_ur,
                                       t_u_right=kf_u_right, inv_level_sigma2=inv_level_sigma2)

    def SearchBySim3(self, mp1_desc, uv1in2, level1in2, valid1, kps1, desc1, mp2_desc, uv2in1, level2in1, valid2, kps2, desc2, scale_factors, th, bounds):
        """SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (src/ORBmatcher.cc:1454-1682): KF1's MapPoints searched in KF2 and KF2's in KF1 (each
        point on its own, TH_HIGH), kept where the two directions agree (:1661-1673).  Returns (nFound, idx2_of_idx1[N1])."""
        _, m1 = self._search_in_windows(mp1_desc, uv1in2, level1in2, kps2, desc2, scale_factors, th, bounds, self.TH_HIGH, 0, q_valid=valid1)
        _, m2 = self._search_in_windows(mp2_desc, uv2in1, level2in1, kps1, desc1, scale_factors, th, bounds, self.TH_HIGH, 0, q_valid=valid2)
        out = np.full(len(m1), -1, np.int32)
        ok = m1 >= 0
        ok[ok] = m2[m1[ok]] == np.nonzero(ok)[0]
        out[ok] = m1[ok]
        return int(ok.sum()), out

    def SearchForTriangulation(self, kps1, desc1, has_mp1, featvec1, kps2, desc2, has_mp2, featvec2, F12, ep, scale_factors, level_sigma2,
                               stereo1=None, stereo2=None, bOnlyStereo=False, bCoarse=False):
        """SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:916-1147) for keyframes of one pinhole
        camera: featvec = {node: [features]} (or pre-flattened groups passed as featvec1 with featvec2 = None); F12 = K1^-T [t12]x R12 K2^-1
        (3 x 3), ep = KF1's centre projected into KF2.  Returns (nmatches, vMatchedPairs as an (n, 2) array)."""
        groups = featvec1 if featvec2 is None else flatten_feature_vectors(featvec1, featvec2)
        qo, qi, to, ti = (np.ascontiguousarray(g, np.int32) for g in groups)
        kp1, kp2 = np.ascontiguousarray(kps1, KP_DTYPE), np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        opt = lambda x: None if x is None else np.ascontiguousarray(x, np.uint8)
        m1, m2, s1, s2 = opt(has_mp1), opt(has_mp2), opt(stereo1), opt(stereo2)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        e = np.ascontiguousarray(ep, np.float32).reshape(2)
        sf, sg = np.ascontiguousarray(scale_factors, np.float32), np.ascontiguousarray(level_sigma2, np.float32)
        m12 = np.zeros(len(kp1), np.int32)
        n = check(
            _lib.lib().orbm_search_for_triangulation(
                self._h, ptr(d1), ptr(kp1), ptr(m1), ptr(s1), len(kp1), ptr(d2), ptr(kp2), ptr(m2), ptr(s2), len(kp2), len(qo) - 1, ptr(qo), ptr(qi), ptr(to),
                ptr(ti), ptr(F), ptr(e), ptr(sf), ptr(sg), len(sf), int(bOnlyStereo), int(bCoarse), self.TH_LOW, int(self.mbCheckOrientation), ptr(m12),
            ),
            "orbm_search_for_triangulation",
        )
        i1 = np.nonzero(m12 >= 0)[0]
        return n, np.stack([i1, m12[i1]], 1).astype(np.int64)

    def match_bruteforce_device(self, d_q, d_q_kp, d_nq, cap_q, d_t, d_t_kp, d_nt, cap_t, d_match, d_nmatches, th=None, strict_lt=False, stream=None):
        """Device-resident brute-force SearchByBoW (one group): enqueue only."""
        return check(
            _lib.lib().orbm_match_bruteforce_device(
                self._h, ptr(d_q), ptr(d_q_kp), ptr(d_nq), cap_q, ptr(d_t), ptr(d_t_kp), ptr(d_nt), cap_t, self.TH_LOW if th is None else th, int(strict_lt),
                self.mfNNratio, int(self.mbCheckOrientation), ptr(d_match), ptr(d_nmatches), C.c_void_p(stream) if stream else None,
            ),
            "orbm_match_bruteforce_device",
        )

    def match_pairs_device(self, d_desc, d_kps, d_counts, cap, d_pairs, npairs, d_match, d_nmatches, th=None, strict_lt=False, stream=None):
        """Brute-force SearchByBoW for `npairs` (query frame, train frame) pairs of one extraction batch: enqueue only."""
        return check(
            _lib.lib().orbm_match_pairs_device(
                self._h, ptr(d_desc), ptr(d_kps), ptr(d_counts), cap, ptr(d_pairs), npairs, self.TH_LOW if th is None else th, int(strict_lt),
                self.mfNNratio, int(self.mbCheckOrientation), ptr(d_match), ptr(d_nmatches), C.c_void_p(stream) if stream else None,
            ),
            "orbm_match_pairs_device",
        )
