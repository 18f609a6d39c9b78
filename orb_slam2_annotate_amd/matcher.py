"""Python mirror of ``ORB_SLAM2::ORBmatcher`` (reference: include/ORBmatcher.h:38-118) for the
searches on the hot path, plus ``ComputeStereoMatches`` (src/Frame.cc:512-686).  SLAM objects
(KeyFrame/Frame/MapPoint) are replaced by the flat arrays they hold; the pointer bookkeeping
stays with the caller exactly as in INTEGRATION.md."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FeatVecC, FrameViewC, check, ptr


class FeatureVector:
    """DBoW2::FeatureVector as CSR (Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-45):
    node ids ascending, each node's feature indices in insertion (ascending feature) order."""

    def __init__(self, node_ids, offsets, indices):
        self.node_ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.c = FeatVecC(len(self.node_ids), ptr(self.node_ids), ptr(self.offsets), ptr(self.indices))

    @classmethod
    def from_node_of_feature(cls, node_of_feature):
        """addFeature(node, i) for i = 0..N-1 (src/Frame.cc:433-440 via DBoW2 transform)."""
        nof = np.asarray(node_of_feature)
        ids, counts = np.unique(nof, return_counts=True)
        order = np.argsort(nof, kind="stable")
        return cls(ids, np.concatenate([[0], np.cumsum(counts)]), order)


class FrameView:
    """The fields of ``ORB_SLAM2::Frame`` the projection searches read (include/Frame.h:120-190):
    mvKeysUn as arrays, mvuRight, mDescriptors, image bounds mnMinX..mnMaxY."""

    def __init__(self, x, y, octave, desc, bounds, angle=None, u_right=None):
        self.x, self.y = _f32(x), _f32(y)
        self.octave = np.ascontiguousarray(octave, dtype=np.int32)
        self.desc = _u8(desc).reshape(-1, 32)
        self.angle = None if angle is None else _f32(angle)
        self.u_right = None if u_right is None else _f32(u_right)
        self.N = len(self.x)
        self.bounds = tuple(float(b) for b in bounds)  # (mnMinX, mnMaxX, mnMinY, mnMaxY)
        self.c = FrameViewC(self.N, ptr(self.x), ptr(self.y), ptr(self.octave),
                            None if self.angle is None else ptr(self.angle),
                            None if self.u_right is None else ptr(self.u_right), ptr(self.desc), *self.bounds, None)

    def upload(self, fv: "FeatureVector | None" = None, device: int = 0) -> "ResidentFrame":
        """orbfe_frame_upload: keypoint arrays, descriptors, grid (built once) and the FeatureVector's index list move to
        the device; the returned handle can stand in for this view in every search and feeds the *_resident / *_multi ones."""
        return ResidentFrame(self, fv, device)

    @classmethod
    def from_keypoints(cls, kps, desc, width, height, u_right=None):
        """kps: structured/[:,7] keypoint records as the extractor returns them; undistorted bounds of a
        distortion-free camera (src/Frame.cc:719-725)."""
        k = np.asarray(kps)
        if k.dtype.names:
            x, y, ang, octv = k["x"], k["y"], k["angle"], k["octave"]
        else:
            x, y, ang, octv = k[:, 0], k[:, 1], k[:, 3], k[:, 5]
        return cls(x, y, octv, desc, (0.0, float(width), 0.0, float(height)), angle=ang, u_right=u_right)

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1, capacity: int = 64, device: int = 0):
        """Frame::GetFeaturesInArea (src/Frame.cc:358-415) for arrays of windows ->
        list of index arrays in the reference's order."""
        x, y, r = _f32(np.atleast_1d(x)), _f32(np.atleast_1d(y)), _f32(np.atleast_1d(r))
        nq = len(x)
        lo = np.ascontiguousarray(np.broadcast_to(np.asarray(minLevel, dtype=np.int32), (nq,)))
        hi = np.ascontiguousarray(np.broadcast_to(np.asarray(maxLevel, dtype=np.int32), (nq,)))
        while True:
            count = np.zeros(max(nq, 1), dtype=np.int32)
            idx = np.zeros((max(nq, 1), max(capacity, 1)), dtype=np.int32)
            rc = _lib.load().orbfe_features_in_area(device, C.byref(self.c), nq, ptr(x), ptr(y), ptr(r), ptr(lo),
                                                    ptr(hi), capacity, ptr(count), ptr(idx))
            if rc == _lib.ERR_CAPACITY:
                capacity = int(count.max())
                continue
            check(rc)
            return [idx[q, :count[q]].copy() for q in range(nq)]


class ResidentFrame:
    """Device-resident operands of a Frame / KeyFrame (include/orbfe.h orbfe_frame).  `.c` is the handle's own
    orbfe_frame_view (host copies inside the handle, `resident` set), so an instance is accepted wherever a FrameView is."""

    XY_FROM_VIEW = 1  # ORBFE_FRAME_XY_FROM_VIEW

    def __init__(self, view: FrameView, fv=None, device: int = 0, extractor=None, frame: int = 0, d_keypoints: int = 0,
                 d_descriptors: int = 0, flags: int = 0):
        """default: orbfe_frame_upload (everything from the host arrays of `view`).  extractor=...: orbfe_frame_from_extractor
        (records and descriptors of frame `frame` of the handle's last host-buffer call, still in HBM); d_keypoints /
        d_descriptors (device addresses in the extractor's output layout): orbfe_frame_from_device."""
        self._L = _lib.load()
        self._handle = C.c_void_p()
        fvp = C.byref(fv.c) if fv is not None else None
        if extractor is not None:
            check(self._L.orbfe_frame_from_extractor(extractor._h, int(frame), C.byref(view.c), fvp, int(flags), C.byref(self._handle)))
        elif d_keypoints:
            check(self._L.orbfe_frame_from_device(device, C.c_void_p(d_keypoints), C.c_void_p(d_descriptors), C.byref(view.c), fvp,
                                                  int(flags), C.byref(self._handle)))
        else:
            check(self._L.orbfe_frame_upload(device, C.byref(view.c), fvp, C.byref(self._handle)))
        self._view = self._L.orbfe_frame_get_view(self._handle).contents  # aliases memory INSIDE the handle
        self.N, self.device = view.N, device
        # the searches' Python wrappers read these for the outputs' shapes only
        self.x, self.y, self.octave, self.angle, self.u_right, self.desc, self.bounds = (view.x, view.y, view.octave, view.angle,
                                                                                       view.u_right, view.desc, view.bounds)

    # `.c` (the view) and `._h` (the handle) are what every wrapper hands to the library: after close() they point into
    # freed memory, so they raise instead (round-3 ADVICE: a closed frame was a use-after-free, not a Python error)
    @property
    def c(self):
        if self._view is None:
            raise ValueError("ResidentFrame is closed")
        return self._view

    @property
    def _h(self):
        if self._handle is None:
            raise ValueError("ResidentFrame is closed")
        return self._handle

    @property
    def closed(self):
        return self._handle is None

    def set_featvec(self, fv):
        """Frame::ComputeBoW after the constructor (src/Tracking.cc:836-843): attach the FeatureVector."""
        check(self._L.orbfe_frame_set_featvec(self._h, C.byref(fv.c)))

    def close(self):
        if getattr(self, "_handle", None):
            self._L.orbfe_frame_release(self._handle)
        self._handle = None
        self._view = None

    __del__ = close
    GetFeaturesInArea = FrameView.GetFeaturesInArea  # (reads self.c only: no frame data travels)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class ORBmatcher:
    TH_HIGH = 100  # src/ORBmatcher.cc:37-39
    TH_LOW = 50
    HISTO_LENGTH = 30

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True, device: int = 0):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self.device = device
        self._L = _lib.load()

    @staticmethod
    def DescriptorDistance(a, b, device: int = 0):
        """src/ORBmatcher.cc:1828-1844.  a, b: [32] or [n,32] uint8."""
        a = _u8(a).reshape(-1, 32)
        b = _u8(b).reshape(-1, 32)
        out = np.zeros(len(a), dtype=np.int32)
        check(_lib.load().orbfe_descriptor_distance(device, ptr(a), ptr(b), len(a), ptr(out)))
        return int(out[0]) if len(out) == 1 else out

    @staticmethod
    def HammingMatrix(d1, d2, device: int = 0):
        d1 = _u8(d1).reshape(-1, 32)
        d2 = _u8(d2).reshape(-1, 32)
        out = np.zeros((len(d1), len(d2)), dtype=np.int32)
        check(_lib.load().orbfe_hamming_matrix(device, ptr(d1), len(d1), ptr(d2), len(d2), ptr(out)))
        return out

    def SearchByBoW(self, desc1, has_mp1, angle1, fv1: FeatureVector, desc2, angle2, fv2: FeatureVector,
                    has_mp2=None):
        """KF-Frame form (src/ORBmatcher.cc:185-325) when has_mp2 is None -> (nmatches, match_f[n2]);
        KF-KF form (:610-743) otherwise -> (nmatches, match12[n1])."""
        desc1 = _u8(desc1).reshape(-1, 32)
        desc2 = _u8(desc2).reshape(-1, 32)
        n1, n2 = len(desc1), len(desc2)
        m1, a1, a2 = _u8(has_mp1), _f32(angle1), _f32(angle2)
        if has_mp2 is None:
            out = np.full(max(n2, 1), -1, dtype=np.int32)
            n = check(self._L.orbfe_search_by_bow(self.device, ptr(desc1), ptr(m1), ptr(a1), n1, C.byref(fv1.c),
                                                  ptr(desc2), ptr(a2), n2, C.byref(fv2.c), self.mfNNratio,
                                                  int(self.mbCheckOrientation), ptr(out)))
            return n, out[:n2]
        m2 = _u8(has_mp2)
        out = np.full(max(n1, 1), -1, dtype=np.int32)
        n = check(self._L.orbfe_search_by_bow_kf(self.device, ptr(desc1), ptr(m1), ptr(a1), n1, C.byref(fv1.c),
                                                 ptr(desc2), ptr(m2), ptr(a2), n2, C.byref(fv2.c),
                                                 self.mfNNratio, int(self.mbCheckOrientation), ptr(out)))
        return n, out[:n1]

    def SearchByBoWResident(self, kf: "ResidentFrame", has_mp1, f: "ResidentFrame", has_mp2=None):
        """SearchByBoW on frames uploaded with their FeatureVector: KF-Frame form when has_mp2 is None, else KF-KF."""
        m1 = _u8(has_mp1)
        if has_mp2 is None:
            out = np.full(max(f.N, 1), -1, dtype=np.int32)
            n = check(self._L.orbfe_search_by_bow_resident(kf._h, ptr(m1), f._h, self.mfNNratio, int(self.mbCheckOrientation), ptr(out)))
            return n, out[:f.N]
        m2 = _u8(has_mp2)
        out = np.full(max(kf.N, 1), -1, dtype=np.int32)
        n = check(self._L.orbfe_search_by_bow_kf_resident(kf._h, ptr(m1), f._h, ptr(m2), self.mfNNratio,
                                                          int(self.mbCheckOrientation), ptr(out)))
        return n, out[:kf.N]

    def SearchByBoWMulti(self, kfs, has_mp_kfs, f: "ResidentFrame"):
        """Tracking::Relocalization (src/Tracking.cc:1478-1498): SearchByBoW(pKF_k, mCurrentFrame) for every candidate key
        frame in ONE call -> (n_matches[K], match_f[K, f.N])."""
        K = len(kfs)
        masks = [_u8(m) for m in has_mp_kfs]
        hs = (C.c_void_p * max(K, 1))(*[k._h for k in kfs])
        ms = (C.c_void_p * max(K, 1))(*[m.ctypes.data for m in masks])
        out = np.full((max(K, 1), max(f.N, 1)), -1, dtype=np.int32)
        cnt = np.zeros(max(K, 1), dtype=np.int32)
        check(self._L.orbfe_search_by_bow_multi(K, hs, ms, f._h, self.mfNNratio, int(self.mbCheckOrientation), ptr(out), ptr(cnt)))
        return cnt[:K], out[:K, :f.N]

    def SearchByBoWKFMulti(self, kf1: "ResidentFrame", has_mp1, kf2s, has_mp2s):
        """LoopClosing::ComputeSim3 (src/LoopClosing.cc:294-321): SearchByBoW(mpCurrentKF, pKF_k) for every candidate in ONE
        call -> (n_matches[K], match12[K, kf1.N])."""
        K = len(kf2s)
        m1 = _u8(has_mp1)
        masks = [_u8(m) for m in has_mp2s]
        hs = (C.c_void_p * max(K, 1))(*[k._h for k in kf2s])
        ms = (C.c_void_p * max(K, 1))(*[m.ctypes.data for m in masks])
        out = np.full((max(K, 1), max(kf1.N, 1)), -1, dtype=np.int32)
        cnt = np.zeros(max(K, 1), dtype=np.int32)
        check(self._L.orbfe_search_by_bow_kf_multi(kf1._h, ptr(m1), K, hs, ms, self.mfNNratio, int(self.mbCheckOrientation),
                                                   ptr(out), ptr(cnt)))
        return cnt[:K], out[:K, :kf1.N]

    def SearchByProjectionKeyFrameMulti(self, Cur: FrameView, scale_factors, candidates):
        """SearchByProjection(CurrentFrame, pKF_k, sFound, th_k, ORBdist_k) (src/Tracking.cc:1577,1595) for K candidates in ONE
        call.  candidates: list of dicts with valid, u, v, level, kf_angle, mp_desc, th, ORBdist and optionally blocked ->
        (n_matches[K], match_cur[K, Cur.N])."""
        K = len(candidates)
        sf = _f32(scale_factors)
        keep = []  # the arrays must outlive the call

        def col(key, conv, optional=False):
            arr = []
            for c in candidates:
                a = c.get(key)
                if a is None:
                    if not optional:
                        raise ValueError(key)
                    arr.append(None)
                else:
                    a = conv(a)
                    keep.append(a)
                    arr.append(a.ctypes.data)
            return (C.c_void_p * max(K, 1))(*arr)

        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        va, uu, vv, lv = col("valid", _u8), col("u", _f32), col("v", _f32), col("level", i32)
        ka, md = col("kf_angle", _f32), col("mp_desc", lambda a: _u8(a).reshape(-1, 32))
        blk = col("blocked", _u8, optional=True)
        n = np.array([len(_u8(c["valid"])) for c in candidates] or [0], dtype=np.int32)
        th = np.array([c["th"] for c in candidates] or [0], dtype=np.float32)
        od = np.array([c["ORBdist"] for c in candidates] or [0], dtype=np.int32)
        match = np.full((max(K, 1), max(Cur.N, 1)), -1, dtype=np.int32)
        cnt = np.zeros(max(K, 1), dtype=np.int32)
        check(self._L.orbfe_search_by_projection_keyframe_multi(
            self.device, C.byref(Cur.c), ptr(sf), len(sf), K, blk, ptr(n), va, uu, vv, lv, ka, md, ptr(th), ptr(od),
            int(self.mbCheckOrientation), ptr(match), ptr(cnt)))
        return cnt[:K], match[:K, :Cur.N]

    def SearchForTriangulationMulti(self, kf1: "ResidentFrame", has_mp1, neighbours, has_mp2_list, F12s, epipoles,
                                    scale_factors2, level_sigma2_2, only_stereo: bool = False):
        """SearchForTriangulation of kf1 against every neighbour in one call (LocalMapping::CreateNewMapPoints):
        returns (n_matches[K], match12[K, n1])."""
        K, n1 = len(neighbours), kf1.N
        m1 = _u8(has_mp1)
        masks = [_u8(m) for m in has_mp2_list]
        hs = (C.c_void_p * max(K, 1))(*[nb._h for nb in neighbours])
        ms = (C.c_void_p * max(K, 1))(*[m.ctypes.data for m in masks])
        F = _f32(np.asarray(F12s, dtype=np.float32).reshape(K, 9))
        ep = _f32(np.asarray(epipoles, dtype=np.float32).reshape(K, 2))
        ex, ey = _f32(ep[:, 0]), _f32(ep[:, 1])
        sf, sg = _f32(scale_factors2), _f32(level_sigma2_2)
        out = np.full((max(K, 1), max(n1, 1)), -1, dtype=np.int32)
        cnt = np.zeros(max(K, 1), dtype=np.int32)
        check(self._L.orbfe_search_for_triangulation_multi(kf1._h, ptr(m1), K, hs, ms, ptr(F), ptr(ex), ptr(ey), ptr(sf), ptr(sg),
                                                           len(sf), int(bool(only_stereo)), int(self.mbCheckOrientation),
                                                           ptr(out), ptr(cnt)))
        return cnt[:K], out[:K, :n1]

    def FuseSearchMulti(self, KFs, scale_factors, valid, u, v, level, mp_desc, th: float = 3.0, inv_level_sigma2=None,
                        ur=None):
        """The per-point search of Fuse for the same map points against K key frames (views or resident frames):
        valid / u / v / level / ur are [K, n]; returns best_idx[K, n]."""
        K = len(KFs)
        valid = _u8(np.asarray(valid).reshape(K, -1))
        n = valid.shape[1] if K else 0
        u, v = _f32(np.asarray(u).reshape(K, n)), _f32(np.asarray(v).reshape(K, n))
        level = np.ascontiguousarray(np.asarray(level).reshape(K, n), dtype=np.int32)
        sf = _f32(scale_factors)
        d = _u8(mp_desc).reshape(-1, 32)
        gate = inv_level_sigma2 is not None
        isg = _f32(inv_level_sigma2) if gate else None
        urr = _f32(np.asarray(ur).reshape(K, n)) if ur is not None else None
        views = (C.POINTER(FrameViewC) * max(K, 1))(*[C.pointer(k.c) for k in KFs])
        out = np.full((max(K, 1), max(n, 1)), -1, dtype=np.int32)
        check(self._L.orbfe_fuse_search_multi(self.device, K, views, ptr(sf), ptr(isg) if gate else None, len(sf), n, ptr(valid),
                                              ptr(u), ptr(v), ptr(urr) if urr is not None else None, ptr(level), ptr(d),
                                              float(th), int(gate), ptr(out)))
        return out[:K, :n]

    def SearchByProjection(self, F: FrameView, scale_factors, in_view, level, view_cos, proj_x, proj_y, mp_desc,
                           th: float = 1.0, proj_xr=None, blocked=None, mp_obs_positive=None):
        """SearchByProjection(Frame&, vector<MapPoint*>&, th) (src/ORBmatcher.cc:51-138) ->
        (nmatches, match[F.N]) with match[idx] = map point index or -1."""
        sf = _f32(scale_factors)
        iv, lv, vc = _u8(in_view), np.ascontiguousarray(level, dtype=np.int32), _f32(view_cos)
        px, py, md = _f32(proj_x), _f32(proj_y), _u8(mp_desc).reshape(-1, 32)
        pxr = None if proj_xr is None else _f32(proj_xr)
        blk = None if blocked is None else _u8(blocked)
        obs = None if mp_obs_positive is None else _u8(mp_obs_positive)
        match = np.full(max(F.N, 1), -1, dtype=np.int32)
        n = C.c_int32(0)
        check(self._L.orbfe_search_by_projection(
            self.device, C.byref(F.c), ptr(sf), len(sf), None if blk is None else ptr(blk), len(iv), ptr(iv), ptr(lv),
            ptr(vc), ptr(px), ptr(py), None if pxr is None else ptr(pxr), ptr(md), None if obs is None else ptr(obs),
            float(th), self.mfNNratio, ptr(match), C.byref(n)))
        return n.value, match[:F.N]

    def SearchByProjectionLastFrame(self, Cur: FrameView, scale_factors, valid, u, v, last_octave, last_angle,
                                    mp_desc, th: float, mode: int = 0, mbf: float = 0.0, invzc=None,
                                    obs_positive=None, blocked=None):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1484-1633) after the
        caller's projection -> (nmatches, match_cur[Cur.N]) with match_cur[i2] = last-frame index or -1."""
        sf = _f32(scale_factors)
        va, uu, vv = _u8(valid), _f32(u), _f32(v)
        lo, la, md = np.ascontiguousarray(last_octave, dtype=np.int32), _f32(last_angle), _u8(mp_desc).reshape(-1, 32)
        iz = None if invzc is None else _f32(invzc)
        obs = None if obs_positive is None else _u8(obs_positive)
        blk = None if blocked is None else _u8(blocked)
        match = np.full(max(Cur.N, 1), -1, dtype=np.int32)
        n = C.c_int32(0)
        check(self._L.orbfe_search_by_projection_last_frame(
            self.device, C.byref(Cur.c), ptr(sf), len(sf), float(mbf), len(va), ptr(va), ptr(uu), ptr(vv),
            None if iz is None else ptr(iz), ptr(lo), ptr(la), ptr(md), None if obs is None else ptr(obs),
            None if blk is None else ptr(blk), int(mode),
            float(th), int(self.mbCheckOrientation), ptr(match), C.byref(n)))
        return n.value, match[:Cur.N]

    def SearchByProjectionKeyFrame(self, Cur: FrameView, scale_factors, valid, u, v, level, kf_angle, mp_desc,
                                   th: float, ORBdist: int, blocked=None):
        """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1641-1775)
        after the caller's projection -> (nmatches, match_cur[Cur.N])."""
        sf, va, uu, vv = _f32(scale_factors), _u8(valid), _f32(u), _f32(v)
        lv, ka, md = np.ascontiguousarray(level, dtype=np.int32), _f32(kf_angle), _u8(mp_desc).reshape(-1, 32)
        blk = None if blocked is None else _u8(blocked)
        match = np.full(max(Cur.N, 1), -1, dtype=np.int32)
        n = C.c_int32(0)
        check(self._L.orbfe_search_by_projection_keyframe(
            self.device, C.byref(Cur.c), ptr(sf), len(sf), ptr(blk), len(va), ptr(va), ptr(uu), ptr(vv), ptr(lv),
            ptr(ka), ptr(md), float(th), int(ORBdist), int(self.mbCheckOrientation), ptr(match), C.byref(n)))
        return n.value, match[:Cur.N]

    def SearchByProjectionSim3(self, KF: FrameView, scale_factors, valid, u, v, level, mp_desc, th: float,
                               matched=None):
        """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:335-449) after the
        caller's projection -> (nmatches, match[KF.N]) (new matches only)."""
        sf, va, uu, vv = _f32(scale_factors), _u8(valid), _f32(u), _f32(v)
        lv, md = np.ascontiguousarray(level, dtype=np.int32), _u8(mp_desc).reshape(-1, 32)
        mt = None if matched is None else _u8(matched)
        match = np.full(max(KF.N, 1), -1, dtype=np.int32)
        n = C.c_int32(0)
        check(self._L.orbfe_search_by_projection_sim3(self.device, C.byref(KF.c), ptr(sf), len(sf), ptr(mt), len(va),
                                                      ptr(va), ptr(uu), ptr(vv), ptr(lv), ptr(md), float(th),
                                                      ptr(match), C.byref(n)))
        return n.value, match[:KF.N]

    def SearchForInitialization(self, F1: FrameView, F2: FrameView, vbPrevMatched, windowSize: int = 10):
        """src/ORBmatcher.cc:469-603 -> (nmatches, vnMatches12[F1.N]); vbPrevMatched ([N1,2] float32) is
        updated in place."""
        prev = np.asarray(vbPrevMatched)
        px, py = _f32(prev[:, 0]).copy(), _f32(prev[:, 1]).copy()
        match = np.full(max(F1.N, 1), -1, dtype=np.int32)
        n = C.c_int32(0)
        check(self._L.orbfe_search_for_initialization(self.device, C.byref(F1.c), C.byref(F2.c), ptr(px), ptr(py),
                                                      int(windowSize), self.mfNNratio, int(self.mbCheckOrientation),
                                                      ptr(match), C.byref(n)))
        prev[:, 0], prev[:, 1] = px, py
        return n.value, match[:F1.N]

    def FuseSearch(self, KF: FrameView, scale_factors, valid, u, v, level, mp_desc, th: float = 3.0,
                   inv_level_sigma2=None, ur=None):
        """The per-map-point search of Fuse (src/ORBmatcher.cc:940-1110; the Sim3 overload :1112-1249 when
        inv_level_sigma2 is None) -> bestIdx[n] (-1: no keypoint within TH_LOW)."""
        sf, va, uu, vv = _f32(scale_factors), _u8(valid), _f32(u), _f32(v)
        lv, md = np.ascontiguousarray(level, dtype=np.int32), _u8(mp_desc).reshape(-1, 32)
        sg = None if inv_level_sigma2 is None else _f32(inv_level_sigma2)
        r = None if ur is None else _f32(ur)
        best = np.full(max(len(va), 1), -1, dtype=np.int32)
        check(self._L.orbfe_fuse_search(self.device, C.byref(KF.c), ptr(sf), ptr(sg), len(sf), len(va), ptr(va),
                                        ptr(uu), ptr(vv), ptr(r), ptr(lv), ptr(md), float(th), int(sg is not None),
                                        ptr(best)))
        return best[:len(va)]

    def SearchBySim3(self, KF1: FrameView, KF2: FrameView, sf1, sf2, valid1, u1, v1, level1, desc1, valid2, u2, v2,
                     level2, desc2, th: float):
        """src/ORBmatcher.cc:1251-1482 after the caller's two projections -> (nFound, match12[KF1.N])."""
        sf1, sf2 = _f32(sf1), _f32(sf2)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        a = [_u8(valid1), _f32(u1), _f32(v1), i32(level1), _u8(desc1), _u8(valid2), _f32(u2), _f32(v2), i32(level2),
             _u8(desc2)]
        match = np.full(max(KF1.N, 1), -1, dtype=np.int32)
        n = C.c_int32(0)
        check(self._L.orbfe_search_by_sim3(self.device, C.byref(KF1.c), C.byref(KF2.c), ptr(sf1), ptr(sf2), len(sf1),
                                           *[ptr(x) for x in a], float(th), ptr(match), C.byref(n)))
        return n.value, match[:KF1.N]

    def SearchForTriangulation(self, desc1, has_mp1, x1, y1, angle1, stereo1, fv1, desc2, has_mp2, x2, y2,
                               angle2, octave2, stereo2, fv2, F12, ex, ey, scale_factors2, level_sigma2_2,
                               bOnlyStereo=False):
        """src/ORBmatcher.cc:754-928 -> (nmatches, vMatchedPairs[k,2] ascending in idx1)."""
        desc1 = _u8(desc1).reshape(-1, 32)
        desc2 = _u8(desc2).reshape(-1, 32)
        n1, n2 = len(desc1), len(desc2)
        a = [_u8(has_mp1), _f32(x1), _f32(y1), _f32(angle1), _u8(stereo1)]
        b = [_u8(has_mp2), _f32(x2), _f32(y2), _f32(angle2), np.ascontiguousarray(octave2, dtype=np.int32),
             _u8(stereo2)]
        F = _f32(F12).reshape(9)
        sf, sg = _f32(scale_factors2), _f32(level_sigma2_2)
        out = np.full(max(n1, 1), -1, dtype=np.int32)
        n = check(self._L.orbfe_search_for_triangulation(
            self.device, ptr(desc1), *[ptr(v) for v in a], n1, C.byref(fv1.c), ptr(desc2), *[ptr(v) for v in b],
            n2, C.byref(fv2.c), ptr(F), float(ex), float(ey), ptr(sf), ptr(sg), len(sf), int(bool(bOnlyStereo)),
            int(self.mbCheckOrientation), ptr(out)))
        out = out[:n1]
        idx = np.nonzero(out >= 0)[0]
        return n, np.stack([idx, out[idx]], axis=1) if len(idx) else np.zeros((0, 2), dtype=np.int64)


def ComputeStereoMatches(extractorLeft, extractorRight, kpL, descL, kpR, descR, mbf: float, mb: float,
                         frameL: int = 0, frameR: int = 0):
    """Frame::ComputeStereoMatches (src/Frame.cc:512-686) -> (mvuRight, mvDepth).  The two extractor
    handles must hold the pyramids of the frames the keypoints came from (their last call)."""
    L = _lib.load()
    kpL = np.ascontiguousarray(kpL, dtype=_lib.KP_DTYPE)
    kpR = np.ascontiguousarray(kpR, dtype=_lib.KP_DTYPE)
    descL = _u8(descL).reshape(-1, 32)
    descR = _u8(descR).reshape(-1, 32)
    N, Nr = len(kpL), len(kpR)
    u = np.full(max(N, 1), -1, dtype=np.float32)
    d = np.full(max(N, 1), -1, dtype=np.float32)
    check(L.orbfe_compute_stereo_matches(extractorLeft._h, frameL, extractorRight._h, frameR, ptr(kpL), ptr(descL),
                                         N, ptr(kpR), ptr(descR), Nr, float(mbf), float(mb), ptr(u), ptr(d)))
    return u[:N], d[:N]
