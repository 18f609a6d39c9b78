"""Python mirror of ``ORB_SLAM2::ORBmatcher`` (reference: include/ORBmatcher.h:38-118) for the
searches on the hot path, plus ``ComputeStereoMatches`` (src/Frame.cc:512-686).  SLAM objects
(KeyFrame/Frame/MapPoint) are replaced by the flat arrays they hold; the pointer bookkeeping
stays with the caller exactly as in INTEGRATION.md."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import FeatVecC, check, ptr


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
