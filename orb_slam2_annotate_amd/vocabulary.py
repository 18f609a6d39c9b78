"""Python mirror of ``ORB_SLAM2::ORBVocabulary`` (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>,
reference: include/ORBVocabulary.h, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) for the calls next to
the hot path: loadFromTextFile and transform (Frame::ComputeBoW, src/Frame.cc:433-440)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr
from .matcher import FeatureVector


class ORBVocabulary:
    def __init__(self, device: int = 0):
        self._L = _lib.load()
        self._h = None
        self.device = device

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orbfe_vocabulary_destroy(self._h)
            self._h = None

    def loadFromTextFile(self, filename: str) -> bool:
        h = C.c_void_p()
        rc = self._L.orbfe_vocabulary_load_text(str(filename).encode(), self.device, C.byref(h))
        if rc != 0:
            return False
        if self._h:
            self._L.orbfe_vocabulary_destroy(self._h)
        self._h = h
        return True

    def createFromArrays(self, arrays) -> bool:
        """arrays = (k, L, parent[n], is_leaf[n], descriptors[n,32], weight[n]): the node lines of a text vocabulary
        (node i+1 = entry i) without the text file -- see synthetic_vocabulary_arrays."""
        k, L, parent, leaf, desc, weight = arrays
        parent = np.ascontiguousarray(parent, np.int32)
        leaf = np.ascontiguousarray(leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        weight = np.ascontiguousarray(weight, np.float64)
        h = C.c_void_p()
        rc = self._L.orbfe_vocabulary_create(int(k), int(L), 0, 0, len(parent), ptr(parent), ptr(leaf), ptr(desc), ptr(weight),
                                             self.device, C.byref(h))
        if rc != 0:
            return False
        if self._h:
            self._L.orbfe_vocabulary_destroy(self._h)
        self._h = h
        return True

    def info(self):
        k, L, n, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(self._L.orbfe_vocabulary_info(self._h, C.byref(k), C.byref(L), C.byref(n), C.byref(w)))
        return dict(k=k.value, L=L.value, nodes=n.value, words=w.value)

    def transform_features(self, descriptors, levelsup: int = 4):
        """Per-feature (word_id, weight, node_id) of transform(feature, ...)."""
        d = np.ascontiguousarray(descriptors, dtype=np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(max(n, 1), np.uint32)
        weight = np.zeros(max(n, 1), np.float64)
        node = np.zeros(max(n, 1), np.uint32)
        check(self._L.orbfe_vocabulary_transform(self._h, ptr(d), n, levelsup, ptr(word), ptr(weight), ptr(node)))
        return word[:n], weight[:n], node[:n]

    def transform(self, descriptors, levelsup: int = 4):
        """transform(features, BowVector&, FeatureVector&, levelsup): returns (bow, featvec) where bow
        is {word_id: value} (TF-IDF weights summed in feature order, L1-normalised for the L1 scoring
        the ORB vocabulary uses) and featvec a FeatureVector (CSR)."""
        word, weight, node = self.transform_features(descriptors, levelsup)
        bow = {}
        used = np.nonzero(weight > 0)[0]
        for i in used:  # BowVector::addWeight in feature order (BowVector.cpp:34-46)
            bow[int(word[i])] = bow.get(int(word[i]), 0.0) + float(weight[i])
        norm = sum(abs(x) for _, x in sorted(bow.items()))  # normalize(L1), ascending word order
        if norm > 0:
            bow = {k: x / norm for k, x in bow.items()}
        nodes = node[used]
        ids, counts = np.unique(nodes, return_counts=True)
        order = used[np.argsort(nodes, kind="stable")]
        fv = FeatureVector(ids, np.concatenate([[0], np.cumsum(counts)]), order)
        return bow, fv

    def featvec_batch_device(self, d_desc, d_n, n_frames, capacity, d_nodes, d_offsets, d_indices, d_count,
                             levelsup: int = 4, d_word=0, d_weight=0):
        check(self._L.orbfe_vocabulary_featvec_batch_device(self._h, C.c_void_p(d_desc), C.c_void_p(d_n), n_frames,
                                                            capacity, levelsup, C.c_void_p(d_nodes),
                                                            C.c_void_p(d_offsets), C.c_void_p(d_indices),
                                                            C.c_void_p(d_count), C.c_void_p(d_word or None),
                                                            C.c_void_p(d_weight or None)))

    def bow_match_consecutive_batch_device(self, n_frames, d_kp, d_desc, d_n, capacity, d_match, d_nmatches,
                                           nnratio: float = 0.7, check_orientation: bool = True, levelsup: int = 4,
                                           extractor=None, stereo: bool = False):
        """extractor given: enqueued on that extractor's stream behind its last extract call, returns at once.
        stereo: n_frames counts PAIRS of an (L0, R0, L1, R1, ...) batch and the left frames are matched (needs extractor)."""
        if stereo:
            check(self._L.orbfe_bow_match_consecutive_stereo_batch_device_async(
                self._h, extractor._h, n_frames, C.c_void_p(d_kp), C.c_void_p(d_desc), C.c_void_p(d_n), capacity,
                levelsup, float(nnratio), int(bool(check_orientation)), C.c_void_p(d_match), C.c_void_p(d_nmatches)))
            return
        if extractor is not None:
            check(self._L.orbfe_bow_match_consecutive_batch_device_async(
                self._h, extractor._h, n_frames, C.c_void_p(d_kp), C.c_void_p(d_desc), C.c_void_p(d_n), capacity,
                levelsup, float(nnratio), int(bool(check_orientation)), C.c_void_p(d_match), C.c_void_p(d_nmatches)))
            return
        check(self._L.orbfe_bow_match_consecutive_batch_device(self._h, n_frames, C.c_void_p(d_kp), C.c_void_p(d_desc),
                                                               C.c_void_p(d_n), capacity, levelsup, float(nnratio),
                                                               int(bool(check_orientation)), C.c_void_p(d_match),
                                                               C.c_void_p(d_nmatches)))


def synthetic_vocabulary_arrays(k: int = 10, L: int = 6, seed: int = 0):
    """A seeded full k-ary, L-level ORB vocabulary as arrays, breadth-first like DBoW2 saves its nodes -- k = 10,
    L = 6 is the shape of ORBvoc.txt (1 111 110 nodes below the root, 10^6 words; src/Frame.cc:438 descends it with
    levelsup = 4).  Level-1 centroids are random; a deeper child is its parent's centroid with about a quarter of the
    bits flipped (byte masks r1 & r2); leaves carry a positive IDF-like weight.  Returns (k, L, parent, is_leaf,
    descriptors, weight) for ORBVocabulary.createFromArrays / the oracle's Vocabulary.from_arrays."""
    rng = np.random.default_rng([0xB0B6, seed])
    parents, leafs, descs, weights = [], [], [], []
    prev_ids = np.zeros(1, np.int64)
    prev_desc = None
    next_id = 1
    for depth in range(1, L + 1):
        n = len(prev_ids) * k
        par = np.repeat(prev_ids, k)
        if depth == 1:
            d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        else:
            mask = rng.integers(0, 256, size=(n, 32), dtype=np.uint8) & rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
            d = np.repeat(prev_desc, k, axis=0) ^ mask
        leaf = depth == L
        parents.append(par.astype(np.int32))
        leafs.append(np.full(n, 1 if leaf else 0, np.uint8))
        descs.append(d)
        weights.append(np.round(rng.uniform(0.5, 6.0, size=n), 6) if leaf else np.zeros(n))
        prev_ids = np.arange(next_id, next_id + n, dtype=np.int64)
        prev_desc = d
        next_id += n
    return k, L, np.concatenate(parents), np.concatenate(leafs), np.concatenate(descs), np.concatenate(weights)


def write_vocabulary_text(path, arrays):
    """DBoW2 text format (TemplatedVocabulary::saveToTextFile layout) of such arrays."""
    k, L, parent, leaf, desc, weight = arrays
    with open(path, "w") as f:
        f.write(f"{k} {L} 0 0\n")  # scoring L1_NORM, weighting TF_IDF
        for i in range(len(parent)):
            f.write(f"{int(parent[i])} {int(leaf[i])} " + " ".join(map(str, desc[i].tolist())) + f" {weight[i]:.6f}\n")
    return len(parent) + 1


def write_synthetic_vocabulary(path, k: int = 10, L: int = 2, seed: int = 0, flip_bits: int = 60):
    """A seeded k-ary, L-level ORB vocabulary in DBoW2's text format (the real ORBvoc.txt is absent from
    the reference, .MISSING_LARGE_BLOBS): children are their parent's centroid with `flip_bits` random bits
    flipped, leaves carry a positive IDF-like weight.  Nodes are written breadth-first like DBoW2 saves them."""
    rng = np.random.default_rng([0xB0B0, seed])
    lines = [f"{k} {L} 0 0"]  # scoring L1_NORM, weighting TF_IDF
    level = [(0, rng.integers(0, 256, 32, dtype=np.uint8))]  # (node id, descriptor); root descriptor unused
    next_id = 1
    for depth in range(1, L + 1):
        nxt = []
        for pid, pdesc in level:
            for _ in range(k):
                bits = np.unpackbits(pdesc)
                flip = rng.choice(256, size=flip_bits if depth > 1 else 128, replace=False)
                bits[flip] ^= 1
                d = np.packbits(bits)
                leaf = 1 if depth == L else 0
                w = float(np.round(rng.uniform(0.5, 6.0), 6)) if leaf else 0.0
                lines.append(f"{pid} {leaf} " + " ".join(str(int(x)) for x in d) + f" {w:.6f}")
                nxt.append((next_id, d))
                next_id += 1
        level = nxt
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return next_id
