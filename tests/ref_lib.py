"""ctypes binding of oracle/_ref/libdbow2_ref.so: the REFERENCE's own DBoW2 FeatureVector / BowVector
(Thirdparty/DBoW2/DBoW2/{FeatureVector,BowVector}.cpp compiled in place by `make -C oracle ref`).
Test infrastructure only.  In the build container the library is (re)built on demand; on the GPU box
/root/reference does not exist and the prebuilt file that travelled with the snapshot is used."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
SO = ROOT / "oracle" / "_ref" / "libdbow2_ref.so"
REF_SRC = Path("/root/reference/Thirdparty/DBoW2/DBoW2/FeatureVector.cpp")
_lib = None


def available() -> bool:
    return SO.exists() or REF_SRC.exists()


def lib():
    global _lib
    if _lib is None:
        if REF_SRC.exists():
            subprocess.run(["make", "-C", str(ROOT / "oracle"), "ref"], check=True, capture_output=True)
        _lib = C.CDLL(str(SO))
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def featvec(node_of_feature):
    """(node_ids, offsets, indices) of the reference's FeatureVector after addFeature(node[i], i), i = 0..n-1."""
    nof = np.ascontiguousarray(node_of_feature, dtype=np.uint32)
    n = len(nof)
    nodes = np.zeros(max(n, 1), np.uint32)
    offs = np.zeros(n + 1, np.int32)
    idx = np.zeros(max(n, 1), np.uint32)
    k = lib().ref_featvec_build(_p(nof), n, _p(nodes), _p(offs), _p(idx))
    return nodes[:k].copy(), offs[:k + 1].copy(), idx[:n].copy()


def bowvec(word, weight, l1_normalize=True):
    word = np.ascontiguousarray(word, dtype=np.uint32)
    weight = np.ascontiguousarray(weight, dtype=np.float64)
    n = len(word)
    ids = np.zeros(max(n, 1), np.uint32)
    val = np.zeros(max(n, 1), np.float64)
    k = lib().ref_bowvec_build(_p(word), _p(weight), n, int(bool(l1_normalize)), _p(ids), _p(val))
    return ids[:k].copy(), val[:k].copy()
