"""The OpenCV-free C++ host layer (include/orbfe_classes.hpp) compiled with g++ against
liborbfe.so: two extractor instances on two threads + stereo matching, checked against the oracle."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _fnv(b: bytes, h=1469598103934665603):
    for x in b:
        h ^= x
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_classes_match_oracle(tmp_path):
    z = np.load(ROOT / "tests/golden/stereo_320x200.npz")
    L, R = z["left"], z["right"]
    (tmp_path / "l.raw").write_bytes(L.tobytes())
    (tmp_path / "r.raw").write_bytes(R.tobytes())
    exe = tmp_path / "test_classes"
    lib = ROOT / "orb_slam2_annotate_amd"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", f"-I{ROOT / 'include'}", str(ROOT / "tests/cpp/test_classes.cpp"),
                    "-o", str(exe), f"-L{lib}", "-lorbfe", f"-Wl,-rpath,{lib}"], check=True)
    out = subprocess.run([str(exe), str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "320", "200"], check=True,
                         capture_output=True, text=True).stdout
    kv = dict(t.split("=") for t in out.split())  # three lines of key=value tokens
    o = orc.Oracle(600, 1.2, 8, 20, 7)
    kL, dL, pL = o.extract(L, want_pyramid=True)
    kR, dR, pR = o.extract(R, want_pyramid=True)
    assert int(kv["nL"]) == len(kL) and int(kv["nR"]) == len(kR)
    assert int(kv["kp"], 16) == _fnv(kL.tobytes())
    assert int(kv["desc"], 16) == _fnv(dL.tobytes())
    assert int(kv["u"], 16) == _fnv(z["uRight"].tobytes())
    assert int(kv["d"], 16) == _fnv(z["depth"].tobytes())
    assert int(kv["pyr"], 16) == _fnv(pL.tobytes())
    assert int(kv["levels"]) == 8
    assert int(kv["dist"]) == orc.descriptor_distance(dL[0], dL[1])
    # tracking-thread searches through the C++ layer == oracle on the same keypoints
    b = (0.0, 320.0, 0.0, 200.0)
    F1 = orc.Frame(kL["x"], kL["y"], kL["octave"], dL, b, angle=kL["angle"])
    F2 = orc.Frame(kR["x"], kR["y"], kR["octave"], dR, b, angle=kR["angle"])
    prev = np.stack([kL["x"], kL["y"]], axis=1).astype(np.float32)
    n_init, m12, _ = orc.search_for_initialization(F1, F2, prev, 100, 0.9, True)
    assert int(kv["ninit"]) == n_init and int(kv["init"], 16) == _fnv(m12.astype(np.int32).tobytes())
    sf = np.array(o.scale_factors(), np.float32)
    n_proj, mc = orc.search_by_projection_lastframe(F2, sf, 0.0, np.ones(len(kL), np.uint8), kL["x"], kL["y"], None,
                                                    kL["octave"], kL["angle"], dL, None, 0, 15.0, True)
    assert int(kv["nproj"]) == n_proj and int(kv["proj"], 16) == _fnv(mc.astype(np.int32).tobytes())
    area = F2.features_in_area(160.0, 100.0, 40.0, 0, 2)
    assert int(kv["narea"]) == len(area) and int(kv["area"], 16) == _fnv(area.astype(np.int32).tobytes())
    assert n_init > 0 and n_proj > 0
    # the six searches the C++ class gained in round 2 (same derived inputs as tests/cpp/test_classes.cpp)
    sig2, isig2 = np.array(o.level_sigma2(), np.float32), np.array(o.inv_level_sigma2(), np.float32)
    iL, iR = np.arange(len(kL)), np.arange(len(kR))
    urL = np.where(iL % 5 == 0, kL["x"] - np.float32(3.0), np.float32(-1.0)).astype(np.float32)
    urR = np.where(iR % 7 == 0, kR["x"] - np.float32(2.0), np.float32(-1.0)).astype(np.float32)
    has1, has2 = (iL % 3 == 0).astype(np.uint8), (iR % 4 == 0).astype(np.uint8)
    fv1 = orc.FeatVec(((dL[:, 0] ^ dL[:, 7]) % 37).astype(np.uint32) * 5 + 2)
    fv2 = orc.FeatVec(((dR[:, 0] ^ dR[:, 7]) % 37).astype(np.uint32) * 5 + 2)
    F12 = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32)
    tri = lambda only: orc.search_for_triangulation(dL, has1, kL["x"], kL["y"], kL["angle"], urL >= 0, fv1, dR, has2,
                                                    kR["x"], kR["y"], kR["angle"], kR["octave"], urR >= 0, fv2, F12,
                                                    1000.0, 100.0, sf, sig2, only, False)
    n_tri, m_tri = tri(False)
    idx = np.nonzero(m_tri >= 0)[0]
    flat = np.stack([idx, m_tri[idx]], axis=1).astype(np.int32)
    assert int(kv["ntri"]) == n_tri > 0 and int(kv["tri"], 16) == _fnv(flat.tobytes())
    assert int(kv["ntris"]) == tri(True)[0]
    M2 = orc.Frame(kR["x"], kR["y"], kR["octave"], dR, b, angle=kR["angle"])
    K1 = orc.Frame(kL["x"], kL["y"], kL["octave"], dL, b, angle=kL["angle"], u_right=urL)
    K2 = orc.Frame(kR["x"], kR["y"], kR["octave"], dR, b, angle=kR["angle"], u_right=urR)
    valid = (iL % 11 != 0).astype(np.uint8)
    u, v = (kL["x"] - np.float32(2.0)).astype(np.float32), kL["y"]
    u2, v2 = (kR["x"] + np.float32(2.0)).astype(np.float32), kR["y"]
    n_rel, m_rel = orc.search_by_projection_reloc(M2, sf, valid, u, v, kL["octave"], kL["angle"], dL,
                                                  np.zeros(len(kR), np.uint8), 12.0, 100, True)
    assert int(kv["nreloc"]) == n_rel > 0 and int(kv["reloc"], 16) == _fnv(m_rel.astype(np.int32).tobytes())
    n_sim, m_sim = orc.search_by_projection_sim3(M2, sf, valid, u, v, kL["octave"], dL, np.zeros(len(kR), np.uint8), 10.0)
    assert int(kv["nsim"]) == n_sim > 0 and int(kv["sim"], 16) == _fnv(m_sim.astype(np.int32).tobytes())
    fa = orc.fuse_search(M2, sf, isig2, valid, u, v, np.full(len(kL), -1.0, np.float32), kL["octave"], dL, 12.0, 1)
    fb = orc.fuse_search(M2, sf, isig2, valid, u, v, np.full(len(kL), -1.0, np.float32), kL["octave"], dL, 12.0, 0)
    assert int(kv["fusea"], 16) == _fnv(fa.astype(np.int32).tobytes()) and (fa >= 0).sum() > 0
    assert int(kv["fuseb"], 16) == _fnv(fb.astype(np.int32).tobytes()) and (fb >= 0).sum() >= (fa >= 0).sum()
    n_s3, m_s3 = orc.search_by_sim3(K1, K2, sf, sf, valid, u, v, kL["octave"], dL, np.ones(len(kR), np.uint8), u2, v2,
                                    kR["octave"], dR, 12.0)
    assert int(kv["ns3"]) == n_s3 > 0 and int(kv["s3"], 16) == _fnv(m_s3.astype(np.int32).tobytes())
    # SearchByBoW x2 through the C++ class (host arrays; the resident and threaded variants are compared with these inside
    # the C++ program: tests/cpp/test_classes.cpp "resident frames, the multi-neighbour calls and RE-ENTRANCY")
    n_bow, m_bow = orc.search_by_bow(dL, np.ones(len(kL), np.uint8), kL["angle"], fv1, dR, kR["angle"], fv2, 0.7, True)
    assert int(kv["nbow"]) == n_bow > 0 and int(kv["bow"], 16) == _fnv(m_bow.astype(np.int32).tobytes())
    n_bkf, m_bkf = orc.search_by_bow_kf(dL, has1, kL["angle"], fv1, dR, has2, kR["angle"], fv2, 0.7, True)
    assert int(kv["nbowkf"]) == n_bkf and int(kv["bowkf"], 16) == _fnv(m_bkf.astype(np.int32).tobytes())
    # three matcher threads + two extractor threads (+ a matcher thread that exits and is replaced): no result differed
    assert int(kv["reentrancy_mismatches"]) == 0
