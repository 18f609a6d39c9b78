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
    kv = dict(t.split("=") for t in out.split())
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
