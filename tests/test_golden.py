"""Golden-vector tests.  tests/golden/*.npz are regression pins produced by tools/gen_golden.py
from the CPU oracle (the reference has no vectors of its own and is unbuildable here): the CPU
tests re-run the oracle against them, the GPU tests run the HIP path through the C-ABI."""
from pathlib import Path

import numpy as np
import pytest

import oracle_lib as orc

G = Path(__file__).resolve().parent / "golden"
CASES = ["extract_160x120", "extract_240x180_kitti_thr", "extract_noise_128x96"]


def _params(z):
    p = z["params"]
    return int(p[0]), float(p[1]), int(p[2]), int(p[3]), int(p[4])


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    z = np.load(G / f"{name}.npz")
    o = orc.Oracle(*_params(z))
    img = z["image"]
    kps, desc, pyr = o.extract(img, want_pyramid=True)
    assert np.array_equal(kps.view(np.uint8).reshape(-1, 28), z["keypoints"])
    assert np.array_equal(desc, z["descriptors"])
    lv = o.split_pyramid(pyr, img.shape[1], img.shape[0])
    assert [int(l.astype(np.uint64).sum()) for l in lv] == list(z["level_sums"])


def test_oracle_reproduces_golden_stereo():
    z = np.load(G / "stereo_320x200.npz")
    o = orc.Oracle(600, 1.2, 8, 20, 7)
    kL, dL, pL = o.extract(z["left"], want_pyramid=True)
    kR, dR, pR = o.extract(z["right"], want_pyramid=True)
    u, d = o.stereo(320, 200, kL, dL, kR, dR, pL, pR, float(z["mbf"]), float(z["mb"]))
    assert np.array_equal(u, z["uRight"]) and np.array_equal(d, z["depth"])
    n, m = orc.search_by_bow(dL, z["has1"], kL["angle"], orc.FeatVec(z["node1"]), dR, kR["angle"],
                             orc.FeatVec(z["node2"]), 0.7, True)
    assert n == int(z["bow_n"]) and np.array_equal(m, z["bow_match"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_reproduces_golden(name):
    import orb_slam2_annotate_amd as amd
    z = np.load(G / f"{name}.npz")
    e = amd.ORBextractor(*_params(z))
    kps, desc = e(z["image"])
    assert np.array_equal(kps.view(np.uint8).reshape(-1, 28), z["keypoints"])
    assert np.array_equal(desc, z["descriptors"])
    for l in range(e.GetLevels()):
        assert int(e.pyramid_level(l).astype(np.uint64).sum()) == int(z["level_sums"][l])
        assert int(e.debug_blurred_level(l).astype(np.uint64).sum()) == int(z["blur_sums"][l])


@pytest.mark.gpu
def test_gpu_reproduces_golden_stereo():
    import orb_slam2_annotate_amd as amd
    z = np.load(G / "stereo_320x200.npz")
    eL = amd.ORBextractor(600, 1.2, 8, 20, 7)
    eR = amd.ORBextractor(600, 1.2, 8, 20, 7)
    kL, dL = eL(z["left"])
    kR, dR = eR(z["right"])
    u, d = amd.ComputeStereoMatches(eL, eR, kL, dL, kR, dR, float(z["mbf"]), float(z["mb"]))
    assert np.array_equal(u, z["uRight"]) and np.array_equal(d, z["depth"])
    m = amd.ORBmatcher(0.7, True)
    n, mm = m.SearchByBoW(dL, z["has1"], kL["angle"], amd.FeatureVector.from_node_of_feature(z["node1"]), dR,
                          kR["angle"], amd.FeatureVector.from_node_of_feature(z["node2"]))
    assert n == int(z["bow_n"]) and np.array_equal(mm, z["bow_match"])
