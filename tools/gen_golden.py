#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- small input/output vectors of the hot path.

The reference ships no tests/golden vectors and cannot be built here (OpenCV/Eigen absent), so
these vectors are produced by the CPU oracle (oracle/orb_oracle.c) and act as REGRESSION PINS of
the frozen canonical spec, not as reference-derived truth ("parity unpinned").  The
fixtures hold data only: input images (synthetic, seeded) and expected outputs.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle_lib as orc  # noqa: E402
from orb_slam2_annotate_amd import synth  # noqa: E402

OUT = ROOT / "tests" / "golden"


def extract_case(name, img, params):
    o = orc.Oracle(*params)
    H, W = img.shape
    kps, desc, pyr = o.extract(img, want_pyramid=True)
    lv = o.split_pyramid(pyr, W, H)
    np.savez_compressed(OUT / f"{name}.npz", image=img, params=np.array(params, dtype=np.float64),
                        keypoints=kps.view(np.uint8).reshape(-1, 28), descriptors=desc,
                        level_sums=np.array([int(l.astype(np.uint64).sum()) for l in lv], dtype=np.int64),
                        blur_sums=np.array([int(orc.gaussian_blur7(l).astype(np.uint64).sum()) for l in lv],
                                           dtype=np.int64))
    print(name, len(kps))
    return o, kps, desc, pyr


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    extract_case("extract_160x120", synth.render_frame(101, 160, 120, n_shapes=60), (300, 1.2, 8, 20, 7))
    extract_case("extract_240x180_kitti_thr", synth.render_frame(102, 240, 180, n_shapes=120), (500, 1.2, 8, 12, 7))
    extract_case("extract_noise_128x96", synth.adversarial("noise", 128, 96, seed=3), (200, 1.2, 6, 20, 7))
    # stereo + matching on a small pair
    L, R = synth.render_stereo(103, 320, 200, n_shapes=160, max_disp=40)
    o = orc.Oracle(600, 1.2, 8, 20, 7)
    kL, dL, pL = o.extract(L, want_pyramid=True)
    kR, dR, pR = o.extract(R, want_pyramid=True)
    mbf = np.float32(80.0)
    mb = np.float32(mbf / np.float32(200.0))
    u, d = o.stereo(320, 200, kL, dL, kR, dR, pL, pR, float(mbf), float(mb))
    rng = np.random.default_rng(7)
    node1 = rng.integers(0, 12, size=len(kL)).astype(np.uint32)
    node2 = rng.integers(0, 12, size=len(kR)).astype(np.uint32)
    has1 = (rng.random(len(kL)) < 0.7).astype(np.uint8)
    n_bow, m_bow = orc.search_by_bow(dL, has1, kL["angle"], orc.FeatVec(node1), dR, kR["angle"], orc.FeatVec(node2),
                                     0.7, True)
    np.savez_compressed(OUT / "stereo_320x200.npz", left=L, right=R, mbf=mbf, mb=mb, uRight=u, depth=d,
                        node1=node1, node2=node2, has1=has1, bow_n=n_bow, bow_match=m_bow)
    print("stereo", int((u >= 0).sum()), "bow", n_bow)


if __name__ == "__main__":
    main()
