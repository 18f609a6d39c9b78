#!/usr/bin/env python3
"""Randomised differential test of Frame::ComputeStereoMatches and both SearchByBoW forms against the CPU oracle:
random stereo frame sizes, feature counts, camera baselines, vocabulary node counts (1 node = hundreds of features
per node, the LDS-claim path; 150 nodes = the register-resident path), MapPoint masks, ratios.
    python tools/fuzz_stereo_bow.py [seconds] [seed]
Exits non-zero on the first mismatch and prints the failing configuration."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as orc  # noqa: E402
import orb_slam2_annotate_amd as amd  # noqa: E402
from orb_slam2_annotate_amd import synth  # noqa: E402


def nodes_of(desc, rng, n_nodes):
    cent = rng.integers(0, 256, size=(n_nodes, 32), dtype=np.uint8)
    x = np.unpackbits(desc, axis=1).astype(np.int16)
    c = np.unpackbits(cent, axis=1).astype(np.int16)
    d = (x[:, None, :] != c[None, :, :]).sum(axis=2)
    return d.argmin(axis=1).astype(np.uint32) * 5 + 2


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    last = t0
    n_cases = n_st = n_bow = 0
    while time.time() - t0 < seconds:
        w, h = int(rng.integers(160, 1300)), int(rng.integers(120, 520))
        if rng.random() < 0.3:
            w, h = [(1241, 376), (752, 480), (640, 480)][rng.integers(0, 3)]
        nf = int(rng.choice([300, 1000, 1200, 2000]))
        nl = int(rng.integers(1, 9))
        sf = float(rng.choice([1.2, 1.2, 1.1, 1.4]))
        scene = synth.STEREO_SCENES[str(rng.choice(["shapes", "textured"]))]  # round 3: both generators
        left, right = scene(int(rng.integers(1 << 30)), w, h)
        if rng.random() < 0.3:  # noise on top: more ambiguous descriptors, SAD minima at the window edge
            left = np.clip(left.astype(np.int16) + rng.normal(0, 6, left.shape), 0, 255).astype(np.uint8)
            right = np.clip(right.astype(np.int16) + rng.normal(0, 6, right.shape), 0, 255).astype(np.uint8)
        eL = amd.ORBextractor(nf, sf, nl, 20, 7)
        eR = amd.ORBextractor(nf, sf, nl, 20, 7)
        kL, dL = eL(left)
        kR, dR = eR(right)
        o = orc.Oracle(nf, sf, nl, 20, 7)
        krL, drL, pL = o.extract(left, want_pyramid=True)
        krR, drR, pR = o.extract(right, want_pyramid=True)
        cfg = f"seed={seed} case={n_cases} {w}x{h} nf={nf} sf={sf} nl={nl}"
        if not (np.array_equal(krL, kL) and np.array_equal(krR, kR) and np.array_equal(drL, dL) and np.array_equal(drR, dR)):
            print("EXTRACT MISMATCH", cfg)
            sys.exit(1)
        n_cases += 1
        if len(kL) and len(kR):
            fx = float(rng.uniform(150, 900))
            mbf = np.float32(fx * float(rng.uniform(0.05, 0.6)))
            mb = np.float32(mbf / np.float32(fx))
            u_ref, d_ref = o.stereo(w, h, krL, drL, krR, drR, pL, pR, float(mbf), float(mb))
            u, d = amd.ComputeStereoMatches(eL, eR, kL, dL, kR, dR, float(mbf), float(mb))
            if not (np.array_equal(u_ref, u) and np.array_equal(d_ref, d)):
                bad = np.flatnonzero((u_ref != u) | (d_ref != d))
                print("STEREO MISMATCH", cfg, f"mbf={mbf} mb={mb} first bad {bad[:5]}", u_ref[bad[:5]], u[bad[:5]])
                sys.exit(1)
            n_st += 1
            if rng.random() < 0.5:  # round 4: the same stereo frame in ONE call on the left handle (orbfe_extract_stereo_frame)
                k1, d1, k2, d2, u1, dd1 = eL.extract_stereo_frame(left, right, float(mbf), float(mb))
                if not (np.array_equal(k1, krL) and np.array_equal(k2, krR) and np.array_equal(d1, drL) and np.array_equal(d2, drR) and
                        np.array_equal(u1, u_ref) and np.array_equal(dd1, d_ref)):
                    print("STEREO FRAME (one call) MISMATCH", cfg, f"mbf={mbf} mb={mb}")
                    sys.exit(1)
        if len(kL) > 4 and len(kR) > 4:
            n_nodes = int(rng.choice([1, 2, 5, 8, 12, 20, 100, 150]))  # round 4: 8 / 12 / 20 land in the 2- and 4-slot register paths
            n1, n2 = nodes_of(dL, rng, n_nodes), nodes_of(dR, rng, n_nodes)
            has1 = (rng.random(len(kL)) < rng.uniform(0.3, 1.0)).astype(np.uint8)
            has2 = (rng.random(len(kR)) < rng.uniform(0.3, 1.0)).astype(np.uint8)
            nnr = float(rng.choice([0.6, 0.7, 0.75, 0.9]))
            ori = bool(rng.random() < 0.7)
            m = amd.ORBmatcher(nnr, ori)
            fv1, fv2 = amd.FeatureVector.from_node_of_feature(n1), amd.FeatureVector.from_node_of_feature(n2)
            rn, r = orc.search_by_bow(dL, has1, kL["angle"], orc.FeatVec(n1), dR, kR["angle"], orc.FeatVec(n2), nnr, ori)
            gn, g = m.SearchByBoW(dL, has1, kL["angle"], fv1, dR, kR["angle"], fv2)
            if rn != gn or not np.array_equal(r, g):
                print("BOW (KF, F) MISMATCH", cfg, f"nodes={n_nodes} nnratio={nnr} ori={ori}")
                sys.exit(1)
            rn, r = orc.search_by_bow_kf(dL, has1, kL["angle"], orc.FeatVec(n1), dR, has2, kR["angle"], orc.FeatVec(n2), nnr, ori)
            gn, g = m.SearchByBoW(dL, has1, kL["angle"], fv1, dR, kR["angle"], fv2, has_mp2=has2)
            if rn != gn or not np.array_equal(r, g):
                print("BOW (KF, KF) MISMATCH", cfg, f"nodes={n_nodes} nnratio={nnr} ori={ori}")
                sys.exit(1)
            # round 3: the same two searches and a 3-neighbour triangulation on RESIDENT frames (orbfe_frame_upload)
            b = (0.0, float(w), 0.0, float(h))
            urL = np.where(rng.random(len(kL)) < 0.5, kL["x"] - 4.0, -1.0).astype(np.float32)
            urR = np.where(rng.random(len(kR)) < 0.5, kR["x"] - 4.0, -1.0).astype(np.float32)
            R1 = amd.FrameView(kL["x"], kL["y"], kL["octave"], dL, b, angle=kL["angle"], u_right=urL).upload(fv1)
            R2 = amd.FrameView(kR["x"], kR["y"], kR["octave"], dR, b, angle=kR["angle"], u_right=urR).upload(fv2)
            rn, r = orc.search_by_bow(dL, has1, kL["angle"], orc.FeatVec(n1), dR, kR["angle"], orc.FeatVec(n2), nnr, ori)
            gn, g = m.SearchByBoWResident(R1, has1, R2)
            rn2, r2 = orc.search_by_bow_kf(dL, has1, kL["angle"], orc.FeatVec(n1), dR, has2, kR["angle"], orc.FeatVec(n2), nnr, ori)
            gn2, g2 = m.SearchByBoWResident(R1, has1, R2, has_mp2=has2)
            if rn != gn or not np.array_equal(r, g) or rn2 != gn2 or not np.array_equal(r2, g2):
                print("RESIDENT BOW MISMATCH", cfg, f"nodes={n_nodes} nnratio={nnr} ori={ori}")
                sys.exit(1)
            F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32) * np.float32(rng.uniform(0.005, 0.02))).astype(np.float32)
            ex, ey = float(rng.uniform(1000, 6000)), float(rng.uniform(0, h))
            only = bool(rng.random() < 0.3)
            tr_n, tr = orc.search_for_triangulation(dL, has1, kL["x"], kL["y"], kL["angle"], urL >= 0, orc.FeatVec(n1), dR, has2, kR["x"],
                                                    kR["y"], kR["angle"], kR["octave"], urR >= 0, orc.FeatVec(n2), F12, ex, ey,
                                                    o.scale_factors(), o.level_sigma2(), only, ori)
            cnt, mt = m.SearchForTriangulationMulti(R1, has1, [R2, R2, R2], [has2] * 3, [F12] * 3, [(ex, ey)] * 3, o.scale_factors(),
                                                    o.level_sigma2(), only)
            if any(int(cnt[k]) != tr_n or not np.array_equal(mt[k], tr) for k in range(3)):
                print("TRIANGULATION MULTI MISMATCH", cfg, f"nodes={n_nodes} only_stereo={only} ori={ori}")
                sys.exit(1)
            # round 4: one frame / key frame against K candidates in ONE call, and a frame built from the extractor's own
            # device records (orbfe_frame_from_extractor; eR's output block still holds kR / dR)
            from orb_slam2_annotate_amd.matcher import ResidentFrame
            X2 = ResidentFrame(amd.FrameView(kR["x"], kR["y"], kR["octave"], dR, b, angle=kR["angle"], u_right=urR), None, extractor=eR, frame=0)
            X2.set_featvec(fv2)
            K = int(rng.integers(1, 5))
            cands = [R2, X2, R2, X2][:K]
            cnt, mm = m.SearchByBoWKFMulti(R1, has1, cands, [has2] * K)
            if any(int(cnt[k]) != rn2 or not np.array_equal(mm[k], r2) for k in range(K)):
                print("BOW MULTI (KF, KF_k) MISMATCH", cfg, f"nodes={n_nodes} nnratio={nnr} ori={ori} K={K}")
                sys.exit(1)
            rn3, r3 = orc.search_by_bow(dR, has2, kR["angle"], orc.FeatVec(n2), dL, kL["angle"], orc.FeatVec(n1), nnr, ori)
            cnt, mm = m.SearchByBoWMulti(cands, [has2] * K, R1)
            if any(int(cnt[k]) != rn3 or not np.array_equal(mm[k], r3) for k in range(K)):
                print("BOW MULTI (KF_k, F) MISMATCH", cfg, f"nodes={n_nodes} nnratio={nnr} ori={ori} K={K}")
                sys.exit(1)
            X2.close()
            R1.close()
            R2.close()
            n_bow += 5 + 2 * K
        if time.time() - last > 50:
            print(f"... {n_cases} frames pairs, {n_st} stereo, {n_bow} bow searches, {time.time() - t0:.0f} s", flush=True)
            last = time.time()
    print(f"fuzz_stereo_bow: {n_cases} random stereo pairs: {n_st} ComputeStereoMatches and {n_bow} SearchByBoW calls "
          f"identical to the oracle in {seconds:.0f} s (seed {seed})")


if __name__ == "__main__":
    main()
