#!/usr/bin/env python3
"""Per-call latency of the ORBmatcher entry points as Tracking / LocalMapping call them: ONE call on host arrays
(H2D of the operands, kernels, D2H of the result), median of `reps` calls, next to the CPU oracle's time for the same
call (one core).  These calls are latency-bound by construction (a few thousand descriptors per call); the batched,
device-resident forms are what bench.py measures.
    python tools/matcher_latency.py [reps]"""
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

SF = (1.2 ** np.arange(8)).astype(np.float32)


def med(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def nodes_of(desc, seed, n_nodes=100):
    rng = np.random.default_rng(seed)
    cent = rng.integers(0, 256, size=(n_nodes, 32), dtype=np.uint8)
    x = np.unpackbits(desc, axis=1).astype(np.int16)
    c = np.unpackbits(cent, axis=1).astype(np.int16)
    return (x[:, None, :] != c[None, :, :]).sum(axis=2).argmin(axis=1).astype(np.uint32)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(1)
    w, h, nf = 1241, 376, 2000
    left, right = synth.render_stereo(3, w, h)
    eL, eR = amd.ORBextractor(nf, 1.2, 8, 20, 7), amd.ORBextractor(nf, 1.2, 8, 20, 7)
    kL, dL = eL(left)
    kR, dR = eR(right)
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    _, _, pL = o.extract(left, want_pyramid=True)
    _, _, pR = o.extract(right, want_pyramid=True)
    rows = []

    def row(name, gpu, cpu):
        g, c = med(gpu, reps), med(cpu, max(3, reps // 6))
        rows.append((name, g, c))
        print(f"{name:44s} GPU {g:7.3f} ms   CPU oracle {c:7.3f} ms   x{c / g:5.1f}", flush=True)

    print(f"frame: {w}x{h}, {len(kL)} / {len(kR)} keypoints; map points projected: 1000")
    # --- stereo
    mbf = np.float32(386.1448)
    mb = np.float32(mbf / np.float32(718.856))
    row("ComputeStereoMatches", lambda: amd.ComputeStereoMatches(eL, eR, kL, dL, kR, dR, float(mbf), float(mb)),
        lambda: o.stereo(w, h, kL, dL, kR, dR, pL, pR, float(mbf), float(mb)))
    # --- BoW family
    n1, n2 = nodes_of(dL, 5), nodes_of(dR, 5)
    has1 = (rng.random(len(kL)) < 0.7).astype(np.uint8)
    has2 = (rng.random(len(kR)) < 0.7).astype(np.uint8)
    M = amd.ORBmatcher(0.7, True)
    fv1, fv2 = amd.FeatureVector.from_node_of_feature(n1), amd.FeatureVector.from_node_of_feature(n2)
    fo1, fo2 = orc.FeatVec(n1), orc.FeatVec(n2)
    row("SearchByBoW(KeyFrame, Frame)", lambda: M.SearchByBoW(dL, has1, kL["angle"], fv1, dR, kR["angle"], fv2),
        lambda: orc.search_by_bow(dL, has1, kL["angle"], fo1, dR, kR["angle"], fo2, 0.7, True))
    row("SearchByBoW(KeyFrame, KeyFrame)", lambda: M.SearchByBoW(dL, has1, kL["angle"], fv1, dR, kR["angle"], fv2, has_mp2=has2),
        lambda: orc.search_by_bow_kf(dL, has1, kL["angle"], fo1, dR, has2, kR["angle"], fo2, 0.7, True))
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float32) * 0.01
    st1 = (rng.random(len(kL)) < 0.5).astype(np.uint8)
    st2 = (rng.random(len(kR)) < 0.5).astype(np.uint8)
    sf, sg = o.scale_factors(), o.level_sigma2()
    h1 = (rng.random(len(kL)) < 0.4).astype(np.uint8)
    h2 = (rng.random(len(kR)) < 0.4).astype(np.uint8)
    row("SearchForTriangulation",
        lambda: M.SearchForTriangulation(dL, h1, kL["x"], kL["y"], kL["angle"], st1, fv1, dR, h2, kR["x"], kR["y"], kR["angle"],
                                         kR["octave"], st2, fv2, F12, 5000.0, 240.0, sf, sg, False),
        lambda: orc.search_for_triangulation(dL, h1, kL["x"], kL["y"], kL["angle"], st1, fo1, dR, h2, kR["x"], kR["y"],
                                             kR["angle"], kR["octave"], st2, fo2, F12, 5000.0, 240.0, sf, sg, False, True))
    # --- projection family: 1000 map points projected into the right frame
    x, y, octv, ang = kR["x"].copy(), kR["y"].copy(), kR["octave"].astype(np.int32), kR["angle"].copy()
    bounds = (0.0, float(w), 0.0, float(h))
    ur = (x - rng.uniform(1, 40, len(x))).astype(np.float32)
    F = amd.FrameView(x, y, octv, dR, bounds, angle=ang, u_right=ur)
    Fo = orc.Frame(x, y, octv, dR, bounds, angle=ang, u_right=ur)
    m = 1000
    src = rng.integers(0, len(x), m)
    u = (x[src] + rng.normal(0, 3, m)).astype(np.float32)
    v = (y[src] + rng.normal(0, 3, m)).astype(np.float32)
    md = dR[src] ^ (rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8) &
                    rng.integers(0, 256, (m, 32), dtype=np.uint8))
    lv = np.clip(octv[src] + rng.integers(-1, 2, m), 0, 7).astype(np.int32)
    a = ((ang[src] + rng.normal(0, 8, m)) % 360).astype(np.float32)
    valid = (rng.random(m) < 0.85).astype(np.uint8)
    vc = rng.uniform(0.99, 1.0, m).astype(np.float32)
    pxr = (u - rng.uniform(1, 40, m)).astype(np.float32)
    invz = rng.uniform(0.02, 0.5, m).astype(np.float32)
    th = 3.0
    row("SearchByProjection(Frame, MapPoints)", lambda: M.SearchByProjection(F, SF, valid, lv, vc, u, v, md, th=th, proj_xr=pxr),
        lambda: orc.search_by_projection_mappoints(Fo, SF, None, valid, lv, vc, u, v, pxr, md, None, th, 0.7))
    row("SearchByProjection(Frame, LastFrame)",
        lambda: M.SearchByProjectionLastFrame(F, SF, valid, u, v, lv, a, md, 7.0, mode=0, mbf=40.0, invzc=invz),
        lambda: orc.search_by_projection_lastframe(Fo, SF, 40.0, valid, u, v, invz, lv, a, md, None, 0, 7.0, True))
    row("SearchByProjection(Frame, KeyFrame) reloc", lambda: M.SearchByProjectionKeyFrame(F, SF, valid, u, v, lv, a, md, th, 100),
        lambda: orc.search_by_projection_reloc(Fo, SF, valid, u, v, lv, a, md, None, th, 100, True))
    row("SearchByProjection(KeyFrame, Scw) Sim3", lambda: M.SearchByProjectionSim3(F, SF, valid, u, v, lv, md, th),
        lambda: orc.search_by_projection_sim3(Fo, SF, valid, u, v, lv, md, None, th))
    inv_s2 = (1.0 / (SF * SF)).astype(np.float32)
    row("Fuse", lambda: M.FuseSearch(F, SF, valid, u, v, lv, md, th=th, inv_level_sigma2=inv_s2, ur=pxr),
        lambda: orc.fuse_search(Fo, SF, inv_s2, valid, u, v, pxr, lv, md, th, True))
    print("| call | GPU ms | CPU oracle ms |\n|---|---|---|")
    for name, g, c in rows:
        print(f"| `{name}` | {g:.3f} | {c:.3f} |")


if __name__ == "__main__":
    main()
