#!/usr/bin/env python3
"""Per-call latency of the ORBmatcher entry points as Tracking / LocalMapping call them, median of `reps` calls, next to
the CPU oracle's time for the same call (one core):
  * ONE call on host arrays (H2D of every operand, kernels, D2H of the result) -- what a drop-in without any change to the
    callers pays;
  * the same call on RESIDENT frames (orbfe_frame_upload once per Frame / KeyFrame: keypoints, descriptors, grid and
    FeatureVector indices stay on the device);
  * the multi-neighbour patterns of LocalMapping in ONE call: SearchForTriangulation of a key frame against 20 neighbours
    (src/LocalMapping.cc:283-315), Fuse of its map points into 10 neighbours (:542-549) -- per-neighbour cost reported;
  * one stereo frame through the class API: operator() on L and R (two handles, two threads, src/Frame.cc:78-81) +
    ComputeStereoMatches.
bench.py embeds measure() as its `latency` block.     python tools/matcher_latency.py [reps]"""
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


def measure(reps=30, verbose=True):
    """-> list of dicts {call, gpu_ms, cpu_oracle_ms, [per_neighbour_gpu_ms]}"""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(1)
    w, h, nf = 1241, 376, 2000
    left, right = synth.render_stereo_textured(3, w, h)
    eL, eR = amd.ORBextractor(nf, 1.2, 8, 20, 7), amd.ORBextractor(nf, 1.2, 8, 20, 7)
    kL, dL = eL(left)
    kR, dR = eR(right)
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    _, _, pL = o.extract(left, want_pyramid=True)
    _, _, pR = o.extract(right, want_pyramid=True)
    rows = []

    def row(name, gpu, cpu, per=1):
        g, c = med(gpu, reps), med(cpu, max(3, reps // 6))
        rows.append({"call": name, "gpu_ms": g, "cpu_oracle_ms": c, "units_per_call": per, "gpu_ms_per_unit": g / per,
                     "cpu_ms_per_unit": c / per})
        if verbose:
            print(f"{name:58s} GPU {g:7.3f} ms   CPU oracle {c:7.3f} ms   x{c / g:5.1f}" +
                  (f"   ({g / per:.4f} / {c / per:.4f} ms per neighbour)" if per > 1 else ""), flush=True)

    if verbose:
        print(f"frame: {w}x{h}, {len(kL)} / {len(kR)} keypoints; map points projected: 1000")
    # --- one stereo frame through the class API: L || R on two handles + ComputeStereoMatches
    mbf = np.float32(386.1448)
    mb = np.float32(mbf / np.float32(718.856))
    with ThreadPoolExecutor(2) as pool:
        def gpu_frame():
            fl, fr = pool.submit(eL, left), pool.submit(eR, right)
            (a1, b1), (a2, b2) = fl.result(), fr.result()
            return amd.ComputeStereoMatches(eL, eR, a1, b1, a2, b2, float(mbf), float(mb))
        o2 = orc.Oracle(nf, 1.2, 8, 20, 7)
        def cpu_frame():
            fl, fr = pool.submit(o.extract, left, None, True), pool.submit(o2.extract, right, None, True)
            (a1, b1, p1), (a2, b2, p2) = fl.result(), fr.result()
            return o.stereo(w, h, a1, b1, a2, b2, p1, p2, float(mbf), float(mb))
        row("stereo Frame: operator() L || R + ComputeStereoMatches", gpu_frame, cpu_frame)
    # --- stereo
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
    # the same on frames made resident once
    R1 = amd.FrameView(kL["x"], kL["y"], kL["octave"], dL, (0.0, float(w), 0.0, float(h)), angle=kL["angle"],
                       u_right=np.where(st1 > 0, kL["x"] - 5, -1).astype(np.float32)).upload(fv1)
    R2 = amd.FrameView(kR["x"], kR["y"], kR["octave"], dR, (0.0, float(w), 0.0, float(h)), angle=kR["angle"],
                       u_right=np.where(st2 > 0, kR["x"] - 5, -1).astype(np.float32)).upload(fv2)
    row("SearchByBoW(KeyFrame, Frame), resident", lambda: M.SearchByBoWResident(R1, has1, R2),
        lambda: orc.search_by_bow(dL, has1, kL["angle"], fo1, dR, kR["angle"], fo2, 0.7, True))
    row("SearchByBoW(KeyFrame, KeyFrame), resident", lambda: M.SearchByBoWResident(R1, has1, R2, has_mp2=has2),
        lambda: orc.search_by_bow_kf(dL, has1, kL["angle"], fo1, dR, has2, kR["angle"], fo2, 0.7, True))
    row("SearchForTriangulation, resident (1 neighbour)",
        lambda: M.SearchForTriangulationMulti(R1, h1, [R2], [h2], [F12], [(5000.0, 240.0)], sf, sg, False),
        lambda: orc.search_for_triangulation(dL, h1, kL["x"], kL["y"], kL["angle"], st1, fo1, dR, h2, kR["x"], kR["y"],
                                             kR["angle"], kR["octave"], st2, fo2, F12, 5000.0, 240.0, sf, sg, False, True))
    NC = 10  # Relocalization / ComputeSim3: the candidate key frames DetectRelocalizationCandidates / the covisibility groups return
    row(f"SearchByBoW(KeyFrame_k, Frame) x {NC} candidates, ONE call (resident)",
        lambda: M.SearchByBoWMulti([R1] * NC, [has1] * NC, R2),
        lambda: [orc.search_by_bow(dL, has1, kL["angle"], fo1, dR, kR["angle"], fo2, 0.7, True) for _ in range(NC)], per=NC)
    row(f"SearchByBoW(KeyFrame, KeyFrame_k) x {NC} candidates, ONE call (resident)",
        lambda: M.SearchByBoWKFMulti(R1, has1, [R2] * NC, [has2] * NC),
        lambda: [orc.search_by_bow_kf(dL, has1, kL["angle"], fo1, dR, has2, kR["angle"], fo2, 0.7, True) for _ in range(NC)], per=NC)
    NB = 20  # CreateNewMapPoints: 20 neighbours for monocular, 10 for stereo (src/LocalMapping.cc:256-259)
    row(f"SearchForTriangulation x {NB} neighbours, ONE call (resident)",
        lambda: M.SearchForTriangulationMulti(R1, h1, [R2] * NB, [h2] * NB, [F12] * NB, [(5000.0, 240.0)] * NB, sf, sg, False),
        lambda: [orc.search_for_triangulation(dL, h1, kL["x"], kL["y"], kL["angle"], st1, fo1, dR, h2, kR["x"], kR["y"],
                                              kR["angle"], kR["octave"], st2, fo2, F12, 5000.0, 240.0, sf, sg, False, True)
                 for _ in range(NB)], per=NB)
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
    FR = F.upload()
    row("SearchByProjection(Frame, MapPoints), resident", lambda: M.SearchByProjection(FR, SF, valid, lv, vc, u, v, md, th=th, proj_xr=pxr),
        lambda: orc.search_by_projection_mappoints(Fo, SF, None, valid, lv, vc, u, v, pxr, md, None, th, 0.7))
    row("SearchByProjection(Frame, LastFrame), resident",
        lambda: M.SearchByProjectionLastFrame(FR, SF, valid, u, v, lv, a, md, 7.0, mode=0, mbf=40.0, invzc=invz),
        lambda: orc.search_by_projection_lastframe(Fo, SF, 40.0, valid, u, v, invz, lv, a, md, None, 0, 7.0, True))
    row("SearchByProjection(Frame, KeyFrame) reloc, resident", lambda: M.SearchByProjectionKeyFrame(FR, SF, valid, u, v, lv, a, md, th, 100),
        lambda: orc.search_by_projection_reloc(Fo, SF, valid, u, v, lv, a, md, None, th, 100, True))
    row("SearchByProjection(KeyFrame, Scw) Sim3, resident", lambda: M.SearchByProjectionSim3(FR, SF, valid, u, v, lv, md, th),
        lambda: orc.search_by_projection_sim3(Fo, SF, valid, u, v, lv, md, None, th))
    row("Fuse, resident", lambda: M.FuseSearch(FR, SF, valid, u, v, lv, md, th=th, inv_level_sigma2=inv_s2, ur=pxr),
        lambda: orc.fuse_search(Fo, SF, inv_s2, valid, u, v, pxr, lv, md, th, True))
    NK = 10  # SearchInNeighbors: the current key frame's map points fused into every neighbour (src/LocalMapping.cc:542-549)
    st = lambda a_: np.stack([a_] * NK)  # noqa: E731
    row(f"Fuse into {NK} neighbours, ONE call (resident)",
        lambda: M.FuseSearchMulti([FR] * NK, SF, st(valid), st(u), st(v), st(lv), md, th=th, inv_level_sigma2=inv_s2, ur=st(pxr)),
        lambda: [orc.fuse_search(Fo, SF, inv_s2, valid, u, v, pxr, lv, md, th, True) for _ in range(NK)], per=NK)
    NR = 5  # Relocalization: SearchByProjection(mCurrentFrame, pKF_k, sFound, 10, 100) per surviving candidate (src/Tracking.cc:1577)
    cand = [dict(valid=valid, u=u, v=v, level=lv, kf_angle=a, mp_desc=md, th=th, ORBdist=100)] * NR
    row(f"SearchByProjection(Frame, KeyFrame_k) reloc x {NR} candidates, ONE call (resident)",
        lambda: M.SearchByProjectionKeyFrameMulti(FR, SF, cand),
        lambda: [orc.search_by_projection_reloc(Fo, SF, valid, u, v, lv, a, md, None, th, 100, True) for _ in range(NR)], per=NR)

    def timed(name, fn, n=20):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        rows.append({"call": name, "gpu_ms": 1e3 * (time.perf_counter() - t0) / n, "cpu_oracle_ms": None, "units_per_call": 1})
        if verbose:
            print(f"{name:58s} GPU {rows[-1]['gpu_ms']:7.3f} ms", flush=True)

    timed("orbfe_frame_upload + release (2000 keypoints, FeatureVector, grid build; pooled slab, no host wait)",
          lambda: F.upload(fv2).close())
    # Frame::Frame without the features travelling twice: resident operands from the extractor's own device records
    from orb_slam2_annotate_amd.matcher import ResidentFrame
    eR(right)  # (the handle's output block = this frame)
    VR = amd.FrameView(kR["x"], kR["y"], kR["octave"], dR, bounds, angle=kR["angle"])
    timed("orbfe_frame_from_extractor + release (records and descriptors stay in HBM)", lambda: ResidentFrame(VR, fv2, extractor=eR, frame=0).close())
    # what a NEW key frame costs in steady state: the pool holds slabs and events of released key frames (culling), the
    # upload does not wait for the device (the first search on another stream is ordered behind it by the frame's event)
    warm = [F.upload(fv2) for _ in range(24)]
    M.SearchByBoWMulti(warm, [has2] * len(warm), R1)  # (settles them: releasing an unused frame waits for its own build)
    for k_ in warm:
        k_.close()
    keep = []
    timed("orbfe_frame_upload alone, warm pool (what a new key frame costs)", lambda: keep.append(F.upload(fv2)), n=10)
    timed("orbfe_frame_from_extractor alone, warm pool", lambda: keep.append(ResidentFrame(VR, fv2, extractor=eR, frame=0)), n=10)
    M.SearchByBoWMulti(keep, [has2] * len(keep), R1)
    for k_ in keep:
        k_.close()
    # the stereo Frame on ONE handle and one thread: both eyes as a 2-frame batch (two sub-batch streams), then
    # ComputeStereoMatches on frames 0 and 1 of that handle -- instead of two handles on two threads
    def one_handle():
        (a1, b1), (a2, b2) = eL.extract_batch(np.stack([left, right]))
        return amd.ComputeStereoMatches(eL, eL, a1, b1, a2, b2, float(mbf), float(mb), frameL=0, frameR=1)
    timed("stereo Frame on ONE handle: extract_batch(L, R) + ComputeStereoMatches(frames 0, 1)", one_handle, n=40)
    timed("stereo Frame in ONE call: orbfe_extract_stereo_frame (both eyes + stereo matcher, one download)",
          lambda: eL.extract_stereo_frame(left, right, float(mbf), float(mb)), n=40)
    # ONE frame through operator() -- the live camera: upload, the single-frame kernel chain, download (tools/single_frame_latency.py)
    timed("operator() on one 1241x376 frame, 2000 features (pageable image, Python call)", lambda: eL(left), n=60)
    import ctypes as C
    from orb_slam2_annotate_amd import _lib
    pimg, pk, pd, _pn = eL.pinned_buffers(1, h, w)
    np.copyto(pimg[0], left)
    L_, n_ = _lib.load(), C.c_int(0)
    timed("operator() on one 1241x376 frame, page-locked image and outputs (orbfe_host_alloc), C-ABI call",
          lambda: _lib.check(L_.orbfe_extract(eL._h, _lib.ptr(pimg), w, h, w, _lib.ptr(pk), _lib.ptr(pd), pk.shape[1], C.byref(n_))), n=60)
    return rows


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rows = measure(reps)
    print("| call | GPU ms | CPU oracle ms | GPU ms per neighbour |\n|---|---|---|---|")
    for r in rows:
        c = "" if r["cpu_oracle_ms"] is None else f"{r['cpu_oracle_ms']:.3f}"
        pn = f"{r['gpu_ms_per_unit']:.4f}" if r["units_per_call"] > 1 else ""
        print(f"| `{r['call']}` | {r['gpu_ms']:.3f} | {c} | {pn} |")


if __name__ == "__main__":
    main()
