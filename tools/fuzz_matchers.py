#!/usr/bin/env python3
"""Randomised differential test of the projection-search family (device frame grid + window search +
device claim loops, k_window_claim) against the CPU oracle.
    python tools/fuzz_matchers.py [seconds] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as orc  # noqa: E402
import orb_slam2_annotate_amd as amd  # noqa: E402

SF = (1.2 ** np.arange(8)).astype(np.float32)


def frame(rng, n, w, h, stereo):
    x = rng.uniform(-3, w + 3, n).astype(np.float32)
    y = rng.uniform(-3, h + 3, n).astype(np.float32)
    if n and rng.random() < 0.3:  # clustered: many features per grid cell, long candidate lists
        cx, cy = rng.uniform(0, w, 5), rng.uniform(0, h, 5)
        k = rng.integers(0, 5, n)
        x = (cx[k] + rng.normal(0, 12, n)).astype(np.float32)
        y = (cy[k] + rng.normal(0, 12, n)).astype(np.float32)
    octv = rng.integers(0, 8, n).astype(np.int32)
    ang = rng.uniform(0, 360, n).astype(np.float32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ur = np.where(rng.random(n) < 0.7, x - rng.uniform(1, 40, n), -1).astype(np.float32) if stereo else None
    return x, y, octv, ang, desc, ur


def queries(rng, x, y, octv, ang, desc, m, noise):
    n = len(x)
    if n == 0:
        return (rng.uniform(0, 600, m).astype(np.float32), rng.uniform(0, 400, m).astype(np.float32),
                rng.integers(0, 256, (m, 32), dtype=np.uint8), rng.integers(0, 8, m).astype(np.int32),
                rng.uniform(0, 360, m).astype(np.float32), np.zeros(m, np.int64))
    src = rng.integers(0, n, m)
    u = (x[src] + rng.normal(0, noise, m)).astype(np.float32)
    v = (y[src] + rng.normal(0, noise, m)).astype(np.float32)
    md = desc[src] ^ (rng.integers(0, 256, (m, 32), dtype=np.uint8) & rng.integers(0, 256, (m, 32), dtype=np.uint8) &
                      rng.integers(0, 256, (m, 32), dtype=np.uint8))
    lv = np.clip(octv[src] + rng.integers(-1, 2, m), 0, 7).astype(np.int32)
    a = ((ang[src] + rng.normal(0, 8, m)) % 360).astype(np.float32)
    return u, v, md, lv, a, src


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n_cases = 0
    while time.time() - t0 < seconds:
        w, h = float(rng.choice([640, 752, 1241, 320])), float(rng.choice([480, 376, 240]))
        bounds = (0.0, w, 0.0, h) if rng.random() < 0.7 else (-rng.uniform(0, 20), w + rng.uniform(0, 20),
                                                                 -rng.uniform(0, 20), h + rng.uniform(0, 20))
        stereo = bool(rng.random() < 0.5)
        n = int(rng.choice([0, 1, 7, 300, 1000, 2000, 4000]))
        m = int(rng.choice([0, 1, 50, 1000, 2500]))
        x, y, octv, ang, desc, ur = frame(rng, n, w, h, stereo)
        F = amd.FrameView(x, y, octv, desc, bounds, angle=ang, u_right=ur)
        resident = bool(rng.random() < 0.5)  # round 3: the same searches on a frame uploaded once (orbfe_frame_upload)
        if resident:
            F = F.upload()
        Fo = orc.Frame(x, y, octv, desc, bounds, angle=ang, u_right=ur)
        u, v, md, lv, a, src = queries(rng, x, y, octv, ang, desc, m, float(rng.choice([0.5, 3.0, 10.0])))
        valid = (rng.random(m) < 0.85).astype(np.uint8)
        th = float(rng.choice([1.0, 3.0, 7.0, 15.0]))
        ratio = float(rng.choice([0.6, 0.8, 0.9]))
        ori = bool(rng.random() < 0.6)
        M = amd.ORBmatcher(ratio, ori)
        tag = "?"
        try:
            tag = "mappoints"
            vc = rng.uniform(0.99, 1.0, m).astype(np.float32)
            pxr = (u - rng.uniform(1, 40, m)).astype(np.float32) if stereo else None
            blocked = (rng.random(max(n, 1)) < 0.1).astype(np.uint8)[:n] if rng.random() < 0.5 else None
            obs = (rng.random(m) < 0.9).astype(np.uint8) if rng.random() < 0.5 else None
            r = orc.search_by_projection_mappoints(Fo, SF, blocked, valid, lv, vc, u, v, pxr, md, obs, th, ratio)
            g = M.SearchByProjection(F, SF, valid, lv, vc, u, v, md, th=th, proj_xr=pxr, blocked=blocked, mp_obs_positive=obs)
            assert (g[0], g[1].tolist()) == (r[0], r[1].tolist())
            tag = "lastframe"
            mode = int(rng.integers(0, 3))
            invz = rng.uniform(0.02, 0.5, m).astype(np.float32)
            r = orc.search_by_projection_lastframe(Fo, SF, 40.0, valid, u, v, invz, lv, a, md, obs, mode, th, ori, blocked)
            g = M.SearchByProjectionLastFrame(F, SF, valid, u, v, lv, a, md, th, mode=mode, mbf=40.0, invzc=invz, obs_positive=obs,
                                              blocked=blocked)
            assert (g[0], g[1].tolist()) == (r[0], r[1].tolist())
            tag = "keyframe"
            od = int(rng.choice([64, 100]))
            r = orc.search_by_projection_reloc(Fo, SF, valid, u, v, lv, a, md, blocked, th, od, ori)
            g = M.SearchByProjectionKeyFrame(F, SF, valid, u, v, lv, a, md, th, od, blocked=blocked)
            assert (g[0], g[1].tolist()) == (r[0], r[1].tolist())
            tag = "sim3proj"
            r = orc.search_by_projection_sim3(Fo, SF, valid, u, v, lv, md, blocked, th)
            g = M.SearchByProjectionSim3(F, SF, valid, u, v, lv, md, th, matched=blocked)
            assert (g[0], g[1].tolist()) == (r[0], r[1].tolist())
            tag = "fuse"
            inv_s2 = (1.0 / (SF * SF)).astype(np.float32)
            chi = bool(rng.random() < 0.5)
            r = orc.fuse_search(Fo, SF, inv_s2, valid, u, v, pxr if stereo else u, lv, md, th, chi)
            g = M.FuseSearch(F, SF, valid, u, v, lv, md, th=th, inv_level_sigma2=inv_s2 if chi else None,
                             ur=pxr if stereo else u)
            assert g.tolist() == r.tolist()
            if n and m:
                tag = "fuse_multi"  # the same points against 3 key frames (this one, resident or not, three times over)
                K3 = 3
                g3 = M.FuseSearchMulti([F] * K3, SF, np.stack([valid] * K3), np.stack([u] * K3), np.stack([v] * K3), np.stack([lv] * K3), md,
                                       th=th, inv_level_sigma2=inv_s2 if chi else None, ur=np.stack([pxr if stereo else u] * K3))
                assert all(g3[k].tolist() == r.tolist() for k in range(K3))
            if n and m:
                tag = "area"
                q = min(m, 64)
                rad = rng.choice(np.array([0.5, 5.0, 40.0, 500.0], np.float32), q)
                lo = rng.integers(-1, 4, q).astype(np.int32)
                hi = rng.integers(-1, 6, q).astype(np.int32)
                got = F.GetFeaturesInArea(u[:q], v[:q], rad, lo, hi, capacity=8)
                for i in range(q):
                    assert got[i].tolist() == Fo.features_in_area(u[i], v[i], rad[i], lo[i], hi[i]).tolist()
        except AssertionError:
            print(f"MISMATCH in {tag}: seed={seed} case={n_cases} n={n} m={m} stereo={stereo} th={th} bounds={bounds} resident={resident}")
            sys.exit(1)
        if resident:
            F.close()
        n_cases += 1
    print(f"fuzz_matchers: {n_cases} random cases x 5-6 searches identical to the oracle in {time.time() - t0:.0f} s (seed {seed})")


if __name__ == "__main__":
    main()
