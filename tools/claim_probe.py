#!/usr/bin/env python3
"""Latency and fixed-point rounds of the device claim loops (k_window_claim) on the projection-search calls of
tools/matcher_latency.py -- no oracle, a few hundred calls, fit for `rocprofv3 --kernel-trace --stats`.
    python tools/claim_probe.py [reps]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import orb_slam2_annotate_amd as amd  # noqa: E402
from orb_slam2_annotate_amd import _lib, synth  # noqa: E402

SF = (1.2 ** np.arange(8)).astype(np.float32)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(1)
    w, h, nf = 1241, 376, 2000
    _, right = synth.render_stereo_textured(3, w, h)
    kR, dR = amd.ORBextractor(nf, 1.2, 8, 20, 7)(right)
    x, y, octv, ang = kR["x"].copy(), kR["y"].copy(), kR["octave"].astype(np.int32), kR["angle"].copy()
    ur = (x - rng.uniform(1, 40, len(x))).astype(np.float32)
    F = amd.FrameView(x, y, octv, dR, (0.0, float(w), 0.0, float(h)), angle=ang, u_right=ur)
    L = _lib.load()
    for m, dup in [(1000, True), (1000, False), (2000, False)]:
        src = rng.integers(0, len(x), m) if dup else rng.permutation(len(x))[:m]
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
        M = amd.ORBmatcher(0.7, True)
        FR = F.upload()
        print(f"# {m} points ({'random features, with repeats' if dup else 'distinct features'}) into {len(x)} key points")
        for name, fn in [
            ("map points", lambda fr: M.SearchByProjection(fr, SF, valid, lv, vc, u, v, md, th=3.0, proj_xr=pxr)),
            ("last frame", lambda fr: M.SearchByProjectionLastFrame(fr, SF, valid, u, v, lv, a, md, 7.0, mode=0, mbf=40.0, invzc=invz)),
            ("key frame", lambda fr: M.SearchByProjectionKeyFrame(fr, SF, valid, u, v, lv, a, md, 3.0, 100)),
            ("Sim3", lambda fr: M.SearchByProjectionSim3(fr, SF, valid, u, v, lv, md, 3.0)),
        ]:
            for kind, fr in (("host arrays", F), ("resident", FR)):
                n = fn(fr)[0]
                rounds = L.orbfe_debug_last_claim_rounds()
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    fn(fr)
                    ts.append(time.perf_counter() - t0)
                print(f"{name:12s} {kind:12s} {1e3 * float(np.median(ts)):7.3f} ms   matches {n:5d}   rounds {rounds}", flush=True)
        FR.close()


if __name__ == "__main__":
    main()
