#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer API (orbfe_extract / orbfe_extract_batch): frames start and
end in pageable host memory.  Reported in DESIGN.md next to the HBM-resident headline, never as `value`."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import orb_slam2_annotate_amd as amd  # noqa: E402
from orb_slam2_annotate_amd import synth  # noqa: E402

frames = np.stack(synth.render_sequence(3, 128, 640, 480))
e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
e.extract_batch(frames[:8])
for B in (1, 8, 32, 128):
    t0 = time.perf_counter()
    reps = max(1, 256 // B)
    for _ in range(reps):
        if B == 1:
            e(frames[0])
        else:
            e.extract_batch(frames[:B])
    dt = time.perf_counter() - t0
    print(f"host API, batch {B:3d}: {reps * B / dt:9.0f} frames/s  ({1e3 * dt / reps / B:.3f} ms/frame)")

# where one single-frame call spends its time (stage events + wall clock)
e.profile(True)
t0 = time.perf_counter()
for _ in range(200):
    e(frames[0])
dt = (time.perf_counter() - t0) / 200
p = e.profile_get()
e.profile(False)
print("single frame with stage events: %.3f ms/call;" % (1e3 * dt),
      " ".join(f"{k}={v[0] / 200 * 1e3:.0f}us" for k, v in p.items()))

# tracking-thread matcher calls (host arrays in, host arrays out): per-call latency
import numpy as _np
from orb_slam2_annotate_amd import FrameView, ORBmatcher
_rng = _np.random.default_rng(0)
_n = 1000
_x = _rng.uniform(0, 640, _n).astype(_np.float32); _y = _rng.uniform(0, 480, _n).astype(_np.float32)
_oct = _rng.integers(0, 8, _n).astype(_np.int32); _ang = _rng.uniform(0, 360, _n).astype(_np.float32)
_desc = _rng.integers(0, 256, (_n, 32), dtype=_np.uint8)
_F = FrameView(_x, _y, _oct, _desc, (0.0, 640.0, 0.0, 480.0), angle=_ang)
_sf = (1.2 ** _np.arange(8)).astype(_np.float32)
_src = _rng.integers(0, _n, _n)
_u = (_x[_src] + _rng.normal(0, 3, _n)).astype(_np.float32); _v = (_y[_src] + _rng.normal(0, 3, _n)).astype(_np.float32)
_valid = _np.ones(_n, _np.uint8)
_m = ORBmatcher(0.9, True)
for _name, _fn in (("SearchByProjection(Frame, LastFrame) th=15", lambda: _m.SearchByProjectionLastFrame(_F, _sf, _valid, _u, _v, _oct[_src], _ang[_src], _desc[_src], 15.0)),
                   ("SearchByProjection(Frame, MapPoints) th=3", lambda: _m.SearchByProjection(_F, _sf, _valid, _oct[_src], _np.full(_n, 0.9995, _np.float32), _u, _v, _desc[_src], th=3.0)),
                   ("GetFeaturesInArea x1000 r=15", lambda: _F.GetFeaturesInArea(_u, _v, _np.full(_n, 15.0, _np.float32)))):
    _fn()
    t0 = time.perf_counter()
    for _ in range(100):
        _fn()
    print(f"{_name}: {1e3 * (time.perf_counter() - t0) / 100:.3f} ms/call (1000 features, 1000 queries, incl. the Python wrapper)")
