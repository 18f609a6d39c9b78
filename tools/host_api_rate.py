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
