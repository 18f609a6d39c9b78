#!/usr/bin/env python3
"""One frame at a time through operator() -- the live camera's pattern: median wall time per call next to the sum of the
kernel durations of that call (rocprofv3 --kernel-trace gives the latter: run this script under it and read the stats).
    python tools/single_frame_latency.py [reps] [kitti|vga]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import orb_slam2_annotate_amd as amd  # noqa: E402
from orb_slam2_annotate_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
kind = sys.argv[2] if len(sys.argv) > 2 else "kitti"
if kind == "kitti":
    img, _ = synth.render_stereo_textured(3, 1241, 376)
    e = amd.ORBextractor(2000, 1.2, 8, 20, 7)
else:
    img = synth.render_frame(5, 640, 480)
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
k, d = e(img)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    e(img)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print(f"{kind}: {img.shape[1]}x{img.shape[0]}, {len(k)} keypoints: median {np.median(ts):.3f} ms, min {ts.min():.3f} ms over {reps} calls", flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "pinned":  # the same frame from page-locked memory (orbfe_host_alloc), outputs page-locked too
    import ctypes as C
    from orb_slam2_annotate_amd import _lib
    H, W = img.shape
    pimg, pk, pd, pn = e.pinned_buffers(1, H, W)
    np.copyto(pimg[0], img)
    L = _lib.load()
    n = C.c_int(0)
    call = lambda: _lib.check(L.orbfe_extract(e._h, _lib.ptr(pimg), W, H, W, _lib.ptr(pk), _lib.ptr(pd), pk.shape[1], C.byref(n)))
    call()
    tp = []
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        tp.append(time.perf_counter() - t0)
    tp = np.array(tp) * 1e3
    print(f"  page-locked image and outputs, C-ABI call only: median {np.median(tp):.3f} ms, min {tp.min():.3f} ms ({n.value} keypoints)", flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "stages":  # HIP-event time per stage of the same calls (the events cost a few microseconds themselves)
    e.profile(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        e(img)
    wall = (time.perf_counter() - t0) / reps * 1e3
    st = e.profile_get()
    print(f"  with stage events: {wall:.3f} ms per call; " + "  ".join(f"{s} {v[0] / reps:.3f}" for s, v in st.items() if v[1]) +
          f"  | sum {sum(v[0] for v in st.values()) / reps:.3f} ms", flush=True)
