#!/usr/bin/env python3
"""200 single-frame orbfe_extract calls (run under rocprofv3 --kernel-trace --stats to see the per-kernel
durations of the live-camera case)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import orb_slam2_annotate_amd as amd  # noqa: E402
from orb_slam2_annotate_amd import synth  # noqa: E402

img = synth.render_frame(5, 640, 480)
e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
for _ in range(200):
    e(img)
