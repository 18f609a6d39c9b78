#!/usr/bin/env python3
"""One resident SearchByBoW(KeyFrame, Frame) call on a KITTI pair whose FeatureVectors have a skewed node (129 x 129 features):
the launch lasts as long as the sequential query loop of the largest node -- run under rocprofv3 --kernel-trace --stats to
read k_search_by_bow's duration.     python tools/bow_probe.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import orb_slam2_annotate_amd as amd
from orb_slam2_annotate_amd import synth
from matcher_latency import nodes_of
w, h, nf = 1241, 376, 2000
left, right = synth.render_stereo_textured(3, w, h)
eL, eR = amd.ORBextractor(nf, 1.2, 8, 20, 7), amd.ORBextractor(nf, 1.2, 8, 20, 7)
kL, dL = eL(left); kR, dR = eR(right)
n1, n2 = nodes_of(dL, 5), nodes_of(dR, 5)
fv1, fv2 = amd.FeatureVector.from_node_of_feature(n1), amd.FeatureVector.from_node_of_feature(n2)
print("largest nodes", np.bincount(n1).max(), np.bincount(n2).max(), "nodes", len(np.unique(n1)))
b = (0.0, float(w), 0.0, float(h))
F1 = amd.FrameView(kL["x"], kL["y"], kL["octave"].astype(np.int32), dL, b, angle=kL["angle"]).upload(fv1)
F2 = amd.FrameView(kR["x"], kR["y"], kR["octave"].astype(np.int32), dR, b, angle=kR["angle"]).upload(fv2)
has1 = np.ones(len(kL), np.uint8)
M = amd.ORBmatcher(0.7, True)
M.SearchByBoWResident(F1, has1, F2)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); M.SearchByBoWResident(F1, has1, F2); ts.append(time.perf_counter() - t0)
print("SearchByBoW resident median ms", 1e3 * float(np.median(ts)))
