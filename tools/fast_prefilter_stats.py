#!/usr/bin/env python3
"""What fraction of the pixels does each necessary test of k_fast_cells' phase A let through?  (CPU, numpy)
A 9-of-16 arc always contains ring point 0 (S) or 8 (N) -- its 7-point complement cannot hold both -- and likewise 4 (E) or
12 (W); and it always covers two ADJACENT cardinal points.  So, with b(p) = I(p) > v + t and d(p) = I(p) < v - t:
  one pair   : (bS|bN) | (dS|dN)                                   -- ONE compare pair per polarity
  two pairs  : ((bS|bN) & (bE|bW)) | ((dS|dN) & (dE|dW))           -- the kernel's exact packed-16 form
  quantised  : the same on x >> 2 with t >> 2 (the SWAR form used for t >= 16; a weaker condition)
  adjacent   : two adjacent cardinals in one polarity (what "two pairs" over-approximates)
  corner     : the exact FAST-9/16 test
on the pyramid levels of the bench's synthetic frames (synth.render_stereo_textured / render_sequence), detection rectangles only."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle_lib as orc  # noqa: E402
from orb_slam2_annotate_amd import synth  # noqa: E402

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def stats(img, t):
    I = img.astype(np.int16)
    h, w = I.shape
    c = I[19:h - 19, 19:w - 19]
    ring = [I[19 + dy:h - 19 + dy, 19 + dx:w - 19 + dx] for dx, dy in RING]
    b = [r > c + t for r in ring]
    d = [r < c - t for r in ring]
    S, E, N, W = 0, 4, 8, 12
    one = (b[S] | b[N]) | (d[S] | d[N])
    two = ((b[S] | b[N]) & (b[E] | b[W])) | ((d[S] | d[N]) & (d[E] | d[W]))
    q = lambda x: x >> 2  # noqa: E731
    t6 = t >> 2
    bq = [q(r) >= q(c) + t6 for r in ring]
    dq = [q(r) <= q(c) - t6 for r in ring]
    quant = ((bq[S] | bq[N]) & (bq[E] | bq[W])) | ((dq[S] | dq[N]) & (dq[E] | dq[W]))
    adj = np.zeros_like(one)
    for x, y in ((S, E), (E, N), (N, W), (W, S)):
        adj |= (b[x] & b[y]) | (d[x] & d[y])
    corner = np.zeros_like(one)
    for pol in (b, d):
        m = np.stack(pol + pol[:8])  # 24 planes: arcs of 9 starting at 0..15
        run = np.zeros(one.shape, np.int8)
        for k in range(24):
            run = np.where(m[k], run + 1, 0).astype(np.int8)
            corner |= run >= 9
    n = one.size
    return n, [int(x.sum()) for x in (one, two, quant, adj, corner)]


def main():
    rows = []
    for name, frames, nf in (("kitti textured stereo 1241x376", [synth.render_stereo_textured(5000 + i, 1241, 376)[0] for i in range(3)], 2000),
                             ("tum 640x480 sequence", synth.render_sequence(1000, 3, 640, 480, step=1.5), 1000)):
        o = orc.Oracle(nf, 1.2, 8, 20, 7)
        for t in (20, 7):
            tot, acc = 0, np.zeros(5, np.int64)
            for fr in frames:
                _, _, pyr = o.extract(fr, want_pyramid=True)
                off = 0
                for (lw, lh) in o.level_sizes(fr.shape[1], fr.shape[0]):
                    lv = np.frombuffer(pyr, np.uint8, lw * lh, off).reshape(lh, lw)
                    off += lw * lh
                    if lw < 45 or lh < 45:
                        continue
                    n, s = stats(lv, t)
                    tot += n
                    acc += s
            rows.append((name, t, tot, acc / tot))
    print("# fraction of the detection-rectangle pixels (all pyramid levels) that PASS each test of FAST phase A; tools/fast_prefilter_stats.py")
    print("%-34s %3s  %9s %9s %9s %9s %9s" % ("frames", "t", "one pair", "two pairs", "quantised", "adjacent", "corner"))
    for name, t, tot, f in rows:
        print("%-34s %3d  %9.4f %9.4f %9.4f %9.4f %9.4f" % (name, t, *f))


if __name__ == "__main__":
    main()
