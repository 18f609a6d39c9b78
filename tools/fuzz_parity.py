#!/usr/bin/env python3
"""Randomised differential test of the whole extractor against the CPU oracle: random image sizes,
extractor parameters and image content (rendered shapes, noise, blends, gradients, blocky textures).
    python tools/fuzz_parity.py [seconds] [seed] [batch]
With a third argument every case is a BATCH of 9..24 frames on 1..8 sub-batch streams (the throughput
kernels: global-memory octree, 64-keypoint descriptor workgroups, multi-copy D2H).
Exits non-zero on the first mismatch and prints the failing configuration."""
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


def content(rng, w, h):
    kind = rng.integers(0, 6)
    if kind == 0:
        return synth.render_frame(int(rng.integers(1 << 30)), w, h, n_shapes=int(rng.integers(5, 400)))
    if kind == 1:
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 2:  # shapes + noise
        a = synth.render_frame(int(rng.integers(1 << 30)), w, h).astype(np.int16)
        return np.clip(a + rng.normal(0, rng.uniform(1, 25), (h, w)), 0, 255).astype(np.uint8)
    if kind == 3:  # blocky texture (many equal scores -> NMS / octree ties)
        b = int(rng.integers(2, 9))
        t = rng.integers(0, 256, ((h + b - 1) // b, (w + b - 1) // b), dtype=np.uint8)
        return np.kron(t, np.ones((b, b), np.uint8))[:h, :w].copy()
    if kind == 4:  # low-contrast gradient with sparse impulses (threshold fallback cells)
        y, x = np.mgrid[0:h, 0:w]
        a = ((x * 0.11 + y * 0.07) % 256).astype(np.int16)
        m = rng.random((h, w)) < 0.002
        a[m] += rng.integers(8, 60, m.sum())
        return np.clip(a, 0, 255).astype(np.uint8)
    a = np.full((h, w), int(rng.integers(0, 256)), np.uint8)  # near-constant
    a[rng.random((h, w)) < 0.0005] = int(rng.integers(0, 256))
    return a


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    batch_mode = len(sys.argv) > 3
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n = 0
    skipped = 0
    last_report = t0
    while time.time() - t0 < seconds:
        w, h = int(rng.integers(64, 900)), int(rng.integers(64, 700))
        if rng.random() < 0.2:
            w, h = [(640, 480), (752, 480), (1241, 376), (320, 240)][rng.integers(0, 4)]
        params = (int(rng.choice([100, 500, 1000, 2000, 3000])), float(rng.choice([1.1, 1.2, 1.3, 1.5])),
                  int(rng.integers(1, 9)), int(rng.choice([20, 30, 12, 7, 16, 17, 19, 23, 64, 120, 255])), int(rng.choice([7, 5, 10, 3, 16, 24])))  # (thresholds on both sides of the FAST pre-test's exact / quantised switch at 16)
        img = content(rng, w, h)
        # round-2 knobs: GaussianBlur arithmetic variant, blur fused into the FAST kernel, lane schedule, host path
        spec = int(rng.choice([0, 0, 1, 2]))
        fused, lanes, piped = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        fmode = str(rng.choice(["auto", "high", "low"]))
        pyrblur = bool(rng.integers(0, 2))
        tiles = bool(rng.integers(0, 2))  # round 3: orientation + descriptors per 128 x 128 tile (k_orient_desc_tiles)
        knobs = dict(spec=spec, fused=fused, lanes=lanes, pipelined=piped, fast=fmode, pyrblur=pyrblur, desc_tiles=tiles)
        try:
            o = orc.Oracle(*params, blur_spec=spec)
            e = amd.ORBextractor(*params)
            e.set_blur_spec(spec)
            e.set_fused(fused)
            e.set_pyramid_blur(pyrblur)
            e.set_fast_mode(fmode)
            e.set_desc_tiles(tiles)
            if batch_mode:
                w, h = min(w, 500), min(h, 400)
                imgs = np.stack([content(rng, w, h) for _ in range(int(rng.integers(9, 25)))])
                img = imgs[0]
                e.set_streams(int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 32])))
                e.set_schedule(lanes)
                res = e.extract_batch_pipelined(imgs, chunk_frames=int(rng.integers(1, 12))) if piped else e.extract_batch(imgs)
                ok = True
                for i in range(len(imgs)):
                    kr, dr = o.extract(imgs[i])
                    kg, dg = res[i]
                    ok = ok and len(kr) == len(kg) and np.array_equal(dr, dg) and all(
                        np.array_equal(kr[f], kg[f]) for f in ("x", "y", "size", "angle", "response", "octave"))
            else:
                kr, dr = o.extract(img)
                kg, dg = e(img)
                ok = len(kr) == len(kg) and np.array_equal(dr, dg) and all(
                    np.array_equal(kr[f], kg[f]) for f in ("x", "y", "size", "angle", "response", "octave"))
        except amd.OrbfeError as ex:
            print("EXCEPTION", ex)
            ok = False
        except Exception as ex:  # noqa: BLE001
            print("EXCEPTION", type(ex).__name__, ex)
            ok = False
        if not ok:
            print(f"MISMATCH seed={seed} case={n} size={w}x{h} params={params} knobs={knobs}")
            np.save(ROOT / "gpurun_out" / f"fuzz_fail_{seed}_{n}.npy", img)
            sys.exit(1)
        n += 1
        if time.time() - last_report > 60:  # progress line (long runs must not look hung)
            last_report = time.time()
            print(f"  ... {n} cases so far", flush=True)
    print(f"fuzz_parity: {n} random cases bit-exact in {time.time() - t0:.0f} s (seed {seed}; blur specs 0/1/2, fused / separate blur, pyramid+blur fused / separate, tile / per-keypoint descriptors, stream / lane schedule, direct / pipelined host path drawn at random)")


if __name__ == "__main__":
    main()
