"""CPU tests of the oracle's Frame grid / projection searches (oracle/orb_oracle.c, restating
src/Frame.cc:246-267,358-427 and src/ORBmatcher.cc:51-138,1484-1633) against a literal
pure-Python transcription of the same loops on small inputs."""
import math

import numpy as np
import pytest

import oracle_lib as O


def c_round(v):  # C round(): half away from zero
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def py_grid(x, y, bounds):
    minx, maxx, miny, maxy = bounds
    f32 = np.float32
    winv = f32(64.0) / (f32(maxx) - f32(minx))
    hinv = f32(48.0) / (f32(maxy) - f32(miny))
    grid = [[[] for _ in range(48)] for _ in range(64)]
    for i in range(len(x)):
        px = c_round(float((f32(x[i]) - f32(minx)) * winv))
        py = c_round(float((f32(y[i]) - f32(miny)) * hinv))
        if 0 <= px < 64 and 0 <= py < 48:
            grid[px][py].append(i)
    return grid, winv, hinv


def py_area(grid, winv, hinv, X, Y, octv, bounds, x, y, r, lo, hi):
    f32 = np.float32
    minx, _, miny, _ = [f32(b) for b in bounds]
    x, y, r = f32(x), f32(y), f32(r)
    out = []
    a = max(0, int(math.floor(float((x - minx - r) * winv))))
    if a >= 64:
        return out
    b = min(63, int(math.ceil(float((x - minx + r) * winv))))
    if b < 0:
        return out
    c = max(0, int(math.floor(float((y - miny - r) * hinv))))
    if c >= 48:
        return out
    d = min(47, int(math.ceil(float((y - miny + r) * hinv))))
    if d < 0:
        return out
    chk = lo > 0 or hi >= 0
    for ix in range(a, b + 1):
        for iy in range(c, d + 1):
            for i in grid[ix][iy]:
                if chk:
                    if octv[i] < lo:
                        continue
                    if hi >= 0 and octv[i] > hi:
                        continue
                if abs(f32(X[i]) - x) < r and abs(f32(Y[i]) - y) < r:
                    out.append(i)
    return out


def random_frame(rng, n, w=640, h=480, stereo=False):
    x = rng.uniform(-5, w + 5, n).astype(np.float32)
    y = rng.uniform(-5, h + 5, n).astype(np.float32)
    octv = rng.integers(0, 8, n).astype(np.int32)
    ang = rng.uniform(0, 360, n).astype(np.float32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ur = None
    if stereo:
        ur = np.where(rng.random(n) < 0.7, x - rng.uniform(1, 40, n), -1).astype(np.float32)
    return x, y, octv, ang, desc, ur


@pytest.mark.parametrize("seed,n", [(0, 0), (1, 1), (2, 300), (3, 1500)])
def test_features_in_area_matches_python(seed, n):
    rng = np.random.default_rng(seed)
    bounds = (0.0, 640.0, 0.0, 480.0) if seed % 2 == 0 else (-12.5, 655.25, -8.0, 490.5)
    x, y, octv, ang, desc, _ = random_frame(rng, n)
    F = O.Frame(x, y, octv, desc, bounds, angle=ang)
    grid, winv, hinv = py_grid(x, y, bounds)
    for _ in range(60):
        qx, qy = rng.uniform(-30, 700), rng.uniform(-30, 520)
        r = rng.choice([0.5, 3.0, 15.0, 64.0, 900.0])
        lo, hi = [(-1, -1), (0, -1), (2, -1), (0, 3), (1, 2), (3, 1)][rng.integers(0, 6)]
        want = py_area(grid, winv, hinv, x, y, octv, bounds, qx, qy, r, lo, hi)
        got = F.features_in_area(qx, qy, r, lo, hi)
        assert got.tolist() == want


def test_grid_drops_features_outside_bounds():
    x = np.array([-10.0, 0.0, 639.9, 645.0, 320.0], np.float32)
    y = np.array([10.0, 0.0, 479.9, 10.0, 500.0], np.float32)
    F = O.Frame(x, y, np.zeros(5, np.int32), np.zeros((5, 32), np.uint8), (0, 640, 0, 480))
    # round(639.9*0.1) = 64 -> outside: PosInGrid rejects the right-most half cell (src/Frame.cc:422-424)
    assert F.features_in_area(320, 240, 2000).tolist() == [1]


def _scene(rng, n, stereo):
    x, y, octv, ang, desc, ur = random_frame(rng, n, stereo=stereo)
    x = np.clip(x, 0, 639).astype(np.float32)
    y = np.clip(y, 0, 479).astype(np.float32)
    return x, y, octv, ang, desc, ur


def test_lastframe_python_transcription():
    """literal transcription of src/ORBmatcher.cc:1525-1628 on a small case"""
    rng = np.random.default_rng(5)
    sf = (1.2 ** np.arange(8)).astype(np.float32)
    for mode, stereo in [(0, False), (1, True), (2, True), (0, True)]:
        x, y, octv, ang, desc, ur = _scene(rng, 400, stereo)
        Cur = O.Frame(x, y, octv, desc, (0, 640, 0, 480), angle=ang, u_right=ur)
        nl = 350
        src = rng.integers(0, 400, nl)
        u = (x[src] + rng.normal(0, 3, nl)).astype(np.float32)
        v = (y[src] + rng.normal(0, 3, nl)).astype(np.float32)
        lo = np.clip(octv[src] + rng.integers(-1, 2, nl), 0, 7).astype(np.int32)
        la = ((ang[src] + rng.normal(0, 5, nl)) % 360).astype(np.float32)
        md = desc[src].copy()
        flip = rng.integers(0, 256, (nl, 32), dtype=np.uint8) & rng.integers(0, 256, (nl, 32), dtype=np.uint8) & \
            rng.integers(0, 256, (nl, 32), dtype=np.uint8)
        md ^= flip
        valid = (rng.random(nl) < 0.9).astype(np.uint8)
        invzc = rng.uniform(0.02, 0.5, nl).astype(np.float32)
        obs = (rng.random(nl) < 0.8).astype(np.uint8)
        mbf, th = 40.0, 7.0 if stereo else 15.0
        n, m = O.search_by_projection_lastframe(Cur, sf, mbf, valid, u, v, invzc, lo, la, md, obs, mode, th, True)
        # transcription
        grid, winv, hinv = py_grid(x, y, (0, 640, 0, 480))
        match = np.full(400, -1, np.int64)
        blocked = np.zeros(400, bool)
        hist = [[] for _ in range(30)]
        nm = 0
        for i in range(nl):
            if not valid[i]:
                continue
            radius = np.float32(th) * sf[lo[i]]
            if mode == 1:
                cand = py_area(grid, winv, hinv, x, y, octv, (0, 640, 0, 480), u[i], v[i], radius, lo[i], -1)
            elif mode == 2:
                cand = py_area(grid, winv, hinv, x, y, octv, (0, 640, 0, 480), u[i], v[i], radius, 0, lo[i])
            else:
                cand = py_area(grid, winv, hinv, x, y, octv, (0, 640, 0, 480), u[i], v[i], radius, lo[i] - 1, lo[i] + 1)
            best, bi = 256, -1
            for i2 in cand:
                if blocked[i2]:
                    continue
                if ur is not None and ur[i2] > 0:
                    uu = np.float32(u[i]) - np.float32(mbf) * invzc[i]
                    if abs(np.float32(uu - ur[i2])) > radius:
                        continue
                d = int(np.unpackbits(md[i] ^ desc[i2]).sum())
                if d < best:
                    best, bi = d, i2
            if best <= 100:
                match[bi] = i
                blocked[bi] = bool(obs[i])
                nm += 1
                rot = np.float32(la[i]) - np.float32(ang[bi])
                if rot < 0:
                    rot = np.float32(rot + np.float32(360.0))
                b = c_round(float(np.float32(rot * np.float32(1.0 / 30))))
                if b == 30:
                    b = 0
                hist[b].append(bi)
        i1, i2_, i3 = O.three_maxima([len(h) for h in hist])
        for b in range(30):
            if b in (i1, i2_, i3):
                continue
            for j in hist[b]:
                match[j] = -1
                nm -= 1
        assert n == nm
        assert m.tolist() == match.tolist()
        assert n > 20


def test_initialization_python_transcription():
    """literal transcription of src/ORBmatcher.cc:469-603 on a small case (duplicates, re-assignment,
    the 'only still-matched entries are pruned' rule)"""
    rng = np.random.default_rng(9)
    n1 = n2 = 500
    x, y, octv, ang, desc, _ = _scene(rng, n2, False)
    octv = np.where(rng.random(n2) < 0.6, 0, octv).astype(np.int32)
    F2 = O.Frame(x, y, octv, desc, (0, 640, 0, 480), angle=ang)
    src = rng.integers(0, n2, n1)
    x1 = (x[src] + rng.normal(0, 4, n1)).astype(np.float32)
    y1 = (y[src] + rng.normal(0, 4, n1)).astype(np.float32)
    o1 = np.where(rng.random(n1) < 0.7, 0, 1).astype(np.int32)
    a1 = ((ang[src] + rng.normal(0, 8, n1)) % 360).astype(np.float32)
    d1 = desc[src] ^ (rng.integers(0, 256, (n1, 32), dtype=np.uint8) & rng.integers(0, 256, (n1, 32), dtype=np.uint8) &
                      rng.integers(0, 256, (n1, 32), dtype=np.uint8))
    F1 = O.Frame(x1, y1, o1, d1, (0, 640, 0, 480), angle=a1)
    prev = np.stack([x1, y1], axis=1).copy()
    n, m12, prev_out = O.search_for_initialization(F1, F2, prev, 30, 0.9, True)
    grid, winv, hinv = py_grid(x, y, (0, 640, 0, 480))
    BIG = 2 ** 31 - 1
    match12 = [-1] * n1
    match21 = [-1] * n2
    mdist = [BIG] * n2
    hist = [[] for _ in range(30)]
    nm = 0
    for i1 in range(n1):
        if o1[i1] > 0:
            continue
        cand = py_area(grid, winv, hinv, x, y, octv, (0, 640, 0, 480), prev[i1, 0], prev[i1, 1], 30.0, 0, 0)
        best, best2, bi = BIG, BIG, -1
        for i2 in cand:
            d = int(np.unpackbits(d1[i1] ^ desc[i2]).sum())
            if mdist[i2] <= d:
                continue
            if d < best:
                best2, best, bi = best, d, i2
            elif d < best2:
                best2 = d
        if best <= 50 and np.float32(best) < np.float32(best2) * np.float32(0.9):
            if match21[bi] >= 0:
                match12[match21[bi]] = -1
                nm -= 1
            match12[i1] = bi
            match21[bi] = i1
            mdist[bi] = best
            nm += 1
            rot = np.float32(a1[i1]) - np.float32(ang[bi])
            if rot < 0:
                rot = np.float32(rot + np.float32(360.0))
            b = c_round(float(np.float32(rot * np.float32(1.0 / 30))))
            hist[0 if b == 30 else b].append(i1)
    k1, k2, k3 = O.three_maxima([len(h) for h in hist])
    for b in range(30):
        if b in (k1, k2, k3):
            continue
        for i1 in hist[b]:
            if match12[i1] >= 0:
                match12[i1] = -1
                nm -= 1
    assert (n, m12.tolist()) == (nm, match12)
    assert n > 50
    for i1 in range(n1):
        want = (x[match12[i1]], y[match12[i1]]) if match12[i1] >= 0 else (prev[i1, 0], prev[i1, 1])
        assert tuple(prev_out[i1]) == want
