"""CPU tests of the PRODUCT's host logic (liborbfe.so, no GPU needed) against the oracle:
constructor tables, pyramid/grid geometry, cv::resize coefficient tables, and the library's host
DistributeOctTree (the twin of the device kernel) on random candidate sets."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc
from orb_slam2_annotate_amd import _lib, synth


def _geometry(params, W, H):
    L = _lib.load()
    nl = params[2]
    lev = np.zeros((nl, 9), np.int32)
    tab = np.zeros((nl, 4), np.float32)
    n = L.orbfe_debug_geometry(params[0], params[1], nl, params[3], params[4], W, H, _lib.ptr(lev), _lib.ptr(tab),
                               None, 0)
    assert n >= 0
    cells = np.zeros((max(n, 1), 5), np.int16)
    assert L.orbfe_debug_geometry(params[0], params[1], nl, params[3], params[4], W, H, _lib.ptr(lev), _lib.ptr(tab),
                                  _lib.ptr(cells), n) == n
    return lev, tab, cells[:n]


@pytest.mark.parametrize("params", [(1000, 1.2, 8, 20, 7), (2000, 1.2, 8, 20, 7), (1200, 1.2, 8, 20, 7),
                                    (777, 1.3, 5, 25, 9), (100, 2.0, 3, 20, 7)])
@pytest.mark.parametrize("wh", [(640, 480), (752, 480), (1241, 376), (321, 243), (96, 80)])
def test_tables_and_geometry_match_oracle(params, wh):
    W, H = wh
    o = orc.Oracle(*params)
    lev, tab, cells = _geometry(params, W, H)
    assert np.array_equal(tab[:, 0], o.scale_factors()) and np.array_equal(tab[:, 1], o.inv_scale_factors())
    assert np.array_equal(tab[:, 2], o.level_sigma2()) and np.array_equal(tab[:, 3], o.inv_level_sigma2())
    assert [tuple(x) for x in lev[:, :2]] == o.level_sizes(W, H)
    assert list(lev[:, 7]) == o.features_per_level()
    # SURVEY.md 8 table for the three benchmark resolutions (level 0 grid)
    known = {(640, 480): (20, 14, 31, 32), (1241, 376): (40, 11, 31, 32)}
    if wh in known and params[1] == 1.2:
        assert tuple(lev[0, 2:6]) == known[wh]
    # the cells tile the detection area [19, w-20] x [19, h-20] of every level exactly once
    for l in range(params[2]):
        w, h = lev[l, 0], lev[l, 1]
        cover = np.zeros((h, w), np.int32)
        for c in cells[cells[:, 0] == l]:
            cover[c[2]:c[2] + c[4], c[1]:c[1] + c[3]] += 1
        if lev[l, 2] > 0 and lev[l, 3] > 0:
            assert (cover[19:h - 19, 19:w - 19] == 1).all() and cover.sum() == max(w - 38, 0) * max(h - 38, 0)
        else:
            assert cover.sum() == 0


@pytest.mark.parametrize("sizes", [(640, 480, 533, 400), (1241, 376, 1034, 313), (179, 134, 149, 112), (50, 40, 100, 80)])
def test_resize_tables_reproduce_oracle_resize(sizes):
    """The product's coefficient tables, applied with the kernel's integer formula in numpy, must
    reproduce the oracle's cv::resize byte for byte."""
    sw, sh, dw, dh = sizes
    L = _lib.load()
    xofs = np.zeros(dw, np.int32); alpha = np.zeros(2 * dw, np.int16)
    yofs = np.zeros(dh, np.int32); beta = np.zeros(2 * dh, np.int16)
    assert L.orbfe_debug_resize_tables(sw, sh, dw, dh, _lib.ptr(xofs), _lib.ptr(alpha), _lib.ptr(yofs), _lib.ptr(beta)) == 0
    img = synth.adversarial("noise", sw, sh, seed=2).astype(np.int64)
    sx1 = np.minimum(xofs + 1, sw - 1)
    a0, a1 = alpha[0::2].astype(np.int64), alpha[1::2].astype(np.int64)
    hrow = img[:, xofs] * a0 + img[:, sx1] * a1  # [sh, dw]
    r0 = np.clip(yofs, 0, sh - 1); r1 = np.clip(yofs + 1, 0, sh - 1)
    b0, b1 = beta[0::2].astype(np.int64)[:, None], beta[1::2].astype(np.int64)[:, None]
    out = ((((b0 * (hrow[r0] >> 4)) >> 16) + ((b1 * (hrow[r1] >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
    assert np.array_equal(out, orc.resize_linear(img.astype(np.uint8), dw, dh))


@pytest.mark.parametrize("sizes", [(640, 480, 533, 400), (1241, 376, 1034, 313), (214, 161, 179, 134), (752, 480, 627, 400),
                                   (900, 700, 600, 467), (300, 200, 273, 182), (70, 66, 58, 55)])
def test_fused_blur_resize_tile_ownership(sizes):
    """Every 4-column group / output row of the next pyramid level belongs to exactly one 64 x 64 source tile, its tap
    window (8 bytes from group_start; rows row_upper and row_upper + 1) lies inside what that tile stages (columns
    bx-4 .. bx+75, rows by-3 .. by+66), and a tile owns at most 16 groups and 80 rows (the kernel's LDS slots)."""
    import ctypes as C
    sw, sh, dw, dh = sizes
    L = _lib.load()
    tx, ty = (sw + 63) // 64, (sh + 63) // 64
    gx = np.zeros(tx + 1, np.int32); dy = np.zeros(ty + 1, np.int32)
    ngx = (dw + 3) // 4
    gs = np.zeros(ngx, np.int32); ru = np.zeros(dh, np.int32)
    nx, ny = C.c_int(0), C.c_int(0)
    assert L.orbfe_debug_resize_tiles(sw, sh, dw, dh, _lib.ptr(gx), C.byref(nx), _lib.ptr(dy), C.byref(ny), _lib.ptr(gs), _lib.ptr(ru)) == 0
    assert (nx.value, ny.value) == (tx, ty)
    assert gx[0] == 0 and gx[-1] == ngx and dy[0] == 0 and dy[-1] == dh
    assert (np.diff(gx) >= 0).all() and (np.diff(dy) >= 0).all()
    assert np.diff(gx).max() <= 16 and np.diff(dy).max() <= 80
    for t in range(tx):
        g = np.arange(gx[t], gx[t + 1])
        assert ((gs[g] >= 64 * t) & (gs[g] < 64 * (t + 1))).all()          # window start inside the tile's core
        assert (gs[g] + 7 <= 64 * t + 75).all() and (gs[g] + 7 < sw).all()  # whole window staged, and inside the image
    for t in range(ty):
        r = np.arange(dy[t], dy[t + 1])
        assert ((ru[r] >= 64 * t) & (ru[r] < 64 * (t + 1))).all()
        assert (np.minimum(ru[r] + 1, sh - 1) <= 64 * t + 66).all()


def _octree_product(xs, ys, rs, minX, maxX, minY, maxY, N):
    L = _lib.load()
    n = len(xs)
    cap = N + 64
    ox = np.zeros(cap, np.uint16); oy = np.zeros(cap, np.uint16); orr = np.zeros(cap, np.uint8)
    xs = np.ascontiguousarray(xs, np.uint16); ys = np.ascontiguousarray(ys, np.uint16); rs = np.ascontiguousarray(rs, np.uint8)
    k = L.orbfe_debug_octree_host(_lib.ptr(xs), _lib.ptr(ys), _lib.ptr(rs), n, minX, maxX, minY, maxY, N,
                                  _lib.ptr(ox), _lib.ptr(oy), _lib.ptr(orr), cap)
    assert 0 <= k <= cap
    return ox[:k], oy[:k], orr[:k]


@pytest.mark.parametrize("seed", range(12))
def test_host_octree_equals_oracle_on_random_sets(seed):
    rng = np.random.default_rng(seed)
    w, h = [(640, 480), (1241, 376), (300, 100), (2000, 150)][seed % 4]
    minX = minY = 16
    maxX, maxY = w - 16, h - 16
    n = int(rng.integers(1, 4000))
    N = int(rng.integers(1, 500))
    # distinct pixels; clustered half the time (deep splits, many equal counts -> tie-break paths)
    if seed % 2:
        cx = rng.integers(0, maxX - minX, size=6); cy = rng.integers(0, maxY - minY, size=6)
        pts = np.stack([np.clip(cx[rng.integers(0, 6, n)] + rng.integers(-25, 25, n), 0, maxX - minX - 1),
                        np.clip(cy[rng.integers(0, 6, n)] + rng.integers(-25, 25, n), 0, maxY - minY - 1)], 1)
    else:
        pts = np.stack([rng.integers(0, maxX - minX, n), rng.integers(0, maxY - minY, n)], 1)
    pts = np.unique(pts, axis=0)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]  # raster-ish emission order
    rs = rng.integers(7, 60, len(pts))  # few distinct responses -> response ties inside nodes
    sel = orc.distribute_octtree(pts[:, 0], pts[:, 1], rs, minX, maxX, minY, maxY, N)
    ox, oy, orr = _octree_product(pts[:, 0], pts[:, 1], rs, minX, maxX, minY, maxY, N)
    assert len(sel) == len(ox)
    assert np.array_equal(ox, pts[sel, 0] + minX) and np.array_equal(oy, pts[sel, 1] + minY)
    assert np.array_equal(orr, rs[sel])
    assert len(sel) <= max(N + 2, 4 * round((maxX - minX) / (maxY - minY)))


def test_host_octree_edge_cases():
    one = _octree_product([5], [7], [30], 16, 624, 16, 464, 100)
    assert list(one[0]) == [21] and list(one[1]) == [23] and list(one[2]) == [30]
    none = _octree_product([], [], [], 16, 624, 16, 464, 100)
    assert len(none[0]) == 0
    # quota 0: the first pass still splits the roots once (Appendix A3)
    xs = np.arange(0, 600, 7); ys = (xs * 3) % 440
    sel = orc.distribute_octtree(xs, ys, np.full(len(xs), 20), 16, 624, 16, 464, 0)
    got = _octree_product(xs, ys, np.full(len(xs), 20), 16, 624, 16, 464, 0)
    assert len(got[0]) == len(sel) and np.array_equal(got[0], xs[sel] + 16)


def test_udiv_magic_is_exact_for_all_32_bit_operands():
    """csrc/kernels.h udiv_magic: q = umulhi(n, M) (+1 if n - q*d >= d) with M = floor(2^32/d) (2^32-1 for
    d = 1) must equal n // d for every 32-bit n -- checked on the extremes and a random sample."""
    rng = np.random.default_rng(0)
    ds = np.concatenate([np.arange(1, 70), rng.integers(1, 1 << 16, 400), rng.integers(1, 1 << 32, 400),
                         np.array([(1 << 32) - 1, (1 << 31), (1 << 31) + 1, 65535, 65536, 65537])]).astype(np.uint64)
    for d in ds:
        d = int(d)
        M = 0xFFFFFFFF if d <= 1 else (1 << 32) // d
        ns = np.concatenate([np.array([0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 32) - 1, (1 << 32) - d,
                                       ((1 << 32) - 1) // d * d, ((1 << 32) - 1) // d * d - 1], dtype=np.int64),
                             rng.integers(0, 1 << 32, 300)])
        ns = ns[(ns >= 0) & (ns < (1 << 32))].astype(np.uint64)
        q = (ns * np.uint64(M)) >> np.uint64(32)
        q = q + ((ns - q * np.uint64(d)) >= np.uint64(d)).astype(np.uint64)
        assert np.array_equal(q, ns // np.uint64(d)), d


def test_textured_stereo_scene_is_matchable():
    """The bench's default stereo input (synth.render_stereo_textured): >= 50 % of the left keypoints obtain a stereo
    match in the oracle's Frame::ComputeStereoMatches, with sub-pixel disparities -- the round-2 'shapes' scene gave 14 %."""
    import oracle_lib as orc
    from orb_slam2_annotate_amd import synth
    w, h = 1241, 376
    left, right = synth.render_stereo_textured(5001, w, h)  # bench.py's second pair (seeds 5000..: 0.50-0.60, mean 0.56)
    o = orc.Oracle(2000, 1.2, 8, 20, 7)
    kL, dL, pL = o.extract(left, want_pyramid=True)
    kR, dR, pR = o.extract(right, want_pyramid=True)
    mbf = np.float32(386.1448)
    mb = np.float32(mbf / np.float32(718.856))
    orc.distance_calls_reset()
    u, dep = o.stereo(w, h, kL, dL, kR, dR, pL, pR, float(mbf), float(mb))
    m = u >= 0
    assert m.sum() >= 0.5 * len(kL) and m.sum() >= 1000
    disp = kL["x"][m] - u[m]
    assert (np.abs(disp - np.rint(disp)) > 0.05).mean() > 0.5  # sub-pixel, not integer shifts
    scanned, sads = orc.stereo_counters()
    assert scanned > 5e4 and sads >= m.sum() and orc.distance_calls() > 1e4
