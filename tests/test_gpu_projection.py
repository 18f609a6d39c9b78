"""GPU parity tests of the Frame grid / GetFeaturesInArea primitive and the two tracking-thread
projection searches (include/orbfe.h, "Tracking-thread projection searches") against the CPU
oracle -- bit-exact match arrays and counts."""
import numpy as np
import pytest

import oracle_lib as orc
from orb_slam2_annotate_amd import synth

pytestmark = pytest.mark.gpu

SF = (1.2 ** np.arange(8)).astype(np.float32)
BOUNDS = (0.0, 640.0, 0.0, 480.0)


@pytest.fixture(scope="module")
def amd():
    import orb_slam2_annotate_amd as m
    return m


def _random_frame(rng, n, stereo=False, spread=5.0):
    x = rng.uniform(-spread, 640 + spread, n).astype(np.float32)
    y = rng.uniform(-spread, 480 + spread, n).astype(np.float32)
    octv = rng.integers(0, 8, n).astype(np.int32)
    ang = rng.uniform(0, 360, n).astype(np.float32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ur = np.where(rng.random(n) < 0.7, x - rng.uniform(1, 40, n), -1).astype(np.float32) if stereo else None
    return x, y, octv, ang, desc, ur


RESIDENT = False


@pytest.fixture(autouse=True, params=["host_arrays", "resident_frame"])
def _frame_kind(request):
    """every search of this file runs twice: on host-array views (everything uploaded per call) and on frames made
    resident with orbfe_frame_upload (keypoints, descriptors and the grid already on the device)"""
    global RESIDENT
    RESIDENT = request.param == "resident_frame"
    yield
    RESIDENT = False


def _both(amd, x, y, octv, ang, desc, ur, bounds=BOUNDS):
    F = amd.FrameView(x, y, octv, desc, bounds, angle=ang, u_right=ur)
    if RESIDENT:
        F = F.upload()
    return F, orc.Frame(x, y, octv, desc, bounds, angle=ang, u_right=ur)


@pytest.mark.parametrize("seed,n,bounds", [(0, 1000, BOUNDS), (1, 2300, (-12.5, 655.25, -8.0, 490.5)), (2, 1, BOUNDS),
                                           (3, 16384, BOUNDS)])
def test_features_in_area(amd, seed, n, bounds):
    rng = np.random.default_rng(seed)
    x, y, octv, ang, desc, _ = _random_frame(rng, n)
    F, Fo = _both(amd, x, y, octv, ang, desc, None, bounds)
    nq = 400
    qx = rng.uniform(-30, 700, nq).astype(np.float32)
    qy = rng.uniform(-30, 520, nq).astype(np.float32)
    r = rng.choice(np.array([0.5, 3.0, 15.0, 64.0, 900.0], np.float32), nq)
    lv = np.array([(-1, -1), (0, -1), (2, -1), (0, 3), (1, 2), (3, 1)], np.int32)[rng.integers(0, 6, nq)]
    got = F.GetFeaturesInArea(qx, qy, r, lv[:, 0], lv[:, 1], capacity=16)  # grows past 16 by itself
    for q in range(nq):
        want = Fo.features_in_area(qx[q], qy[q], r[q], lv[q, 0], lv[q, 1])
        assert got[q].tolist() == want.tolist(), q


def test_features_in_area_capacity_error_and_empty(amd):
    import ctypes as C
    from orb_slam2_annotate_amd import _lib
    rng = np.random.default_rng(4)
    x, y, octv, ang, desc, _ = _random_frame(rng, 500)
    F = amd.FrameView(x, y, octv, desc, BOUNDS, angle=ang)
    f32 = lambda v: np.array([v], np.float32)
    i32 = lambda v: np.array([v], np.int32)
    count = np.zeros(1, np.int32); idx = np.zeros(4, np.int32)
    L = _lib.load()
    rc = L.orbfe_features_in_area(0, C.byref(F.c), 1, _lib.ptr(f32(320)), _lib.ptr(f32(240)), _lib.ptr(f32(1000)),
                                  _lib.ptr(i32(-1)), _lib.ptr(i32(-1)), 4, _lib.ptr(count), _lib.ptr(idx))
    assert rc == _lib.ERR_CAPACITY
    gridded = ((np.round((x - 0) * np.float32(0.1)) < 64) & (np.round(x * np.float32(0.1)) >= 0) &
               (np.round(y * np.float32(0.1)) < 48) & (np.round(y * np.float32(0.1)) >= 0))
    assert count[0] == orc.Frame(x, y, octv, desc, BOUNDS).features_in_area(320, 240, 1000).size <= gridded.sum()
    # empty frame, no queries
    E = amd.FrameView(np.zeros(0), np.zeros(0), np.zeros(0), np.zeros((0, 32)), BOUNDS)
    assert [a.tolist() for a in E.GetFeaturesInArea([10.0], [10.0], [50.0])] == [[]]
    assert F.GetFeaturesInArea(np.zeros(0), np.zeros(0), np.zeros(0)) == []
    # degenerate bounds are rejected
    with pytest.raises(amd.OrbfeError):
        amd.FrameView(x, y, octv, desc, (0, 0, 0, 480)).GetFeaturesInArea([1.0], [1.0], [1.0])


def _map_points(rng, x, y, octv, desc, n_mp, noise_px, bit_noise=3):
    n = len(x)
    src = rng.integers(0, n, n_mp)
    px = (x[src] + rng.normal(0, noise_px, n_mp)).astype(np.float32)
    py = (y[src] + rng.normal(0, noise_px, n_mp)).astype(np.float32)
    md = desc[src].copy()
    flip = rng.integers(0, 256, (n_mp, 32), dtype=np.uint8)
    for _ in range(bit_noise - 1):
        flip &= rng.integers(0, 256, (n_mp, 32), dtype=np.uint8)
    md ^= flip
    return src, px, py, md


@pytest.mark.parametrize("seed,stereo,th", [(10, False, 1.0), (11, True, 1.0), (12, False, 3.0), (13, True, 5.0)])
def test_search_by_projection_mappoints(amd, seed, stereo, th):
    rng = np.random.default_rng(seed)
    x, y, octv, ang, desc, ur = _random_frame(rng, 1500, stereo, spread=0.0)
    F, Fo = _both(amd, x, y, octv, ang, desc, ur)
    n_mp = 2500
    src, px, py, md = _map_points(rng, x, y, octv, desc, n_mp, 1.5)
    level = np.clip(octv[src] + rng.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    in_view = (rng.random(n_mp) < 0.85).astype(np.uint8)
    view_cos = rng.uniform(0.99, 1.0, n_mp).astype(np.float32)
    pxr = (px - rng.uniform(1, 40, n_mp)).astype(np.float32) if stereo else None
    if stereo:  # make most stereo projections consistent with the feature they came from
        ok = ur[src] > 0
        pxr[ok] = ur[src][ok] + rng.normal(0, 1.0, ok.sum()).astype(np.float32)
    blocked = (rng.random(1500) < 0.1).astype(np.uint8)
    obs = (rng.random(n_mp) < 0.9).astype(np.uint8)
    m = amd.ORBmatcher(0.8, True)
    for b, o in [(blocked, obs), (None, None)]:
        n_ref, ref = orc.search_by_projection_mappoints(Fo, SF, b, in_view, level, view_cos, px, py, pxr, md, o, th, 0.8)
        n_got, got = m.SearchByProjection(F, SF, in_view, level, view_cos, px, py, md, th=th, proj_xr=pxr, blocked=b,
                                          mp_obs_positive=o)
        assert n_got == n_ref
        assert got.tolist() == ref.tolist()
        assert n_ref > 300


@pytest.mark.parametrize("seed,stereo,mode,check_ori", [(20, False, 0, True), (21, True, 0, True), (22, True, 1, True),
                                                        (23, True, 2, False), (24, False, 1, False)])
def test_search_by_projection_last_frame(amd, seed, stereo, mode, check_ori):
    rng = np.random.default_rng(seed)
    x, y, octv, ang, desc, ur = _random_frame(rng, 1800, stereo, spread=0.0)
    Cur, Co = _both(amd, x, y, octv, ang, desc, ur)
    nl = 1700
    src, u, v, md = _map_points(rng, x, y, octv, desc, nl, 3.0)
    lo = np.clip(octv[src] + rng.integers(-1, 2, nl), 0, 7).astype(np.int32)
    la = ((ang[src] + rng.normal(0, 6, nl)) % 360).astype(np.float32)
    la[rng.random(nl) < 0.15] = rng.uniform(0, 360)  # outliers for the rotation histogram
    valid = (rng.random(nl) < 0.9).astype(np.uint8)
    mbf = 40.0
    invzc = rng.uniform(0.02, 0.5, nl).astype(np.float32)
    if stereo:
        ok = ur[src] > 0
        invzc[ok] = ((u[ok] - ur[src][ok]) / mbf + rng.normal(0, 0.02, ok.sum())).astype(np.float32)
    obs = (rng.random(nl) < 0.8).astype(np.uint8)
    th = 7.0 if stereo else 15.0
    m = amd.ORBmatcher(0.9, check_ori)
    for o in (obs, None):
        n_ref, ref = orc.search_by_projection_lastframe(Co, SF, mbf, valid, u, v, invzc, lo, la, md, o, mode, th, check_ori)
        n_got, got = m.SearchByProjectionLastFrame(Cur, SF, valid, u, v, lo, la, md, th, mode=mode, mbf=mbf, invzc=invzc,
                                                   obs_positive=o)
        assert n_got == n_ref
        assert got.tolist() == ref.tolist()
        assert n_ref > 200
    # CurrentFrame keypoints that hold a map point with observations at entry are skipped (src/ORBmatcher.cc:1572-1574)
    blocked = (rng.random(len(x)) < 0.3).astype(np.uint8)
    n_ref, ref = orc.search_by_projection_lastframe(Co, SF, mbf, valid, u, v, invzc, lo, la, md, obs, mode, th, check_ori, blocked)
    n_got, got = m.SearchByProjectionLastFrame(Cur, SF, valid, u, v, lo, la, md, th, mode=mode, mbf=mbf, invzc=invzc,
                                               obs_positive=obs, blocked=blocked)
    assert n_got == n_ref and got.tolist() == ref.tolist()
    assert not (got[blocked > 0] >= 0).any() and n_ref < n_got + 1 and (ref >= 0).sum() > 100


def test_projection_on_extracted_frames(amd):
    """End to end on real extractor output: frame t projected into frame t+1 with the identity motion
    model (TrackWithMotionModel with zero velocity, src/Tracking.cc:865-918 -> th = 15, mono)."""
    fr = synth.render_sequence(31, 2, 640, 480, step=2.0)
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    (k1, d1), (k2, d2) = e.extract_batch(np.stack(fr))
    Cur = amd.FrameView.from_keypoints(k2, d2, 640, 480)
    Co = orc.Frame(k2["x"], k2["y"], k2["octave"], d2, BOUNDS, angle=k2["angle"])
    valid = np.ones(len(k1), np.uint8)
    m = amd.ORBmatcher(0.9, True)
    n_ref, ref = orc.search_by_projection_lastframe(Co, SF, 0.0, valid, k1["x"], k1["y"], None, k1["octave"], k1["angle"],
                                                    d1, None, 0, 15.0, True)
    n_got, got = m.SearchByProjectionLastFrame(Cur, SF, valid, k1["x"], k1["y"], k1["octave"], k1["angle"], d1, 15.0)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    assert n_ref > 100
    # and the local-map search on the same pair
    level = k1["octave"].astype(np.int32)
    vc = np.full(len(k1), 0.9995, np.float32)
    n_ref, ref = orc.search_by_projection_mappoints(Co, SF, None, valid, level, vc, k1["x"], k1["y"], None, d1, None, 3.0, 0.8)
    n_got, got = amd.ORBmatcher(0.8).SearchByProjection(Cur, SF, valid, level, vc, k1["x"], k1["y"], d1, th=3.0)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    assert n_ref > 100


def test_projection_invalid_arguments(amd):
    rng = np.random.default_rng(40)
    x, y, octv, ang, desc, _ = _random_frame(rng, 50)
    F = amd.FrameView(x, y, octv, desc, BOUNDS, angle=ang)
    m = amd.ORBmatcher(0.8)
    one = np.ones(1, np.uint8)
    with pytest.raises(amd.OrbfeError):  # predicted level outside the pyramid
        m.SearchByProjection(F, SF, one, [9], [1.0], [10.0], [10.0], desc[:1])
    with pytest.raises(amd.OrbfeError):
        m.SearchByProjectionLastFrame(F, SF, one, [1.0], [1.0], [8], [0.0], desc[:1], 15.0)
    with pytest.raises(amd.OrbfeError):
        m.SearchByProjectionLastFrame(F, SF, one, [1.0], [1.0], [1], [0.0], desc[:1], 15.0, mode=3)
    # nothing in view -> no matches, all -1
    n, match = m.SearchByProjection(F, SF, np.zeros(3, np.uint8), [0, 0, 0], [1.0] * 3, [1.0] * 3, [1.0] * 3, desc[:3])
    assert n == 0 and (match == -1).all()


def _projected(rng, x, y, octv, ang, desc, n, noise_px=2.0):
    src, u, v, md = _map_points(rng, x, y, octv, desc, n, noise_px)
    level = np.clip(octv[src] + rng.integers(0, 2, n), 0, 7).astype(np.int32)
    valid = (rng.random(n) < 0.85).astype(np.uint8)
    a = ((ang[src] + rng.normal(0, 6, n)) % 360).astype(np.float32)
    a[rng.random(n) < 0.15] = rng.uniform(0, 360)
    return src, u, v, md, level, valid, a


@pytest.mark.parametrize("seed,th,orb_dist,check_ori", [(50, 10.0, 100, True), (51, 3.0, 64, True), (52, 10.0, 100, False)])
def test_search_by_projection_keyframe(amd, seed, th, orb_dist, check_ori):
    rng = np.random.default_rng(seed)
    x, y, octv, ang, desc, _ = _random_frame(rng, 1600, spread=0.0)
    Cur, Co = _both(amd, x, y, octv, ang, desc, None)
    src, u, v, md, level, valid, ka = _projected(rng, x, y, octv, ang, desc, 1400)
    blocked = (rng.random(1600) < 0.3).astype(np.uint8)
    m = amd.ORBmatcher(0.9, check_ori)
    for b in (blocked, None):
        n_ref, ref = orc.search_by_projection_reloc(Co, SF, valid, u, v, level, ka, md, b, th, orb_dist, check_ori)
        n_got, got = m.SearchByProjectionKeyFrame(Cur, SF, valid, u, v, level, ka, md, th, orb_dist, blocked=b)
        assert (n_got, got.tolist()) == (n_ref, ref.tolist())
        assert n_ref > 150


@pytest.mark.parametrize("seed,th", [(60, 10.0), (61, 3.0)])
def test_search_by_projection_sim3(amd, seed, th):
    rng = np.random.default_rng(seed)
    x, y, octv, ang, desc, _ = _random_frame(rng, 1600, spread=0.0)
    KF, Ko = _both(amd, x, y, octv, ang, desc, None)
    src, u, v, md, level, valid, _ = _projected(rng, x, y, octv, ang, desc, 2000)
    matched = (rng.random(1600) < 0.3).astype(np.uint8)
    m = amd.ORBmatcher(0.75, True)
    for mt in (matched, None):
        n_ref, ref = orc.search_by_projection_sim3(Ko, SF, valid, u, v, level, md, mt, th)
        n_got, got = m.SearchByProjectionSim3(KF, SF, valid, u, v, level, md, th, matched=mt)
        assert (n_got, got.tolist()) == (n_ref, ref.tolist())
        assert n_ref > 150


@pytest.mark.parametrize("seed,window,check_ori", [(70, 100, True), (71, 30, True), (72, 100, False)])
def test_search_for_initialization(amd, seed, window, check_ori):
    """mono initialisation extracts 2*nFeatures (src/Tracking.cc:124-127); window 100 is the reference's
    call (src/Tracking.cc:632)."""
    fr = synth.render_sequence(seed, 2, 640, 480, step=4.0)
    e = amd.ORBextractor(2000, 1.2, 8, 20, 7)
    (k1, d1), (k2, d2) = e.extract_batch(np.stack(fr))
    F1 = amd.FrameView.from_keypoints(k1, d1, 640, 480)
    F2 = amd.FrameView.from_keypoints(k2, d2, 640, 480)
    O1 = orc.Frame(k1["x"], k1["y"], k1["octave"], d1, BOUNDS, angle=k1["angle"])
    O2 = orc.Frame(k2["x"], k2["y"], k2["octave"], d2, BOUNDS, angle=k2["angle"])
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
    n_ref, ref, prev_ref = orc.search_for_initialization(O1, O2, prev.copy(), window, 0.9, check_ori)
    prev_got = prev.copy()
    n_got, got = amd.ORBmatcher(0.9, check_ori).SearchForInitialization(F1, F2, prev_got, window)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    assert np.array_equal(prev_got, prev_ref)
    assert n_ref > 100
    # second call with the updated vbPrevMatched, as Tracking::MonocularInitialization does frame after frame
    n_ref2, ref2, _ = orc.search_for_initialization(O1, O2, prev_ref.copy(), window, 0.9, check_ori)
    n_got2, got2 = amd.ORBmatcher(0.9, check_ori).SearchForInitialization(F1, F2, prev_got, window)
    assert (n_got2, got2.tolist()) == (n_ref2, ref2.tolist())


@pytest.mark.parametrize("seed,stereo,chi2", [(80, False, True), (81, True, True), (82, True, False)])
def test_fuse_search(amd, seed, stereo, chi2):
    rng = np.random.default_rng(seed)
    x, y, octv, ang, desc, ur = _random_frame(rng, 1500, stereo, spread=0.0)
    if stereo:
        ur = np.where(ur > 0, ur, -1.0).astype(np.float32)
    KF, Ko = _both(amd, x, y, octv, ang, desc, ur)
    src, u, v, md, level, valid, _ = _projected(rng, x, y, octv, ang, desc, 2500, noise_px=1.2)
    pur = (u - rng.uniform(1, 40, len(u))).astype(np.float32)
    if stereo:
        ok = ur[src] >= 0
        pur[ok] = (ur[src][ok] + rng.normal(0, 1.0, ok.sum())).astype(np.float32)
    inv_sigma2 = (1.0 / (SF * SF)).astype(np.float32)
    ref = orc.fuse_search(Ko, SF, inv_sigma2, valid, u, v, pur, level, md, 3.0, chi2)
    got = amd.ORBmatcher(0.6).FuseSearch(KF, SF, valid, u, v, level, md, th=3.0,
                                         inv_level_sigma2=inv_sigma2 if chi2 else None, ur=pur)
    assert got.tolist() == ref.tolist()
    assert (ref >= 0).sum() > 300
    if chi2:  # the gate really rejects something the ungated search accepts
        ungated = orc.fuse_search(Ko, SF, inv_sigma2, valid, u, v, pur, level, md, 3.0, False)
        assert (ungated != ref).any()


def test_search_by_sim3(amd):
    rng = np.random.default_rng(90)
    x1, y1, o1, a1, d1, _ = _random_frame(rng, 1400, spread=0.0)
    # KF2 sees the same structure shifted, with descriptor noise
    n2 = 1500
    src = rng.integers(0, 1400, n2)
    x2 = np.clip(x1[src] + 6 + rng.normal(0, 1.0, n2), 0, 639).astype(np.float32)
    y2 = np.clip(y1[src] - 4 + rng.normal(0, 1.0, n2), 0, 479).astype(np.float32)
    o2 = o1[src].copy()
    a2 = a1[src].copy()
    d2 = d1[src] ^ (rng.integers(0, 256, (n2, 32), dtype=np.uint8) & rng.integers(0, 256, (n2, 32), dtype=np.uint8) &
                    rng.integers(0, 256, (n2, 32), dtype=np.uint8))
    K1, O1 = _both(amd, x1, y1, o1, a1, d1, None)
    K2, O2 = _both(amd, x2, y2, o2, a2, d2, None)
    # projections: KF1 points land in KF2 at +(6,-4); KF2 points land in KF1 at -(6,-4)
    u1 = (x1 + 6 + rng.normal(0, 1.5, 1400)).astype(np.float32)
    v1 = (y1 - 4 + rng.normal(0, 1.5, 1400)).astype(np.float32)
    u2 = (x2 - 6 + rng.normal(0, 1.5, n2)).astype(np.float32)
    v2 = (y2 + 4 + rng.normal(0, 1.5, n2)).astype(np.float32)
    l1 = np.clip(o1 + rng.integers(0, 2, 1400), 0, 7).astype(np.int32)
    l2 = np.clip(o2 + rng.integers(0, 2, n2), 0, 7).astype(np.int32)
    va1 = (rng.random(1400) < 0.8).astype(np.uint8)
    va2 = (rng.random(n2) < 0.8).astype(np.uint8)
    n_ref, ref = orc.search_by_sim3(O1, O2, SF, SF, va1, u1, v1, l1, d1, va2, u2, v2, l2, d2, 7.5)
    n_got, got = amd.ORBmatcher(0.75).SearchBySim3(K1, K2, SF, SF, va1, u1, v1, l1, d1, va2, u2, v2, l2, d2, 7.5)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    assert n_ref > 100


def test_fuse_search_multi(amd):
    """The per-point search of Fuse for the same map points against K key frames in one call (LocalMapping::SearchInNeighbors,
    src/LocalMapping.cc:542-549) == K single orbfe_fuse_search calls of the oracle; a mix of resident and host-array key frames."""
    rng = np.random.default_rng(85)
    K, n = 5, 1800
    inv_sigma2 = (1.0 / (SF * SF)).astype(np.float32)
    md = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    KFs, refs, U, V, UR, LV, VA = [], [], [], [], [], [], []
    for k in range(K):
        x, y, octv, ang, desc, ur = _random_frame(rng, 1200 + 100 * k, True, spread=0.0)
        ur = np.where(ur > 0, ur, -1.0).astype(np.float32)
        src = rng.integers(0, len(x), n)
        desc[src[: n // 2]] = md[: n // 2] ^ (rng.integers(0, 256, (n // 2, 32), dtype=np.uint8) & rng.integers(0, 256, (n // 2, 32), dtype=np.uint8) &
                                              rng.integers(0, 256, (n // 2, 32), dtype=np.uint8))
        KF, Ko = _both(amd, x, y, octv, ang, desc, ur)
        if k == 2 and RESIDENT:
            KF = amd.FrameView(x, y, octv, desc, BOUNDS, angle=ang, u_right=ur)  # one host-array frame in the group
        u = (x[src] + rng.normal(0, 1.2, n)).astype(np.float32)
        v = (y[src] + rng.normal(0, 1.2, n)).astype(np.float32)
        pur = np.where(ur[src] >= 0, ur[src] + rng.normal(0, 1.0, n), u - 10).astype(np.float32)
        level = np.clip(octv[src] + rng.integers(0, 2, n), 0, 7).astype(np.int32)
        valid = (rng.random(n) < 0.85).astype(np.uint8)
        refs.append(orc.fuse_search(Ko, SF, inv_sigma2, valid, u, v, pur, level, md, 3.0, True))
        KFs.append(KF); U.append(u); V.append(v); UR.append(pur); LV.append(level); VA.append(valid)
    got = amd.ORBmatcher(0.6).FuseSearchMulti(KFs, SF, np.stack(VA), np.stack(U), np.stack(V), np.stack(LV), md, th=3.0,
                                              inv_level_sigma2=inv_sigma2, ur=np.stack(UR))
    for k in range(K):
        assert got[k].tolist() == refs[k].tolist(), k
    assert sum(int((r >= 0).sum()) for r in refs) > 500


def test_search_by_projection_keyframe_multi(amd):
    """ONE current frame against the projected map points of K candidate key frames in one call (Tracking::Relocalization,
    src/Tracking.cc:1577,1595: th = 10 / ORBdist = 100, then th = 3 / ORBdist = 64) == K single calls of the oracle;
    candidates of different sizes, one empty, one without `blocked`."""
    rng = np.random.default_rng(90)
    x, y, octv, ang, desc, _ = _random_frame(rng, 1600, spread=0.0)
    Cur, Co = _both(amd, x, y, octv, ang, desc, None)
    cands, refs = [], []
    for k, (nk, th, od) in enumerate([(1400, 10.0, 100), (900, 3.0, 64), (0, 10.0, 100), (1200, 10.0, 100), (300, 3.0, 64)]):
        if nk:
            src, u, v, md, level, valid, ka = _projected(rng, x, y, octv, ang, desc, nk)
        else:
            u = v = ka = np.zeros(0, np.float32); level = np.zeros(0, np.int32); valid = np.zeros(0, np.uint8); md = np.zeros((0, 32), np.uint8)
        blocked = None if k == 3 else (rng.random(1600) < 0.3).astype(np.uint8)
        cands.append(dict(valid=valid, u=u, v=v, level=level, kf_angle=ka, mp_desc=md, th=th, ORBdist=od, blocked=blocked))
        if nk:
            refs.append(orc.search_by_projection_reloc(Co, SF, valid, u, v, level, ka, md, blocked, th, od, True))
        else:
            refs.append((0, np.full(1600, -1, np.int32)))
    cnt, got = amd.ORBmatcher(0.9, True).SearchByProjectionKeyFrameMulti(Cur, SF, cands)
    for k, (n_ref, ref) in enumerate(refs):
        assert (int(cnt[k]), got[k].tolist()) == (n_ref, ref.tolist()), k
    assert sum(r[0] for r in refs) > 400


def _cluster(rng, n, base, octave, bits):
    """n features inside a 6-px disc around (320, 240), descriptors a few bits away from `base`"""
    a = rng.uniform(0, 2 * np.pi, n)
    rr = rng.uniform(0, 6, n)
    x = (320 + rr * np.cos(a)).astype(np.float32)
    y = (240 + rr * np.sin(a)).astype(np.float32)
    d = np.repeat(base[None, :], n, axis=0).copy()
    for i in range(n):
        for b in rng.integers(0, 256, bits):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return x, y, np.full(n, octave, np.int32), rng.uniform(0, 360, n).astype(np.float32), d


def test_claim_loops_with_long_dependency_chains(amd):
    """The claim loops run on the device as a fixed-point iteration (k_window_claim): worst case for it are queries that
    ALL want the same few features -- every one of them depends on every earlier one (src/ORBmatcher.cc:77-78, 1572-1574,
    1726-1727, 431-432, 516-533).  60 features in one 6-px cluster, 400 queries on top of it."""
    rng = np.random.default_rng(90)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    x, y, octv, ang, desc = _cluster(rng, 60, base, 2, 20)
    F, Fo = _both(amd, x, y, octv, ang, desc, None)
    nq = 400
    u, v, lv, qa, md = _cluster(rng, nq, base, 2, 24)
    valid = np.ones(nq, np.uint8)
    # key frame (reloc) form, with and without the rotation histogram
    for check_ori in (True, False):
        m = amd.ORBmatcher(0.9, check_ori)
        n_ref, ref = orc.search_by_projection_reloc(Fo, SF, valid, u, v, lv, qa, md, None, 10.0, 100, check_ori)
        n_got, got = m.SearchByProjectionKeyFrame(F, SF, valid, u, v, lv, qa, md, 10.0, 100)
        assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    assert (ref >= 0).sum() > 10
    from orb_slam2_annotate_amd import _lib
    assert _lib.load().orbfe_debug_last_claim_rounds() > 10  # (the chains are really there: random frames take 2-4 rounds)
    # Sim3 form (TH_LOW)
    n_ref, ref = orc.search_by_projection_sim3(Fo, SF, valid, u, v, lv, md, None, 10.0)
    n_got, got = amd.ORBmatcher(0.75, True).SearchByProjectionSim3(F, SF, valid, u, v, lv, md, 10.0)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    # last-frame form: a third of the matches do not hide their feature (MapPoint::Observations() == 0), so later points
    # overwrite them and a feature is pushed into the rotation histogram more than once
    obs = (rng.random(nq) < 0.66).astype(np.uint8)
    invzc = np.zeros(nq, np.float32)
    for check_ori in (True, False):
        n_ref, ref = orc.search_by_projection_lastframe(Fo, SF, 40.0, valid, u, v, invzc, lv, qa, md, obs, 0, 15.0, check_ori)
        n_got, got = amd.ORBmatcher(0.9, check_ori).SearchByProjectionLastFrame(F, SF, valid, u, v, lv, qa, md, 15.0, mode=0, mbf=40.0,
                                                                               invzc=invzc, obs_positive=obs)
        assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    # map-point form (best / second-best level + ratio test)
    view_cos = np.full(nq, 0.9, np.float32)
    for o in (obs, None):
        n_ref, ref = orc.search_by_projection_mappoints(Fo, SF, None, valid, lv, view_cos, u, v, None, md, o, 3.0, 0.8)
        n_got, got = amd.ORBmatcher(0.8, True).SearchByProjection(F, SF, valid, lv, view_cos, u, v, md, th=3.0, mp_obs_positive=o)
        assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    # SearchForInitialization: 300 level-0 key points of F1 over the same 60 key points of F2 -- a later, closer key point
    # takes a feature back from an earlier one (:529-533)
    x2, y2, o2, a2, d2 = _cluster(rng, 60, base, 0, 12)
    x1, y1, o1, a1, d1 = _cluster(rng, 300, base, 0, 16)
    o1[rng.random(300) < 0.1] = 1  # (not level 0: skipped, :482-484)
    F1, O1 = _both(amd, x1, y1, o1, a1, d1, None)
    F2, O2 = _both(amd, x2, y2, o2, a2, d2, None)
    prev = np.stack([x1, y1], axis=1).astype(np.float32)
    for check_ori in (True, False):
        n_ref, ref, prev_ref = orc.search_for_initialization(O1, O2, prev.copy(), 100, 0.9, check_ori)
        prev_got = prev.copy()
        n_got, got = amd.ORBmatcher(0.9, check_ori).SearchForInitialization(F1, F2, prev_got, 100)
        assert (n_got, got.tolist()) == (n_ref, ref.tolist())
        assert np.array_equal(prev_got, prev_ref)
    assert n_ref > 5


def test_claim_loops_on_a_frame_too_large_for_lds(amd):
    """more than 12288 key points: k_window_claim keeps owner[] in HBM instead of LDS (and a thread owns several points)"""
    rng = np.random.default_rng(91)
    x, y, octv, ang, desc, _ = _random_frame(rng, 14000, spread=0.0)
    F, Fo = _both(amd, x, y, octv, ang, desc, None)
    n_mp = 5000
    src, px, py, md = _map_points(rng, x, y, octv, desc, n_mp, 1.5)
    level = np.clip(octv[src] + rng.integers(0, 2, n_mp), 0, 7).astype(np.int32)
    in_view = (rng.random(n_mp) < 0.9).astype(np.uint8)
    view_cos = rng.uniform(0.99, 1.0, n_mp).astype(np.float32)
    obs = (rng.random(n_mp) < 0.8).astype(np.uint8)
    blocked = (rng.random(14000) < 0.1).astype(np.uint8)
    n_ref, ref = orc.search_by_projection_mappoints(Fo, SF, blocked, in_view, level, view_cos, px, py, None, md, obs, 3.0, 0.8)
    n_got, got = amd.ORBmatcher(0.8, True).SearchByProjection(F, SF, in_view, level, view_cos, px, py, md, th=3.0, blocked=blocked,
                                                              mp_obs_positive=obs)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist()) and n_ref > 1000
    ka = ((ang[src] + rng.normal(0, 6, n_mp)) % 360).astype(np.float32)
    n_ref, ref = orc.search_by_projection_reloc(Fo, SF, in_view, px, py, level, ka, md, blocked, 10.0, 100, True)
    n_got, got = amd.ORBmatcher(0.9, True).SearchByProjectionKeyFrame(F, SF, in_view, px, py, level, ka, md, 10.0, 100, blocked=blocked)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist()) and n_ref > 1000


def test_claim_chain_longer_than_the_round_stamp(amd):
    """A dependency chain of 2100 points: point k sees key points k and k+1 and prefers k; a point in front of all takes key
    point 0, so point 0 moves to 1, which pushes point 1 to 2, ... one link per round of k_window_claim -- more rounds than
    the 11-bit round field of its owner stamps counts, so the re-basing path runs (BEST and RATIO form)."""
    n = 2101
    bounds = (-8.0, 8.0 * n + 8.0, 0.0, 480.0)
    base = np.random.default_rng(92).integers(0, 256, 32, dtype=np.uint8)

    def onehot(k):
        d = np.zeros(32, np.uint8)
        d[(k % 256) >> 3] = 1 << (k & 7)
        return d

    D = np.stack([base ^ onehot(k) for k in range(n)])
    fdesc = np.stack([D[k] ^ onehot(k + 128) for k in range(n)])          # key point k: one bit from D[k]
    fx = (8.0 * np.arange(n)).astype(np.float32)
    fy = np.full(n, 240.0, np.float32)
    octv = np.zeros(n, np.int32)
    ang = np.zeros(n, np.float32)
    F, Fo = _both(amd, fx, fy, octv, ang, fdesc, None, bounds)
    # point 0 sits left of key point 0 and sees only it; point k+1 sits between key points k and k+1
    u = np.concatenate([[-3.0], 8.0 * np.arange(n - 1) + 4.0]).astype(np.float32)
    v = np.full(n, 240.0, np.float32)
    md = np.concatenate([D[:1], D[: n - 1]])
    lv = np.zeros(n, np.int32)
    valid = np.ones(n, np.uint8)
    qa = np.zeros(n, np.float32)
    from orb_slam2_annotate_amd import _lib
    n_ref, ref = orc.search_by_projection_reloc(Fo, SF, valid, u, v, lv, qa, md, None, 5.0, 100, False)
    n_got, got = amd.ORBmatcher(0.9, False).SearchByProjectionKeyFrame(F, SF, valid, u, v, lv, qa, md, 5.0, 100)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist())
    # key point k went to point k -- the shifted chain (the key points of the last half grid cell are not in the grid:
    # PosInGrid rounds, src/Frame.cc:417-427)
    assert n_ref > 2060 and ref[:2060].tolist() == list(range(2060))
    assert _lib.load().orbfe_debug_last_claim_rounds() > 2046
    view_cos = np.full(n, 0.9, np.float32)                                  # radius 4 * th
    n_ref, ref = orc.search_by_projection_mappoints(Fo, SF, None, valid, lv, view_cos, u, v, None, md, None, 1.25, 0.8)
    n_got, got = amd.ORBmatcher(0.8, True).SearchByProjection(F, SF, valid, lv, view_cos, u, v, md, th=1.25)
    assert (n_got, got.tolist()) == (n_ref, ref.tolist()) and n_ref > 2060
    assert _lib.load().orbfe_debug_last_claim_rounds() > 2046


@pytest.mark.parametrize("n_crowd", [40, 65, 700])
def test_grid_with_a_crowded_cell(amd, n_crowd):
    """The frame grid is built by counting (k_grid_build_count) up to 64 key points per cell and by the bitonic network
    beyond: n_crowd key points in ONE 10 x 10 px cell plus a spread-out rest -- the windows come back in the reference's
    scan order either way (mGrid's push_back order inside a cell, src/Frame.cc:246-259)."""
    rng = np.random.default_rng(93 + n_crowd)
    n_rest = 900
    x = np.concatenate([rng.uniform(301.0, 309.0, n_crowd), rng.uniform(0, 640, n_rest)]).astype(np.float32)
    y = np.concatenate([rng.uniform(201.0, 209.0, n_crowd), rng.uniform(0, 480, n_rest)]).astype(np.float32)
    perm = rng.permutation(len(x))  # the crowd is not contiguous in index order
    x, y = x[perm], y[perm]
    n = len(x)
    octv = rng.integers(0, 8, n).astype(np.int32)
    ang = rng.uniform(0, 360, n).astype(np.float32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    F, Fo = _both(amd, x, y, octv, ang, desc, None)
    qx = np.array([305.0, 305.0, 100.0, 320.0, 304.0], np.float32)
    qy = np.array([205.0, 205.0, 100.0, 240.0, 206.0], np.float32)
    r = np.array([3.0, 60.0, 30.0, 500.0, 12.0], np.float32)
    lo = np.array([-1, 0, -1, 2, 1], np.int32)
    hi = np.array([-1, 3, -1, -1, 5], np.int32)
    got = F.GetFeaturesInArea(qx, qy, r, lo, hi, capacity=16)
    for q in range(len(qx)):
        assert got[q].tolist() == Fo.features_in_area(qx[q], qy[q], r[q], lo[q], hi[q]).tolist(), q
    assert len(got[1]) > n_crowd // 3
