"""GPU parity tests of the Hamming matchers and the stereo matcher against the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as orc
from orb_slam2_annotate_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import orb_slam2_annotate_amd as m
    return m


def _nodes(desc, seed, n_nodes=100):
    """Synthetic 2-level k=10 vocabulary stand-in (the real ORBvoc.txt is absent):
    node = index of the nearest of 100 seeded 256-bit centroids."""
    rng = np.random.default_rng(seed)
    cent = rng.integers(0, 256, size=(n_nodes, 32), dtype=np.uint8)
    x = np.unpackbits(desc, axis=1).astype(np.int16)
    c = np.unpackbits(cent, axis=1).astype(np.int16)
    d = (x[:, None, :] != c[None, :, :]).sum(axis=2)
    return d.argmin(axis=1).astype(np.uint32) * 7 + 3  # sparse, non-contiguous node ids


def _two_frames(amd, seed, w=640, h=480, nf=1000):
    fr = synth.render_sequence(seed, 2, w, h, step=3.0)
    e = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    (k1, d1), (k2, d2) = e.extract_batch(np.stack(fr))
    return k1, d1, k2, d2


def test_descriptor_distance(amd):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, size=(1000, 32), dtype=np.uint8)
    b = rng.integers(0, 256, size=(1000, 32), dtype=np.uint8)
    got = amd.ORBmatcher.DescriptorDistance(a, b)
    ref = np.unpackbits(a ^ b, axis=1).sum(axis=1)
    assert np.array_equal(got, ref)
    assert amd.ORBmatcher.DescriptorDistance(a[0], a[0]) == 0
    assert amd.ORBmatcher.DescriptorDistance(a[0], ~a[0]) == 256
    m = amd.ORBmatcher.HammingMatrix(a[:37], b[:129])
    assert np.array_equal(m, np.unpackbits(a[:37, None, :] ^ b[None, :129, :], axis=2).sum(axis=2))


@pytest.mark.parametrize("seed,nnratio,ori", [(1, 0.7, True), (2, 0.75, True), (3, 0.6, False)])
def test_search_by_bow_frame(amd, seed, nnratio, ori):
    k1, d1, k2, d2 = _two_frames(amd, seed)
    rng = np.random.default_rng(seed)
    has1 = (rng.random(len(k1)) < 0.7).astype(np.uint8)
    n1, n2 = _nodes(d1, 11), _nodes(d2, 11)
    ref_n, ref = orc.search_by_bow(d1, has1, k1["angle"], orc.FeatVec(n1), d2, k2["angle"], orc.FeatVec(n2), nnratio, ori)
    m = amd.ORBmatcher(nnratio, ori)
    fv1, fv2 = amd.FeatureVector.from_node_of_feature(n1), amd.FeatureVector.from_node_of_feature(n2)
    got_n, got = m.SearchByBoW(d1, has1, k1["angle"], fv1, d2, k2["angle"], fv2)
    assert ref_n == got_n and np.array_equal(ref, got)
    assert ref_n > 20


@pytest.mark.parametrize("n_nodes", [3, 7, 12, 20])
def test_search_by_bow_large_nodes(amd, n_nodes):
    """Few vocabulary nodes -> hundreds of features per node.  Round 4: nodes of up to 256 candidates stay in registers
    (1, 2 or 4 candidates per lane, the queries pass through the wave 64 at a time); beyond that the LDS-claim path runs.
    3 nodes: > 256 per node (LDS path); 7: 128..256 (4 slots); 12 and 20 mix 1 / 2 / 4 slots in one call.  Both SearchByBoW
    forms, and the resident / multi entry points on the same nodes."""
    k1, d1, k2, d2 = _two_frames(amd, 6)
    n1, n2 = _nodes(d1, 13, n_nodes), _nodes(d2, 13, n_nodes)
    assert np.bincount(n1).max() > 64
    rng = np.random.default_rng(n_nodes)
    has1 = np.ones(len(k1), np.uint8)
    ref_n, ref = orc.search_by_bow(d1, has1, k1["angle"], orc.FeatVec(n1), d2, k2["angle"], orc.FeatVec(n2), 0.7, True)
    m = amd.ORBmatcher(0.7, True)
    fv1, fv2 = amd.FeatureVector.from_node_of_feature(n1), amd.FeatureVector.from_node_of_feature(n2)
    got_n, got = m.SearchByBoW(d1, has1, k1["angle"], fv1, d2, k2["angle"], fv2)
    assert ref_n == got_n and np.array_equal(ref, got)
    assert ref_n > 20
    h1, h2 = (rng.random(len(k1)) < 0.6).astype(np.uint8), (rng.random(len(k2)) < 0.6).astype(np.uint8)
    ref_n, ref = orc.search_by_bow_kf(d1, h1, k1["angle"], orc.FeatVec(n1), d2, h2, k2["angle"], orc.FeatVec(n2), 0.7, True)
    got_n, got = m.SearchByBoW(d1, h1, k1["angle"], fv1, d2, k2["angle"], fv2, has_mp2=h2)
    assert ref_n == got_n and np.array_equal(ref, got)
    R1, R2 = _resident(amd, k1, d1, n1), _resident(amd, k2, d2, n2)
    cnt, multi = m.SearchByBoWKFMulti(R1, h1, [R2, R2], [h2, h2])
    assert cnt.tolist() == [ref_n, ref_n] and np.array_equal(multi[0], ref) and np.array_equal(multi[1], ref)
    R1.close()
    R2.close()


@pytest.mark.parametrize("seed", [4, 5])
def test_search_by_bow_kf(amd, seed):
    k1, d1, k2, d2 = _two_frames(amd, seed)
    rng = np.random.default_rng(seed)
    has1 = (rng.random(len(k1)) < 0.6).astype(np.uint8)
    has2 = (rng.random(len(k2)) < 0.6).astype(np.uint8)
    n1, n2 = _nodes(d1, 12, 40), _nodes(d2, 12, 40)
    ref_n, ref = orc.search_by_bow_kf(d1, has1, k1["angle"], orc.FeatVec(n1), d2, has2, k2["angle"], orc.FeatVec(n2), 0.75, True)
    m = amd.ORBmatcher(0.75, True)
    got_n, got = m.SearchByBoW(d1, has1, k1["angle"], amd.FeatureVector.from_node_of_feature(n1), d2, k2["angle"],
                               amd.FeatureVector.from_node_of_feature(n2), has_mp2=has2)
    assert ref_n == got_n and np.array_equal(ref, got)
    assert ref_n > 5


@pytest.mark.parametrize("seed,only_stereo", [(6, False), (7, True), (8, False)])
def test_search_for_triangulation(amd, seed, only_stereo):
    k1, d1, k2, d2 = _two_frames(amd, seed)
    rng = np.random.default_rng(seed)
    has1 = (rng.random(len(k1)) < 0.4).astype(np.uint8)
    has2 = (rng.random(len(k2)) < 0.4).astype(np.uint8)
    st1 = (rng.random(len(k1)) < 0.5).astype(np.uint8)
    st2 = (rng.random(len(k2)) < 0.5).astype(np.uint8)
    n1, n2 = _nodes(d1, 13, 60), _nodes(d2, 13, 60)
    # pure x-translation between the views (3 px per frame): F = [t]_x with t=(1,0,0)
    F12 = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float32) * 0.01
    ex, ey = 5000.0, 240.0
    o = orc.Oracle()
    sf, sg = o.scale_factors(), o.level_sigma2()
    ref_n, ref = orc.search_for_triangulation(d1, has1, k1["x"], k1["y"], k1["angle"], st1, orc.FeatVec(n1), d2, has2,
                                              k2["x"], k2["y"], k2["angle"], k2["octave"], st2, orc.FeatVec(n2), F12,
                                              ex, ey, sf, sg, only_stereo, seed != 8)
    m = amd.ORBmatcher(0.6, seed != 8)
    got_n, pairs = m.SearchForTriangulation(d1, has1, k1["x"], k1["y"], k1["angle"], st1,
                                            amd.FeatureVector.from_node_of_feature(n1), d2, has2, k2["x"], k2["y"],
                                            k2["angle"], k2["octave"], st2, amd.FeatureVector.from_node_of_feature(n2),
                                            F12, ex, ey, sf, sg, only_stereo)
    idx = np.nonzero(ref >= 0)[0]
    assert ref_n == got_n
    assert np.array_equal(pairs, np.stack([idx, ref[idx]], axis=1))
    assert ref_n > 5


def test_bow_edge_cases(amd):
    m = amd.ORBmatcher(0.7, True)
    d = np.zeros((0, 32), np.uint8)
    fv = amd.FeatureVector.from_node_of_feature(np.zeros(0, np.uint32))
    n, out = m.SearchByBoW(d, np.zeros(0, np.uint8), np.zeros(0, np.float32), fv, d, np.zeros(0, np.float32), fv)
    assert n == 0 and len(out) == 0
    # identical descriptors everywhere: ties, ratio test rejects everything (best == second)
    d = np.zeros((50, 32), np.uint8)
    nodes = np.zeros(50, np.uint32)
    fvo = orc.FeatVec(nodes)
    ang = np.linspace(0, 359, 50).astype(np.float32)
    rn, r = orc.search_by_bow(d, np.ones(50, np.uint8), ang, fvo, d, ang, fvo, 0.7, True)
    gn, g = m.SearchByBoW(d, np.ones(50, np.uint8), ang, amd.FeatureVector.from_node_of_feature(nodes), d, ang,
                          amd.FeatureVector.from_node_of_feature(nodes))
    assert rn == gn and np.array_equal(r, g)
    # disjoint node sets: nothing to match
    gn, g = m.SearchByBoW(d, np.ones(50, np.uint8), ang, amd.FeatureVector.from_node_of_feature(nodes), d, ang,
                          amd.FeatureVector.from_node_of_feature(nodes + 1))
    assert gn == 0 and (g == -1).all()


@pytest.mark.parametrize("scene", ["shapes", "textured"])
@pytest.mark.parametrize("seed,shape,nf,bf,fx", [(1, (1241, 376), 2000, 386.1448, 718.856),
                                                   (2, (752, 480), 1200, 47.90639384423901, 435.2046959714599)])
def test_compute_stereo_matches(amd, seed, shape, nf, bf, fx, scene):
    w, h = shape
    left, right = synth.STEREO_SCENES[scene](seed, w, h)
    eL = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    eR = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    kL, dL = eL(left)
    kR, dR = eR(right)
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    krL, drL, pL = o.extract(left, want_pyramid=True)
    krR, drR, pR = o.extract(right, want_pyramid=True)
    assert np.array_equal(krL, kL) and np.array_equal(krR, kR)
    mbf = np.float32(bf)
    mb = np.float32(mbf / np.float32(fx))
    u_ref, d_ref = o.stereo(w, h, krL, drL, krR, drR, pL, pR, float(mbf), float(mb))
    u, d = amd.ComputeStereoMatches(eL, eR, kL, dL, kR, dR, float(mbf), float(mb))
    assert np.array_equal(u_ref, u)
    assert np.array_equal(d_ref, d)
    assert (u >= 0).sum() > (0.5 * len(kL) if scene == "textured" else 100)


def test_compute_stereo_matches_border_keypoints(amd):
    """Keypoints no extractor would emit (a few pixels from the right / lower edges, every octave): the SAD rows fall
    back from the 16-byte row requests to byte loads of exactly the pixels the reference reads; still bit-exact."""
    w, h, nf = 752, 480, 1200
    left, right = synth.render_stereo(7, w, h)
    eL = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    eR = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    kL0, _ = eL(left)
    eR(right)
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    _, _, pL = o.extract(left, want_pyramid=True)
    _, _, pR = o.extract(right, want_pyramid=True)
    rng = np.random.default_rng(11)
    sc = np.array(eL.GetScaleFactors(), dtype=np.float32)
    kl, kr, dl, dr = [], [], [], []
    for octave in range(8):
        lw, lh = o.level_sizes(w, h)[octave]
        for k in range(24):
            # level coordinates: left x in the last 6..14 columns, right x 0..6 columns further left, rows anywhere legal
            xl = lw - 6 - int(rng.integers(0, 9))
            xr = xl - int(rng.integers(0, 7))
            y = int(rng.integers(6, lh - 6)) if k % 3 else lh - 6 - int(rng.integers(0, 3))
            d = rng.integers(0, 256, 32, dtype=np.uint8)
            for lst, x in ((kl, xl), (kr, xr)):
                kp = np.zeros((), dtype=kL0.dtype)
                names = kp.dtype.names
                kp[names[0]] = np.float32(x) * sc[octave]
                kp[names[1]] = np.float32(y) * sc[octave]
                kp[names[2]] = 31.0 * sc[octave]
                kp[names[3]] = 0.0
                kp[names[4]] = 50.0
                kp[names[5]] = octave
                kp[names[6]] = -1
                lst.append(kp)
            dl.append(d)
            d2 = d.copy()
            d2[int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))  # distance 1
            dr.append(d2)
    kl, kr = np.array(kl, dtype=kL0.dtype), np.array(kr, dtype=kL0.dtype)
    dl, dr = np.stack(dl), np.stack(dr)
    mbf = np.float32(47.90639384423901)
    mb = np.float32(mbf / np.float32(435.2046959714599))
    u_ref, d_ref = o.stereo(w, h, kl, dl, kr, dr, pL, pR, float(mbf), float(mb))
    u, d = amd.ComputeStereoMatches(eL, eR, kl, dl, kr, dr, float(mbf), float(mb))
    assert np.array_equal(u_ref, u)
    assert np.array_equal(d_ref, d)
    assert (u >= 0).sum() > 20


def test_stereo_batch_device_resident(amd):
    """orbfe_stereo_match_batch_device: pairs (2p, 2p+1) of one device batch, everything in HBM."""
    torch = pytest.importorskip("torch")
    w, h, nf = 640, 240, 800
    pairs = [synth.render_stereo(20 + p, w, h, n_shapes=300, max_disp=48) for p in range(3)]
    imgs = np.stack([im for pr in pairs for im in pr])  # L0,R0,L1,R1,L2,R2
    e = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    cap = e.max_keypoints()
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(imgs).to(dev)
    B = len(imgs)
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_u = torch.zeros((B // 2, cap), dtype=torch.float32, device=dev)
    d_d = torch.zeros((B // 2, cap), dtype=torch.float32, device=dev)
    d_ns = torch.zeros((B // 2,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    mbf = np.float32(120.0)
    mb = np.float32(mbf / np.float32(400.0))
    for streams in (1, 2):
        e.set_streams(streams)
        e.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                               d_n.data_ptr(), wait=False)
        e.stereo_match_batch_device(B // 2, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, float(mbf),
                                    float(mb), d_u.data_ptr(), d_d.data_ptr(), d_ns.data_ptr())
        e.synchronize()
        o = orc.Oracle(nf, 1.2, 8, 20, 7)
        n = d_n.cpu().numpy()
        for p in range(B // 2):
            kL, dL, pL = o.extract(imgs[2 * p], want_pyramid=True)
            kR, dR, pR = o.extract(imgs[2 * p + 1], want_pyramid=True)
            assert n[2 * p] == len(kL) and n[2 * p + 1] == len(kR)
            u_ref, d_ref = o.stereo(w, h, kL, dL, kR, dR, pL, pR, float(mbf), float(mb))
            u = d_u[p].cpu().numpy()
            d = d_d[p].cpu().numpy()
            assert np.array_equal(u[: len(kL)], u_ref) and np.array_equal(d[: len(kL)], d_ref)
            assert (u[len(kL):] == -1).all()
            assert int(d_ns[p].item()) == int((u_ref >= 0).sum()) > 30


def _poison_records(kp, desc, rng, W, H):
    """records no extractor writes, planted in copies of real keypoint arrays: (index, field values) -> what a zeroed /
    stale / foreign device buffer looks like to the stereo kernels"""
    bad = {
        3: dict(x=0.0, y=0.0, octave=0),                       # a zeroed record (torch.zeros: round 3's three GPU faults)
        5: dict(x=np.nan, y=50.0, octave=1),
        7: dict(x=100.0, y=np.inf, octave=0),
        11: dict(x=120.0, y=60.0, octave=9),                   # not a pyramid level
        13: dict(x=120.0, y=60.0, octave=-3),
        17: dict(x=1.0e9, y=60.0, octave=0),                   # far outside the image
        19: dict(x=200.0, y=-40.0, octave=2),
        23: dict(x=300.0, y=1.0e6, octave=0),
        29: dict(x=float(W - 2), y=float(H - 2), octave=0),    # inside the image, the 11 x 11 patch is not
        31: dict(x=2.0, y=2.0, octave=0),
    }
    kp, desc = kp.copy(), desc.copy()
    names = kp.dtype.names
    for i, f in bad.items():
        kp[names[0]][i], kp[names[1]][i], kp[names[5]][i] = np.float32(f["x"]), np.float32(f["y"]), f["octave"]
    return kp, desc, sorted(bad)


def test_stereo_device_operands_are_not_trusted(amd):
    """`orbfe_stereo_match_batch_device` takes arbitrary device keypoint records.  Zeroed / NaN /
    octave-9 / out-of-image / border records (left AND right side) must give "no stereo" (-1) -- the reference's own guards
    are `iniu < 0 || endu >= cols` (src/Frame.cc:619-622) and the rowRange / colRange asserts (:609-610, :626) -- never a
    read outside a pyramid level; every other keypoint keeps the oracle's result."""
    torch = pytest.importorskip("torch")
    w, h, nf = 640, 240, 800
    pairs = [synth.render_stereo(60 + p, w, h, n_shapes=300, max_disp=48) for p in range(2)]
    imgs = np.stack([im for pr in pairs for im in pr])
    e = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    cap = e.max_keypoints()
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(imgs).to(dev)
    B = len(imgs)
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_u = torch.zeros((B // 2, cap), dtype=torch.float32, device=dev)
    d_d = torch.zeros((B // 2, cap), dtype=torch.float32, device=dev)
    d_ns = torch.zeros((B // 2,), dtype=torch.int32, device=dev)
    e.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), wait=True)
    mbf = np.float32(120.0)
    mb = np.float32(mbf / np.float32(400.0))
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    rng = np.random.default_rng(5)
    n = d_n.cpu().numpy()
    kps = d_kp.cpu().numpy()
    descs = d_desc.cpu().numpy()
    from orb_slam2_annotate_amd import _lib
    host = []
    for fi in range(B):  # poison the left frame of pair 0, the right frame of pair 1, and both frames' twins of a border match
        k = np.ascontiguousarray(kps[fi, :n[fi]]).view(_lib.KP_DTYPE).reshape(-1)
        dsc = descs[fi, :n[fi]].copy()
        if fi in (0, 3):
            k, dsc, _ = _poison_records(k, dsc, rng, w, h)
        host.append([k, dsc])
    # a border PAIR that really matches (distance 0, right keypoint left of the left one): without the window guard the SAD
    # would read rows -3.. of level 0
    nm = host[0][0].dtype.names
    for fi, x in ((0, 2.0), (1, 1.0)):
        host[fi][0][nm[0]][31], host[fi][0][nm[1]][31], host[fi][0][nm[5]][31] = x, 2.0, 0
        host[fi][1][31] = 0xA5
    for fi in range(B):
        kps[fi, :n[fi]] = host[fi][0].view(np.float32).reshape(-1, 7)
        descs[fi, :n[fi]] = host[fi][1]
    d_kp.copy_(torch.from_numpy(kps))
    d_desc.copy_(torch.from_numpy(descs))
    # all-zero records for a whole pair as well: counts say 700 keypoints, the buffers hold nothing
    e.stereo_match_batch_device(B // 2, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, float(mbf), float(mb),
                                d_u.data_ptr(), d_d.data_ptr(), d_ns.data_ptr())
    e.synchronize()
    pyr = [o.extract(imgs[fi], want_pyramid=True)[2] for fi in range(B)]
    for p in range(B // 2):
        (kL, dL), (kR, dR) = host[2 * p], host[2 * p + 1]
        u_ref, d_ref = o.stereo(w, h, kL, dL, kR, dR, pyr[2 * p], pyr[2 * p + 1], float(mbf), float(mb))
        u, dd = d_u[p].cpu().numpy()[: len(kL)], d_d[p].cpu().numpy()[: len(kL)]
        assert np.array_equal(u, u_ref) and np.array_equal(dd, d_ref)
        assert (u >= 0).sum() > 30
    bad = _poison_records(host[0][0], host[0][1], rng, w, h)[2]
    assert (d_u[0].cpu().numpy()[bad] == -1).all() and (d_d[0].cpu().numpy()[bad] == -1).all()
    # buffers that hold NOTHING (zeros) while the counts say hundreds of keypoints
    d_kp.zero_()
    e.stereo_match_batch_device(B // 2, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, float(mbf), float(mb),
                                d_u.data_ptr(), d_d.data_ptr(), d_ns.data_ptr())
    e.synchronize()
    assert (d_u.cpu().numpy() == -1).all() and (d_ns.cpu().numpy() == 0).all()


def test_compute_stereo_matches_rejects_bad_host_records(amd):
    """the host-operand form can look at its operands: ORBFE_ERR_INVALID for an octave that is no pyramid level, a
    non-finite or out-of-image position (left or right); a border record inside the image is "no stereo" like the oracle"""
    from orb_slam2_annotate_amd import _lib
    w, h, nf = 640, 240, 800
    left, right = synth.render_stereo(61, w, h, n_shapes=300, max_disp=48)
    eL, eR = amd.ORBextractor(nf, 1.2, 8, 20, 7), amd.ORBextractor(nf, 1.2, 8, 20, 7)
    (kL, dL), (kR, dR) = eL(left), eR(right)
    mbf = np.float32(120.0)
    mb = np.float32(mbf / np.float32(400.0))
    nm = kL.dtype.names
    for side in (0, 1):
        for field, val in ((nm[5], 9), (nm[5], -1), (nm[0], np.nan), (nm[1], np.inf), (nm[0], float(w)), (nm[1], -1.0)):
            a, b = kL.copy(), kR.copy()
            (b if side else a)[field][10] = val
            with pytest.raises(_lib.OrbfeError):
                amd.ComputeStereoMatches(eL, eR, a, dL, b, dR, float(mbf), float(mb))
    a, b, da, db = kL.copy(), kR.copy(), dL.copy(), dR.copy()
    for k_, d_, x in ((a, da, 2.0), (b, db, 1.0)):
        k_[nm[0]][31], k_[nm[1]][31], k_[nm[5]][31] = x, 2.0, 0
        d_[31] = 0x5A
    a[nm[0]][40], a[nm[1]][40] = w - 1.0, h - 1.0
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    pL, pR = o.extract(left, want_pyramid=True)[2], o.extract(right, want_pyramid=True)[2]
    u_ref, d_ref = o.stereo(w, h, a, da, b, db, pL, pR, float(mbf), float(mb))
    u, d = amd.ComputeStereoMatches(eL, eR, a, da, b, db, float(mbf), float(mb))
    assert np.array_equal(u, u_ref) and np.array_equal(d, d_ref) and u[31] == -1 and u[40] == -1 and (u >= 0).sum() > 30


def test_stereo_then_next_async_extract_does_not_race(amd):
    """async extract(A) on 4 sub-batch streams, stereo(A), async extract(B != A) enqueued
    right behind it without any host wait -- stereo(A) must still equal the oracle (the sub-batch streams
    of extract(B) overwrite the pyramid slabs and keypoint buffers stereo(A) reads)."""
    torch = pytest.importorskip("torch")
    w, h, nf = 640, 240, 800
    P = 8
    pairsA = [synth.render_stereo(300 + p, w, h, n_shapes=300, max_disp=48) for p in range(P)]
    pairsB = [synth.render_stereo(900 + p, w, h, n_shapes=200, max_disp=32) for p in range(P)]
    imgsA = np.stack([im for pr in pairsA for im in pr])
    imgsB = np.stack([im for pr in pairsB for im in pr])
    e = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    e.set_streams(4)
    cap = e.max_keypoints()
    dev = torch.device("cuda", 0)
    B = 2 * P
    d_A, d_B = torch.from_numpy(imgsA).to(dev), torch.from_numpy(imgsB).to(dev)
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_u = torch.zeros((2, P, cap), dtype=torch.float32, device=dev)
    d_d = torch.zeros((2, P, cap), dtype=torch.float32, device=dev)
    d_ns = torch.zeros((2, P), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    mbf = np.float32(120.0)
    mb = np.float32(mbf / np.float32(400.0))
    for rep in range(3):  # several back-to-back rounds keep all four streams busy
        for k, d_img in enumerate((d_A, d_B)):
            e.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                                   d_n.data_ptr(), wait=False)
            e.stereo_match_batch_device(P, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, float(mbf),
                                        float(mb), d_u[k].data_ptr(), d_d[k].data_ptr(), d_ns[k].data_ptr())
    e.synchronize()
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    for k, imgs in enumerate((imgsA, imgsB)):
        for p in range(P):
            kL, dL, pL = o.extract(imgs[2 * p], want_pyramid=True)
            kR, dR, pR = o.extract(imgs[2 * p + 1], want_pyramid=True)
            u_ref, d_ref = o.stereo(w, h, kL, dL, kR, dR, pL, pR, float(mbf), float(mb))
            assert np.array_equal(d_u[k, p, :len(kL)].cpu().numpy(), u_ref), (k, p)
            assert np.array_equal(d_d[k, p, :len(kL)].cpu().numpy(), d_ref), (k, p)
            assert int(d_ns[k, p].item()) == int((u_ref >= 0).sum())


def _resident(amd, k, d, nodes, u_right=None):
    v = amd.FrameView(k["x"], k["y"], k["octave"], d, (0.0, 640.0, 0.0, 480.0), angle=k["angle"], u_right=u_right)
    return v.upload(amd.FeatureVector.from_node_of_feature(nodes))


@pytest.mark.parametrize("seed,ori", [(31, True), (32, False)])
def test_search_by_bow_resident_frames(amd, seed, ori):
    """SearchByBoW (KF, F) and (KF, KF) on frames uploaded once (orbfe_frame_upload): only the shared-node list, the
    MapPoint masks and the result travel per call -- same outputs as the oracle; the handles are reused across calls."""
    k1, d1, k2, d2 = _two_frames(amd, seed)
    rng = np.random.default_rng(seed)
    n1, n2 = _nodes(d1, 13, 60), _nodes(d2, 13, 60)
    R1, R2 = _resident(amd, k1, d1, n1), _resident(amd, k2, d2, n2)
    m = amd.ORBmatcher(0.7, ori)
    for rep in range(3):  # masks change between calls (map points get created), the frames stay
        has1 = (rng.random(len(k1)) < 0.6).astype(np.uint8)
        has2 = (rng.random(len(k2)) < 0.6).astype(np.uint8)
        rn, r = orc.search_by_bow(d1, has1, k1["angle"], orc.FeatVec(n1), d2, k2["angle"], orc.FeatVec(n2), 0.7, ori)
        gn, g = m.SearchByBoWResident(R1, has1, R2)
        assert rn == gn and np.array_equal(r, g)
        rn, r = orc.search_by_bow_kf(d1, has1, k1["angle"], orc.FeatVec(n1), d2, has2, k2["angle"], orc.FeatVec(n2), 0.7, ori)
        gn, g = m.SearchByBoWResident(R1, has1, R2, has_mp2=has2)
        assert rn == gn and np.array_equal(r, g)
        assert rn > 5
    R1.close()
    R2.close()


def test_resident_frame_closed_and_empty_featvec(amd):
    """round-3 ADVICE: (1) a closed ResidentFrame raises instead of handing the library a dangling view / handle;
    (2) orbfe_frame_upload with an EMPTY FeatureVector given as NULL arrays (n_nodes = 0) must not read through them."""
    import ctypes as C
    from orb_slam2_annotate_amd import _lib
    from orb_slam2_annotate_amd._lib import FeatVecC
    k1, d1, k2, d2 = _two_frames(amd, 33)
    n1, n2 = _nodes(d1, 13, 60), _nodes(d2, 13, 60)
    R1, R2 = _resident(amd, k1, d1, n1), _resident(amd, k2, d2, n2)
    m = amd.ORBmatcher(0.7, True)
    has1 = np.ones(len(k1), np.uint8)
    gn, _ = m.SearchByBoWResident(R1, has1, R2)
    assert gn > 5 and not R2.closed
    R2.close()
    assert R2.closed
    with pytest.raises(ValueError):
        m.SearchByBoWResident(R1, has1, R2)
    with pytest.raises(ValueError):
        R2.GetFeaturesInArea(100.0, 100.0, 20.0)
    R2.close()  # idempotent
    v = amd.FrameView(k2["x"], k2["y"], k2["octave"], d2, (0.0, 640.0, 0.0, 480.0), angle=k2["angle"])
    empty = FeatVecC(0, None, None, None)
    h = C.c_void_p()
    _lib.check(_lib.load().orbfe_frame_upload(0, C.byref(v.c), C.byref(empty), C.byref(h)))
    out = np.full(len(k2), 7, dtype=np.int32)
    n = _lib.check(_lib.load().orbfe_search_by_bow_resident(R1._h, _lib.ptr(has1), h, C.c_float(0.7), 1, _lib.ptr(out)))
    assert n == 0 and (out == -1).all()  # no node in common with an empty FeatureVector
    _lib.load().orbfe_frame_release(h)
    R1.close()


@pytest.mark.parametrize("ori", [True, False])
def test_search_by_bow_multi(amd, ori):
    """SearchByBoW of one frame / key frame against K candidate key frames in ONE call (Tracking::Relocalization,
    src/Tracking.cc:1478-1498; LoopClosing::ComputeSim3, src/LoopClosing.cc:294-321): every candidate's match array and
    count equal the oracle's single-pair call; includes an empty candidate and one sharing no vocabulary node."""
    fr = synth.render_sequence(45, 7, 640, 480, step=1.5)
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    ext = e.extract_batch(np.stack(fr))
    rng = np.random.default_rng(45)
    kF, dF = ext[0]
    nF = _nodes(dF, 13, 60)
    RF = _resident(amd, kF, dF, nF)
    hasF = (rng.random(len(kF)) < 0.6).astype(np.uint8)
    cands, masks, ref_f, ref_kf = [], [], [], []
    for k in range(6):
        k2, d2 = ext[1 + k]
        if k == 3:
            k2, d2 = k2[:0], d2[:0]
        n2 = _nodes(d2, 13, 60) if len(d2) else np.zeros(0, np.int64)
        if k == 4:
            n2 = n2 + 1000
        has2 = (rng.random(len(k2)) < 0.6).astype(np.uint8)
        cands.append(_resident(amd, k2, d2, n2))
        masks.append(has2)
        if len(d2):
            # (KF_k, F): the candidate is the key-frame side; (KF, KF_k): the current key frame is side 1
            ref_f.append(orc.search_by_bow(d2, has2, k2["angle"], orc.FeatVec(n2), dF, kF["angle"], orc.FeatVec(nF), 0.75, ori))
            ref_kf.append(orc.search_by_bow_kf(dF, hasF, kF["angle"], orc.FeatVec(nF), d2, has2, k2["angle"], orc.FeatVec(n2), 0.75, ori))
        else:
            ref_f.append((0, np.full(len(kF), -1, np.int32)))
            ref_kf.append((0, np.full(len(kF), -1, np.int32)))
    m = amd.ORBmatcher(0.75, ori)
    cnt, got = m.SearchByBoWMulti(cands, masks, RF)
    for k in range(6):
        assert (int(cnt[k]), got[k].tolist()) == (int(ref_f[k][0]), ref_f[k][1].tolist()), k
    cnt, got = m.SearchByBoWKFMulti(RF, hasF, cands, masks)
    for k in range(6):
        assert (int(cnt[k]), got[k].tolist()) == (int(ref_kf[k][0]), ref_kf[k][1].tolist()), k
    assert sum(int(r[0]) for r in ref_f) > 50 and sum(int(r[0]) for r in ref_kf) > 50
    cnt, got = m.SearchByBoWMulti([], [], RF)
    assert len(cnt) == 0
    for r in cands + [RF]:
        r.close()


def test_frame_from_extractor_and_device(amd):
    """orbfe_frame_from_extractor / orbfe_frame_from_device: the resident frame built from the extractor's OWN device
    records (no re-upload of keypoints and descriptors) behaves exactly like the uploaded one -- grid (GetFeaturesInArea),
    SearchByBoW after set_featvec, and with ORBFE_FRAME_XY_FROM_VIEW for caller-undistorted positions."""
    torch = pytest.importorskip("torch")
    fr = synth.render_sequence(46, 2, 640, 480, step=2.0)
    e1, e2 = amd.ORBextractor(1000, 1.2, 8, 20, 7), amd.ORBextractor(1000, 1.2, 8, 20, 7)
    k1, d1 = e1(fr[0])
    k2, d2 = e2(fr[1])
    n1, n2 = _nodes(d1, 13, 60), _nodes(d2, 13, 60)
    bounds = (0.0, 640.0, 0.0, 480.0)
    v1 = amd.FrameView(k1["x"], k1["y"], k1["octave"], d1, bounds, angle=k1["angle"])
    v2 = amd.FrameView(k2["x"], k2["y"], k2["octave"], d2, bounds, angle=k2["angle"])
    from orb_slam2_annotate_amd.matcher import ResidentFrame
    A = ResidentFrame(v1, amd.FeatureVector.from_node_of_feature(n1), extractor=e1, frame=0)
    B = ResidentFrame(v2, None, extractor=e2, frame=0)              # Frame::Frame first ...
    B.set_featvec(amd.FeatureVector.from_node_of_feature(n2))        # ... Frame::ComputeBoW later
    U1, U2 = _resident(amd, k1, d1, n1), _resident(amd, k2, d2, n2)  # the uploaded twins
    rng = np.random.default_rng(46)
    qx, qy = rng.uniform(0, 640, 300).astype(np.float32), rng.uniform(0, 480, 300).astype(np.float32)
    qr = rng.uniform(5, 60, 300).astype(np.float32)
    for X, Y in ((A, U1), (B, U2)):
        for a, b in zip(X.GetFeaturesInArea(qx, qy, qr), Y.GetFeaturesInArea(qx, qy, qr)):
            assert a.tolist() == b.tolist()
    m = amd.ORBmatcher(0.7, True)
    has1 = (rng.random(len(k1)) < 0.7).astype(np.uint8)
    rn, r = orc.search_by_bow(d1, has1, k1["angle"], orc.FeatVec(n1), d2, k2["angle"], orc.FeatVec(n2), 0.7, True)
    gn, g = m.SearchByBoWResident(A, has1, B)
    assert rn == gn and np.array_equal(r, g) and rn > 20
    # a frame that is released before anything used it (its build may still be in flight), then the slab reused at once
    for _ in range(3):
        ResidentFrame(v1, None, extractor=e1, frame=0).close()
    C_ = ResidentFrame(v1, amd.FeatureVector.from_node_of_feature(n1), extractor=e1, frame=0)
    gn, g = m.SearchByBoWResident(C_, has1, B)
    assert rn == gn and np.array_equal(r, g)
    # caller-undistorted positions: x / y from the view, descriptors / angles / octaves still from the device records
    xs, ys = (k1["x"] + 0.37).astype(np.float32), (k1["y"] - 0.21).astype(np.float32)
    vs = amd.FrameView(xs, ys, k1["octave"], d1, bounds, angle=k1["angle"])
    S = ResidentFrame(vs, None, extractor=e1, frame=0, flags=ResidentFrame.XY_FROM_VIEW)
    T = vs.upload()
    for a, b in zip(S.GetFeaturesInArea(qx, qy, qr), T.GetFeaturesInArea(qx, qy, qr)):
        assert a.tolist() == b.tolist()
    # device-batch outputs are the caller's: orbfe_frame_from_device on (d_kp + f * cap, d_desc + f * cap * 32)
    cap = e1.max_keypoints()
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(np.stack(fr)).to(dev)
    d_kp = torch.zeros((2, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((2,), dtype=torch.int32, device=dev)
    e1.extract_batch_device(d_img.data_ptr(), 2, 640, 480, 640, 640 * 480, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), wait=True)
    with pytest.raises(Exception):  # the handle's own output block is gone after a device-batch call
        ResidentFrame(v1, None, extractor=e1, frame=0)
    D = ResidentFrame(v2, amd.FeatureVector.from_node_of_feature(n2), d_keypoints=d_kp[1].data_ptr(), d_descriptors=d_desc[1].data_ptr())
    gn, g = m.SearchByBoWResident(A, has1, D)
    assert rn == gn and np.array_equal(r, g)
    with pytest.raises(Exception):  # more keypoints in the view than the extractor produced
        big = amd.FrameView(np.tile(k2["x"], 2), np.tile(k2["y"], 2), np.tile(k2["octave"], 2), np.tile(d2, (2, 1)), bounds)
        ResidentFrame(big, None, extractor=e2, frame=0)
    for r_ in (A, B, U1, U2, C_, S, T, D):
        r_.close()


def test_resident_frame_used_on_another_thread_right_after_the_upload(amd):
    """orbfe_frame_upload / orbfe_frame_from_extractor return WITHOUT waiting for the device (round 4): the slab is filled and
    the grid built on the uploading thread's stream.  A search that uses the frame at once on ANOTHER thread (its own
    stream) must be ordered behind that build by the frame's event -- 40 frames uploaded and consumed back to back, slabs
    recycled through the pool in between, every result equal to the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    from orb_slam2_annotate_amd.matcher import ResidentFrame
    k1, d1, k2, d2 = _two_frames(amd, 34)
    n1, n2 = _nodes(d1, 13, 60), _nodes(d2, 13, 60)
    rng = np.random.default_rng(34)
    has1 = (rng.random(len(k1)) < 0.7).astype(np.uint8)
    rn, r = orc.search_by_bow(d1, has1, k1["angle"], orc.FeatVec(n1), d2, k2["angle"], orc.FeatVec(n2), 0.7, True)
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    fr = synth.render_sequence(34, 2, 640, 480, step=3.0)
    kx, dx = e(fr[1])  # the handle's output block = frame 2's records (same extractor settings as _two_frames)
    assert np.array_equal(dx, d2)
    v2 = amd.FrameView(k2["x"], k2["y"], k2["octave"], d2, (0.0, 640.0, 0.0, 480.0), angle=k2["angle"])
    fv2 = amd.FeatureVector.from_node_of_feature(n2)
    R1 = _resident(amd, k1, d1, n1)
    m = amd.ORBmatcher(0.7, True)
    qx = rng.uniform(0, 640, 64).astype(np.float32)
    qy = rng.uniform(0, 480, 64).astype(np.float32)
    qr = rng.uniform(5, 50, 64).astype(np.float32)
    area_ref = [a.tolist() for a in v2.GetFeaturesInArea(qx, qy, qr)]

    def consume(R):
        gn, g = m.SearchByBoWResident(R1, has1, R)
        area = [a.tolist() for a in R.GetFeaturesInArea(qx, qy, qr)]
        return gn == rn and np.array_equal(g, r) and area == area_ref

    with ThreadPoolExecutor(2) as pool:
        for i in range(40):
            R = v2.upload(fv2) if i % 2 == 0 else ResidentFrame(v2, fv2, extractor=e, frame=0)
            ok = pool.submit(consume, R).result()  # used on a worker thread's stream right after the asynchronous build
            R.close()                              # the slab goes back to the pool and is reused by the next upload
            assert ok, i
    R1.close()


@pytest.mark.parametrize("only_stereo,ori", [(False, True), (True, False)])
def test_search_for_triangulation_multi(amd, only_stereo, ori):
    """One key frame against K neighbours in ONE call (LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:283-315): every
    neighbour's match array equals the oracle's single-pair SearchForTriangulation; includes an empty neighbour and one
    sharing no vocabulary node."""
    o = orc.Oracle()
    sf, sg = o.scale_factors(), o.level_sigma2()
    fr = synth.render_sequence(40, 7, 640, 480, step=1.0)  # the key frame and six neighbours of ONE scene
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    ext = e.extract_batch(np.stack(fr))
    k1, d1 = ext[0]
    rng = np.random.default_rng(40)
    n1 = _nodes(d1, 13, 60)
    ur1 = np.where(rng.random(len(k1)) < 0.5, k1["x"] - 5.0, -1.0).astype(np.float32)
    has1 = (rng.random(len(k1)) < 0.4).astype(np.uint8)
    R1 = _resident(amd, k1, d1, n1, ur1)
    neigh, masks, Fs, eps, refs = [], [], [], [], []
    for k in range(6):
        k2, d2 = ext[1 + k]
        if k == 4:
            k2, d2 = k2[:0], d2[:0]  # a neighbour without keypoints
        n2 = _nodes(d2, 13, 60) if len(d2) else np.zeros(0, np.int64)
        if k == 5:
            n2 = n2 + 1000  # no node in common
        ur2 = np.where(rng.random(len(k2)) < 0.5, k2["x"] - 4.0, -1.0).astype(np.float32)
        has2 = (rng.random(len(k2)) < 0.4).astype(np.uint8)
        F12 = (np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float32) * (0.01 + 0.002 * k)).astype(np.float32)
        ex, ey = 5000.0 - 100 * k, 240.0 + k
        neigh.append(_resident(amd, k2, d2, n2, ur2))
        masks.append(has2); Fs.append(F12); eps.append((ex, ey))
        if len(d2):
            refs.append(orc.search_for_triangulation(d1, has1, k1["x"], k1["y"], k1["angle"], (ur1 >= 0).astype(np.uint8),
                                                     orc.FeatVec(n1), d2, has2, k2["x"], k2["y"], k2["angle"], k2["octave"],
                                                     (ur2 >= 0).astype(np.uint8), orc.FeatVec(n2), F12, ex, ey, sf, sg,
                                                     only_stereo, ori))
        else:
            refs.append((0, np.full(len(k1), -1, np.int32)))
    cnt, match = amd.ORBmatcher(0.6, ori).SearchForTriangulationMulti(R1, has1, neigh, masks, Fs, eps, sf, sg, only_stereo)
    for k in range(6):
        assert int(cnt[k]) == refs[k][0], k
        assert np.array_equal(match[k], refs[k][1]), k
    assert sum(r[0] for r in refs) > 20 and refs[0][0] > 5 and refs[4][0] == 0 and refs[5][0] == 0



@pytest.mark.parametrize("shape,nf,bf,fx,scene", [((1241, 376), 2000, 386.1448, 718.856, "textured"), ((752, 480), 1200, 47.90639384423901, 435.2, "shapes")])
def test_extract_stereo_frame_is_two_extractions_and_the_stereo_matcher(amd, shape, nf, bf, fx, scene):
    """orbfe_extract_stereo_frame = ExtractORB x2 + ComputeStereoMatches of the stereo Frame constructor (src/Frame.cc:78-96)
    in one call on one handle: keypoints / descriptors of both eyes, mvuRight and mvDepth equal the oracle's, and the
    handle then holds the pair (a resident frame built from frame 0 equals one built from host arrays)."""
    w, h = shape
    left, right = synth.STEREO_SCENES[scene](5, w, h)
    e = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    mbf = np.float32(bf)
    mb = np.float32(mbf / np.float32(fx))
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    krL, drL, pL = o.extract(left, want_pyramid=True)
    krR, drR, pR = o.extract(right, want_pyramid=True)
    u_ref, d_ref = o.stereo(w, h, krL, drL, krR, drR, pL, pR, float(mbf), float(mb))
    for _ in range(2):  # (the second call reuses the handle's buffers and streams)
        kL, dL, kR, dR, u, d = e.extract_stereo_frame(left, right, float(mbf), float(mb))
        assert np.array_equal(krL, kL) and np.array_equal(krR, kR)
        assert np.array_equal(drL, dL) and np.array_equal(drR, dR)
        assert np.array_equal(u_ref, u) and np.array_equal(d_ref, d)
    assert (u >= 0).sum() > 100
    # the ordinary calls still work on the same handle afterwards, and a different size re-plans it
    k1, d1 = e(left)
    assert np.array_equal(k1, krL) and np.array_equal(d1, drL)
    small_l, small_r = np.ascontiguousarray(left[:240, :320]), np.ascontiguousarray(right[:240, :320])
    ks, ds, _, _, us, _ = e.extract_stereo_frame(small_l, small_r, float(mbf), float(mb))
    ko, do, po = o.extract(small_l, want_pyramid=True)
    assert np.array_equal(ks, ko) and np.array_equal(ds, do) and len(us) == len(ks)
