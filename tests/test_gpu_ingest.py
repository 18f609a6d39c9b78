"""Gray conversion and ComputeDistinctiveDescriptors against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("channels,rgb", [(3, True), (3, False), (4, True), (4, False)])
def test_cvt_gray(channels, rgb):
    import orb_slam2_annotate_amd as amd
    rng = np.random.default_rng(channels * 2 + rgb)
    for (w, h) in [(640, 480), (123, 45), (5, 3)]:
        img = rng.integers(0, 256, size=(h, w, channels), dtype=np.uint8)
        ref = np.zeros((h, w), np.uint8)
        orc.lib().orc_cvt_gray(orc._p(img), w, h, w * channels, channels, int(rgb), orc._p(ref), w)
        assert np.array_equal(amd.cvtColorToGray(img, rgb), ref)
    # extremes: pure white stays 255, pure colours follow the 4899/9617/1868 split
    one = np.zeros((1, 4, 3), np.uint8)
    one[0, 0] = 255; one[0, 1, 0] = 255; one[0, 2, 1] = 255; one[0, 3, 2] = 255
    assert list(amd.cvtColorToGray(one, True)[0]) == [255, 76, 150, 29]


def test_distinctive_descriptors():
    import orb_slam2_annotate_amd as amd
    rng = np.random.default_rng(11)
    lists = []
    for n in [1, 2, 3, 4, 7, 20, 65, 130, 0, 300]:
        base = rng.integers(0, 256, size=32, dtype=np.uint8)
        d = np.repeat(base[None], n, 0)
        if n:
            noise = (rng.random((n, 256)) < 0.15).astype(np.uint8)
            d = d ^ np.packbits(noise, axis=1)
        lists.append(d)
    lists.append(np.zeros((6, 32), np.uint8))  # all identical: every median 0 -> first row wins
    got = amd.ComputeDistinctiveDescriptors(lists)
    L = orc.lib()
    for d, g in zip(lists, got):
        d = np.ascontiguousarray(d)
        ref = L.orc_distinctive_descriptor(orc._p(d), len(d)) if len(d) else -1
        assert int(g) == ref


EUROC = dict(w=752, h=480, fx=458.654, fy=457.296, cx=367.215, cy=248.375, k1=-0.28340811, k2=0.07395907,
             p1=0.00019359, p2=1.76187114e-05)  # shape of the EuRoC cam0 calibration (Examples/Stereo/EuRoC.yaml)


@pytest.mark.parametrize("seed,w,h", [(0, 752, 480), (1, 640, 480), (2, 131, 77), (3, 5, 3)])
def test_remap_matches_oracle(seed, w, h):
    import orb_slam2_annotate_amd as amd
    from orb_slam2_annotate_amd import synth
    rng = np.random.default_rng(seed)
    src = synth.render_frame(seed, w, h) if w >= 64 else rng.integers(0, 256, (h, w), dtype=np.uint8)
    s = w / 752.0
    mx, my = orc.rectify_maps(w, h, EUROC["fx"] * s, EUROC["fy"] * s, EUROC["cx"] * s, h / 2 + 3.0, EUROC["k1"],
                              EUROC["k2"], EUROC["p1"], EUROC["p2"])
    r = amd.Rectifier(mx, my)
    assert np.array_equal(r(src), orc.remap_linear(src, mx, my))
    # a rougher map: sub-pixel noise, excursions far outside, NaN / inf entries
    mx2 = (mx + rng.normal(0, 0.7, mx.shape)).astype(np.float32)
    my2 = (my + rng.normal(0, 0.7, my.shape)).astype(np.float32)
    bad = rng.random(mx.shape) < 0.02
    mx2[bad] = rng.choice(np.array([np.nan, np.inf, -np.inf, 1e12, -3.0, -1.0, -0.5, w - 1.0, w - 0.5, w + 40.0],
                                   np.float32), bad.sum())
    bad = rng.random(mx.shape) < 0.02
    my2[bad] = rng.choice(np.array([np.nan, -1e12, -1.0, -0.25, h - 1.0, h - 0.75, h + 0.0, 4e4], np.float32), bad.sum())
    r2 = amd.Rectifier(mx2, my2)
    assert np.array_equal(r2(src), orc.remap_linear(src, mx2, my2))
    # destination (map) size differs from the source size, strided source view
    big = np.zeros((h + 9, w + 13), np.uint8)
    big[:] = rng.integers(0, 256, big.shape, dtype=np.uint8)
    view = big[4:4 + h, 6:6 + w]
    assert np.array_equal(r2(view), orc.remap_linear(np.ascontiguousarray(view), mx2, my2))
    crop = amd.Rectifier(mx2[: max(h // 2, 1), : max(w // 3, 1)], my2[: max(h // 2, 1), : max(w // 3, 1)])
    assert np.array_equal(crop(src), orc.remap_linear(src, mx2[: max(h // 2, 1), : max(w // 3, 1)],
                                                      my2[: max(h // 2, 1), : max(w // 3, 1)]))


def test_remap_identity_and_invalid():
    import orb_slam2_annotate_amd as amd
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, (48, 100), dtype=np.uint8)
    x, y = np.meshgrid(np.arange(100, dtype=np.float32), np.arange(48, dtype=np.float32))
    r = amd.Rectifier(x, y)
    assert np.array_equal(r(im), im)
    with pytest.raises(ValueError):
        r(np.zeros((4, 4, 3), np.uint8))
    with pytest.raises(ValueError):
        amd.Rectifier(x, y[:-1])
    L = amd._lib.load()
    h = C.c_void_p()
    assert L.orbfe_rectifier_create(0, None, amd._lib.ptr(y), 100, 48, 100, C.byref(h)) == amd._lib.ERR_INVALID
    assert L.orbfe_rectifier_create(0, amd._lib.ptr(x), amd._lib.ptr(y), 100, 48, 99, C.byref(h)) == amd._lib.ERR_INVALID


def test_remap_batch_device_feeds_the_extractor():
    """EuRoC stereo ingest as the reference runs it (remap left + right, then extract), device-resident:
    rectified frames never visit the host; keypoints == oracle extractor on the oracle-rectified frames."""
    torch = pytest.importorskip("torch")
    import orb_slam2_annotate_amd as amd
    from orb_slam2_annotate_amd import synth
    w, h, B = 752, 480, 4
    raw = np.stack([synth.render_frame(40 + i, w, h) for i in range(B)])
    mx, my = orc.rectify_maps(w, h, EUROC["fx"], EUROC["fy"], EUROC["cx"], EUROC["cy"], EUROC["k1"], EUROC["k2"],
                              EUROC["p1"], EUROC["p2"])
    r = amd.Rectifier(mx, my)
    dev = torch.device("cuda", 0)
    d_raw = torch.from_numpy(raw).to(dev)
    d_rect = torch.zeros((B, h, w), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    r.batch_device(d_raw.data_ptr(), B, w, h, w, w * h, d_rect.data_ptr(), w, w * h)
    rect = d_rect.cpu().numpy()
    for i in range(B):
        assert np.array_equal(rect[i], orc.remap_linear(raw[i], mx, my))
    e = amd.ORBextractor(1200, 1.2, 8, 20, 7)
    cap = e.max_keypoints()
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e.extract_batch_device(d_rect.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    o = orc.Oracle(1200, 1.2, 8, 20, 7)
    k0, dsc0 = o.extract(rect[0])
    n0 = int(d_n[0].item())
    assert n0 == len(k0)
    assert np.array_equal(d_desc[0, :n0].cpu().numpy(), dsc0)


TUM1_K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)  # Examples/Monocular/TUM1.yaml
TUM1_D = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


@pytest.mark.parametrize("n_dist", [0, 4, 5, 8])
def test_undistort_points_bit_exact(n_dist):
    import orb_slam2_annotate_amd as amd
    rng = np.random.default_rng(60 + n_dist)
    pts = np.stack([rng.uniform(-20, 660, 5000), rng.uniform(-20, 500, 5000)], axis=1).astype(np.float32)
    d = np.concatenate([TUM1_D, [0.01, -0.02, 0.003]]).astype(np.float32)[:n_dist]
    ref = orc.undistort_points(pts, TUM1_K, d)
    got = amd.undistortPoints(pts, TUM1_K, d)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # 3x3 K accepted too
    K = np.array([[TUM1_K[0], 0, TUM1_K[2]], [0, TUM1_K[1], TUM1_K[3]], [0, 0, 1]], np.float32)
    assert np.array_equal(amd.undistortPoints(pts[:10], K, d), ref[:10])
    assert amd.undistortPoints(np.zeros((0, 2), np.float32), K, d).shape == (0, 2)


def test_undistort_keypoints_bounds_and_grid():
    """TUM1 mono pipeline after extraction: UndistortKeyPoints, ComputeImageBounds, AssignFeaturesToGrid."""
    import orb_slam2_annotate_amd as amd
    from orb_slam2_annotate_amd import synth
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    kps, desc = e(synth.render_frame(77, 640, 480))
    un = amd.UndistortKeyPoints(kps, TUM1_K, TUM1_D)
    ref = orc.undistort_points(np.stack([kps["x"], kps["y"]], axis=1), TUM1_K, TUM1_D)
    assert np.array_equal(un["x"], ref[:, 0]) and np.array_equal(un["y"], ref[:, 1])
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(un[f], kps[f])
    assert amd.UndistortKeyPoints(kps, TUM1_K, np.zeros(5, np.float32)) is not kps
    assert np.array_equal(amd.UndistortKeyPoints(kps, TUM1_K, np.zeros(5, np.float32)), kps)
    b = amd.ComputeImageBounds(640, 480, TUM1_K, TUM1_D)
    assert b == orc.image_bounds(640, 480, TUM1_K, TUM1_D)
    assert b != (0.0, 640.0, 0.0, 480.0) and b[0] < b[1] and b[2] < b[3]
    assert amd.ComputeImageBounds(640, 480, TUM1_K, np.zeros(4, np.float32)) == (0.0, 640.0, 0.0, 480.0)
    # the undistorted frame on its real bounds through the grid
    F = amd.FrameView(un["x"], un["y"], un["octave"], desc, b, angle=un["angle"])
    Fo = orc.Frame(un["x"], un["y"], un["octave"], desc, b, angle=un["angle"])
    got = F.GetFeaturesInArea([100.0, 320.0, 600.0], [80.0, 240.0, 400.0], [40.0, 25.0, 60.0])
    for g, (x, y, r) in zip(got, [(100, 80, 40), (320, 240, 25), (600, 400, 60)]):
        assert g.tolist() == Fo.features_in_area(x, y, r).tolist() and len(g) > 0


def test_undistort_keypoints_batch_device_and_rgbd():
    torch = pytest.importorskip("torch")
    import orb_slam2_annotate_amd as amd
    from orb_slam2_annotate_amd import synth
    B, w, h = 3, 640, 480
    imgs = np.stack([synth.render_frame(90 + i, w, h) for i in range(B)])
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    cap = e.max_keypoints()
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(imgs).to(dev)
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_un = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    L = amd._lib.load()
    amd._lib.check(L.orbfe_undistort_keypoints_batch_device(0, C.c_void_p(d_kp.data_ptr()), C.c_void_p(d_n.data_ptr()), B,
                                                            cap, amd._lib.ptr(TUM1_K), amd._lib.ptr(TUM1_D), 5,
                                                            C.c_void_p(d_un.data_ptr())))
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy()
    un = d_un.cpu().numpy()
    for f in range(B):
        ref = orc.undistort_points(kp[f, :n[f], :2], TUM1_K, TUM1_D)
        assert np.array_equal(un[f, :n[f], :2], ref)
        assert np.array_equal(un[f, :n[f], 2:].view(np.uint32), kp[f, :n[f], 2:].view(np.uint32))
        assert not un[f, n[f]:].any()
    # ComputeStereoFromRGBD on frame 0
    rng = np.random.default_rng(3)
    depth = np.where(rng.random((h, w)) < 0.8, rng.uniform(0.4, 8.0, (h, w)), 0.0).astype(np.float32)
    k0 = np.zeros(n[0], dtype=amd.KP_DTYPE); k0["x"], k0["y"] = kp[0, :n[0], 0], kp[0, :n[0], 1]
    u0 = np.zeros(n[0], dtype=amd.KP_DTYPE); u0["x"], u0["y"] = un[0, :n[0], 0], un[0, :n[0], 1]
    ur, dp = amd.ComputeStereoFromRGBD(k0, u0, depth, 40.0)
    ur_ref, dp_ref = orc.stereo_from_rgbd(k0["x"], k0["y"], u0["x"], depth, 40.0)
    assert np.array_equal(ur, ur_ref) and np.array_equal(dp, dp_ref)
    assert (dp > 0).sum() > 100 and (dp < 0).sum() > 10
    with pytest.raises(amd.OrbfeError):
        bad = k0.copy(); bad["x"][0] = 640.0
        amd.ComputeStereoFromRGBD(bad, u0, depth, 40.0)


@pytest.mark.parametrize("streams", [1, 2])
def test_euroc_stereo_chain_matches_oracle(streams):
    """The whole EuRoC stereo frame of the reference, device-resident and without a host wait inside: cv::remap of both
    eyes (Examples/Stereo/stereo_euroc.cc:136-137) -> ExtractORB L + R (src/Frame.cc:78-81) -> ComputeStereoMatches
    (:90) -> ComputeBoW on ORBvoc's shape with levelsup 4 (:433-440) -> SearchByBoW(pair t-1, pair t) on the left
    keypoints (src/ORBmatcher.cc:185-325); two batches back to back on the same buffers, each step against the oracle."""
    torch = pytest.importorskip("torch")
    import orb_slam2_annotate_amd as amd
    from orb_slam2_annotate_amd import synth
    from orb_slam2_annotate_amd.vocabulary import synthetic_vocabulary_arrays
    w, h, P, NF = 752, 480, 4, 1200
    mapsL, mapsR = synth.rectify_maps(w, h, **synth.EUROC_CAM0), synth.rectify_maps(w, h, **synth.EUROC_CAM1)
    rl, rr = amd.Rectifier(*mapsL), amd.Rectifier(*mapsR)
    arrays = synthetic_vocabulary_arrays(10, 6, 5)
    vo = orc.Vocabulary.from_arrays(arrays)
    voc = amd.ORBVocabulary()
    assert voc.createFromArrays(arrays)
    e = amd.ORBextractor(NF, 1.2, 8, 20, 7)
    e.set_streams(streams)
    cap = e.max_keypoints(w, h)
    dev = torch.device("cuda", 0)
    batches = [[synth.render_stereo_raw(700 + 10 * b + p, w, h) for p in range(P)] for b in range(2)]
    d_rawL = [torch.from_numpy(np.stack([x[0] for x in bt])).to(dev) for bt in batches]
    d_rawR = [torch.from_numpy(np.stack([x[1] for x in bt])).to(dev) for bt in batches]
    d_rect = torch.zeros((2 * P, h, w), dtype=torch.uint8, device=dev)
    d_kp = torch.zeros((2 * P, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((2 * P, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((2 * P,), dtype=torch.int32, device=dev)
    d_u = torch.zeros((2, P, cap), dtype=torch.float32, device=dev)
    d_dep = torch.zeros((2, P, cap), dtype=torch.float32, device=dev)
    d_ns = torch.zeros((2, P), dtype=torch.int32, device=dev)
    d_match = torch.zeros((2, P - 1, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros((2, P - 1), dtype=torch.int32, device=dev)
    mbf = np.float32(47.90639384423901)
    mb = np.float32(mbf / np.float32(435.2046959714599))
    torch.cuda.synchronize()
    outs = []
    for b in range(2):
        e.extract_stereo_rectified_batch_device(rl, rr, d_rawL[b].data_ptr(), d_rawR[b].data_ptr(), P, w, h, w, w * h,
                                                d_rect.data_ptr(), d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
        e.stereo_match_batch_device(P, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, float(mbf), float(mb),
                                    d_u[b].data_ptr(), d_dep[b].data_ptr(), d_ns[b].data_ptr())
        voc.bow_match_consecutive_batch_device(P, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap,
                                               d_match[b].data_ptr(), d_nm[b].data_ptr(), nnratio=0.7,
                                               check_orientation=True, levelsup=4, extractor=e, stereo=True)
        if b == 1:
            e.synchronize()
            outs.append((d_rect.cpu().numpy(), d_kp.cpu().numpy(), d_desc.cpu().numpy(), d_n.cpu().numpy()))
    o = orc.Oracle(NF, 1.2, 8, 20, 7)
    for b in range(2):
        prev, tot_st, tot_bow = None, 0, 0
        for p in range(P):
            rectL, rectR = orc.remap_linear(batches[b][p][0], *mapsL), orc.remap_linear(batches[b][p][1], *mapsR)
            kL, dL, pL = o.extract(rectL, want_pyramid=True)
            kR, dR, pR = o.extract(rectR, want_pyramid=True)
            if b == 1:  # the buffers hold the last batch
                rect, kp, desc, n = outs[0]
                assert np.array_equal(rect[2 * p], rectL) and np.array_equal(rect[2 * p + 1], rectR)
                assert n[2 * p] == len(kL) and n[2 * p + 1] == len(kR)
                assert np.array_equal(desc[2 * p, :len(kL)], dL) and np.array_equal(desc[2 * p + 1, :len(kR)], dR)
                assert np.array_equal(kp[2 * p, :len(kL)].view(np.uint8).reshape(-1, 28), kL.view(np.uint8).reshape(-1, 28))
            u_ref, dep_ref = o.stereo(w, h, kL, dL, kR, dR, pL, pR, float(mbf), float(mb))
            assert np.array_equal(d_u[b, p, :len(kL)].cpu().numpy(), u_ref), (b, p)
            assert np.array_equal(d_dep[b, p, :len(kL)].cpu().numpy(), dep_ref), (b, p)
            tot_st += int((u_ref >= 0).sum())
            fv = orc.FeatVec(vo.transform(dL, 4)[3])
            if prev is not None:
                k0, d0, fv0 = prev
                rn, rm = orc.search_by_bow(d0, np.ones(len(k0), np.uint8), k0["angle"], fv0, dL, kL["angle"], fv, 0.7, True)
                assert int(d_nm[b, p - 1].item()) == rn, (b, p)
                assert np.array_equal(d_match[b, p - 1, :len(kL)].cpu().numpy(), rm), (b, p)
                tot_bow += rn
            prev = (kL, dL, fv)
        assert tot_st > 0.4 * P * NF
