"""GPU parity tests of the extractor: every stage and the whole ORBextractor::operator()
through the C-ABI (liborbfe.so) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import oracle_lib as orc
from orb_slam2_annotate_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import orb_slam2_annotate_amd as m
    return m


def _kp_equal(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for name in a.dtype.names:
        assert np.array_equal(a[name], b[name]), name


@pytest.mark.parametrize("shape", [(640, 480), (752, 480), (1241, 376), (533, 400), (70, 67)])
def test_resize_bit_exact(amd, shape):
    w, h = shape
    img = synth.render_frame(3, w, h)
    o = orc.Oracle()
    sizes = o.level_sizes(w, h)
    cur = img
    for (dw, dh) in sizes[1:]:
        ref = orc.resize_linear(cur, dw, dh)
        got = amd.resize_linear(cur, dw, dh)
        assert np.array_equal(ref, got)
        cur = ref


@pytest.mark.parametrize("shape", [(640, 480), (179, 134), (65, 130), (64, 16), (7, 9)])
def test_blur_bit_exact(amd, shape):
    w, h = shape
    img = synth.adversarial("noise", w, h, seed=5)
    assert np.array_equal(orc.gaussian_blur7(img), amd.gaussian_blur7(img))
    img = synth.render_frame(8, w, h) if w > 32 and h > 32 else img
    assert np.array_equal(orc.gaussian_blur7(img), amd.gaussian_blur7(img))


def _check_frame(amd, img, params):
    nf, sf, nl, ini, mn = params
    o = orc.Oracle(nf, sf, nl, ini, mn)
    e = amd.ORBextractor(nf, sf, nl, ini, mn)
    H, W = img.shape
    kps_ref, desc_ref, pyr = o.extract(img, want_pyramid=True)
    kps, desc = e(img)
    levels = o.split_pyramid(pyr, W, H)
    # pyramid (mvImagePyramid)
    for l, ref in enumerate(levels):
        assert np.array_equal(ref, e.pyramid_level(l)), f"pyramid level {l}"
    # grid-stage candidates, in emission order
    for l, ref in enumerate(levels):
        xr, yr, rr = orc.grid_candidates(o, ref)
        xg, yg, rg = e.debug_candidates(l)
        assert len(xr) == len(xg), f"level {l}: {len(xr)} vs {len(xg)} candidates"
        assert np.array_equal(xr, xg) and np.array_equal(yr, yg) and np.array_equal(rr, rg), f"candidates level {l}"
    # blurred levels
    for l, ref in enumerate(levels):
        assert np.array_equal(orc.gaussian_blur7(ref), e.debug_blurred_level(l)), f"blur level {l}"
    _kp_equal(kps_ref, kps)
    assert np.array_equal(desc_ref, desc)
    return len(kps)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_extract_tum_640x480(amd, seed):
    n = _check_frame(amd, synth.render_frame(seed), (1000, 1.2, 8, 20, 7))
    assert n > 500


def test_extract_kitti_1241x376(amd):
    l, r = synth.render_stereo(4)
    assert _check_frame(amd, l, (2000, 1.2, 8, 20, 7)) > 1000
    assert _check_frame(amd, r, (2000, 1.2, 8, 12, 7)) > 1000  # KITTI04-12 iniThFAST


def test_extract_euroc_752x480(amd):
    assert _check_frame(amd, synth.render_frame(9, 752, 480), (1200, 1.2, 8, 20, 7)) > 600


@pytest.mark.parametrize("seed", [4, 5])
def test_extract_mono_initialisation_extractor(amd, seed):
    """mpIniORBextractor = ORBextractor(2 * nFeatures, ...) (src/Tracking.cc:128): 2000 features on a 640 x 480 TUM frame,
    single frame and a batch on several streams."""
    params = (2000, 1.2, 8, 20, 7)
    n = _check_frame(amd, synth.render_frame(seed), params)
    assert n > 1200
    frames = np.stack(synth.render_sequence(seed, 9, 640, 480))
    e = amd.ORBextractor(*params)
    e.set_streams(3)
    o = orc.Oracle(*params)
    for (k, d), f in zip(e.extract_batch(frames), frames):
        kr, dr = o.extract(f)
        _kp_equal(kr, k)
        assert np.array_equal(dr, d)


@pytest.mark.parametrize("kind", ["constant", "noise", "checker"])
def test_extract_adversarial(amd, kind):
    n = _check_frame(amd, synth.adversarial(kind, 640, 480, seed=1), (1000, 1.2, 8, 20, 7))
    if kind == "constant":
        assert n == 0


def test_extract_other_params(amd):
    img = synth.render_frame(21, 400, 300)
    _check_frame(amd, img, (500, 1.2, 8, 20, 7))
    _check_frame(amd, img, (300, 1.5, 4, 15, 5))
    _check_frame(amd, img, (2000, 1.1, 6, 30, 10))


@pytest.mark.parametrize("ini,mn", [(15, 7), (16, 7), (17, 16), (19, 3), (40, 24), (120, 16), (255, 7), (7, 20)])
def test_fast_pretest_forms_at_the_threshold_switch(amd, ini, mn):
    """k_fast_cells phase A takes the quantised 4-pixels-per-op pre-test for thresholds >= 16 and the exact packed-16 form
    below (csrc/k_fast.hip); the candidates in emission order, their responses and everything downstream must not depend
    on which one listed the pixels -- thresholds on both sides of the switch, for the first attempt (iniThFAST) and for
    the fallback (minThFAST), incl. minThFAST > iniThFAST, on a textured and on a low-contrast frame."""
    _check_frame(amd, synth.render_frame(40 + ini, 400, 300), (800, 1.2, 5, ini, mn))
    rng = np.random.default_rng(ini * 256 + mn)
    y, x = np.mgrid[0:240, 0:320]
    soft = ((x * 0.31 + y * 0.17) % 256 + rng.normal(0, 6, (240, 320))).clip(0, 255).astype(np.uint8)  # few cells reach iniThFAST
    _check_frame(amd, soft, (500, 1.2, 4, ini, mn))


def test_extract_strided_and_batch(amd):
    imgs = np.stack([synth.render_frame(30 + i, 320, 240) for i in range(5)])
    o = orc.Oracle(500, 1.2, 8, 20, 7)
    e = amd.ORBextractor(500, 1.2, 8, 20, 7)
    res = e.extract_batch(imgs)
    for f in range(5):
        kr, dr = o.extract(imgs[f])
        _kp_equal(kr, res[f][0])
        assert np.array_equal(dr, res[f][1])
    # strided view (cv::Mat ROI): every other column block of a wider buffer
    wide = np.zeros((240, 512), dtype=np.uint8)
    wide[:, 100:420] = imgs[0]
    k2, d2 = e(wide[:, 100:420])
    _kp_equal(res[0][0], k2)
    assert np.array_equal(res[0][1], d2)


def test_empty_and_errors(amd):
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    k, d = e(np.zeros((0, 0), dtype=np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    with pytest.raises(AssertionError):
        e(np.zeros((480, 640), dtype=np.float32))
    with pytest.raises(amd.OrbfeError) as ei:
        e(synth.render_frame(1), capacity=10)
    assert ei.value.code == -2
    with pytest.raises(amd.OrbfeError):  # node rectangles are signed 16-bit
        e(np.zeros((40, 32800), dtype=np.uint8))
    k, d = e(np.zeros((40, 8200), dtype=np.uint8))  # (round 1 refused anything beyond 8191 px)
    assert len(k) == 0


@pytest.mark.parametrize("params,shape,kind", [((3000, 1.3, 1, 20, 7), (640, 480), "noise"),
                                               ((3000, 1.3, 1, 20, 7), (320, 240), "render"),
                                               ((9000, 1.2, 2, 20, 7), (752, 480), "noise")])
def test_octree_node_list_in_global_memory(amd, params, shape, kind):
    """More than ~2 890 keypoints asked of ONE pyramid level: the octree node list no longer fits in LDS and the
    same generation passes run with it in global memory (k_octree_global) -- still on the device, still the
    reference's selection, keypoint for keypoint (round 1 refused these configurations)."""
    w, h = shape
    img = synth.adversarial("noise", w, h, seed=3) if kind == "noise" else synth.render_frame(2, w, h)
    n = _check_frame(amd, img, params)
    if kind == "noise":
        assert n > 2900  # the quota of the level really exceeds the LDS form's capacity
    # a small batch on two streams through the same path
    e = amd.ORBextractor(*params)
    e.set_streams(2)
    imgs = np.stack([synth.adversarial("noise", w, h, seed=10 + i) for i in range(3)])
    o = orc.Oracle(*params)
    for (kp, desc), im in zip(e.extract_batch(imgs), imgs):
        kr, dr = o.extract(im)
        _kp_equal(kr, kp)
        assert np.array_equal(dr, desc)


def test_tables_match_oracle(amd):
    for p in [(1000, 1.2, 8, 20, 7), (2000, 1.2, 8, 20, 7), (1200, 1.2, 8, 20, 7), (777, 1.3, 5, 20, 7)]:
        o = orc.Oracle(*p)
        e = amd.ORBextractor(*p)
        assert np.array_equal(o.scale_factors(), e.GetScaleFactors())
        assert np.array_equal(o.inv_scale_factors(), e.GetInverseScaleFactors())
        assert np.array_equal(o.level_sigma2(), e.GetScaleSigmaSquares())
        assert np.array_equal(o.inv_level_sigma2(), e.GetInverseScaleSigmaSquares())
        assert o.features_per_level() == list(e.features_per_level())
        assert o.umax() == list(e.umax())
        assert e.GetLevels() == p[2]


def test_device_octree_equals_host_octree(amd):
    """k_octree (device, generation form) vs octree_host.cpp (independent host implementation)."""
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    for seed in (41, 42):
        img = synth.render_frame(seed)
        e.debug_host_octree(False)
        kd, dd = e(img)
        e.debug_host_octree(True)
        kh, dh = e(img)
        e.debug_host_octree(False)
        _kp_equal(kd, kh)
        assert np.array_equal(dd, dh)


def test_octree_small_quota_and_wide_image(amd):
    # tiny quotas (phase 1 overshoot, N=0 levels) and a very wide image (nIni > 2 roots)
    _check_frame(amd, synth.render_frame(50, 640, 480), (20, 1.2, 8, 20, 7))
    _check_frame(amd, synth.render_frame(51, 1000, 200), (800, 1.2, 6, 20, 7))
    _check_frame(amd, synth.render_frame(52, 2000, 150), (1500, 1.2, 4, 20, 7))


def test_streams_and_async_give_identical_results(amd):
    """Sub-batch streams (1..4) and the asynchronous entry point must not change a single byte."""
    import ctypes as C
    imgs = np.stack([synth.render_frame(60 + i, 320, 240) for i in range(7)])
    o = orc.Oracle(400, 1.2, 8, 20, 7)
    ref = [o.extract(im) for im in imgs]
    for n in (1, 2, 3, 4):
        e = amd.ORBextractor(400, 1.2, 8, 20, 7)
        e.set_streams(n)
        for _ in range(2):  # second call reuses the workspace slices
            res = e.extract_batch(imgs)
            for f in range(len(imgs)):
                _kp_equal(ref[f][0], res[f][0])
                assert np.array_equal(ref[f][1], res[f][1])
        # batch size change with streams in flight
        res = e.extract_batch(imgs[:3])
        for f in range(3):
            _kp_equal(ref[f][0], res[f][0])


def test_very_wide_image_starts_the_octree_with_many_roots(amd):
    """813x77 at 3 levels: nIni = 17 roots per level, 4*nIni children can exceed the quota-based bound; the
    wrappers size their buffers with orbfe_extractor_max_keypoints_for (found by tools/fuzz_parity.py)."""
    img = synth.render_frame(7, 813, 77, n_shapes=120)
    _check_frame(amd, img, (100, 1.1, 3, 30, 10))
    e = amd.ORBextractor(100, 1.1, 3, 30, 10)
    assert e.max_keypoints(813, 77) > e.max_keypoints()


def test_large_batch_uses_the_throughput_kernels(amd):
    """Launches of more than 8 frames take the global-memory form of k_octree (the register form serves
    small launches); 1 stream = one 20-frame launch, 8 streams = sub-batches of 3."""
    imgs = np.stack([synth.render_frame(160 + i, 320, 240) for i in range(20)])
    o = orc.Oracle(400, 1.2, 8, 20, 7)
    ref = [o.extract(im) for im in imgs]
    for n in (1, 8):
        e = amd.ORBextractor(400, 1.2, 8, 20, 7)
        e.set_streams(n)
        res = e.extract_batch(imgs)
        for f in range(len(imgs)):
            _kp_equal(ref[f][0], res[f][0])
            assert np.array_equal(ref[f][1], res[f][1])


def test_large_and_tiny_images(amd):
    # full-HD frame with a large quota (octree node list in LDS ~1.1k nodes), and images so small
    # that the upper pyramid levels have no FAST grid at all
    assert _check_frame(amd, synth.render_frame(70, 1920, 1080, n_shapes=2500), (5000, 1.2, 8, 20, 7)) > 3000
    _check_frame(amd, synth.render_frame(71, 96, 80, n_shapes=30), (200, 1.2, 8, 20, 7))
    _check_frame(amd, synth.render_frame(72, 66, 64, n_shapes=20), (100, 1.2, 8, 20, 7))
    _check_frame(amd, synth.adversarial("noise", 100, 72, seed=9), (500, 1.2, 8, 20, 7))


def test_images_beyond_8191_pixels(amd):
    # round 1 refused images wider / higher than 8191 px (13-bit fields in the octree's spatial slot key)
    wide = np.tile(synth.render_frame(73, 1500, 96, n_shapes=300), (1, 6))[:, :8700].copy()
    assert _check_frame(amd, wide, (1500, 1.2, 4, 20, 7)) > 500
    # (much taller than wide gives round(width / height) = 0 octree roots and no keypoints, src/ORBextractor.cc:570)
    tall = np.tile(synth.render_frame(74, 850, 1700, n_shapes=300), (5, 5))[:8400, :4250].copy()
    assert _check_frame(amd, tall, (800, 1.2, 2, 20, 7)) > 100


def test_noise_fullsize_stresses_cell_capacity(amd):
    # pure noise: corners everywhere, work queues and per-cell slot ranges near their bounds
    _check_frame(amd, synth.adversarial("noise", 640, 480, seed=3), (1000, 1.2, 8, 7, 7))
    img = np.zeros((480, 640), np.uint8)
    img[::2, ::2] = 255  # isolated bright pixels on a 2-px lattice: maximal NMS-survivor density
    _check_frame(amd, img, (1000, 1.2, 8, 20, 7))


def test_full_size_device_batch_properties(amd):
    """BASELINE configs[1] shape at batch scale (device-resident, 4 streams, asynchronous): size-independent
    properties for every frame, and equality with the oracle-checked single-frame path for a sample."""
    torch = pytest.importorskip("torch")
    B, w, h = 96, 640, 480
    imgs = np.stack(synth.render_sequence(500, B, w, h, step=1.5))
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    e.set_streams(4)
    cap = e.max_keypoints()
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(imgs).to(dev)
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for _ in range(2):  # second pass: workspace reuse, asynchronous entry point
        e.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                               d_n.data_ptr(), wait=False)
    e.synchronize()
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy()
    desc = d_desc.cpu().numpy()
    sf = np.array(e.GetScaleFactors(), np.float32)
    assert (n > 800).all() and (n <= cap).all()
    for f in range(B):
        k = kp[f, :n[f]]
        octv = k[:, 5].view(np.int32)
        assert (np.diff(octv) >= 0).all() and octv.min() >= 0 and octv.max() <= 7   # level-major output order
        assert (k[:, 6].view(np.int32) == -1).all()                                  # class_id
        assert (k[:, 0] >= 19 * 0.99).all() and (k[:, 0] < w).all() and (k[:, 1] >= 19 * 0.99).all() and (k[:, 1] < h).all()
        assert np.array_equal(k[:, 2], np.floor(31 * sf[octv]))                      # size = (int)(31 * scale)
        assert (k[:, 3] >= 0).all() and (k[:, 3] < 360).all()                        # fastAtan2 range
        assert (k[:, 4] >= 7).all() and (k[:, 4] <= 255).all()                       # FAST response >= minThFAST
    single = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    for f in (0, 1, 31, 32, 47, 64, 95):
        ks, ds = single(imgs[f])
        assert len(ks) == n[f]
        assert np.array_equal(ds, desc[f, :n[f]])
        assert np.array_equal(np.stack([ks["x"], ks["y"], ks["angle"], ks["response"]], 1), kp[f, :n[f]][:, [0, 1, 3, 4]])


@pytest.mark.parametrize("shape,params", [((640, 480), (1000, 1.2, 8, 20, 7)), ((1241, 376), (2000, 1.2, 8, 20, 7)),
                                          ((752, 480), (1200, 1.2, 8, 20, 7)), ((333, 217), (500, 1.2, 8, 20, 7)),
                                          ((97, 81), (300, 1.2, 8, 20, 7)), ((200, 64), (200, 1.5, 4, 20, 7))])
def test_fused_and_separate_blur_agree_with_the_oracle(amd, shape, params):
    """The blur runs inside the FAST kernel by default (detection cells + blur-only frame cells with
    reflect-101 staging); the separate k_blur7 launch stays selectable.  Both must give the oracle's blurred
    levels (every byte of every level, including the 19-px frame and levels too small for any FAST cell)
    and identical keypoints / descriptors."""
    w, h = shape
    nf, sf, nl, ini, mn = params
    for seed, img in ((1, synth.render_frame(40, w, h)), (2, synth.adversarial("noise", w, h, seed=9))):
        o = orc.Oracle(nf, sf, nl, ini, mn)
        kr, dr, pyr = o.extract(img, want_pyramid=True)
        levels = o.split_pyramid(pyr, w, h)
        for fused, pyrblur in ((True, True), (False, True), (False, False)):
            e = amd.ORBextractor(nf, sf, nl, ini, mn)
            e.set_fused(fused)
            e.set_pyramid_blur(pyrblur)  # (ignored with the fused FAST + blur kernel)
            kps, desc = e(img)
            for l, ref in enumerate(levels):
                got = e.debug_blurred_level(l)
                assert np.array_equal(orc.gaussian_blur7(ref), got), f"blur level {l} fused={fused} pyrblur={pyrblur} seed={seed}"
                assert np.array_equal(ref, e.pyramid_level(l)), f"pyramid level {l} fused={fused} pyrblur={pyrblur}"
            _kp_equal(kr, kps)
            assert np.array_equal(dr, desc)
    # batch of frames on several streams, strided caller-owned input (odd pitch = byte staging of level 0)
    torch = pytest.importorskip("torch")
    B = 5
    frames = np.stack([synth.render_frame(60 + i, w, h) for i in range(B)])
    dev = torch.device("cuda", 0)
    stride = w + 3
    buf = torch.zeros((B, h, stride), dtype=torch.uint8, device=dev)
    buf[:, :, :w] = torch.from_numpy(frames).to(dev)
    e = amd.ORBextractor(nf, sf, nl, ini, mn)
    e.set_streams(2)
    cap = e.max_keypoints(w, h)
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e.extract_batch_device(buf.data_ptr(), B, w, h, stride, stride * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    o = orc.Oracle(nf, sf, nl, ini, mn)
    for f in range(B):
        kr, dr = o.extract(frames[f])
        n = int(d_n[f].item())
        assert n == len(kr)
        assert np.array_equal(d_kp[f, :n].cpu().numpy().view(np.uint8).reshape(n, 28), kr.view(np.uint8).reshape(-1, 28))
        assert np.array_equal(d_desc[f, :n].cpu().numpy(), dr)


@pytest.mark.parametrize("spec", [1, 2])
def test_blur_spec_variants_match_the_oracle(amd, spec):
    """GaussianBlur arithmetic of OpenCV 2.4 / 3.0-3.3 (taps 18 34 49 55, saturating; spec 2 = SSE2 column pass with
    round-half-even on the first w & ~3 columns) selectable on the extractor: blurred levels, keypoints and
    descriptors equal the oracle built with the same spec; bright images exercise the saturation (257-sum taps)."""
    rng = np.random.default_rng(4)
    for shape in ((640, 480), (333, 217), (70, 66), (5, 9)):
        w, h = shape
        for img in (rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(250, 256, (h, w), dtype=np.uint8),
                    np.full((h, w), 255, np.uint8)):
            assert np.array_equal(orc.gaussian_blur7(img, spec), amd.gaussian_blur7(img, spec=spec)), (shape, spec)
    img = synth.render_frame(12, 640, 480)
    bright = np.clip(img.astype(np.int32) + 120, 0, 255).astype(np.uint8)  # large saturated regions
    for im in (img, bright):
        o = orc.Oracle(1000, 1.2, 8, 20, 7, blur_spec=spec)
        kr, dr, pyr = o.extract(im, want_pyramid=True)
        e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
        e.set_blur_spec(spec)
        e.set_fused(True)  # the fused kernel implements spec 0 only: the extractor must fall back to k_blur7 by itself
        kps, desc = e(im)
        for l, ref in enumerate(o.split_pyramid(pyr, 640, 480)):
            assert np.array_equal(orc.gaussian_blur7(ref, spec), e.debug_blurred_level(l)), f"level {l}"
        _kp_equal(kr, kps)
        assert np.array_equal(dr, desc)
    # and the default spec really is a different function of the image
    d0 = orc.Oracle(1000, 1.2, 8, 20, 7).extract(img)[1]
    ds = orc.Oracle(1000, 1.2, 8, 20, 7, blur_spec=spec).extract(img)[1]
    assert d0.shape == ds.shape and not np.array_equal(d0, ds)


def test_blur_pass_orders_give_the_oracle_bytes(amd):
    """k_blur7 runs its two separable passes horizontal-first (bytes through v_dot4, then row pairs through v_dot2: the
    default since round 4) or vertical-first (rounds 1-3), orbfe_set_blur_pass_order / $ORBFE_BLUR_HFIRST.  Both are exact
    integer sums: every spec, every tile shape case (partial tiles, one-tile images, odd last row pair, images narrower than a
    group) and the fused pyramid + blur form give the oracle's bytes in both orders."""
    rng = np.random.default_rng(11)
    was = amd.set_blur_pass_order(-1)
    try:
        for order in (0, 1):
            assert amd.set_blur_pass_order(order) == order
            for spec in (0, 1, 2):
                for (w, h) in ((640, 480), (1241, 376), (333, 217), (130, 65), (70, 66), (64, 64), (65, 7), (5, 9), (7, 9)):
                    imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8), np.full((h, w), 255, np.uint8)]
                    if spec:
                        imgs.append(rng.integers(250, 256, (h, w), dtype=np.uint8))
                    for img in imgs:
                        assert np.array_equal(orc.gaussian_blur7(img, spec), amd.gaussian_blur7(img, spec=spec)), (order, spec, w, h)
            # the extractor: single frame (separate launches) and a 12-frame batch (fused pyramid + blur kernel per level)
            for (w, h), params in (((1241, 376), (2000, 1.2, 8, 20, 7)), ((333, 217), (500, 1.2, 8, 20, 7))):
                nf, sf, nl, ini, mn = params
                frames = np.stack([synth.render_frame(900 + i, w, h) for i in range(12)])
                o = orc.Oracle(nf, sf, nl, ini, mn)
                e = amd.ORBextractor(nf, sf, nl, ini, mn)
                kps, desc = e(frames[0])
                kr, dr, pyr = o.extract(frames[0], want_pyramid=True)
                for l, ref in enumerate(o.split_pyramid(pyr, w, h)):
                    assert np.array_equal(orc.gaussian_blur7(ref), e.debug_blurred_level(l)), (order, l)
                _kp_equal(kr, kps)
                assert np.array_equal(dr, desc)
                res = e.extract_batch(frames)
                for f in range(len(frames)):
                    kr, dr = o.extract(frames[f])
                    _kp_equal(kr, res[f][0])
                    assert np.array_equal(dr, res[f][1]), (order, f)
    finally:
        amd.set_blur_pass_order(was)


def test_pipelined_host_batch_equals_oracle(amd):
    """orbfe_extract_batch_pipelined: chunks of the host batch move H2D / through the kernels / D2H on separate streams.
    Several chunks + a short tail, 1 and 3 sub-batch streams, pinned and ordinary (page-locked for the call) buffers,
    an odd width with a non-tight row stride."""
    import ctypes as C
    from orb_slam2_annotate_amd import _lib
    B, W, H = 21, 333, 217
    frames = np.stack([synth.render_frame(500 + i, W, H) for i in range(B)])
    o = orc.Oracle(500, 1.2, 8, 20, 7)
    ref = [o.extract(f) for f in frames]
    for streams, chunk in ((1, 8), (3, 5)):
        e = amd.ORBextractor(500, 1.2, 8, 20, 7)
        e.set_streams(streams)
        out = e.extract_batch_pipelined(frames, chunk_frames=chunk)  # through the extractor's pinned buffers
        for (kr, dr), (kg, dg) in zip(ref, out):
            _kp_equal(kr, kg)
            assert np.array_equal(dr, dg)
        # twice in a row on the same handle (slabs and events are reused)
        out = e.extract_batch_pipelined(frames[::-1].copy(), chunk_frames=chunk)
        for (kr, dr), (kg, dg) in zip(ref[::-1], out):
            _kp_equal(kr, kg)
            assert np.array_equal(dr, dg)
    # ordinary numpy memory + a padded row stride straight into the C entry point
    e = amd.ORBextractor(500, 1.2, 8, 20, 7)
    L = _lib.load()
    stride = W + 11
    padded = np.zeros((B, H, stride), np.uint8)
    padded[:, :, :W] = frames
    cap = e.max_keypoints(W, H)
    kps = np.zeros((B, cap), dtype=_lib.KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    n = np.zeros(B, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.orbfe_extract_batch_pipelined(e._h, p(padded), B, W, H, stride, stride * H, p(kps), p(desc), cap, p(n), 4)
    assert rc == 0
    for f, (kr, dr) in enumerate(ref):
        assert n[f] == len(kr)
        _kp_equal(kr, kps[f, :n[f]])
        assert np.array_equal(dr, desc[f, :n[f]])
    # capacity too small is reported, empty batch is fine
    rc = L.orbfe_extract_batch_pipelined(e._h, p(padded), B, W, H, stride, stride * H, p(kps), p(desc), 10, p(n), 4)
    assert rc == -2
    assert L.orbfe_extract_batch_pipelined(e._h, p(padded), 0, W, H, stride, stride * H, p(kps), p(desc), cap, p(n), 4) == 0


@pytest.mark.parametrize("subs", [2, 4, 8])
def test_lane_schedule_equals_oracle_across_back_to_back_calls(amd, subs):
    """The three-lane software pipeline (pyramid | FAST + blur | octree + descriptors on three shared streams):
    several asynchronous calls on DIFFERENT inputs enqueued back to back -- every workspace slice is reused while
    the previous call's tail lane may still read it -- with the batched stereo matcher in between; all results
    must equal the oracle."""
    torch = pytest.importorskip("torch")
    w, h, nf, P = 480, 200, 600, 9
    sets = [[synth.render_stereo(700 + 50 * s + p, w, h, n_shapes=250, max_disp=40) for p in range(P)] for s in range(3)]
    imgs = [np.stack([im for pr in st for im in pr]) for st in sets]
    dev = torch.device("cuda", 0)
    B = 2 * P
    d_img = [torch.from_numpy(x).to(dev) for x in imgs]
    e = amd.ORBextractor(nf, 1.2, 8, 20, 7)
    e.set_streams(subs)
    e.set_schedule(True)
    cap = e.max_keypoints(w, h)
    d_kp = [torch.zeros((B, cap, 7), dtype=torch.float32, device=dev) for _ in range(3)]
    d_desc = [torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev) for _ in range(3)]
    d_n = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(3)]
    d_u = torch.zeros((3, P, cap), dtype=torch.float32, device=dev)
    d_d = torch.zeros((3, P, cap), dtype=torch.float32, device=dev)
    d_ns = torch.zeros((3, P), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    mbf = np.float32(100.0)
    mb = np.float32(mbf / np.float32(350.0))
    for rep in range(2):
        for s in range(3):
            e.extract_batch_device(d_img[s].data_ptr(), B, w, h, w, w * h, d_kp[s].data_ptr(), d_desc[s].data_ptr(), cap,
                                   d_n[s].data_ptr(), wait=False)
            e.stereo_match_batch_device(P, d_kp[s].data_ptr(), d_desc[s].data_ptr(), d_n[s].data_ptr(), cap, float(mbf),
                                        float(mb), d_u[s].data_ptr(), d_d[s].data_ptr(), d_ns[s].data_ptr())
    e.synchronize()
    o = orc.Oracle(nf, 1.2, 8, 20, 7)
    for s in range(3):
        n = d_n[s].cpu().numpy()
        for p in range(P):
            kL, dL, pL = o.extract(imgs[s][2 * p], want_pyramid=True)
            kR, dR, pR = o.extract(imgs[s][2 * p + 1], want_pyramid=True)
            for fi, (kr, dr) in ((2 * p, (kL, dL)), (2 * p + 1, (kR, dR))):
                assert n[fi] == len(kr), (s, fi)
                assert np.array_equal(d_kp[s][fi, :n[fi]].cpu().numpy().view(np.uint8).reshape(-1, 28), kr.view(np.uint8).reshape(-1, 28))
                assert np.array_equal(d_desc[s][fi, :n[fi]].cpu().numpy(), dr)
            u_ref, d_ref = o.stereo(w, h, kL, dL, kR, dR, pL, pR, float(mbf), float(mb))
            assert np.array_equal(d_u[s, p, :len(kL)].cpu().numpy(), u_ref), (s, p)
            assert np.array_equal(d_d[s, p, :len(kL)].cpu().numpy(), d_ref), (s, p)
    # switching the schedule back on the same handle, and the host-batch paths under the lane schedule
    frames = imgs[0][:7]
    ref = [o.extract(f) for f in frames]
    for out in (e.extract_batch(frames), e.extract_batch_pipelined(frames, chunk_frames=3)):
        for (kr, dr), (kg, dg) in zip(ref, out):
            _kp_equal(kr, kg)
            assert np.array_equal(dr, dg)
    e.set_schedule(False)
    for (kr, dr), (kg, dg) in zip(ref, e.extract_batch(frames)):
        _kp_equal(kr, kg)
        assert np.array_equal(dr, dg)


@pytest.mark.parametrize("mode", ["high", "low", "auto"])
def test_fast_threshold_order_modes_are_equivalent(amd, mode):
    """iniThFAST first with per-cell fallback, one attempt at the lower threshold, or the automatic choice: the grid
    stage must emit the reference's candidates either way -- textured frames (few fallbacks), low-contrast frames (most
    cells fall back), frames where nothing passes, reversed thresholds (iniTh < minTh), and a batch whose content
    flips from call to call so that the automatic mode really switches."""
    low = np.clip(synth.render_frame(3, 640, 480).astype(np.int32) // 6 + 100, 0, 255).astype(np.uint8)  # contrast / 6
    y, x = np.mgrid[0:240, 0:320]
    ramp = ((x * 0.3 + y * 0.2) % 256).astype(np.uint8)
    for img, params in ((synth.render_frame(5, 640, 480), (1000, 1.2, 8, 20, 7)), (low, (1000, 1.2, 8, 20, 7)),
                        (ramp, (500, 1.2, 6, 20, 7)), (synth.render_frame(6, 320, 240), (500, 1.2, 8, 7, 20)),
                        (synth.render_frame(7, 320, 240), (500, 1.2, 8, 12, 12))):
        nf, sf, nl, ini, mn = params
        o = orc.Oracle(*params)
        kr, dr, pyr = o.extract(img, want_pyramid=True)
        e = amd.ORBextractor(*params)
        e.set_fast_mode(mode)
        kps, desc = e(img)
        h, w = img.shape
        for l, ref in enumerate(o.split_pyramid(pyr, w, h)):
            xr, yr, rr = orc.grid_candidates(o, ref)
            xg, yg, rg = e.debug_candidates(l)
            assert np.array_equal(xr, xg) and np.array_equal(yr, yg) and np.array_equal(rr, rg), f"candidates level {l}"
        _kp_equal(kr, kps)
        assert np.array_equal(dr, desc)
    # alternating content on one handle, asynchronous device batches (auto mode reads the statistic of finished launches)
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    tex = np.stack([synth.render_frame(20 + i, 640, 480) for i in range(6)])
    flat = np.stack([np.clip(f.astype(np.int32) // 8 + 90, 0, 255).astype(np.uint8) for f in tex])
    e = amd.ORBextractor(1000, 1.2, 8, 20, 7)
    e.set_fast_mode(mode)
    e.set_streams(2)
    cap = e.max_keypoints(640, 480)
    o = orc.Oracle(1000, 1.2, 8, 20, 7)
    bufs = []
    for rep in range(3):
        for imgs in (tex, flat):
            d_img = torch.from_numpy(imgs).to(dev)
            d_kp = torch.zeros((6, cap, 7), dtype=torch.float32, device=dev)
            d_desc = torch.zeros((6, cap, 32), dtype=torch.uint8, device=dev)
            d_n = torch.zeros((6,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            e.extract_batch_device(d_img.data_ptr(), 6, 640, 480, 640, 640 * 480, d_kp.data_ptr(), d_desc.data_ptr(), cap,
                                   d_n.data_ptr(), wait=False)
            bufs.append((imgs, d_img, d_kp, d_desc, d_n))
    e.synchronize()
    refs = {id(tex): [o.extract(f) for f in tex], id(flat): [o.extract(f) for f in flat]}
    for imgs, _, d_kp, d_desc, d_n in bufs:
        for f, (kr, dr) in enumerate(refs[id(imgs)]):
            n = int(d_n[f].item())
            assert n == len(kr)
            assert np.array_equal(d_kp[f, :n].cpu().numpy().view(np.uint8).reshape(-1, 28), kr.view(np.uint8).reshape(-1, 28))
            assert np.array_equal(d_desc[f, :n].cpu().numpy(), dr)


@pytest.mark.parametrize("shape,params", [((640, 480), (1000, 1.2, 8, 20, 7)), ((1241, 376), (2000, 1.2, 8, 20, 7)),
                                          ((752, 480), (1200, 1.2, 8, 20, 7)), ((333, 217), (500, 1.2, 8, 20, 7)),
                                          ((97, 81), (300, 1.2, 8, 20, 7)), ((200, 64), (200, 1.5, 4, 20, 7)),
                                          ((400, 300), (3000, 1.2, 1, 20, 7))])
def test_orient_desc_tile_form_matches_the_oracle(amd, shape, params):
    """k_orient_desc_tiles (one workgroup per 128 x 128 tile of a level, level and blurred level staged in LDS once) gives
    the oracle's keypoints and descriptors: single frames, batches on several streams with a strided level 0, noise
    (every tile crowded; the 1-level / 3000-feature case puts > 64 keypoints in a tile = several list windows), a
    constant image (all tiles empty)."""
    w, h = shape
    nf, sf, nl, ini, mn = params
    o = orc.Oracle(nf, sf, nl, ini, mn)
    for img in (synth.render_frame(41, w, h), synth.adversarial("noise", w, h, seed=5), synth.adversarial("constant", w, h)):
        e = amd.ORBextractor(nf, sf, nl, ini, mn)
        e.set_desc_tiles(True)
        kps, desc = e(img)
        kr, dr = o.extract(img)
        _kp_equal(kr, kps)
        assert np.array_equal(dr, desc)
    torch = pytest.importorskip("torch")
    B = 11
    frames = np.stack([synth.render_frame(70 + i, w, h) for i in range(B)])
    dev = torch.device("cuda", 0)
    stride = w + 5
    buf = torch.zeros((B, h, stride), dtype=torch.uint8, device=dev)
    buf[:, :, :w] = torch.from_numpy(frames).to(dev)
    e = amd.ORBextractor(nf, sf, nl, ini, mn)
    e.set_desc_tiles(True)
    e.set_streams(3)
    cap = e.max_keypoints(w, h)
    d_kp = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_n = torch.zeros((B,), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e.extract_batch_device(buf.data_ptr(), B, w, h, stride, stride * h, d_kp.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr())
    n = d_n.cpu().numpy()
    for f in range(B):
        kr, dr = o.extract(frames[f])
        assert n[f] == len(kr), f
        assert np.array_equal(d_kp[f, :n[f]].cpu().numpy().view(np.uint8).reshape(-1, 28), kr.view(np.uint8).reshape(-1, 28)), f
        assert np.array_equal(d_desc[f, :n[f]].cpu().numpy(), dr), f


def test_pipelined_host_batch_keeps_the_last_chunk_only(amd):
    """orbfe_extract_batch routes large host batches through the chunked path, which keeps the pyramid /
    blurred levels / candidates of its LAST chunk.  Frame-indexed accessors count in the caller's batch: a retained frame
    answers with ITS pyramid (== oracle), an earlier one fails loudly instead of returning another frame's data."""
    w, h, B, chunk = 160, 120, 40, 16  # chunks [0,16) [16,32) [32,40): the tail chunk is shorter
    frames = np.stack([synth.render_frame(500 + i, w, h) for i in range(B)])
    e = amd.ORBextractor(300, 1.2, 8, 20, 7)
    res = e.extract_batch_pipelined(frames, chunk_frames=chunk)
    o = orc.Oracle(300, 1.2, 8, 20, 7)
    for f in (0, 17, 39):
        kr, dr = o.extract(frames[f])
        _kp_equal(kr, res[f][0])
        assert np.array_equal(dr, res[f][1])
    for f in (32, 39):  # retained
        _, _, pyr = o.extract(frames[f], want_pyramid=True)
        levels = o.split_pyramid(pyr, w, h)
        for l in (0, 3, 7):
            assert np.array_equal(e.pyramid_level(l, frame=f), levels[l]), (f, l)
            assert np.array_equal(e.debug_blurred_level(l, frame=f), orc.gaussian_blur7(levels[l])), (f, l)
    for f in (0, 16, 31):  # not retained: an error, never frame f + 32's pyramid
        with pytest.raises(amd.OrbfeError, match="not retained"):
            e.pyramid_level(0, frame=f)
    with pytest.raises(amd.OrbfeError):
        e.pyramid_level(0, frame=B)
    # the plain host entry point with a pageable batch large enough to be routed (>= 512 frames): same contract
    big = np.stack([frames[i % B] for i in range(520)])
    res = e.extract_batch(big)
    kr, dr = o.extract(big[519])
    _kp_equal(kr, res[519][0])
    _, _, pyr = o.extract(big[519], want_pyramid=True)
    assert np.array_equal(e.pyramid_level(2, frame=519), o.split_pyramid(pyr, w, h)[2])
    with pytest.raises(amd.OrbfeError, match="not retained"):
        e.pyramid_level(0, frame=0)



@pytest.mark.parametrize("shape,cfg", [((1241, 376), (2000, 1.2, 8, 20, 7)), ((640, 480), (1000, 1.2, 8, 20, 7)), ((333, 217), (500, 1.2, 8, 20, 7)),
                                       ((752, 480), (1200, 1.1, 10, 20, 7)), ((640, 480), (800, 1.4, 6, 20, 7))])
def test_pyramid_in_one_launch_matches_the_level_by_level_form(amd, shape, cfg):
    """orbfe_extractor_set_pyramid_chain(1): every tile of every level recomputed from level 0 in LDS (k_pyramid_chain) --
    the pyramid levels, keypoints and descriptors of single-frame calls equal the oracle's (scale 1.4 / deep pyramids whose
    rectangles do not fit LDS fall back to the launches: same results either way)."""
    w, h = shape
    img = synth.render_frame(31, w, h)
    e = amd.ORBextractor(*cfg)
    e.set_pyramid_chain(True)
    o = orc.Oracle(*cfg)
    kr, dr, pyr = o.extract(img, want_pyramid=True)
    for _ in range(2):
        k, d = e(img)
        assert np.array_equal(k, kr) and np.array_equal(d, dr)
    for l, ref in enumerate(o.split_pyramid(pyr, w, h)):
        assert np.array_equal(e.pyramid_level(l), ref), l
    (k2, d2), (k3, d3) = e.extract_batch(np.stack([img, img[::-1].copy()]))  # two frames: still the one-launch form
    assert np.array_equal(k2, kr) and np.array_equal(d2, dr)
    kf, df, _ = o.extract(np.ascontiguousarray(img[::-1]), want_pyramid=True)
    assert np.array_equal(k3, kf) and np.array_equal(d3, df)
