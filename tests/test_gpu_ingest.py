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
