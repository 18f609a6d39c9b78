"""CPU tests pinning the oracle to every fact derivable from the reference source alone
(SURVEY.md 8(c) "known-answer tests"): the reference ships no tests or golden vectors, so
these constants are the only reference-side pins (the OpenCV boundary stays "parity unpinned")."""
import hashlib

import numpy as np
import pytest

import oracle_lib as orc


def test_pattern_table():
    # src/ORBextractor.cc:155-413: first pair 8,-3, 9,5 ; last -1,-6, 0,-11
    p = np.ctypeslib.as_array(orc.lib().orc_pattern(), shape=(1024,)).astype(np.int8)
    assert list(p[:4]) == [8, -3, 9, 5]
    assert list(p[-4:]) == [-1, -6, 0, -11]
    assert list(p[4:8]) == [4, 2, 7, -12]
    assert np.abs(p).max() <= 13 and len(p) == 1024
    assert hashlib.sha256(p.tobytes()).hexdigest() == \
        "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    # every rotated sample stays inside the 37x37 neighbourhood guaranteed by EDGE_THRESHOLD=19
    assert (np.hypot(p[0::2].astype(float), p[1::2].astype(float)) < 18.5).all()


def test_product_pattern_equals_oracle_pattern():
    from pathlib import Path
    import re
    root = Path(orc.ROOT)
    a = re.findall(r"-?\d+", (root / "oracle/orb_pattern_data.h").read_text().split("{", 1)[1])
    b = re.findall(r"-?\d+", (root / "orb_slam2_annotate_amd/csrc/orb_pattern.inc").read_text().split("{", 1)[1])
    assert a == b and len(a) == 1024


def test_umax_and_quotas():
    # src/ORBextractor.cc:448-485, values derived in SURVEY.md 8
    o = orc.Oracle(1000, 1.2, 8, 20, 7)
    assert o.umax() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert sum(2 * u + 1 for u in o.umax()[1:]) * 2 // 2 * 1 + 0 >= 0
    assert (2 * 15 + 1) + 2 * sum(2 * u + 1 for u in o.umax()[1:]) == 749
    assert o.features_per_level() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert orc.Oracle(2000, 1.2, 8, 20, 7).features_per_level() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert orc.Oracle(1200, 1.2, 8, 20, 7).features_per_level() == [261, 217, 181, 151, 126, 105, 87, 72]


def test_scale_tables():
    o = orc.Oracle(1000, 1.2, 8, 20, 7)
    s = o.scale_factors()
    assert s[0] == 1.0 and s[1] == np.float32(1.2)
    # mvScaleFactor[i] = float(double(prev) * double(1.2f))  (double member, include/ORBextractor.h:100)
    exp = [np.float32(1.0)]
    for _ in range(7):
        exp.append(np.float32(np.float64(exp[-1]) * np.float64(np.float32(1.2))))
    assert np.array_equal(s, np.array(exp, dtype=np.float32))
    assert np.array_equal(o.level_sigma2(), s * s)
    assert np.array_equal(o.inv_scale_factors(), np.float32(1.0) / s)


@pytest.mark.parametrize("wh,sizes", [
    ((640, 480), [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    ((752, 480), [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
    ((1241, 376), [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)]),
])
def test_level_sizes(wh, sizes):
    assert orc.Oracle().level_sizes(*wh) == sizes  # SURVEY.md 8 size table


def test_cvround_half_even_and_atan2():
    L = orc.lib()
    assert [L.orc_cvround(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    assert orc.fast_atan2(0.0, 0.0) == 0.0
    for y, x, deg in [(0, 1, 0), (1, 0, 90), (0, -1, 180), (-1, 0, 270), (1, 1, 45), (-1, -1, 225)]:
        assert abs(orc.fast_atan2(float(y), float(x)) - deg) < 0.02
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, size=2)
        a = orc.fast_atan2(float(y), float(x))
        ref = np.degrees(np.arctan2(y, x)) % 360
        assert 0 <= a <= 360 and min(abs(a - ref), 360 - abs(a - ref)) < 0.02


def test_sincos_is_correctly_rounded_and_close_to_glibc():
    """oracle/sincos_audit.c walks float32 angles in [0,360): the project's sincos must equal the
    correctly rounded (float)cos((double)x) everywhere; against this container's glibc cosf/sinf
    the VALUE may differ for ~0.13 % of angles but the rBRIEF sampling GEOMETRY almost never
    (exhaustive run, stride 1: 96 of 1,135,869,952 angles; recorded in DESIGN.md)."""
    import json
    import subprocess
    orc.build_oracle()
    subprocess.run(["make", "-C", str(orc.ORACLE_DIR), "sincos_audit"], check=True, capture_output=True)
    out = subprocess.run([str(orc.ORACLE_DIR / "sincos_audit"), "499"], check=True, capture_output=True, text=True)
    r = json.loads(out.stdout)
    assert r["angles"] > 2_000_000
    assert r["differs_from_double_rounded"] == 0
    assert r["value_differs_from_glibc"] < 0.005 * r["angles"]
    assert r["geometry_differs_from_glibc"] <= 5


def test_descriptor_distance_and_three_maxima():
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, size=(500, 32), dtype=np.uint8)
    b = rng.integers(0, 256, size=(500, 32), dtype=np.uint8)
    for i in range(500):
        assert orc.descriptor_distance(a[i], b[i]) == int(np.unpackbits(a[i] ^ b[i]).sum())
    assert orc.descriptor_distance(a[0], a[0]) == 0
    assert orc.descriptor_distance(a[0], ~a[0]) == 256
    # ComputeThreeMaxima, src/ORBmatcher.cc:1777-1821
    h = [0] * 30
    assert orc.three_maxima(h) == (-1, -1, -1)
    h[3], h[7], h[9] = 100, 50, 20
    assert orc.three_maxima(h) == (3, 7, 9)
    h[9] = 9  # third < 10% of first -> dropped
    assert orc.three_maxima(h) == (3, 7, -1)
    h[7] = 9  # second < 10% -> second and third dropped
    assert orc.three_maxima(h) == (3, -1, -1)
    h = [0] * 30
    h[2] = h[5] = h[8] = 10  # ties: strictly-greater updates keep the first seen in front
    assert orc.three_maxima(h) == (2, 5, 8)


def test_descriptor_bit_order_and_gaussian_kernel():
    # bit k of byte i = test 8i+k (src/ORBextractor.cc:128-149): a vertical step edge, angle 0:
    # pair 0 = (8,-3)->(9,5): both right of a step at x<=0 -> t0<t1 false when image is flat right
    img = np.zeros((64, 64), dtype=np.uint8)
    img[:, 32:] = 200
    import ctypes as C
    d = np.zeros(32, dtype=np.uint8)
    orc.lib().orc_descriptor(orc._p(img), 64, 32, 32, C.c_float(0.0), orc._p(d))
    pat = np.ctypeslib.as_array(orc.lib().orc_pattern(), shape=(256, 4)).astype(int)
    for k in range(256):
        x0, y0, x1, y1 = pat[k]
        t0 = img[32 + y0, 32 + x0]
        t1 = img[32 + y1, 32 + x1]
        assert ((d[k // 8] >> (k % 8)) & 1) == int(t0 < t1)
    # blur kernel [18,34,48,56,48,34,18]/256 with reflect-101, round half up
    imp = np.zeros((15, 15), dtype=np.uint8)
    imp[7, 7] = 255
    b = orc.gaussian_blur7(imp)
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    exp = (np.outer(k, k) * 255 + (1 << 15)) >> 16
    assert np.array_equal(b[4:11, 4:11], exp)
    assert np.array_equal(orc.gaussian_blur7(np.full((20, 30), 77, np.uint8)), np.full((20, 30), 77, np.uint8))


def test_keypoint_bounds_and_ordering():
    from orb_slam2_annotate_amd import synth
    o = orc.Oracle(1000, 1.2, 8, 20, 7)
    img = synth.render_frame(5)
    kps, desc = o.extract(img)
    assert len(kps) == len(desc) and 900 < len(kps) <= 1000 + 3 * 8
    assert (np.diff(kps["octave"]) >= 0).all()  # level-major output order (:1160-1195)
    sizes = o.level_sizes(640, 480)
    sf = o.scale_factors()
    for l, (w, h) in enumerate(sizes):
        m = kps["octave"] == l
        x = kps["x"][m] / sf[l]
        y = kps["y"][m] / sf[l]
        assert (np.rint(x) >= 19).all() and (np.rint(x) <= w - 20).all()
        assert (np.rint(y) >= 19).all() and (np.rint(y) <= h - 20).all()
        assert (kps["size"][m] == np.float32(int(31 * sf[l]))).all()
    assert (kps["class_id"] == -1).all() and (kps["angle"] >= 0).all() and (kps["angle"] <= 360).all()
    assert (kps["response"] >= 7).all()


def test_bow_two_min_and_triangulation_last_wins():
    # SearchByBoW: best = first index of the minimum, second = second smallest of the multiset (:251-260)
    q = np.zeros((1, 32), np.uint8)
    c = np.zeros((4, 32), np.uint8)
    c[0, 0] = 0b111        # dist 3
    c[1, 0] = 0b1          # dist 1  <- best (first of the two minima)
    c[2, 0] = 0b10         # dist 1
    c[3, 0] = 0b1111       # dist 4
    fv1 = orc.FeatVec(np.zeros(1, np.uint32))
    fv2 = orc.FeatVec(np.zeros(4, np.uint32))
    n, m = orc.search_by_bow(q, np.ones(1, np.uint8), np.zeros(1, np.float32), fv1, c, np.zeros(4, np.float32), fv2,
                             0.7, False)
    assert n == 0  # ratio test: 1 < 0.7*1 fails (second best equals best)
    c[2, 0] = 0b11         # dist 2 -> 1 < 0.7*2
    n, m = orc.search_by_bow(q, np.ones(1, np.uint8), np.zeros(1, np.float32), fv1, c, np.zeros(4, np.float32), fv2,
                             0.7, False)
    assert n == 1 and list(m) == [-1, 0, -1, -1]
    # SearchForTriangulation: dist <= bestDist, so the LAST equal-distance candidate wins (:840)
    c2 = np.zeros((3, 32), np.uint8)
    c2[:, 0] = [0b1, 0b10, 0b100]  # all distance 1
    F = np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32)  # epipolar lines: y2 = y1
    z1, z3 = np.zeros(1, np.float32), np.zeros(3, np.float32)
    n, m = orc.search_for_triangulation(q, np.zeros(1, np.uint8), z1 + 10, z1 + 20, z1, np.ones(1, np.uint8), fv1,
                                        c2, np.zeros(3, np.uint8), z3 + 30, z3 + 20, z3, np.zeros(3, np.int32),
                                        np.ones(3, np.uint8), orc.FeatVec(np.zeros(3, np.uint32)), F, 0.0, 0.0,
                                        np.ones(8, np.float32), np.ones(8, np.float32), False, False)
    assert n == 1 and m[0] == 2


def test_blur_spec_variants_known_answers():
    """The three GaussianBlur arithmetic variants (OpenCV is unpinned in the reference):
    facts that follow from the published definitions alone."""
    # getGaussianKernel(7, 2): exp(-x^2/8) normalised; times 256 and rounded tap by tap -> 18 34 49 55 (sum 257)
    g = np.exp(-(np.arange(-3, 4) ** 2) / 8.0)
    g /= g.sum()
    assert list(np.rint(g * 256).astype(int)) == [18, 34, 49, 55, 49, 34, 18]
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    b0, b1, b2 = (orc.gaussian_blur7(img, s) for s in (0, 1, 2))
    # a constant image: spec 0 (taps sum to 256) reproduces it; the 257-sum taps give round(c * 257^2 / 65536), saturated
    for c in (0, 1, 100, 200, 254, 255):
        k = np.full((20, 24), c, np.uint8)
        assert (orc.gaussian_blur7(k, 0) == c).all()
        want = min(255, (c * 257 * 257 + 32768) >> 16)
        assert (orc.gaussian_blur7(k, 1) == want).all()
        assert (orc.gaussian_blur7(k, 2) == want).all()  # no exact .5 ties for these values
    # spec 2 differs from spec 1 only where the 16.16 sum is an exact tie (x mod 2^16 == 2^15) with an even floor, and
    # never on the last width % 4 columns; spec 0 and spec 1 differ by at most 1 grey level on an image below saturation
    assert np.array_equal(b1[:, 53 - 53 % 4:], b2[:, 53 - 53 % 4:])
    assert np.abs(b1.astype(int) - b2.astype(int)).max() <= 1
    assert np.abs(b0.astype(int) - b1.astype(int)).max() <= 2
    # an engineered exact tie: a 7x7 neighbourhood whose weighted sum is k * 65536 + 32768
    # (all pixels p: sum = p * 257^2 = p * 66049 -> p = 64: 4227136 = 64 * 65536 + 32832, not a tie; so search one)
    K = np.array([18, 34, 49, 55, 49, 34, 18])
    W = np.outer(K, K)
    patch = np.random.default_rng(1).integers(0, 200, (7, 7))
    patch[0, 0] = patch[3, 3] = patch[0, 1] = 0
    base = int((W * patch).sum())
    a, b, c = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    tot = base + W[0, 0] * a + W[3, 3] * b + W[0, 1] * c
    hit = np.argwhere((tot % 65536 == 32768) & ((tot >> 16) % 2 == 0))  # exact tie with an EVEN floor: the two roundings differ
    assert len(hit) > 0
    patch[0, 0], patch[3, 3], patch[0, 1] = hit[0]
    big = np.zeros((7, 8), np.uint8)  # width 8: column 3 is a SIMD column of spec 2
    big[:, :7] = patch
    x = int((W * patch).sum())
    assert x % 65536 == 32768
    assert orc.gaussian_blur7(big, 1)[3, 3] == (x >> 16) + 1   # round half up
    assert orc.gaussian_blur7(big, 2)[3, 3] == (x >> 16)       # round half to even (floor is even)
    big7 = np.ascontiguousarray(big[:, :7])                    # width 7: column 3 is in the scalar tail (7 - 7 % 4 = 4 > 3: still SIMD)
    assert orc.gaussian_blur7(big7, 2)[3, 3] == (x >> 16)
    # width 3 has no SIMD columns at all (3 - 3 % 4 = 0): spec 2 must equal spec 1 everywhere
    tiny = np.random.default_rng(2).integers(0, 256, (9, 3), dtype=np.uint8)
    assert np.array_equal(orc.gaussian_blur7(tiny, 1), orc.gaussian_blur7(tiny, 2))
