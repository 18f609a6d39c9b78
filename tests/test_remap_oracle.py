"""CPU known-answer tests of the oracle's cv::remap restatement (oracle/orb_oracle.c,
orc_remap_linear): properties that follow from the published algorithm and do not need OpenCV."""
import numpy as np

import oracle_lib as O


def _img(seed, h=60, w=80):
    return np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)


def test_identity_map_is_a_copy():
    im = _img(0)
    x, y = np.meshgrid(np.arange(80, dtype=np.float32), np.arange(60, dtype=np.float32))
    assert np.array_equal(O.remap_linear(im, x, y), im)


def test_integer_shift_and_zero_border():
    im = _img(1)
    x, y = np.meshgrid(np.arange(80, dtype=np.float32), np.arange(60, dtype=np.float32))
    out = O.remap_linear(im, x + 7, y - 5)
    want = np.zeros_like(im)
    want[5:, :73] = im[:55, 7:]
    assert np.array_equal(out, want)


def test_half_pixel_is_the_rounded_mean_of_four_taps_and_border_taps_read_zero():
    im = _img(2)
    x, y = np.meshgrid(np.arange(80, dtype=np.float32), np.arange(60, dtype=np.float32))
    out = O.remap_linear(im, x + 0.5, y + 0.5)
    p = np.pad(im.astype(np.int64), ((0, 1), (0, 1)))  # taps beyond the last row / column are 0
    want = (p[:-1, :-1] + p[:-1, 1:] + p[1:, :-1] + p[1:, 1:]) * 8192 + (1 << 14) >> 15
    assert np.array_equal(out, want.astype(np.uint8))
    # left / top border: sx = -1 keeps only the taps at x = 0
    out = O.remap_linear(im, x - 0.5, y)
    want = (np.pad(im.astype(np.int64), ((0, 0), (1, 0)))[:, :-1] + im) * 16384 + (1 << 14) >> 15
    assert np.array_equal(out, want.astype(np.uint8))


def test_quantisation_to_one_32nd_pixel_and_half_even_rounding():
    im = np.zeros((4, 4), np.uint8)
    im[1, 1], im[1, 2] = 0, 255
    one = lambda fx: O.remap_linear(im, np.full((1, 1), fx, np.float32), np.full((1, 1), 1.0, np.float32))[0, 0]
    assert one(1 + 3 / 32) == (255 * 3 * 32 * 32 + (1 << 14)) >> 15
    # 1 + 2.5/32 rounds to phase 2 (half to even), 1 + 3.5/32 to phase 4
    assert one(1 + 2.5 / 32) == (255 * 2 * 1024 + (1 << 14)) >> 15
    assert one(1 + 3.5 / 32) == (255 * 4 * 1024 + (1 << 14)) >> 15


def test_far_outside_nan_and_huge_map_values_give_zero():
    im = np.full((10, 10), 200, np.uint8)
    mx = np.array([[np.nan, np.inf, -np.inf, 1e12, -1e12, -1.0, 10.0, 5.0]], np.float32)
    my = np.array([[5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0, 5.0]], np.float32)
    out = O.remap_linear(im, mx, my)
    assert out[0, :5].tolist() == [0] * 5
    assert out[0, 5] == 0          # sx = -1, fx = 0: both weights on the outside tap column... (w1 = 0)
    assert out[0, 6] == 0          # sx = 10 >= width
    assert out[0, 7] == 200
