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
