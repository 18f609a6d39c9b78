"""Host mirrors of the small data-parallel steps next to the hot path:
colour->gray conversion in front of the extractor (src/Tracking.cc:176-262) and
MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:269-333)."""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import check, ptr


def cvtColorToGray(image: np.ndarray, rgb: bool = True, device: int = 0) -> np.ndarray:
    """cv::cvtColor(im, im, CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY) for HxWx3|4 uint8."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w, c = image.shape
    out = np.zeros((h, w), dtype=np.uint8)
    check(_lib.load().orbfe_cvt_gray(device, ptr(image), w, h, w * c, c, int(bool(rgb)), ptr(out), w))
    return out


def ComputeDistinctiveDescriptors(descriptor_lists, device: int = 0):
    """For each map point (a [n_i, 32] uint8 array of its observations' descriptors) the index of the
    descriptor MapPoint::ComputeDistinctiveDescriptors would keep; -1 for an empty list."""
    lens = [len(d) for d in descriptor_lists]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    flat = (np.concatenate([np.asarray(d, dtype=np.uint8).reshape(-1, 32) for d in descriptor_lists])
            if sum(lens) else np.zeros((0, 32), np.uint8))
    flat = np.ascontiguousarray(flat)
    best = np.full(max(len(lens), 1), -1, dtype=np.int32)
    check(_lib.load().orbfe_distinctive_descriptors(device, ptr(flat), ptr(offsets), len(lens), ptr(best)))
    return best[: len(lens)]
