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


class Rectifier:
    """cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) with the two CV_32F maps of
    cv::initUndistortRectifyMap (Examples/Stereo/stereo_euroc.cc:97-98, 136-137).  The maps are
    handed over once; `__call__` rectifies one host image, `batch_device` a device-resident batch."""

    def __init__(self, map_x: np.ndarray, map_y: np.ndarray, device: int = 0):
        import ctypes as C
        mx = np.ascontiguousarray(map_x, dtype=np.float32)
        my = np.ascontiguousarray(map_y, dtype=np.float32)
        if mx.shape != my.shape or mx.ndim != 2:
            raise ValueError("map_x / map_y must be two HxW float32 arrays")
        self.height, self.width = mx.shape
        self._h = C.c_void_p()
        check(_lib.load().orbfe_rectifier_create(device, ptr(mx), ptr(my), self.width, self.height, self.width,
                                                 C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().orbfe_rectifier_destroy(self._h)
            self._h = None

    __del__ = close

    def __call__(self, image: np.ndarray) -> np.ndarray:
        if image.ndim != 2 or image.dtype != np.uint8:
            raise ValueError("remap: 8-bit single-channel image expected")
        h, w = image.shape
        stride = image.strides[0] if image.strides[1] == 1 else None
        if stride is None:
            image = np.ascontiguousarray(image)
            stride = w
        out = np.zeros((self.height, self.width), dtype=np.uint8)
        check(_lib.load().orbfe_remap(self._h, ptr(image), w, h, stride, ptr(out), self.width))
        return out

    def batch_device(self, d_src: int, n_frames: int, src_width: int, src_height: int, src_stride: int,
                     src_frame_stride: int, d_dst: int, dst_stride: int, dst_frame_stride: int):
        check(_lib.load().orbfe_remap_batch_device(self._h, d_src, n_frames, src_width, src_height, src_stride,
                                                   src_frame_stride, d_dst, dst_stride, dst_frame_stride))
