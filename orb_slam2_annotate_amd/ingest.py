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


def initUndistortRectifyMap(K, D, R, P, size, device: int = 0):
    """cv::initUndistortRectifyMap(K, D, R, P[:3, :3], (cols, rows), CV_32F) -> (map_x, map_y) as
    Examples/Stereo/stereo_euroc.cc:97-98 builds the maps cv::remap takes.  K, R, P: 3x3 (P may be 3x4; R / P may be None);
    D: 0, 4, 5 or 8 coefficients; size = (cols, rows)."""
    K = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(3, 3))
    Dv = np.ascontiguousarray(np.asarray(D if D is not None else [], dtype=np.float64).reshape(-1))
    Rm = None if R is None else np.ascontiguousarray(np.asarray(R, dtype=np.float64).reshape(3, 3))
    Pm = None if P is None else np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(3, -1)[:, :3])
    cols, rows = int(size[0]), int(size[1])
    mx = np.zeros((rows, cols), dtype=np.float32)
    my = np.zeros((rows, cols), dtype=np.float32)
    check(_lib.load().orbfe_init_undistort_rectify_map(device, ptr(K), ptr(Dv) if len(Dv) else None, len(Dv), ptr(Rm), ptr(Pm),
                                                       cols, rows, ptr(mx), ptr(my)))
    return mx, my


def _cam(K, dist):
    K = np.asarray(K, dtype=np.float32)
    K4 = np.ascontiguousarray([K[0, 0], K[1, 1], K[0, 2], K[1, 2]] if K.ndim == 2 else K, dtype=np.float32)
    d = np.ascontiguousarray(np.asarray(dist, dtype=np.float32).reshape(-1))
    return K4, d


def undistortPoints(xy, K, dist, device: int = 0) -> np.ndarray:
    """cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) (src/Frame.cc:463): xy [n,2] float32;
    K a 3x3 matrix or (fx, fy, cx, cy); dist = k1,k2,p1,p2[,k3]."""
    pts = np.ascontiguousarray(xy, dtype=np.float32).reshape(-1, 2)
    K4, d = _cam(K, dist)
    out = np.zeros_like(pts)
    check(_lib.load().orbfe_undistort_points(device, ptr(pts), len(pts), ptr(K4), ptr(d), len(d), ptr(out)))
    return out


def UndistortKeyPoints(keypoints, K, dist, device: int = 0):
    """Frame::UndistortKeyPoints (src/Frame.cc:443-475): mvKeys records -> mvKeysUn records."""
    K4, d = _cam(K, dist)
    kps = np.array(keypoints, copy=True)
    if len(d) == 0 or d[0] == 0.0 or len(kps) == 0:
        return kps
    un = undistortPoints(np.stack([kps["x"], kps["y"]], axis=1), K4, d, device)
    kps["x"], kps["y"] = un[:, 0], un[:, 1]
    return kps


def ComputeImageBounds(cols: int, rows: int, K, dist, device: int = 0):
    """Frame::ComputeImageBounds (src/Frame.cc:481-510) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    K4, d = _cam(K, dist)
    b = np.zeros(4, dtype=np.float32)
    check(_lib.load().orbfe_compute_image_bounds(device, cols, rows, ptr(K4), ptr(d), len(d), ptr(b)))
    return tuple(float(v) for v in b)


def ComputeStereoFromRGBD(keypoints, keypoints_un, depth_image, mbf: float, device: int = 0):
    """Frame::ComputeStereoFromRGBD (src/Frame.cc:689-713) -> (mvuRight, mvDepth)."""
    dimg = np.ascontiguousarray(depth_image, dtype=np.float32)
    kx = np.ascontiguousarray(keypoints["x"], dtype=np.float32)
    ky = np.ascontiguousarray(keypoints["y"], dtype=np.float32)
    kux = np.ascontiguousarray(keypoints_un["x"], dtype=np.float32)
    n = len(kx)
    ur, dp = np.full(max(n, 1), -1, np.float32), np.full(max(n, 1), -1, np.float32)
    check(_lib.load().orbfe_stereo_from_rgbd(device, ptr(kx), ptr(ky), ptr(kux), n, ptr(dimg), dimg.shape[1],
                                             dimg.shape[0], dimg.shape[1], float(mbf), ptr(ur), ptr(dp)))
    return ur[:n], dp[:n]
