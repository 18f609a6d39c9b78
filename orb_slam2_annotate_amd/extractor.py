"""Python mirror of ``ORB_SLAM2::ORBextractor`` (reference: include/ORBextractor.h:46-114)
on top of the C-ABI of liborbfe.so.  Same constructor arguments, same getters, ``__call__``
in place of ``operator()``, ``mvImagePyramid`` as a lazily copied property."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, check, ptr


class ORBextractor:
    HARRIS_SCORE = 0  # include/ORBextractor.h:50 (unused by the reference as well)
    FAST_SCORE = 1

    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int,
                 device: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.orbfe_extractor_create(int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST),
                                             int(minThFAST), int(device), C.byref(h)))
        self._h = h
        self.device = device
        self.nfeatures = int(nfeatures)
        self._last_shape = None

    def __del__(self):
        if getattr(self, "_pin_ptrs", None):
            self._free_pinned()
        h = getattr(self, "_h", None)
        if h:
            self._L.orbfe_extractor_destroy(h)
            self._h = None

    # ---- getters, include/ORBextractor.h:66-87 ----
    def GetLevels(self) -> int:
        return self._L.orbfe_extractor_get_levels(self._h)

    def GetScaleFactor(self) -> float:
        return float(self._L.orbfe_extractor_get_scale_factor(self._h))

    def _vec(self, name, dtype=np.float32, n=None):
        out = np.zeros(n or self.GetLevels(), dtype=dtype)
        check(getattr(self._L, name)(self._h, ptr(out)))
        return out

    def GetScaleFactors(self):
        return self._vec("orbfe_extractor_get_scale_factors")

    def GetInverseScaleFactors(self):
        return self._vec("orbfe_extractor_get_inverse_scale_factors")

    def GetScaleSigmaSquares(self):
        return self._vec("orbfe_extractor_get_scale_sigma_squares")

    def GetInverseScaleSigmaSquares(self):
        return self._vec("orbfe_extractor_get_inverse_scale_sigma_squares")

    def features_per_level(self):
        return self._vec("orbfe_extractor_get_features_per_level", np.int32)

    def umax(self):
        return self._vec("orbfe_extractor_get_umax", np.int32, 16)

    def max_keypoints(self, width: int = 0, height: int = 0) -> int:
        """Output bound; with an image size the exact bound for that size (very wide, flat images start the
        octree with many roots and can exceed the size-independent figure)."""
        if width > 0 and height > 0:
            return max(self._L.orbfe_extractor_max_keypoints_for(self._h, int(width), int(height)),
                       self._L.orbfe_extractor_max_keypoints(self._h))
        return self._L.orbfe_extractor_max_keypoints(self._h)

    def level_size(self, width, height, level):
        w, h = C.c_int(), C.c_int()
        check(self._L.orbfe_extractor_level_size(self._h, width, height, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    # ---- operator(), src/ORBextractor.cc:1119-1197 ----
    def __call__(self, image: np.ndarray, mask=None, capacity: int | None = None):
        """Returns (keypoints[KP_DTYPE], descriptors[N,32] uint8).  `mask` is ignored, as in
        the reference.  An empty image returns empty outputs silently."""
        if image is None or image.size == 0:
            return np.zeros(0, dtype=KP_DTYPE), np.zeros((0, 32), dtype=np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise AssertionError("image.type() == CV_8UC1")  # the reference asserts (:1126)
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        H, W = image.shape
        cap = capacity or self.max_keypoints(W, H)
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        check(self._L.orbfe_extract(self._h, ptr(image), W, H, image.strides[0], ptr(kps), ptr(desc), cap,
                                    C.byref(n)))
        self._last_shape = (1, H, W)
        return kps[: n.value].copy(), desc[: n.value].copy()

    def extract_stereo_frame(self, left: np.ndarray, right: np.ndarray, mbf: float, mb: float, capacity: int | None = None):
        """The stereo Frame constructor's front end in one call (src/Frame.cc:78-96): ExtractORB on both eyes +
        ComputeStereoMatches -> (keysLeft, descLeft, keysRight, descRight, mvuRight, mvDepth).  The handle then holds the
        pair as frames 0 / 1 (ResidentFrame(view, fv, extractor=self, frame=0))."""
        if left.dtype != np.uint8 or left.ndim != 2 or right.dtype != np.uint8 or right.shape != left.shape:
            raise AssertionError("image.type() == CV_8UC1, both eyes of one size")
        left, right = np.ascontiguousarray(left), np.ascontiguousarray(right)
        H, W = left.shape
        cap = capacity or self.max_keypoints(W, H)
        kl, kr = np.zeros(cap, dtype=KP_DTYPE), np.zeros(cap, dtype=KP_DTYPE)
        dl, dr = np.zeros((cap, 32), dtype=np.uint8), np.zeros((cap, 32), dtype=np.uint8)
        u, d = np.full(cap, -1, dtype=np.float32), np.full(cap, -1, dtype=np.float32)
        nl, nr = C.c_int(0), C.c_int(0)
        check(self._L.orbfe_extract_stereo_frame(self._h, ptr(left), ptr(right), W, H, W, ptr(kl), ptr(dl), C.byref(nl), ptr(kr),
                                                 ptr(dr), C.byref(nr), cap, float(mbf), float(mb), ptr(u), ptr(d)))
        self._last_shape = (2, H, W)
        a, b = nl.value, nr.value
        return kl[:a].copy(), dl[:a].copy(), kr[:b].copy(), dr[:b].copy(), u[:a].copy(), d[:a].copy()

    def extract_batch(self, images: np.ndarray, capacity: int | None = None):
        """images: [B,H,W] uint8 (host).  Returns list of (keypoints, descriptors) per frame."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        B, H, W = images.shape
        cap = capacity or self.max_keypoints(W, H)
        kps = np.zeros((B, cap), dtype=KP_DTYPE)
        desc = np.zeros((B, cap, 32), dtype=np.uint8)
        n = np.zeros(B, dtype=np.int32)
        check(self._L.orbfe_extract_batch(self._h, ptr(images), B, W, H, W, W * H, ptr(kps), ptr(desc), cap,
                                          ptr(n)))
        self._last_shape = (B, H, W)
        return [(kps[f, : n[f]].copy(), desc[f, : n[f]].copy()) for f in range(B)]

    # ---- end to end: host images in, host keypoints / descriptors out, copies overlapped with the kernels ----
    def pinned_buffers(self, n_frames: int, height: int, width: int, capacity: int | None = None):
        """(images[B,H,W] u8, keypoints[B,cap] KP_DTYPE, descriptors[B,cap,32] u8, n[B] i32) as numpy views of
        PINNED host memory owned by this extractor (orbfe_host_alloc); fill `images`, call extract_pinned()."""
        cap = capacity or self.max_keypoints(width, height)
        key = (n_frames, height, width, cap)
        if getattr(self, "_pin_key", None) != key:
            self._free_pinned()
            sizes = [n_frames * height * width, n_frames * cap * KP_DTYPE.itemsize, n_frames * cap * 32, n_frames * 4]
            self._pin_ptrs = []
            for nb in sizes:
                p = C.c_void_p()
                check(self._L.orbfe_host_alloc(C.byref(p), nb))
                self._pin_ptrs.append((p, nb))
            mk = lambda i, dt, shape: np.frombuffer((C.c_uint8 * self._pin_ptrs[i][1]).from_address(self._pin_ptrs[i][0].value),
                                                    dtype=dt).reshape(shape)
            self._pin = (mk(0, np.uint8, (n_frames, height, width)), mk(1, KP_DTYPE, (n_frames, cap)),
                         mk(2, np.uint8, (n_frames, cap, 32)), mk(3, np.int32, (n_frames,)))
            self._pin_key = key
        return self._pin

    def _free_pinned(self):
        for p, _ in getattr(self, "_pin_ptrs", []):
            self._L.orbfe_host_free(p)
        self._pin_ptrs, self._pin, self._pin_key = [], None, None

    def extract_pinned(self, chunk_frames: int = 0):
        """orbfe_extract_batch_pipelined over the pinned buffers: chunked H2D / kernels / D2H on separate streams."""
        img, kps, desc, n = self._pin
        B, H, W = img.shape
        cap = kps.shape[1]
        check(self._L.orbfe_extract_batch_pipelined(self._h, ptr(img), B, W, H, W, W * H, ptr(kps), ptr(desc), cap, ptr(n),
                                                    int(chunk_frames)))
        self._last_shape = (B, H, W)  # frame indices count in the caller's batch; only the last chunk's pyramids are retained
        return kps, desc, n

    def extract_batch_pipelined(self, images: np.ndarray, chunk_frames: int = 0, capacity: int | None = None):
        """images [B,H,W] uint8 in ordinary host memory -> list of (keypoints, descriptors): staged through the
        extractor's pinned buffers, then the pipelined path."""
        images = np.asarray(images, dtype=np.uint8)
        B, H, W = images.shape
        img, kps, desc, n = self.pinned_buffers(B, H, W, capacity)
        np.copyto(img, images)
        self.extract_pinned(chunk_frames)
        return [(kps[f, : n[f]].copy(), desc[f, : n[f]].copy()) for f in range(B)]

    def extract_batch_device(self, d_images: int, n_frames: int, width: int, height: int, stride: int,
                             frame_stride: int, d_keypoints: int, d_descriptors: int, capacity: int,
                             d_n_out: int, wait: bool = True):
        """All pointers are raw DEVICE addresses (e.g. torch tensor .data_ptr()).  wait=False only
        enqueues the batch on the handle's stream; call synchronize() before reading results."""
        fn = self._L.orbfe_extract_batch_device if wait else self._L.orbfe_extract_batch_device_async
        check(fn(self._h, C.c_void_p(d_images), n_frames, width, height, stride, frame_stride,
                 C.c_void_p(d_keypoints), C.c_void_p(d_descriptors), capacity, C.c_void_p(d_n_out)))
        self._last_shape = (n_frames, height, width)

    def extract_stereo_rectified_batch_device(self, rect_left, rect_right, d_raw_left: int, d_raw_right: int, n_pairs: int,
                                              src_width: int, src_height: int, src_stride: int, src_frame_stride: int,
                                              d_rectified: int, d_keypoints: int, d_descriptors: int, capacity: int,
                                              d_n_out: int):
        """cv::remap of both eyes (Examples/Stereo/stereo_euroc.cc:136-137) + the two ExtractORB calls of the stereo
        Frame (src/Frame.cc:78-81) for a device-resident batch of raw pairs; rectified frames land interleaved
        (L0, R0, L1, R1, ...) in d_rectified.  Enqueues only: synchronize() before reading results."""
        check(self._L.orbfe_extract_stereo_rectified_batch_device_async(
            self._h, rect_left._h, rect_right._h, C.c_void_p(d_raw_left), C.c_void_p(d_raw_right), n_pairs, src_width,
            src_height, src_stride, src_frame_stride, C.c_void_p(d_rectified), C.c_void_p(d_keypoints),
            C.c_void_p(d_descriptors), capacity, C.c_void_p(d_n_out)))
        self._last_shape = (2 * n_pairs, rect_left.height, rect_left.width)

    def stereo_match_batch_device(self, n_pairs: int, d_keypoints: int, d_descriptors: int, d_n: int, capacity: int,
                                  mbf: float, mb: float, d_uRight: int, d_depth: int, d_n_stereo: int):
        """Frame::ComputeStereoMatches for pairs (2p, 2p+1) of the last device batch; device addresses."""
        check(self._L.orbfe_stereo_match_batch_device(self._h, n_pairs, C.c_void_p(d_keypoints), C.c_void_p(d_descriptors),
                                                      C.c_void_p(d_n), capacity, float(mbf), float(mb),
                                                      C.c_void_p(d_uRight), C.c_void_p(d_depth), C.c_void_p(d_n_stereo)))

    def synchronize(self):
        check(self._L.orbfe_extractor_synchronize(self._h))

    # ---- mvImagePyramid, include/ORBextractor.h:86 ----
    def pyramid_level(self, level: int, frame: int = 0) -> np.ndarray:
        if self._last_shape is None:
            raise RuntimeError("no extract call yet")
        _, H, W = self._last_shape
        w, h = self.level_size(W, H, level)
        out = np.zeros((h, w), dtype=np.uint8)
        check(self._L.orbfe_extractor_get_pyramid_level(self._h, frame, level, ptr(out), w))
        return out

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.GetLevels())]

    # ---- diagnostics ----
    def debug_blurred_level(self, level: int, frame: int = 0) -> np.ndarray:
        _, H, W = self._last_shape
        w, h = self.level_size(W, H, level)
        out = np.zeros((h, w), dtype=np.uint8)
        check(self._L.orbfe_extractor_debug_blurred_level(self._h, frame, level, ptr(out), w))
        return out

    def debug_candidates(self, level: int, frame: int = 0):
        _, H, W = self._last_shape
        w, h = self.level_size(W, H, level)
        cap = w * h // 4 + 16
        xs = np.zeros(cap, np.float32); ys = np.zeros(cap, np.float32); rs = np.zeros(cap, np.float32)
        n = check(self._L.orbfe_extractor_debug_candidates(self._h, frame, level, ptr(xs), ptr(ys), ptr(rs), cap))
        return xs[:n].copy(), ys[:n].copy(), rs[:n].copy()

    def debug_host_octree(self, enable: bool):
        check(self._L.orbfe_extractor_debug_host_octree(self._h, int(bool(enable))))

    def profile(self, stages=True):
        """stages: True/False, or an iterable of stage names from _lib.STAGES."""
        if stages is True:
            mask = -1
        elif not stages:
            mask = 0
        else:
            mask = sum(1 << _lib.STAGES.index(s) for s in stages)
        check(self._L.orbfe_extractor_profile(self._h, mask))

    def profile_get(self):
        """{stage: (total ms, launches, frames processed by the timed launches)}"""
        ms = np.zeros(len(_lib.STAGES), dtype=np.float64)
        cnt = np.zeros(len(_lib.STAGES), dtype=np.int64)
        fr = np.zeros(len(_lib.STAGES), dtype=np.float64)
        check(self._L.orbfe_extractor_profile_get(self._h, ptr(ms), ptr(cnt), ptr(fr)))
        return {s: (float(ms[i]), int(cnt[i]), float(fr[i])) for i, s in enumerate(_lib.STAGES)}

    def set_streams(self, n: int):
        check(self._L.orbfe_extractor_set_streams(self._h, int(n)))

    def set_schedule(self, lanes: bool):
        """Sub-batches on independent streams (False) or as a three-lane software pipeline (True); same results."""
        check(self._L.orbfe_extractor_set_schedule(self._h, int(bool(lanes))))

    def set_desc_tiles(self, enable):
        """Orientation + descriptor stage in tile form (True), per-keypoint form (False, default) or $ORBFE_DESC_TILES (None)."""
        check(self._L.orbfe_extractor_set_desc_tiles(self._h, -1 if enable is None else int(bool(enable))))

    def set_fast_mode(self, mode):
        """FAST threshold order: 'auto' (default), 'high' (iniThFAST first, per-cell fallback) or 'low' (one attempt
        at the lower threshold); identical results, different speed depending on the image content."""
        check(self._L.orbfe_extractor_set_fast_mode(self._h, {"auto": 0, "high": 1, "low": 2}.get(mode, mode)))

    def set_blur_spec(self, spec: int):
        """0: OpenCV >= 3.4.1/4.x GaussianBlur arithmetic (default); 1: OpenCV 2.4/3.0-3.3 scalar path;
        2: the same with the SSE2 column pass (include/orbfe.h ORBFE_BLUR_*)."""
        check(self._L.orbfe_extractor_set_blur_spec(self._h, int(spec)))

    def set_pyramid_blur(self, enable: bool):
        """ComputePyramid + GaussianBlur as one kernel per level (default) or as separate launches; identical results."""
        check(self._L.orbfe_extractor_set_pyramid_blur(self._h, int(bool(enable))))

    def set_pyramid_chain(self, enable: bool):
        """calls of <= 8 frames: the whole pyramid in one launch (k_pyramid_chain) instead of n-1 resize launches; identical results."""
        check(self._L.orbfe_extractor_set_pyramid_chain(self._h, int(bool(enable))))

    def set_fused(self, enable: bool):
        """GaussianBlur inside the FAST kernel (default) or as its own launch; identical results."""
        check(self._L.orbfe_extractor_set_fused(self._h, int(bool(enable))))


def resize_linear(src: np.ndarray, dw: int, dh: int, device: int = 0) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    dst = np.zeros((dh, dw), dtype=np.uint8)
    check(_lib.load().orbfe_resize_linear(device, ptr(src), sw, sh, sw, ptr(dst), dw, dh, dw))
    return dst


def set_blur_pass_order(order: int) -> int:
    """Process-wide order of the blur kernel's two separable passes (1: horizontal on bytes first, the default; 0: the
    rounds 1-3 form, vertical packed-16 first; a negative value only queries).  Identical bytes; returns the order in force."""
    return int(_lib.load().orbfe_set_blur_pass_order(int(order)))


def gaussian_blur7(src: np.ndarray, device: int = 0, spec: int = 0) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.zeros_like(src)
    check(_lib.load().orbfe_gaussian_blur7_spec(device, int(spec), ptr(src), w, h, w, ptr(dst), w))
    return dst
