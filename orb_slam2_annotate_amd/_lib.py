"""Loader of liborbfe.so (the HIP/gfx950 product library) and its ctypes prototypes.

There is no CPU fallback: if the library is missing the import fails loudly, and every
compute entry point returns ORBFE_ERR_HIP on a machine without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
import os as _os

# $ORBFE_LIB: an alternative build of the library (A/B runs of compile-time variants, tools/ab_build.sh); default in-tree
LIB_PATH = Path(_os.environ["ORBFE_LIB"]) if _os.environ.get("ORBFE_LIB") else PKG_DIR / "liborbfe.so"

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

STAGES = ["h2d", "pyramid", "fast", "octree", "blur", "orient_desc", "d2h", "match"]
ORBFE_OK, ERR_INVALID, ERR_CAPACITY, ERR_HIP, ERR_NOMEM = 0, -1, -2, -3, -4


class OrbfeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"liborbfe error {code}: {msg}")
        self.code = code


class FrameViewC(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p),
                ("angle", C.c_void_p), ("u_right", C.c_void_p), ("desc", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("resident", C.c_void_p)]


class FeatVecC(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_ids", C.c_void_p), ("offsets", C.c_void_p),
                ("indices", C.c_void_p)]


# every symbol include/orbfe.h declares (tests/test_cabi_symbols.py checks the header against this)
EXPORTS = [
    "orbfe_last_error", "orbfe_device_count", "orbfe_extractor_create", "orbfe_extractor_destroy",
    "orbfe_extractor_get_levels", "orbfe_extractor_get_scale_factor", "orbfe_extractor_get_scale_factors",
    "orbfe_extractor_get_inverse_scale_factors", "orbfe_extractor_get_scale_sigma_squares",
    "orbfe_extractor_get_inverse_scale_sigma_squares", "orbfe_extractor_get_features_per_level",
    "orbfe_extractor_get_umax", "orbfe_extractor_max_keypoints", "orbfe_extractor_max_keypoints_for", "orbfe_extract", "orbfe_extract_batch",
    "orbfe_extract_batch_device", "orbfe_extract_batch_device_async", "orbfe_extract_batch_pipelined", "orbfe_host_alloc", "orbfe_host_free", "orbfe_extract_stereo_frame", "orbfe_extractor_synchronize", "orbfe_extractor_level_size", "orbfe_extractor_get_pyramid_level",
    "orbfe_extractor_pyramid_level_device", "orbfe_extractor_debug_candidates",
    "orbfe_extractor_debug_blurred_level", "orbfe_extractor_debug_host_octree", "orbfe_debug_octree_host", "orbfe_debug_geometry",
    "orbfe_debug_resize_tables", "orbfe_debug_resize_tiles", "orbfe_extractor_set_streams", "orbfe_extractor_set_fused", "orbfe_extractor_set_pyramid_blur", "orbfe_extractor_set_pyramid_chain", "orbfe_set_blur_pass_order", "orbfe_extractor_set_fast_mode", "orbfe_extractor_set_schedule", "orbfe_extractor_set_desc_tiles", "orbfe_extractor_set_blur_spec", "orbfe_gaussian_blur7_spec", "orbfe_extractor_profile", "orbfe_extractor_profile_get",
    "orbfe_stage_name", "orbfe_resize_linear", "orbfe_gaussian_blur7", "orbfe_descriptor_distance",
    "orbfe_hamming_matrix", "orbfe_search_by_bow", "orbfe_search_by_bow_kf",
    "orbfe_search_for_triangulation", "orbfe_frame_upload", "orbfe_frame_release", "orbfe_frame_get_view", "orbfe_search_by_bow_resident", "orbfe_search_by_bow_kf_resident", "orbfe_search_for_triangulation_multi", "orbfe_fuse_search_multi", "orbfe_search_by_bow_multi", "orbfe_search_by_bow_kf_multi", "orbfe_search_by_projection_keyframe_multi", "orbfe_frame_from_extractor", "orbfe_frame_from_device", "orbfe_frame_set_featvec", "orbfe_compute_stereo_matches", "orbfe_stereo_match_batch_device", "orbfe_vocabulary_load_text", "orbfe_vocabulary_create", "orbfe_vocabulary_destroy",
    "orbfe_vocabulary_info", "orbfe_vocabulary_transform", "orbfe_vocabulary_featvec_batch_device",
    "orbfe_bow_match_consecutive_batch_device", "orbfe_bow_match_consecutive_batch_device_async", "orbfe_bow_match_consecutive_stereo_batch_device_async", "orbfe_cvt_gray", "orbfe_cvt_gray_batch_device",
    "orbfe_distinctive_descriptors", "orbfe_features_in_area", "orbfe_search_by_projection",
    "orbfe_search_by_projection_last_frame", "orbfe_search_by_projection_keyframe",
    "orbfe_search_by_projection_sim3", "orbfe_search_for_initialization", "orbfe_debug_last_claim_rounds", "orbfe_fuse_search", "orbfe_search_by_sim3",
    "orbfe_init_undistort_rectify_map", "orbfe_rectifier_create", "orbfe_rectifier_destroy", "orbfe_remap", "orbfe_remap_batch_device", "orbfe_extract_stereo_rectified_batch_device_async",
    "orbfe_undistort_points", "orbfe_undistort_keypoints_batch_device", "orbfe_compute_image_bounds",
    "orbfe_stereo_from_rgbd",
]

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP/HSA
    runtimes in one process cannot both own the GPU, so when torch is installed but not imported
    yet, its runtime is loaded first and liborbfe.so binds to it by SONAME -- `import torch` then
    works in either order.  No torch module is imported here; without torch this is a no-op."""
    import importlib.util
    import os
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    hip = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime


def load():
    """Load liborbfe.so; raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = C.CDLL(str(LIB_PATH))
    L.orbfe_last_error.restype = C.c_char_p
    L.orbfe_stage_name.restype = C.c_char_p
    L.orbfe_extractor_get_scale_factor.restype = C.c_float
    L.orbfe_extractor_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_void_p)]
    L.orbfe_extractor_destroy.argtypes = [C.c_void_p]
    L.orbfe_extractor_destroy.restype = None
    vp, ci, cf, cs = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orbfe_extractor_get_levels.argtypes = [vp]
    L.orbfe_extractor_get_scale_factor.argtypes = [vp]
    for n in ("scale_factors", "inverse_scale_factors", "scale_sigma_squares", "inverse_scale_sigma_squares",
              "features_per_level", "umax"):
        getattr(L, f"orbfe_extractor_get_{n}").argtypes = [vp, vp]
    L.orbfe_extractor_max_keypoints.argtypes = [vp]
    L.orbfe_extract.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, vp]
    L.orbfe_extract_batch.argtypes = [vp, vp, ci, ci, ci, ci, cs, vp, vp, ci, vp]
    L.orbfe_extract_batch_device.argtypes = [vp, vp, ci, ci, ci, ci, cs, vp, vp, ci, vp]
    L.orbfe_extract_batch_pipelined.argtypes = [vp, vp, ci, ci, ci, ci, cs, vp, vp, ci, vp, ci]
    L.orbfe_host_alloc.argtypes = [C.POINTER(vp), cs]
    L.orbfe_host_free.argtypes = [vp]
    L.orbfe_host_free.restype = None
    L.orbfe_extract_batch_device_async.argtypes = [vp, vp, ci, ci, ci, ci, cs, vp, vp, ci, vp]
    L.orbfe_extractor_synchronize.argtypes = [vp]
    L.orbfe_extractor_level_size.argtypes = [vp, ci, ci, ci, vp, vp]
    L.orbfe_extractor_get_pyramid_level.argtypes = [vp, ci, ci, vp, ci]
    L.orbfe_extractor_pyramid_level_device.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    L.orbfe_extractor_debug_candidates.argtypes = [vp, ci, ci, vp, vp, vp, ci]
    L.orbfe_extractor_debug_blurred_level.argtypes = [vp, ci, ci, vp, ci]
    L.orbfe_extractor_profile.argtypes = [vp, ci]
    L.orbfe_extractor_debug_host_octree.argtypes = [vp, ci]
    L.orbfe_extractor_profile_get.argtypes = [vp, vp, vp, vp]
    L.orbfe_extractor_set_streams.argtypes = [vp, ci]
    L.orbfe_stage_name.argtypes = [ci]
    L.orbfe_resize_linear.argtypes = [ci, vp, ci, ci, ci, vp, ci, ci, ci]
    L.orbfe_gaussian_blur7.argtypes = [ci, vp, ci, ci, ci, vp, ci]
    L.orbfe_descriptor_distance.argtypes = [ci, vp, vp, ci, vp]
    L.orbfe_hamming_matrix.argtypes = [ci, vp, ci, vp, ci, vp]
    fvp = C.POINTER(FeatVecC)
    L.orbfe_search_by_bow.argtypes = [ci, vp, vp, vp, ci, fvp, vp, vp, ci, fvp, cf, ci, vp]
    L.orbfe_search_by_bow_kf.argtypes = [ci, vp, vp, vp, ci, fvp, vp, vp, vp, ci, fvp, cf, ci, vp]
    L.orbfe_search_for_triangulation.argtypes = [ci, vp, vp, vp, vp, vp, vp, ci, fvp, vp, vp, vp, vp, vp, vp, vp,
                                                 ci, fvp, vp, cf, cf, vp, vp, ci, ci, ci, vp]
    L.orbfe_compute_stereo_matches.argtypes = [vp, ci, vp, ci, vp, vp, ci, vp, vp, ci, cf, cf, vp, vp]
    L.orbfe_extract_stereo_frame.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp, vp]
    L.orbfe_stereo_match_batch_device.argtypes = [vp, ci, vp, vp, vp, ci, cf, cf, vp, vp, vp]
    L.orbfe_vocabulary_load_text.argtypes = [C.c_char_p, ci, C.POINTER(C.c_void_p)]
    L.orbfe_vocabulary_create.argtypes = [ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, C.POINTER(C.c_void_p)]
    L.orbfe_vocabulary_destroy.argtypes = [vp]
    L.orbfe_vocabulary_destroy.restype = None
    L.orbfe_vocabulary_info.argtypes = [vp, vp, vp, vp, vp]
    L.orbfe_vocabulary_transform.argtypes = [vp, vp, ci, ci, vp, vp, vp]
    L.orbfe_vocabulary_featvec_batch_device.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    L.orbfe_bow_match_consecutive_batch_device.argtypes = [vp, ci, vp, vp, vp, ci, ci, cf, ci, vp, vp]
    L.orbfe_bow_match_consecutive_batch_device_async.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci, cf, ci, vp, vp]
    L.orbfe_cvt_gray.argtypes = [ci, vp, ci, ci, ci, ci, ci, vp, ci]
    L.orbfe_cvt_gray_batch_device.argtypes = [ci, vp, ci, ci, ci, ci, cs, ci, ci, vp, ci, cs]
    L.orbfe_distinctive_descriptors.argtypes = [ci, vp, vp, ci, vp]
    fwp = C.POINTER(FrameViewC)
    L.orbfe_features_in_area.argtypes = [ci, fwp, ci, vp, vp, vp, vp, vp, ci, vp, vp]
    L.orbfe_search_by_projection.argtypes = [ci, fwp, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, cf, vp, vp]
    L.orbfe_search_by_projection_last_frame.argtypes = [ci, fwp, vp, ci, cf, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                        ci, cf, ci, vp, vp]
    L.orbfe_search_by_projection_keyframe.argtypes = [ci, fwp, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, cf, ci, ci, vp, vp]
    L.orbfe_search_by_projection_sim3.argtypes = [ci, fwp, vp, ci, vp, ci, vp, vp, vp, vp, vp, cf, vp, vp]
    L.orbfe_set_blur_pass_order.argtypes = [ci]
    L.orbfe_set_blur_pass_order.restype = ci
    L.orbfe_debug_last_claim_rounds.argtypes = []
    L.orbfe_debug_last_claim_rounds.restype = ci
    L.orbfe_search_for_initialization.argtypes = [ci, fwp, fwp, vp, vp, ci, cf, ci, vp, vp]
    L.orbfe_fuse_search.argtypes = [ci, fwp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, cf, ci, vp]
    L.orbfe_search_by_sim3.argtypes = [ci, fwp, fwp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cf, vp, vp]
    L.orbfe_frame_upload.argtypes = [ci, fwp, vp, C.POINTER(C.c_void_p)]
    L.orbfe_frame_release.argtypes = [vp]
    L.orbfe_frame_release.restype = None
    L.orbfe_frame_get_view.argtypes = [vp]
    L.orbfe_frame_get_view.restype = fwp
    L.orbfe_search_by_bow_resident.argtypes = [vp, vp, vp, cf, ci, vp]
    L.orbfe_search_by_bow_kf_resident.argtypes = [vp, vp, vp, vp, cf, ci, vp]
    L.orbfe_search_for_triangulation_multi.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp]
    L.orbfe_fuse_search_multi.argtypes = [ci, ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, cf, ci, vp]
    L.orbfe_search_by_bow_multi.argtypes = [ci, vp, vp, vp, cf, ci, vp, vp]
    L.orbfe_search_by_bow_kf_multi.argtypes = [vp, vp, ci, vp, vp, cf, ci, vp, vp]
    L.orbfe_search_by_projection_keyframe_multi.argtypes = [ci, fwp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp]
    L.orbfe_frame_from_extractor.argtypes = [vp, ci, fwp, vp, ci, C.POINTER(C.c_void_p)]
    L.orbfe_frame_from_device.argtypes = [ci, vp, vp, fwp, vp, ci, C.POINTER(C.c_void_p)]
    L.orbfe_frame_set_featvec.argtypes = [vp, vp]
    L.orbfe_init_undistort_rectify_map.argtypes = [ci, vp, vp, ci, vp, vp, ci, ci, vp, vp]
    L.orbfe_rectifier_create.argtypes = [ci, vp, vp, ci, ci, ci, C.POINTER(C.c_void_p)]
    L.orbfe_rectifier_destroy.argtypes = [vp]
    L.orbfe_rectifier_destroy.restype = None
    L.orbfe_remap.argtypes = [vp, vp, ci, ci, ci, vp, ci]
    L.orbfe_remap_batch_device.argtypes = [vp, vp, ci, ci, ci, ci, cs, vp, ci, cs]
    L.orbfe_extract_stereo_rectified_batch_device_async.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, cs, vp, vp, vp, ci, vp]
    L.orbfe_bow_match_consecutive_stereo_batch_device_async.argtypes = [vp, vp, ci, vp, vp, vp, ci, ci, cf, ci, vp, vp]
    L.orbfe_undistort_points.argtypes = [ci, vp, ci, vp, vp, ci, vp]
    L.orbfe_undistort_keypoints_batch_device.argtypes = [ci, vp, vp, ci, ci, vp, vp, ci, vp]
    L.orbfe_compute_image_bounds.argtypes = [ci, ci, ci, vp, vp, ci, vp]
    L.orbfe_stereo_from_rgbd.argtypes = [ci, vp, vp, vp, ci, vp, ci, ci, ci, cf, vp, vp]
    L.orbfe_extractor_max_keypoints_for.argtypes = [vp, ci, ci]
    L.orbfe_debug_octree_host.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, ci]
    L.orbfe_debug_geometry.argtypes = [ci, cf, ci, ci, ci, ci, ci, vp, vp, vp, ci]
    L.orbfe_debug_resize_tables.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp]
    L.orbfe_debug_resize_tiles.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise OrbfeError(rc, load().orbfe_last_error().decode("utf-8", "replace"))
    return rc


def ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)
