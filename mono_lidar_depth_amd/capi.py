"""ctypes binding of the C-ABI in include/mld.h (libmld_hip.so).

The library is built in-tree by `mono_lidar_depth_amd/csrc/Makefile` (see `__graft_entry__.build`).
There is no fallback: if the shared library is missing, importing the symbols raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libmld_hip.so"

MLD_ABI_VERSION = 8  # include/mld.h
MLD_OK = 0
MLD_ERR_INVALID_ARG = -1
MLD_ERR_NOT_INITIALIZED = -2
MLD_ERR_UNSUPPORTED_MODE = -3
MLD_ERR_NO_ROAD_ESTIMATOR = -4
MLD_ERR_NO_GROUND_PLANE = -5
MLD_ERR_CLOUD_TOO_SMALL = -6
MLD_ERR_HIP = -7
MLD_ERR_CAPACITY = -8
MLD_RESULT_TYPE_COUNT = 21

# eDepthResultType.h:8-30
RESULT_TYPE_NAMES = {
    0: "Unspecified", 1: "Success", 2: "RadiusSearchInsufficientPoints", 3: "HistogramNoLocalMax",
    4: "TresholdDepthGlobalGreaterMax", 5: "TresholdDepthGlobalSmallerMin", 6: "TresholdDepthLocalGreaterMax",
    7: "TresholdDepthLocalSmallerMin", 8: "TriangleNotPlanar", 9: "TriangleNotPlanarInsufficientPoints",
    10: "CornerBehindCamera", 11: "PlaneViewrayNotOrthogonal", 12: "PcaIsPoint", 13: "PcaIsLine",
    14: "PcaIsCubic", 15: "InsufficientRoadPoints", 16: "SuccessRoad",
    17: "RegionGrowingNearestSeedNotAvailable", 18: "RegionGrowingSeedsOutOfRange",
    19: "RegionGrowingInsufficientPoints", 20: "SuccessRegionGrowing",
}


class MldCamera(C.Structure):
    _fields_ = [
        ("focal_length", C.c_double),
        ("principal_point_x", C.c_double),
        ("principal_point_y", C.c_double),
        ("width", C.c_int32),
        ("height", C.c_int32),
    ]


class MldParams(C.Structure):
    _fields_ = [
        ("histogram_segmentation_bin_witdh", C.c_double),
        ("treshold_depth_local_value", C.c_double),
        ("pca_treshold_3_abs_min", C.c_double),
        ("pca_treshold_3_2_rel_max", C.c_double),
        ("pca_treshold_2_1_rel_min", C.c_double),
        ("ransac_plane_point_distance_treshold", C.c_double),
        ("ransac_plane_distance_treshold", C.c_double),
        ("ransac_plane_min_z", C.c_double),
        ("ransac_plane_max_z", C.c_double),
        ("ransac_plane_refinement_treshold", C.c_double),
        ("ransac_plane_probability", C.c_double),
        ("plane_estimator_z_x_min_relation", C.c_double),
        ("triangleplanar_crossnorm_treshold", C.c_double),
        ("viewray_plane_orthoganality_treshold", C.c_double),
        ("ransac_plane_treshold_camx", C.c_double),
        ("neighbor_search_mode", C.c_int32),
        ("pixelarea_search_witdh", C.c_int32),
        ("pixelarea_search_height", C.c_int32),
        ("radiusSearch_count_min", C.c_int32),
        ("do_use_histogram_segmentation", C.c_int32),
        ("histogram_segmentation_min_pointcount", C.c_int32),
        ("do_use_depth_segmentation", C.c_int32),
        ("treshold_depth_enabled", C.c_int32),
        ("treshold_depth_mode", C.c_int32),
        ("treshold_depth_max", C.c_int32),
        ("treshold_depth_min", C.c_int32),
        ("treshold_depth_local_enabled", C.c_int32),
        ("treshold_depth_local_mode", C.c_int32),
        ("treshold_depth_local_valuetype", C.c_int32),
        ("do_use_PCA", C.c_int32),
        ("do_use_ransac_plane", C.c_int32),
        ("plane_estimator_use_triangle_maximation", C.c_int32),
        ("plane_estimator_use_leastsquares", C.c_int32),
        ("plane_estimator_use_mestimator", C.c_int32),
        ("do_use_cut_behind_camera", C.c_int32),
        ("do_use_triangle_size_maximation", C.c_int32),
        ("do_check_triangleplanar_condition", C.c_int32),
        ("set_all_depths_to_zero", C.c_int32),
        ("ransac_plane_max_iterations", C.c_int32),
        ("ransac_plane_use_refinement", C.c_int32),
        ("ransac_plane_use_camx_treshold", C.c_int32),
    ]

    def copy(self) -> "MldParams":
        out = MldParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(MldParams))
        return out

    def replace(self, **kw) -> "MldParams":
        out = self.copy()
        for k, v in kw.items():
            if not hasattr(out, k):
                raise AttributeError(k)
            setattr(out, k, v)
        return out


MLD_PLANE_RANSAC, MLD_PLANE_SEMANTIC = 0, 1


class MldPlaneRequest(C.Structure):
    """mld_plane_request (include/mld.h): how the not-yet-segmented plane of a one-frame call is estimated."""
    _fields_ = [
        ("kind", C.c_int32),
        ("seed", C.c_uint32),
        ("label_image", C.c_void_p),
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("row_stride_bytes", C.c_int32),
        ("n_labels", C.c_int32),
        ("ground_labels", C.c_void_p),
        ("inlier_threshold", C.c_double),
    ]


class MldPlaneResult(C.Structure):
    _fields_ = [("coeffs", C.c_float * 4), ("n_inliers", C.c_int64), ("status", C.c_int32), ("iterations", C.c_int32)]


# Every symbol include/mld.h declares: (name, restype, argtypes)
_P = C.POINTER
_SIGNATURES = [
    ("mld_params_default", None, [_P(MldParams)]),
    ("mld_params_c0", None, [_P(MldParams)]),
    ("mld_params_from_file", C.c_int, [_P(MldParams), C.c_char_p, C.c_char_p, C.c_int]),
    ("mld_abi_version", C.c_int, []),
    ("mld_create", C.c_void_p, [_P(MldParams), _P(MldCamera), _P(C.c_double), C.c_int, C.c_int, C.c_int64,
                                C.c_int64, _P(C.c_int)]),
    ("mld_create_error", C.c_char_p, []),
    ("mld_destroy", None, [C.c_void_p]),
    ("mld_last_error", C.c_char_p, [C.c_void_p]),
    ("mld_get_stream", C.c_void_p, [C.c_void_p]),
    ("mld_synchronize", C.c_int, [C.c_void_p]),
    ("mld_order_after", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mld_order_after_classify", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mld_pair_contexts", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mld_set_shared_gpu", C.c_int, [C.c_void_p, C.c_int]),
    ("mld_set_list_capacity", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("mld_set_list_budget", C.c_int, [C.c_void_p, C.c_int]),
    ("mld_contexts_concurrent", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mld_get_path_counts", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("mld_set_cloud", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int]),
    ("mld_set_cloud_device", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int]),
    ("mld_set_clouds_device", C.c_int, [C.c_void_p, C.c_int, _P(C.c_void_p), _P(C.c_int64), C.c_int]),
    ("mld_set_clouds_planes_device", C.c_int, [C.c_void_p, C.c_int, _P(C.c_void_p), _P(C.c_int64), C.c_int,
                                               _P(C.c_float), _P(C.c_void_p)]),
    ("mld_set_clouds_estimate_planes_device", C.c_int, [C.c_void_p, C.c_int, _P(C.c_void_p), _P(C.c_int64), C.c_int,
                                                        _P(C.c_uint32)]),
    ("mld_get_estimated_planes", C.c_int, [C.c_void_p, C.c_int, _P(C.c_float), _P(C.c_int64), _P(C.c_int32)]),
    ("mld_set_ground_plane", C.c_int, [C.c_void_p, C.c_int, _P(C.c_float), C.c_void_p, C.c_int64]),
    ("mld_set_ground_plane_device", C.c_int, [C.c_void_p, C.c_int, _P(C.c_float), C.c_void_p, C.c_int64]),
    ("mld_estimate_ground_plane", C.c_int, [C.c_void_p, C.c_int, C.c_uint32, _P(C.c_float), _P(C.c_int64)]),
    ("mld_estimate_semantic_plane", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                              C.c_int, C.c_double, _P(C.c_float), _P(C.c_int64)]),
    ("mld_estimate_semantic_plane_device", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                                     C.c_void_p, C.c_int, C.c_double, _P(C.c_float), _P(C.c_int64)]),
    ("mld_get_ground_plane_inliers", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    ("mld_set_ground_plane_mask_device", C.c_int, [C.c_void_p, C.c_int, _P(C.c_float), C.c_void_p]),
    ("mld_set_ground_planes_mask_device", C.c_int, [C.c_void_p, C.c_int, _P(C.c_float), _P(C.c_void_p)]),
    ("mld_calculate_depth", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("mld_calculate_depth_opts", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_uint32]),
    ("mld_calculate_depth_frame", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, _P(C.c_float), C.c_void_p,
                                            C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("mld_calculate_depth_frame_estimate", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                                     _P(MldPlaneRequest), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                     _P(MldPlaneResult)]),
    ("mld_frame_timing", C.c_int, [C.c_void_p, _P(C.c_double)]),
    ("mld_pack_points_host", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    ("mld_calculate_depth_device", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("mld_calculate_depths_device", C.c_int, [C.c_void_p, C.c_int, _P(C.c_void_p), _P(C.c_int64),
                                              _P(C.c_void_p), _P(C.c_void_p)]),
    ("mld_tracklets_depth_device", C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int64] +
     [C.c_void_p] * 4 + [_P(C.c_int64)]),
    ("mld_tracklets_depth", C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int64] +
     [C.c_void_p] * 4 + [_P(C.c_int64)]),
    ("mld_tracklets_frame", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_int, _P(MldPlaneRequest),
                                      _P(C.c_float), C.c_void_p, C.c_int64] + [C.c_void_p] * 5 + [C.c_int64] +
     [C.c_void_p] * 4 + [_P(C.c_int64), _P(MldPlaneResult)]),
    ("mld_set_clouds_planes_range_device", C.c_int, [C.c_void_p, C.c_int, C.c_int, _P(C.c_void_p), _P(C.c_int64), C.c_int,
                                                     _P(C.c_float), _P(C.c_void_p)]),
    ("mld_tracklets_depths_device", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [_P(C.c_void_p)] * 5 +
     [_P(C.c_int64)] + [_P(C.c_void_p)] * 4),
    ("mld_get_visible_count", C.c_int, [C.c_void_p, C.c_int, _P(C.c_int64)]),
    ("mld_get_visible_image_points", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    ("mld_get_point_index", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    ("mld_get_cloud_camera_cs", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    ("mld_get_pixel_map", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    ("mld_get_point_depth_cam_visible", C.c_int, [C.c_void_p, C.c_int, C.c_int64, _P(C.c_double)]),
    ("mld_calculate_depth_debug", C.c_int,
     [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mld_get_ground_plane_cloud", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, _P(C.c_int64)]),
    ("mld_result_histogram", C.c_int, [C.c_void_p, C.c_int64, _P(C.c_int64)]),
    ("mld_timing_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("mld_timing_reset", C.c_int, [C.c_void_p]),
    ("mld_kernel_time_ms", C.c_int, [C.c_void_p, C.c_int, _P(C.c_double), _P(C.c_int64)]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None
_lib_ab = None
LIB_AB_PATH = _HERE / "lib" / "libmld_hip_ab.so"


def _bind(path: Path) -> C.CDLL:
    # PyTorch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Two HIP runtimes in one process cannot
    # both own the GPU, so when torch is installed it is imported FIRST: the dynamic loader then resolves this
    # library's libamdhip64.so.7 dependency to the runtime torch already loaded.  (A C++ caller without torch
    # simply gets /opt/rocm's runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C mono_lidar_depth_amd/csrc`). There is no CPU fallback for this path.")
    lib = C.CDLL(str(path))
    for name, restype, argtypes in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load() -> C.CDLL:
    """Load libmld_hip.so.  Raises (never falls back) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        _lib = _bind(Path(os.environ.get("MLD_HIP_LIBRARY", LIB_PATH)))
    return _lib


def load_ab() -> C.CDLL:
    """The test / measurement build of the same sources (-DMLD_AB_SWITCHES, see csrc/Makefile): its mld_create reads
    MLD_FORCE_WAVE_PATH etc. from the environment.  Used by the GPU parity suite and the A/B tools, never by the
    product path."""
    global _lib_ab
    if _lib_ab is None:
        _lib_ab = _bind(Path(os.environ.get("MLD_HIP_AB_LIBRARY") or LIB_AB_PATH))  # (the override: A/B tools only)
    return _lib_ab


def params_c0() -> MldParams:
    p = MldParams()
    load().mld_params_c0(C.byref(p))
    return p


def params_default() -> MldParams:
    p = MldParams()
    load().mld_params_default(C.byref(p))
    return p


def params_from_file(path: str) -> MldParams:
    p = MldParams()
    err = C.create_string_buffer(2048)
    rc = load().mld_params_from_file(C.byref(p), str(path).encode(), err, 2048)
    if rc != MLD_OK:
        raise RuntimeError(err.value.decode())
    # mirrored keys the file lacks (they read as 0, as cv::FileStorage reads them in the reference)
    note = err.value.decode()
    p.absent_keys = [k.strip() for k in note.split(":", 1)[1].split(",")] if note.startswith("absent") else []
    return p
