"""ctypes wrapper of the CPU oracle (oracle/libmld_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (mono_lidar_depth_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from mono_lidar_depth_amd.capi import MldCamera, MldParams

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libmld_oracle.so"
_lib = None


class OrcTrace(C.Structure):
    _fields_ = [("type", C.c_int32), ("reached_road", C.c_int32), ("depth", C.c_double),
                ("n_nb", C.c_int32), ("n_seg", C.c_int32), ("n_road", C.c_int32), ("n_road_inl", C.c_int32),
                ("corner_pos", C.c_int32 * 3), ("pad_", C.c_int32), ("plane_n", C.c_double * 3),
                ("plane_offset", C.c_double)]


def build():
    subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        build()
    lib = C.CDLL(str(LIB_PATH))
    P = C.POINTER
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [P(MldParams), P(MldCamera), P(C.c_double), P(C.c_int)]
    lib.orc_destroy.argtypes = [C.c_void_p]
    lib.orc_get_calibration.argtypes = [C.c_void_p, P(C.c_double), P(C.c_double)]
    lib.orc_set_cloud.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    lib.orc_set_ground_plane.argtypes = [C.c_void_p, P(C.c_float), C.c_void_p, C.c_int64]
    lib.orc_calculate_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_num_points.restype = C.c_int64
    lib.orc_num_points.argtypes = [C.c_void_p]
    lib.orc_visible_count.restype = C.c_int64
    lib.orc_visible_count.argtypes = [C.c_void_p]
    for name in ("orc_get_visible_image_points", "orc_get_point_index", "orc_get_cloud_camera_cs",
                 "orc_get_cloud_image_cs", "orc_get_in_range", "orc_get_pixel_map"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p]
        getattr(lib, name).restype = None
    lib.orc_trace_feature.argtypes = [C.c_void_p, C.c_double, C.c_double, P(OrcTrace), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int32]
    lib.orc_estimate_ground_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_uint32, P(C.c_float), P(C.c_int64)]
    lib.orc_estimate_semantic_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_int, C.c_double, P(C.c_float), P(C.c_int64)]
    lib.orc_get_plane_inliers.restype = C.c_int64
    lib.orc_get_plane_inliers.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.orc_tracklets_depth.argtypes = [C.c_void_p, C.c_void_p] + [C.c_void_p] * 5 + [C.c_int64] + [C.c_void_p] * 4 + [C.c_int]
    lib.orc_filter_points_min_dist_blob.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                                    P(C.c_int32), P(C.c_double), P(C.c_double)]
    lib.orc_histogram_counts.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
    lib.orc_histogram_counts.restype = None
    lib.orc_get_nearest_point.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_max_spanning_triangle.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    lib.orc_check_planar.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
    lib.orc_viewing_ray.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    lib.orc_viewing_ray.restype = None
    lib.orc_intersect_triangle.argtypes = [C.c_void_p] * 5 + [C.c_double, C.c_void_p, P(C.c_double)]
    lib.orc_threshold_global.argtypes = [P(MldParams), P(C.c_double)]
    lib.orc_threshold_local.argtypes = [P(MldParams), C.c_void_p, C.c_int, P(C.c_double)]
    lib.orc_mestimator_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, P(C.c_double)]
    lib.orc_mestimator_plane.restype = None
    lib.orc_pca.argtypes = [P(MldParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleDepthEstimator:
    """The restated reference CPU path for one frame (setInputCloud + CalculateDepth)."""

    def __init__(self, params: MldParams, camera: MldCamera, T_cam_lidar):
        self.lib = load()
        T = _f64(np.asarray(T_cam_lidar)[:3, :4])
        st = C.c_int(0)
        self.camera = camera
        h = self.lib.orc_create(C.byref(params), C.byref(camera), T.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))
        if not h:
            raise RuntimeError(f"orc_create failed: status {st.value}")
        self.h = C.c_void_p(h)
        self._keep = None

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_cloud(self, cloud: np.ndarray):
        arr = np.ascontiguousarray(cloud, dtype=np.float32)
        assert arr.ndim == 2 and arr.shape[1] in (4, 8)
        self._keep = arr
        rc = self.lib.orc_set_cloud(self.h, arr.ctypes.data, arr.shape[0], arr.shape[1] * 4)
        assert rc == 0, rc

    def set_ground_plane(self, coeffs, inliers):
        if coeffs is None:
            rc = self.lib.orc_set_ground_plane(self.h, None, None, 0)
        else:
            c = (C.c_float * 4)(*[float(x) for x in coeffs])
            inl = np.ascontiguousarray(inliers, dtype=np.int32)
            rc = self.lib.orc_set_ground_plane(self.h, c, inl.ctypes.data, inl.size)
        assert rc == 0, rc

    def estimate_ground_plane(self, seed: int = 0):
        """RansacPlane::CalculateInliersPlane restated (see orc_estimate_ground_plane); sets the frame's plane."""
        arr = self._keep
        coeffs = (C.c_float * 4)()
        n_inl = C.c_int64(0)
        rc = self.lib.orc_estimate_ground_plane(self.h, arr.ctypes.data, arr.shape[0], arr.shape[1] * 4, seed, coeffs,
                                                C.byref(n_inl))
        if rc != 0:
            raise RuntimeError(f"orc_estimate_ground_plane: status {rc}")
        inl = np.empty(n_inl.value, dtype=np.int32)
        k = self.lib.orc_get_plane_inliers(self.h, inl.ctypes.data, inl.size)  # unique, ascending
        return np.array(list(coeffs), dtype=np.float32), inl[:k].copy()

    def estimate_semantic_plane(self, label_image, labels, inlier_threshold: float):
        """SemanticPlane::CalculateInliersPlane restated (see orc_estimate_semantic_plane); sets the frame's plane."""
        arr = self._keep
        img = np.ascontiguousarray(label_image, dtype=np.uint8)
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        coeffs = (C.c_float * 4)()
        n_inl = C.c_int64(0)
        rc = self.lib.orc_estimate_semantic_plane(self.h, arr.ctypes.data, arr.shape[0], arr.shape[1] * 4, img.ctypes.data,
                                                  img.shape[0], img.shape[1], img.strides[0], lab.ctypes.data, lab.size,
                                                  float(inlier_threshold), coeffs, C.byref(n_inl))
        if rc != 0:
            raise RuntimeError(f"orc_estimate_semantic_plane: status {rc}")
        inl = np.empty(n_inl.value, dtype=np.int32)
        k = self.lib.orc_get_plane_inliers(self.h, inl.ctypes.data, inl.size)
        return np.array(list(coeffs), dtype=np.float32), inl[:k].copy()

    def calculate_depth(self, uv, n_threads: int = 1):
        uv = _f64(uv)
        assert uv.ndim == 2 and uv.shape[1] == 2
        F = uv.shape[0]
        depth = np.empty(F, dtype=np.float64)
        types = np.empty(F, dtype=np.int32)
        rc = self.lib.orc_calculate_depth(self.h, uv.ctypes.data, F, depth.ctypes.data, types.ctypes.data, n_threads)
        assert rc == 0, rc
        return depth, types

    @property
    def n(self):
        return int(self.lib.orc_num_points(self.h))

    @property
    def nvis(self):
        return int(self.lib.orc_visible_count(self.h))

    def visible_image_points(self):
        out = np.empty((self.nvis, 2), dtype=np.float64)
        self.lib.orc_get_visible_image_points(self.h, out.ctypes.data)
        return out.T

    def point_index(self):
        out = np.empty(self.nvis, dtype=np.int32)
        self.lib.orc_get_point_index(self.h, out.ctypes.data)
        return out

    def cloud_camera_cs(self):
        out = np.empty((self.n, 3), dtype=np.float64)
        self.lib.orc_get_cloud_camera_cs(self.h, out.ctypes.data)
        return out.T

    def cloud_image_cs(self):
        out = np.empty((self.n, 2), dtype=np.float64)
        self.lib.orc_get_cloud_image_cs(self.h, out.ctypes.data)
        return out.T

    def in_range(self):
        out = np.empty(self.n, dtype=np.uint8)
        self.lib.orc_get_in_range(self.h, out.ctypes.data)
        return out.astype(bool)

    def pixel_map(self):
        W, H = self.camera.width, self.camera.height
        out = np.empty(W * H, dtype=np.int32)
        self.lib.orc_get_pixel_map(self.h, out.ctypes.data)
        return out.reshape(H, W)

    def calibration(self):
        Tinv = np.empty(12)
        Kinv = np.empty(9)
        self.lib.orc_get_calibration(self.h, Tinv.ctypes.data_as(C.POINTER(C.c_double)),
                                     Kinv.ctypes.data_as(C.POINTER(C.c_double)))
        return Tinv.reshape(3, 4), Kinv.reshape(3, 3)

    def trace_feature(self, u: float, v: float, cap: int = 4096):
        tr = OrcTrace()
        bufs = [np.empty(cap, dtype=np.int32) for _ in range(4)]
        rc = self.lib.orc_trace_feature(self.h, u, v, C.byref(tr), *[b.ctypes.data for b in bufs], cap)
        assert rc == 0, rc
        return {
            "type": int(tr.type), "depth": float(tr.depth), "reached_road": int(tr.reached_road),
            "nb_idx": bufs[0][:tr.n_nb].copy(), "seg_pos": bufs[1][:tr.n_seg].copy(),
            "road_idx": bufs[2][:tr.n_road].copy(), "road_pos": bufs[3][:tr.n_road_inl].copy(),
            "corner_pos": [int(x) for x in tr.corner_pos], "plane_n": [float(x) for x in tr.plane_n],
            "plane_offset": float(tr.plane_offset),
        }

    def viewing_ray(self, u, v):
        out = np.empty(3)
        self.lib.orc_viewing_ray(self.h, u, v, out.ctypes.data)
        return out


def tracklets_depth(cur: OracleDepthEstimator, last, u_new, v_new, u_old, v_old, is_new, n_threads: int = 1):
    """TrackletDepthModule::process feature marshalling on the CPU (see orc_tracklets_depth)."""
    arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (u_new, v_new, u_old, v_old)]
    isn = np.ascontiguousarray(is_new, dtype=np.uint8)
    n = isn.size
    d_cur = np.full(n, np.nan, dtype=np.float32)
    d_last = np.full(n, np.nan, dtype=np.float32)
    t_cur = np.zeros(n, dtype=np.int32)
    t_last = np.zeros(n, dtype=np.int32)
    rc = load().orc_tracklets_depth(cur.h, last.h if last is not None else None, *[a.ctypes.data for a in arrs],
                                    isn.ctypes.data, n, d_cur.ctypes.data, d_last.ctypes.data, t_cur.ctypes.data,
                                    t_last.ctypes.data, n_threads)
    assert rc == 0, rc
    return d_cur, d_last, t_cur, t_last


# ---- component functions -------------------------------------------------------------------------
def filter_points_min_dist_blob(depths, bin_width, min_count):
    d = _f64(depths)
    keep = np.empty(max(1, d.size), dtype=np.int32)
    n = C.c_int32(0)
    lo, hi = C.c_double(0), C.c_double(0)
    ok = load().orc_filter_points_min_dist_blob(d.ctypes.data, d.size, bin_width, min_count, keep.ctypes.data,
                                                C.byref(n), C.byref(lo), C.byref(hi))
    return bool(ok), keep[:n.value].copy(), lo.value, hi.value


def get_nearest_point(depths, neighbors_index):
    d = _f64(depths)
    idx = np.ascontiguousarray(neighbors_index, dtype=np.int32)
    return int(load().orc_get_nearest_point(d.ctypes.data, idx.ctypes.data, d.size))


def max_spanning_triangle(points, thr=0.0):
    p = _f64(points)  # [n,3]
    out = np.empty(3, dtype=np.int32)
    ok = load().orc_max_spanning_triangle(p.ctypes.data, p.shape[0], thr, out.ctypes.data)
    return bool(ok), out


def check_planar(c1, c2, c3, thr):
    a, b, c = _f64(c1), _f64(c2), _f64(c3)
    return bool(load().orc_check_planar(a.ctypes.data, b.ctypes.data, c.ctypes.data, thr))


def intersect_triangle(p1, p2, p3, n0, n1, orth_thr):
    arrs = [_f64(x) for x in (p1, p2, p3, n0, n1)]
    pt = np.empty(3)
    depth = C.c_double(0)
    ok = load().orc_intersect_triangle(*[a.ctypes.data for a in arrs], orth_thr, pt.ctypes.data, C.byref(depth))
    return bool(ok), pt, depth.value


def threshold_global(params, depth):
    d = C.c_double(depth)
    r = load().orc_threshold_global(C.byref(params), C.byref(d))
    return int(r), d.value


def threshold_local(params, points, depth):
    p = _f64(points)
    d = C.c_double(depth)
    r = load().orc_threshold_local(C.byref(params), p.ctypes.data, p.shape[0], C.byref(d))
    return int(r), d.value


def mestimator_plane(points, prior_n, prior_offset):
    p = _f64(points)
    pn = _f64(prior_n)
    n = np.empty(3)
    off = C.c_double(0)
    load().orc_mestimator_plane(p.ctypes.data, p.shape[0], pn.ctypes.data, prior_offset, n.ctypes.data, C.byref(off))
    return n, off.value


def pca(params, points):
    p = _f64(points)
    n, m = np.empty(3), np.empty(3)
    r = load().orc_pca(C.byref(params), p.ctypes.data, p.shape[0], n.ctypes.data, m.ctypes.data)
    return int(r), n, m


def histogram_counts(values, bin_width, bin_count):
    v = _f64(values)
    out = np.zeros(bin_count, dtype=np.int32)
    load().orc_histogram_counts(v.ctypes.data, v.size, float(bin_width), int(bin_count), out.ctypes.data)
    return out


REF_LIB_PATH = _HERE / "_ref" / "libmld_ref.so"


def load_reference_parts():
    """oracle/_ref/libmld_ref.so: Histogram.cpp and TresholdDepthGlobal.cpp of the reference, compiled from
    /root/reference by `make -C oracle ref` (only where that tree exists).  None if it has not been built."""
    if not REF_LIB_PATH.exists():
        return None
    lib = C.CDLL(str(REF_LIB_PATH))
    lib.ref_histogram_counts.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
    lib.ref_threshold_global.argtypes = [C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
    return lib
