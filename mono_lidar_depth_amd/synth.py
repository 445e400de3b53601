"""Seeded synthetic LiDAR frames / features / ground planes.

Shared by the parity tests, the CPU baseline and the GPU benchmark.  A spinning scanner (R rings x A azimuth
steps, ring-major order) ray-casts a scene of a ground plane, 12 axis-aligned boxes and enclosing walls; range
noise N(0, 0.02 m); 2 % drop-outs are emitted as NaN points so that N = R*A stays fixed (a non-dense PCL cloud).
No file, dataset or reference code is involved.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# KITTI-like camera
KITTI_W, KITTI_H = 1242, 375
KITTI_F, KITTI_CU, KITTI_CV = 721.5377, 609.5593, 172.854
GROUND_Z = -1.73

# lidar (x fwd, y left, z up) -> camera (x right, y down, z fwd): (x,y,z)_l -> (-y, -z, x); t = (0, -0.08, -0.27)
T_CAM_LIDAR = np.array([[0.0, -1.0, 0.0, 0.0],
                        [0.0, 0.0, -1.0, -0.08],
                        [1.0, 0.0, 0.0, -0.27]], dtype=np.float64)


@dataclass(frozen=True)
class Scanner:
    rings: int
    azimuth_steps: int
    elev_top_deg: float
    elev_bottom_deg: float
    max_range: float = 120.0


HDL64 = Scanner(64, 2048, 2.0, -24.9)          # config 2: 131 072 points
HDL64_KITTI = Scanner(64, 1875, 2.0, -24.9)    # config 1: ~120 k points
VLP16 = Scanner(16, 1800, 15.0, -15.0)         # config 3: 28 800 points
DENSE128 = Scanner(128, 4096, 15.0, -25.0)     # config 5: 524 288 points


def make_scene(seed: int):
    """12 boxes on the ground (centres 5..60 m ahead / +-20 m lateral) + walls."""
    rng = np.random.default_rng(1000 + seed)
    cx = rng.uniform(5.0, 60.0, 12)
    cy = rng.uniform(-20.0, 20.0, 12)
    sx = rng.uniform(1.0, 5.0, 12)
    sy = rng.uniform(1.0, 5.0, 12)
    h = rng.uniform(1.0, 3.0, 12)
    lo = np.stack([cx - sx / 2, cy - sy / 2, np.full(12, GROUND_Z)], axis=1)
    hi = np.stack([cx + sx / 2, cy + sy / 2, GROUND_Z + h], axis=1)
    walls_lo = np.array([-80.0, -80.0, GROUND_Z - 1.0])
    walls_hi = np.array([180.0, 80.0, 40.0])
    return lo, hi, walls_lo, walls_hi


def make_cloud(scanner: Scanner, seed: int = 0, frame: int = 0, stride_floats: int = 4) -> np.ndarray:
    """One frame as float32 [N, stride_floats]: x,y,z,(pad),intensity(...) in the lidar frame.

    stride_floats = 4: packed xyzi; 8: pcl::PointXYZI memory layout (x,y,z,pad,intensity,pad,pad,pad).
    The sensor advances 1 m per frame along x through the scene `seed`.
    """
    lo, hi, wlo, whi = make_scene(seed)
    R, A = scanner.rings, scanner.azimuth_steps
    el = np.deg2rad(np.linspace(scanner.elev_top_deg, scanner.elev_bottom_deg, R))
    az = 2.0 * np.pi * np.arange(A) / A
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    d = np.stack([ce * np.cos(az)[None, :], ce * np.sin(az)[None, :], np.broadcast_to(se, (R, A))], axis=-1)
    d = d.reshape(-1, 3)  # ring-major
    o = np.array([float(frame), 0.0, 0.0])
    n = d.shape[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        # ground
        t_ground = np.where(d[:, 2] < 0, (GROUND_Z - o[2]) * inv[:, 2], np.inf)
        # walls: exit distance of the enclosing box
        t1 = (wlo[None, :] - o[None, :]) * inv
        t2 = (whi[None, :] - o[None, :]) * inv
        t_wall = np.nanmin(np.maximum(t1, t2), axis=1)
        best = np.minimum(t_ground, t_wall)
        # boxes: entry distance (slab test)
        for b in range(lo.shape[0]):
            ta = (lo[b][None, :] - o[None, :]) * inv
            tb = (hi[b][None, :] - o[None, :]) * inv
            tn = np.nanmax(np.minimum(ta, tb), axis=1)
            tf = np.nanmin(np.maximum(ta, tb), axis=1)
            hit = (tn <= tf) & (tn > 0)
            best = np.where(hit & (tn < best), tn, best)
    rng = np.random.default_rng(seed * 100003 + frame)
    rngs = best + rng.normal(0.0, 0.02, n)
    valid = np.isfinite(best) & (best < scanner.max_range) & (best > 0.5)
    valid &= rng.random(n) >= 0.02  # drop-outs
    pts = d * rngs[:, None]
    out = np.zeros((n, stride_floats), dtype=np.float32)
    out[:, :3] = pts.astype(np.float32)
    out[:, 4 if stride_floats == 8 else 3] = rng.uniform(0.0, 1.0, n).astype(np.float32)
    out[~valid, :3] = np.nan
    return out


def make_features(n: int, seed: int = 0, integer: bool = False, width: int = KITTI_W, height: int = KITTI_H) -> np.ndarray:
    """[F, 2] float64 features, uniform over the image (integer-valued for the tracklet API, config 5)."""
    rng = np.random.default_rng(77000 + seed)
    uv = np.stack([rng.uniform(0, width, n), rng.uniform(0, height, n)], axis=1)
    if integer:
        uv = np.floor(uv)
    return np.ascontiguousarray(uv, dtype=np.float64)


def make_features_near_points(cloud: np.ndarray, n: int, seed: int = 0, jitter_u: float = 3.0, jitter_v: float = 2.0,
                              width: int = KITTI_W, height: int = KITTI_H) -> np.ndarray:
    """[F, 2] float64 features scattered around the image positions of visible LiDAR returns (uniform jitter of a few
    pixels): with a sparse scanner (VLP-16: rings ~25 px apart, search window 9 px high) uniformly random features almost
    never see a neighbour, these always do - the variant of BASELINE config 3 that exercises the paths behind the
    neighbour search (collinear triangles, planarity / orthogonality rejections, thresholds, road fallback)."""
    rng = np.random.default_rng(88000 + seed)
    xyz = cloud[:, :3].astype(np.float64)
    cam = xyz @ T_CAM_LIDAR[:, :3].T + T_CAM_LIDAR[:, 3]
    with np.errstate(all="ignore"):
        u = cam[:, 0] / cam[:, 2] * KITTI_F + KITTI_CU
        v = cam[:, 1] / cam[:, 2] * KITTI_F + KITTI_CV
        vis = np.nonzero((cam[:, 2] > 0.5) & (u > 0) & (u < width) & (v > 0) & (v < height))[0]
    if vis.size == 0:
        return make_features(n, seed, width=width, height=height)
    pick = rng.choice(vis, n, replace=True)
    uv = np.stack([u[pick] + rng.uniform(-jitter_u, jitter_u, n), v[pick] + rng.uniform(-jitter_v, jitter_v, n)], axis=1)
    uv[:, 0] = np.clip(uv[:, 0], 0.0, width - 1e-6)
    uv[:, 1] = np.clip(uv[:, 1], 0.0, height - 1e-6)
    return np.ascontiguousarray(uv, dtype=np.float64)


def make_features_k_neighbours(cloud: np.ndarray, n: int, seed: int = 0, min_neighbours: int = 6,
                               window=(6, 9), width: int = KITTI_W, height: int = KITTI_H) -> np.ndarray:
    """[F, 2] float64 features on (sub-pixel jittered) image positions of LiDAR returns whose search window - `window` =
    (pixelarea_search_witdh, pixelarea_search_height), NeighborFinderPixel.cpp:67-88 - holds at least `min_neighbours`
    returns: BASELINE config 2 "at its stated neighbour count" (k = 7; a uniformly random feature sees 2-3 on a 64 x 2048
    scan, and 40 % of them none).  The pixel occupancy is computed here with the path's own rule (first point per pixel,
    z > 0, strict image bounds)."""
    rng = np.random.default_rng(99000 + seed)
    xyz = cloud[:, :3].astype(np.float64)
    cam = xyz @ T_CAM_LIDAR[:, :3].T + T_CAM_LIDAR[:, 3]
    with np.errstate(all="ignore"):
        u = (KITTI_F * cam[:, 0] + KITTI_CU * cam[:, 2]) / cam[:, 2]
        v = (KITTI_F * cam[:, 1] + KITTI_CV * cam[:, 2]) / cam[:, 2]
        vis = np.nonzero((cam[:, 2] > 0) & (u > 0) & (u < width) & (v > 0) & (v < height))[0]
    if vis.size == 0:
        return make_features(n, seed, width=width, height=height)
    xi, yi = u[vis].astype(np.int64), v[vis].astype(np.int64)
    occ = np.zeros((height, width), dtype=np.int64)
    occ[yi, xi] = 1
    ii = np.zeros((height + 1, width + 1), dtype=np.int64)
    ii[1:, 1:] = occ.cumsum(0).cumsum(1)
    hx, hy = 0.5 * window[0], 0.5 * window[1]
    # candidate features: every return's own pixel at a random sub-pixel position; the window of THAT position is counted
    # (bounds exactly as NeighborFinderPixel.cpp:67-76: clamped doubles, truncation)
    fu, fv = xi + rng.uniform(0.05, 0.95, xi.size), yi + rng.uniform(0.05, 0.95, yi.size)
    x0 = np.maximum(fu - hx, 0.0).astype(np.int64)
    x1 = np.minimum(fu + hx, width - 1.0).astype(np.int64)
    y0 = np.maximum(fv - hy, 0.0).astype(np.int64)
    y1 = np.minimum(fv + hy, height - 1.0).astype(np.int64)
    k = ii[y1 + 1, x1 + 1] - ii[y0, x1 + 1] - ii[y1 + 1, x0] + ii[y0, x0]
    good = np.nonzero(k >= min_neighbours)[0]
    if good.size == 0:
        good = np.arange(vis.size)
    pick = rng.choice(good, n, replace=True)
    return np.ascontiguousarray(np.stack([fu[pick], fv[pick]], axis=1), dtype=np.float64)


def make_ground_plane(cloud: np.ndarray, subsample: int = 0, seed: int = 0):
    """Analytic plane of the scene in the lidar frame (0,0,1,1.73) + inliers |z + 1.73| < 0.3."""
    z = cloud[:, 2]
    inl = np.nonzero(np.abs(z - GROUND_Z) < 0.3)[0].astype(np.int32)
    if subsample and inl.size > subsample:
        rng = np.random.default_rng(5150 + seed)
        inl = np.sort(rng.choice(inl, subsample, replace=False)).astype(np.int32)
    coeffs = np.array([0.0, 0.0, 1.0, -GROUND_Z], dtype=np.float32)
    return coeffs, inl


def make_label_image(cloud: np.ndarray, width: int = KITTI_W, height: int = KITTI_H) -> np.ndarray:
    """Synthetic semantic label image (rows x cols uint8) for SemanticPlane, rendered from the frame itself: pixels
    hit by ground returns carry label 7 (road) or 8 (sidewalk, |y_lidar| > 6 m), pixels hit by anything else label 11,
    the rest 23 (sky).  Object returns are painted last, so that they occlude the ground as in a real segmentation."""
    xyz = cloud[:, :3].astype(np.float64)
    cam = xyz @ T_CAM_LIDAR[:, :3].T + T_CAM_LIDAR[:, 3]
    front = cam[:, 2] > 0.5
    u = np.where(front, cam[:, 0] / np.where(front, cam[:, 2], 1.0) * KITTI_F + KITTI_CU, -1.0)
    v = np.where(front, cam[:, 1] / np.where(front, cam[:, 2], 1.0) * KITTI_F + KITTI_CV, -1.0)
    inside = front & (u >= 0) & (u < width) & (v >= 0) & (v < height)
    ground = np.abs(xyz[:, 2] - GROUND_Z) < 0.1
    img = np.full((height, width), 23, dtype=np.uint8)

    def paint(sel, label, rx, ry):
        ix, iy = u[sel].astype(np.int64), v[sel].astype(np.int64)
        for dy in range(-ry, ry + 1):
            for dx in range(-rx, rx + 1):
                img[np.clip(iy + dy, 0, height - 1), np.clip(ix + dx, 0, width - 1)] = label

    paint(inside & ground & (np.abs(xyz[:, 1]) <= 6.0), 7, 2, 1)
    paint(inside & ground & (np.abs(xyz[:, 1]) > 6.0), 8, 2, 1)
    paint(inside & ~ground, 11, 1, 1)
    return img
