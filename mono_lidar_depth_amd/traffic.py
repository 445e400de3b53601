"""Algorithmic byte counts of the path, from measured per-frame statistics.

    per frame   : B_A = 16*N + 4*W*H + 28*Nvis        (cloud read as float4, map clear, map entry + cam xyz per
                                                      visible point)                      -> k_project_scatter
    per feature : B_f = 16 + 4*P1 + 24*k1 + 12        (uv, window cells, neighbours, depth+type)
                        [+ 4*P2 + 25*k2 if the feature enters the road fallback]          -> k_feature_depth

These are the bytes the ALGORITHM needs, not what the HIP kernels move: the kernels avoid the map clear (tagged
keys) and the camera-frame copy (neighbours are re-derived from the raw point), so the achieved "algorithmic
GB/s" is allowed to exceed what the HBM counters see.
"""
from __future__ import annotations

import numpy as np


def window_bounds(uv: np.ndarray, half_x: float, half_y: float, W: int, H: int):
    """Integer window bounds exactly as NeighborFinderPixel::getNeighbors (NeighborFinderPixel.cpp:67-76)."""
    u, v = uv[:, 0], uv[:, 1]
    left = np.maximum(u - half_x, 0.0)
    right = np.minimum(u + half_x, float(W - 1))
    top = np.maximum(v - half_y, 0.0)
    bottom = np.minimum(v + half_y, float(H - 1))
    x0, x1 = np.trunc(left).astype(np.int64), np.trunc(right).astype(np.int64)
    y0, y1 = np.trunc(top).astype(np.int64), np.trunc(bottom).astype(np.int64)
    return x0, x1, y0, y1


def neighbour_counts(pixel_map: np.ndarray, uv: np.ndarray, half_x: float, half_y: float):
    """Number of occupied map cells in every feature's window (integral image)."""
    H, W = pixel_map.shape
    occ = (pixel_map >= 0).astype(np.int64)
    ii = np.zeros((H + 1, W + 1), dtype=np.int64)
    ii[1:, 1:] = occ.cumsum(0).cumsum(1)
    x0, x1, y0, y1 = window_bounds(uv, half_x, half_y, W, H)
    empty = (x1 < x0) | (y1 < y0)
    x0c, x1c = np.clip(x0, 0, W - 1), np.clip(x1, 0, W - 1)
    y0c, y1c = np.clip(y0, 0, H - 1), np.clip(y1, 0, H - 1)
    cnt = ii[y1c + 1, x1c + 1] - ii[y0c, x1c + 1] - ii[y1c + 1, x0c] + ii[y0c, x0c]
    return np.where(empty, 0, cnt)


def frame_bytes(params, width: int, height: int, n_points: int, n_visible: int, pixel_map: np.ndarray,
                uv: np.ndarray, types: np.ndarray, has_plane: bool = True):
    """Algorithmic bytes of one frame, split by kernel, plus the statistics they were derived from."""
    w, h = params.pixelarea_search_witdh, params.pixelarea_search_height
    hx1, hy1 = w * 0.5, h * 0.5
    hx2, hy2 = w * 0.5 * 2.0, h * 0.5 * 1.5
    P1 = (w + 1) * (h + 1)
    P2 = (int(np.floor(2 * hx2)) + 1) * (int(np.floor(2 * hy2)) + 1)
    k1 = neighbour_counts(pixel_map, uv, hx1, hy1)
    k2 = neighbour_counts(pixel_map, uv, hx2, hy2)
    types = np.asarray(types)
    road_on = bool(params.do_use_ransac_plane) and has_plane
    fallback = road_on & (types != 1) & (k1 >= max(params.radiusSearch_count_min, 0))
    b_project = 16 * n_points + 4 * width * height + 28 * n_visible
    b_road = int((fallback * (4 * P2 + 25 * k2)).sum())
    b_feature = int((16 + 4 * P1 + 24 * k1 + 12).sum()) + b_road
    return {
        "project_bytes": int(b_project),
        "feature_bytes": int(b_feature),
        "road_bytes": int(b_road),
        "n_visible": int(n_visible),
        "k1_mean": float(k1.mean()) if k1.size else 0.0,
        "k2_mean_fallback": float(k2[fallback].mean()) if fallback.any() else 0.0,
        "fallback_features": int(fallback.sum()),
        "P1": int(P1),
        "P2": int(P2),
    }
