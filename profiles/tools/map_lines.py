#!/usr/bin/env python3
"""What k_project_scatter HAS to write per frame in this map layout (CPU only, no GPU): the distinct 64-byte and 128-byte
lines of the row-major pixel map (4 B per cell) and of the occupancy bitmap (column-of-words) that a frame's visible
points fall into.  A line that receives one 4-byte atomic is written back whole, so lines x line size - not entries x
4 B - is the floor of WRITE_SIZE for the scatter; compare with profiles/r5_summary.md (WRITE_SIZE per 1024-frame launch).

    python3 profiles/tools/map_lines.py [frames]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mono_lidar_depth_amd import synth  # noqa: E402

W, H = synth.KITTI_W, synth.KITTI_H
bm_stride = (H + 3 + 3) & ~3


def frame_lines(cloud):
    xyz = cloud[:, :3].astype(np.float64)
    cam = xyz @ synth.T_CAM_LIDAR[:, :3].T + synth.T_CAM_LIDAR[:, 3]
    with np.errstate(all="ignore"):
        u = (synth.KITTI_F * cam[:, 0] + synth.KITTI_CU * cam[:, 2]) / cam[:, 2]
        v = (synth.KITTI_F * cam[:, 1] + synth.KITTI_CV * cam[:, 2]) / cam[:, 2]
        vis = (cam[:, 2] > 0) & (u > 0) & (u < W) & (v > 0) & (v < H)
    xi, yi = u[vis].astype(np.int64), v[vis].astype(np.int64)
    cell = yi * W + xi
    word = (xi >> 5) * bm_stride + yi
    out = {"entries": int(vis.sum()), "cells": int(np.unique(cell).size)}
    for line in (64, 128):
        out[f"map_lines_{line}"] = int(np.unique(cell * 4 // line).size)
        out[f"bitmap_lines_{line}"] = int(np.unique(word * 4 // line).size)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for name, sc in (("config 2 (64 x 2048)", synth.HDL64), ("config 5 (128 x 4096)", synth.DENSE128), ("config 3 (16 x 1800)", synth.VLP16)):
        rows = [frame_lines(synth.make_cloud(sc, seed=0, frame=f)) for f in range(n)]
        m = {k: float(np.mean([r[k] for r in rows])) for k in rows[0]}
        print(f"{name}: {m['entries']:.0f} visible points -> {m['cells']:.0f} cells ({4 * m['cells'] / 1e3:.0f} KB of keys)")
        for line in (64, 128):
            b = (m[f'map_lines_{line}'] + m[f'bitmap_lines_{line}']) * line
            print(f"   {line:3d}-byte lines: map {m[f'map_lines_{line}']:.0f} + bitmap {m[f'bitmap_lines_{line}']:.0f} "
                  f"-> {b / 1e3:.0f} KB per frame, {b * 1024 / 1e6:.0f} MB per 1024 frames "
                  f"({b / (4 * m['cells']):.1f} x the key bytes)")


if __name__ == "__main__":
    main()
