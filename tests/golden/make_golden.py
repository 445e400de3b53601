#!/usr/bin/env python3
"""Generates tests/golden/golden_frames.npz (run from the repo root: python tests/golden/make_golden.py).

The expected outputs come from the independent NumPy restatement (oracle/np_restatement.py) and are only written
if the C++ oracle agrees with it (integers exactly, depths to 1e-9 m).  The reference itself cannot be run here
(Eigen/PCL/OpenCV absent), so these vectors pin the two restatements and the HIP path to each
other, not to a reference binary: end-to-end parity stays "unpinned" as stated in DESIGN.md.

Cases (small camera 320x96, f=180; 32x512 cloud; features chosen so that every live result type appears):
  c0            parameters.yaml values (do_use_depth_segmentation 0)
  tight         global depth window [6, 20] m, absolute local tolerance 0.02 m  -> types 4, 5, 6, 7
  no_thresholds thresholds off, cut-behind-camera on                            -> type 10 (intersection behind the camera)
  find_by_pixel the reference's NeigborFinder.findByPixel layout (100x100, f=600, window 3x5, 50 points)
  vlp16_int     BASELINE config 3 in small: 16-ring cloud (rows of points 10+ pixels apart), integer-pixel features as the
                tracklet caller passes them (tracklet_depth_module.cpp:75-76), thresholds in Adjust mode, relative local
  dense_int     BASELINE config 5 in small: 128-ring cloud cropped to the forward sector, integer-pixel features: windows
                with dozens of neighbours (the long-list path of the HIP build), many points per pixel (first-wins map)
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from mono_lidar_depth_amd import capi, synth  # noqa: E402
from oracle import oracle  # noqa: E402
from oracle.np_restatement import NpDepthEstimator  # noqa: E402

CAM = capi.MldCamera(180.0, 160.0, 40.0, 320, 96)
SCANNER = synth.Scanner(32, 512, 2.0, -24.9)


def run_case(name, P, cam, T, cloud, plane, candidates, per_type=6, max_features=64, cloud_from=None):
    npo = NpDepthEstimator(P, cam, T)
    npo.set_cloud(cloud)
    npo.set_ground_plane(*plane) if plane is not None else npo.set_ground_plane(None, None)
    d, t, tr = npo.calculate_depth(candidates)
    # pick up to per_type features of every result type
    chosen = []
    for ty in sorted(set(t.tolist())):
        chosen += np.nonzero(t == ty)[0][:per_type].tolist()
    chosen = sorted(chosen)[:max_features]
    uv = candidates[chosen]
    d, t, tr = d[chosen], t[chosen], [tr[i] for i in chosen]

    ref = oracle.OracleDepthEstimator(P, cam, T)
    ref.set_cloud(cloud)
    ref.set_ground_plane(*plane) if plane is not None else ref.set_ground_plane(None, None)
    d0, t0 = ref.calculate_depth(uv)
    assert np.array_equal(t, t0), (name, t, t0)
    assert np.allclose(d, d0, rtol=0, atol=1e-9, equal_nan=True), name
    assert np.array_equal(ref.point_index(), npo.point_index) and np.array_equal(ref.pixel_map(), npo.pixel_map)

    def ragged(key):
        lists = [x.get(key, []) for x in tr]
        flat = np.array([v for l in lists for v in l], dtype=np.int32)
        off = np.cumsum([0] + [len(l) for l in lists]).astype(np.int32)
        return flat, off

    # parameters / camera as name -> value (robust against struct layout changes of the C-ABI)
    pj = json.dumps({n: getattr(P, n) for n, _ in capi.MldParams._fields_})
    cj = json.dumps({n: getattr(cam, n) for n, _ in capi.MldCamera._fields_})
    out = {"params_json": np.array(pj), "camera_json": np.array(cj),
           "T": np.asarray(T, dtype=np.float64), "uv": uv,
           "depth": d0, "type": t0,  # depths as produced by the C++ oracle (agreeing with NumPy to 1e-9)
           "point_index": npo.point_index, "pixel_map": npo.pixel_map}
    if cloud_from is None:
        out["cloud"] = cloud
    else:
        out["cloud_from"] = np.array(cloud_from)  # same cloud as that case (kept once to keep the fixture small)
    if plane is not None:
        out["plane_coeffs"], out["plane_inliers"] = plane
    for key in ("nb_idx", "seg_pos", "road_idx", "road_pos"):
        out[key + "_flat"], out[key + "_off"] = ragged(key)
    out["corners"] = np.array([x.get("corners", (-1, -1, -1)) for x in tr], dtype=np.int32)
    print(f"{name}: {len(uv)} features, types {dict(zip(*np.unique(t0, return_counts=True)))}")
    return {f"{name}/{k}": v for k, v in out.items()}


def main():
    rng = np.random.default_rng(2024)
    cloud = synth.make_cloud(SCANNER, seed=4, frame=2)
    plane = synth.make_ground_plane(cloud)
    cand = np.stack([rng.uniform(-4, CAM.width + 4, 4000), rng.uniform(20, CAM.height + 3, 4000)], axis=1)
    cand[::7] = np.floor(cand[::7])
    data = {}
    P0 = capi.params_c0()
    data.update(run_case("c0", P0, CAM, synth.T_CAM_LIDAR, cloud, plane, cand))
    Pt = P0.replace(treshold_depth_min=6, treshold_depth_max=20, treshold_depth_local_valuetype=0,
                    treshold_depth_local_value=0.02)
    data.update(run_case("tight", Pt, CAM, synth.T_CAM_LIDAR, cloud, plane, cand, cloud_from="c0"))
    Pn = P0.replace(treshold_depth_enabled=0, treshold_depth_local_enabled=0, do_check_triangleplanar_condition=0,
                    viewray_plane_orthoganality_treshold=0.0)
    data.update(run_case("no_thresholds", Pn, CAM, synth.T_CAM_LIDAR, cloud, None, cand, cloud_from="c0"))
    # the reference's findByPixel layout
    camf = capi.MldCamera(600.0, 50.0, 50.0, 100, 100)
    Pf = P0.replace(pixelarea_search_witdh=3, pixelarea_search_height=5, do_use_ransac_plane=0)
    Tf = np.hstack([np.eye(3), np.zeros((3, 1))])
    pix = rng.integers(0, 10, size=(50, 2)).astype(np.float64) + 0.25
    dep = rng.integers(1, 11, size=50).astype(np.float64)
    rays = np.stack([(pix[:, 0] - 50) / 600, (pix[:, 1] - 50) / 600, np.ones(50)], axis=1)
    rays /= np.linalg.norm(rays, axis=1, keepdims=True)
    cl = np.zeros((50, 4), np.float32)
    cl[:, :3] = rays * dep[:, None]
    data.update(run_case("find_by_pixel", Pf, camf, Tf, cl, None, pix, per_type=50, max_features=50))
    # config 3 / config 5 shaped mini-cases (integer-pixel features)
    candi = np.floor(np.stack([rng.uniform(0, CAM.width, 3000), rng.uniform(0, CAM.height, 3000)], axis=1))
    cv = synth.make_cloud(synth.Scanner(16, 450, 15.0, -15.0), seed=5, frame=1)
    Pv = P0.replace(treshold_depth_mode=1, treshold_depth_local_mode=1, treshold_depth_local_valuetype=1,
                    treshold_depth_min=3, treshold_depth_max=40)
    data.update(run_case("vlp16_int", Pv, CAM, synth.T_CAM_LIDAR, cv, synth.make_ground_plane(cv), candi))
    cd = synth.make_cloud(synth.Scanner(128, 1536, 15.0, -25.0), seed=6, frame=0)
    fwd = np.nan_to_num(cd[:, 0], nan=-1.0) > np.abs(np.nan_to_num(cd[:, 1], nan=1e9)) * 0.6  # forward sector, order kept
    cd = np.ascontiguousarray(cd[fwd])
    data.update(run_case("dense_int", P0, CAM, synth.T_CAM_LIDAR, cd, synth.make_ground_plane(cd), candi, per_type=8))
    out = Path(__file__).resolve().parent / "golden_frames.npz"
    np.savez_compressed(out, **data)
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
