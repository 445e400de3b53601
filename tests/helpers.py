"""Shared helpers of the parity tests: build the HIP estimator and the oracle on the same inputs."""
import numpy as np

from mono_lidar_depth_amd import CameraPinhole, DepthEstimator, GroundPlane, capi, synth
from oracle import oracle

# Parity bar (BASELINE.json north_star): depths within 1e-4 m, identical result types.
DEPTH_TOL_M = 1e-4


def kitti_camera():
    return CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)


def make_estimator(params, camera=None, T=None, **kw):
    est = DepthEstimator(device=0, **kw)
    est.InitConfig(params)
    est.Initialize(camera or kitti_camera(), synth.T_CAM_LIDAR if T is None else T)
    return est


def make_oracle(params, camera=None, T=None):
    cam = camera or kitti_camera()
    return oracle.OracleDepthEstimator(params, cam.as_struct(), synth.T_CAM_LIDAR if T is None else T)


def run_oracle(params, cloud, uv, plane, camera=None, T=None, n_threads=1):
    ref = make_oracle(params, camera, T)
    ref.set_cloud(cloud)
    if plane is None:
        ref.set_ground_plane(None, None)
    else:
        ref.set_ground_plane(*plane)
    return ref, ref.calculate_depth(uv, n_threads)


def assert_depth_parity(depth, types, depth_ref, types_ref, exact_main=True):
    depth, types = np.asarray(depth), np.asarray(types)
    assert np.array_equal(types, types_ref), (
        f"{(types != types_ref).sum()} result types differ: "
        f"{list(zip(types[types != types_ref][:10], types_ref[types != types_ref][:10]))}")
    nan_a, nan_b = np.isnan(depth), np.isnan(depth_ref)
    assert np.array_equal(nan_a, nan_b)
    diff = np.abs(np.where(nan_a, 0.0, depth) - np.where(nan_b, 0.0, depth_ref))
    assert diff.max(initial=0.0) <= DEPTH_TOL_M, f"max |depth - oracle| = {diff.max():.3e} m"
    if exact_main:
        # the triangle path has no reordered sums: bit-exact with the oracle
        main = types_ref != 16
        assert np.array_equal(depth[main], depth_ref[main]), (
            f"main-path depths not bit-exact: max diff {diff[main].max():.3e}")
    return diff
