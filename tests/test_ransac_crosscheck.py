"""RansacPlane::CalculateInliersPlane: the C++ restatement against an independently written NumPy restatement (same
draw convention, float32 plane model, PCL's sequential stopping rule).  CPU only."""
import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth
from oracle import np_restatement

from helpers import make_oracle


@pytest.mark.parametrize("seed,kw", [
    (0, {}), (1, {}), (7, dict(ransac_plane_use_refinement=0)),
    (3, dict(ransac_plane_min_z=-3.0, ransac_plane_max_z=-0.5)),
    (5, dict(ransac_plane_max_iterations=50, ransac_plane_probability=0.9)),
    (9, dict(ransac_plane_distance_treshold=0.05, ransac_plane_refinement_treshold=0.1)),
])
def test_cpp_ransac_matches_numpy_ransac(seed, kw):
    P = capi.params_c0().replace(**kw)
    scanner = synth.VLP16 if seed % 2 else synth.HDL64_KITTI
    cloud = synth.make_cloud(scanner, seed=30 + seed, frame=seed)
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    c_cpp, inl_cpp = ref.estimate_ground_plane(seed)
    c_np, inl_np = np_restatement.ransac_plane(cloud, P, seed)
    # identical draws and counts -> identical RANSAC model and inlier set; the refined coefficients differ only by the
    # eigen-solver (Jacobi / LAPACK)
    assert np.array_equal(inl_cpp, inl_np)
    sgn = 1.0 if np.dot(c_cpp[:3], c_np[:3]) > 0 else -1.0
    assert np.abs(c_cpp - sgn * c_np).max() < 2e-6
    assert abs(abs(c_cpp[2]) - 1.0) < 0.02 and abs(abs(c_cpp[3]) - 1.73) < 0.2  # the reference test's tolerance
