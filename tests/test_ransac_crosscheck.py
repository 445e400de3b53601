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


@pytest.mark.parametrize("seed", [0, 1, 3, 5, 9, 12])
def test_refinement_partial_sums_stay_close_to_the_reference_summation_order(seed, monkeypatch):
    """The refinement (optimizeModelCoefficients) sums its nine float32 moments in 256 interleaved partials - in the kernel
    and in both restatements -, PCL's computeMeanAndCovarianceMatrix in ONE sequential float32 chain per moment.  Nothing
    pins one association against the other bit for bit, so this BOUNDS the drift (as tests/test_semantic_plane.py does for
    the semantic plane): the same RANSAC model and inlier set refined in PCL's order give the same plane to 3e-8
    (normal) / 2e-6 m (offset); the reference's own test accepts +-0.2 (test_monolidar_fusion.cpp:436-439)."""
    P = capi.params_c0()
    scanner = synth.VLP16 if seed % 2 else synth.HDL64_KITTI
    cloud = synth.make_cloud(scanner, seed=30 + seed, frame=seed % 5)
    c_tree, inl_tree = np_restatement.ransac_plane(cloud, P, seed)
    sequential = lambda values, partials=256: np.add.accumulate(values.astype(np.float32), dtype=np.float32)[-1]  # noqa: E731
    monkeypatch.setattr(np_restatement, "_f32_partials_sum", sequential)
    c_seq, inl_seq = np_restatement.ransac_plane(cloud, P, seed)
    assert np.array_equal(inl_tree, inl_seq)          # (the inlier set is the unrefined model's: same draws)
    assert inl_tree.size > 1000
    sgn = 1.0 if np.dot(c_tree[:3], c_seq[:3]) > 0 else -1.0
    assert np.abs(c_tree[:3] - sgn * c_seq[:3]).max() < 2e-5, (c_tree, c_seq)   # observed: 3e-9 ... 3e-8
    assert abs(c_tree[3] - sgn * c_seq[3]) < 2e-4, (c_tree, c_seq)              # metres; observed: 2e-7 ... 1.6e-6
    assert not np.array_equal(c_tree, c_seq) or inl_tree.size < 256             # (the two orders do differ in the last bits)
