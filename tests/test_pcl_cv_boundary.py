"""COMPILE EVIDENCE for the shim's boundary in the reference's own cloud / image types (host/monolidar_fusion/
DepthEstimator.h: `MLD_HAVE_PCL`, `MLD_HAVE_OPENCV`, `MLD_HAVE_EIGEN`): Cloud = pcl::PointCloud<pcl::PointXYZI>
(reference DepthEstimator.h:62-63,93,174-220), SemanticPlane(const cv::Mat&, Camera, std::set<int>, double)
(RansacPlane.h:173-193), driven as tracklets_depth drives them (tracklet_depth_module.cpp:63-117, 269-284).  This image
has no PCL / OpenCV / Eigen: the demo is compiled against tests-only stand-ins (tests/stubs/{pcl,opencv2,Eigen}: just the
members the shim and its call sites touch).  That proves the overloads are well-formed C++ and forward to the C-ABI
correctly; it says nothing about the real libraries and is NOT a parity pin.  On the GPU the demo's numbers are compared
with the oracle."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth

from helpers import make_oracle

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "mono_lidar_depth_amd" / "host"
LIB = ROOT / "mono_lidar_depth_amd" / "lib"


def _build(tmp_path):
    exe = tmp_path / "pcl_cv_boundary_demo"
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", f"-I{ROOT / 'tests' / 'stubs'}", f"-I{ROOT / 'include'}",
           f"-I{HOST}", "-o", str(exe), str(ROOT / "tests" / "cpp" / "pcl_cv_boundary_demo.cpp"), f"-L{LIB}", "-lmld_hip",
           f"-Wl,-rpath,{LIB}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def test_pcl_and_cv_overloads_compile_and_link(tmp_path):
    """The pcl / cv::Mat / Eigen typed boundary is compiled (the demo #errors and static_asserts if it is not)."""
    assert _build(tmp_path).exists()


@pytest.mark.gpu
def test_pcl_and_cv_overloads_forward_to_the_c_abi(tmp_path):
    exe = _build(tmp_path)
    P = capi.params_c0()
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=31, frame=3, stride_floats=8)   # pcl::PointXYZI records
    cloud_last = synth.make_cloud(synth.HDL64_KITTI, seed=31, frame=2, stride_floats=8)
    uv = np.floor(synth.make_features(900, seed=31))   # integer pixels, as the tracklet caller passes them (:75-76)
    img = synth.make_label_image(cloud)
    (tmp_path / "cloud.bin").write_bytes(cloud.tobytes())
    (tmp_path / "cloud_last.bin").write_bytes(cloud_last.tobytes())
    (tmp_path / "uv.bin").write_bytes(uv.tobytes())
    (tmp_path / "labels.bin").write_bytes(np.ascontiguousarray(img).tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe), str(tmp_path / "cloud.bin"), str(tmp_path / "cloud_last.bin"), str(tmp_path / "uv.bin"),
                        str(tmp_path / "labels.bin"), str(img.shape[0]), str(img.shape[1]), str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"pcl_cv_boundary ok features {uv.shape[0]} points {cloud.shape[0]}" in r.stdout and "refused 11" in r.stdout
    raw = out.read_bytes()
    F = uv.shape[0]
    d_cur = np.frombuffer(raw[:8 * F], dtype=np.float64)
    d_last = np.frombuffer(raw[8 * F:16 * F], dtype=np.float64)
    coeffs = np.frombuffer(raw[16 * F:16 * F + 16], dtype=np.float32)
    n_inl, hits = np.frombuffer(raw[16 * F + 16:16 * F + 24], dtype=np.int32)
    # current frame: SemanticPlane estimated inside the call
    ref = make_oracle(P)
    ref.set_cloud(cloud)
    c0, inl0 = ref.estimate_semantic_plane(img, (6, 7, 8, 9), P.ransac_plane_refinement_treshold)
    d0, t0 = ref.calculate_depth(uv, 8)
    assert np.array_equal(coeffs, c0) and n_inl == inl0.size and hits == inl0.size
    assert np.allclose(d_cur, d0, rtol=0, atol=1e-4) and (t0 == 16).sum() > 0
    # previous frame: null plane pointer -> a RansacPlane (seed 0) is created and estimated
    ref2 = make_oracle(P)
    ref2.set_cloud(cloud_last)
    ref2.estimate_ground_plane(0)
    d1, _ = ref2.calculate_depth(uv, 8)
    assert np.allclose(d_last, d1, rtol=0, atol=1e-4)
