"""COMPILE EVIDENCE for the Eigen-typed overloads of the C++ shim (host/monolidar_fusion/DepthEstimator.h, the
`MLD_HAVE_EIGEN` block) - the signatures tracklets_depth binds (tracklet_depth_module.cpp:80,115; reference
DepthEstimator.h:174-220).  This image has no Eigen, so they are compiled against a tests-only stand-in
(tests/stubs/Eigen: just the members the overloads and their call sites touch).  That proves the overloads are
well-formed C++ and forward to the C-ABI correctly; it says nothing about real Eigen and is NOT a parity pin."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth

from helpers import assert_depth_parity, run_oracle

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "mono_lidar_depth_amd" / "host"
LIB = ROOT / "mono_lidar_depth_amd" / "lib"


def _build(tmp_path):
    exe = tmp_path / "eigen_overloads_demo"
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", f"-I{ROOT / 'tests' / 'stubs'}", f"-I{ROOT / 'include'}",
           f"-I{HOST}", "-o", str(exe), str(ROOT / "tests" / "cpp" / "eigen_overloads_demo.cpp"), f"-L{LIB}", "-lmld_hip",
           f"-Wl,-rpath,{LIB}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def test_eigen_overloads_compile_and_link(tmp_path):
    """The `#ifdef MLD_HAVE_EIGEN` overloads are compiled (the demo #errors if they are not) and link against the C-ABI."""
    assert _build(tmp_path).exists()


@pytest.mark.gpu
def test_eigen_overloads_forward_to_the_c_abi(tmp_path):
    exe = _build(tmp_path)
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=21, frame=2, stride_floats=8)  # pcl::PointXYZI layout
    uv = synth.make_features(700, seed=21)
    coeffs, inl = synth.make_ground_plane(cloud)
    (tmp_path / "cloud.bin").write_bytes(cloud.tobytes())
    (tmp_path / "uv.bin").write_bytes(uv.tobytes())
    (tmp_path / "inl.bin").write_bytes(inl.tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([str(exe), str(tmp_path / "cloud.bin"), str(tmp_path / "uv.bin"), str(tmp_path / "inl.bin"),
                        str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"eigen_overloads ok features {uv.shape[0]} same3 1" in r.stdout
    raw = out.read_bytes()
    F = uv.shape[0]
    off = 0

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(raw[off:off + n * np.dtype(dtype).itemsize], dtype=dtype)
        off += a.nbytes
        return a

    d4 = take(np.float64, F)
    d5, t5 = take(np.float64, F), take(np.int32, F)
    dF, tF = take(np.float64, F), take(np.int32, F)
    d1, t1 = take(np.float64, 1), take(np.int32, 1)
    coeffs[3] = np.float32(1.73)
    _, (d0, t0) = run_oracle(capi.params_c0(), cloud, uv, (coeffs, inl))
    assert_depth_parity(d5, t5, d0, t0)
    assert_depth_parity(dF, tF, d0, t0)
    assert np.array_equal(d4, d5, equal_nan=True)
    assert t1[0] == t0[0] and (abs(d1[0] - d0[0]) <= 1e-4 or (d1[0] == -1.0 and d0[0] == -1.0))
