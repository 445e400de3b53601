"""The C++ shim class (reference method names over the C-ABI) driven the way tracklets_depth drives it."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth

from helpers import assert_depth_parity, run_oracle

ROOT = Path(__file__).resolve().parent.parent
DEMO = ROOT / "mono_lidar_depth_amd" / "lib" / "mld_shim_demo"


def test_shim_demo_is_built():
    assert DEMO.exists(), "run __graft_entry__.build()"


@pytest.mark.gpu
@pytest.mark.parametrize("with_plane", [True, False])
def test_cpp_shim_matches_oracle(tmp_path, with_plane):
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=12, frame=1, stride_floats=8)  # pcl::PointXYZI layout
    uv = synth.make_features(900, seed=12)
    coeffs, inl = synth.make_ground_plane(cloud)
    (tmp_path / "cloud.bin").write_bytes(cloud.tobytes())
    (tmp_path / "uv.bin").write_bytes(uv.tobytes())
    (tmp_path / "inl.bin").write_bytes(inl.tobytes())
    out = tmp_path / "out.bin"
    r = subprocess.run([str(DEMO), "-", str(tmp_path / "cloud.bin"), str(tmp_path / "uv.bin"),
                        str(tmp_path / "inl.bin") if with_plane else "-", str(out)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    assert "usage_error_ok 1" in r.stdout
    raw = out.read_bytes()
    F = uv.shape[0]
    depth = np.frombuffer(raw[:8 * F], dtype=np.float64)
    types = np.frombuffer(raw[8 * F:], dtype=np.int32)
    P = capi.params_c0() if with_plane else capi.params_c0().replace(do_use_ransac_plane=0)
    coeffs[3] = np.float32(1.73)
    _, (d0, t0) = run_oracle(P, cloud, uv, (coeffs, inl) if with_plane else None)
    assert_depth_parity(depth, types, d0, t0)
    # debug-mode leg of the demo: identical result types, 3 corners per found triangle, one interpolated point per
    # valid depth, the ground-plane cloud of the supplied inliers, the reference's always-empty neighbour cloud
    import re
    m = re.search(r"debug same_types (\d+) corners (\d+) plane (\d+) interpolated (\d+) valid (\d+) neighbors (\d+) "
                  r"camcs (\d+)", r.stdout)
    assert m, r.stdout
    same, n_corners, n_plane, n_interp, n_valid, n_nb, n_cam = map(int, m.groups())
    assert same == 1 and n_corners % 3 == 0 and n_corners > 0
    assert n_interp == n_valid == int((d0 >= 0).sum())
    assert n_plane == (np.unique(inl).size if with_plane else 0)
    assert n_nb == 0 and n_cam == cloud.shape[0]
    if with_plane:
        m = re.search(r"semantic segmented (\d+) inliers (\d+) nz (\S+)", r.stdout)
        assert m, r.stdout
        assert int(m.group(1)) == 1 and int(m.group(2)) > 1000 and abs(abs(float(m.group(3))) - 1.0) < 0.05
