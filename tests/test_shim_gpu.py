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
        # per-call plane of the feature-only overloads: null plane -> no road result in that call, the slot's plane
        # is used again afterwards (bit-identical results), another plane object is installed when it is handed in
        m = re.search(r"percall same_after_null (\d+) road_with_null (\d+) road_other (\d+) road_first (\d+)", r.stdout)
        assert m, r.stdout
        same_after, road_null, road_other, road_first = map(int, m.groups())
        assert same_after == 1 and road_null == 0 and road_first > 0 and 0 < road_other < road_first
        m = re.search(r"semantic segmented (\d+) inliers (\d+) nz (\S+) lazy (\d+) counted (\d+) flushed (\d+) "
                      r"ransac_pending (\d+) ransac_inliers (\d+) ransac_counted (\d+)", r.stdout)
        assert m, r.stdout
        assert int(m.group(1)) == 1 and int(m.group(2)) > 1000 and abs(abs(float(m.group(3))) - 1.0) < 0.05
        # the production call is one C call: the plane object had its coefficients and inlier COUNT at once, the index
        # list stayed on the GPU until the slot was reused by the next frame (then it was fetched: nothing is lost)
        assert int(m.group(4)) == 1 and int(m.group(5)) == int(m.group(2)) and int(m.group(6)) == 1
        assert int(m.group(7)) == 1 and int(m.group(8)) == int(m.group(9))
        # that next frame's RANSAC plane (seed 5) equals the restatement's
        ref, _ = run_oracle(P, cloud, uv, None)
        _, inl5 = ref.estimate_ground_plane(5)
        assert int(m.group(8)) == inl5.size


TRACKLET_DEMO = ROOT / "mono_lidar_depth_amd" / "lib" / "mld_tracklet_demo"


def test_tracklet_demo_is_built():
    assert TRACKLET_DEMO.exists(), "run __graft_entry__.build()"


@pytest.mark.gpu
def test_cpp_tracklet_module_matches_oracle(tmp_path):
    """tracklets_depth::TrackletDepthModule (C++ shim, plain-struct messages) over four frames: 10 % new tracks per
    frame, previous frame served from its resident slot; against the oracle's restatement of the module's marshalling."""
    from oracle import oracle
    from helpers import make_oracle
    P = capi.params_c0()
    rng = np.random.default_rng(8)
    n_tracks, n_frames = 3000, 4
    ids = np.arange(n_tracks, dtype=np.uint64)
    next_id = n_tracks
    frames = []
    for k in range(n_frames):
        cloud = synth.make_cloud(synth.HDL64_KITTI, seed=21, frame=2 * k, stride_floats=8)
        coeffs, inl = synth.make_ground_plane(cloud)
        if k > 0:
            repl = rng.choice(n_tracks, n_tracks // 10, replace=False)
            ids = ids.copy()
            ids[repl] = np.arange(next_id, next_id + repl.size, dtype=np.uint64)
            next_id += repl.size
        u0 = rng.uniform(-2, synth.KITTI_W + 2, n_tracks).astype(np.float32)
        v0 = rng.uniform(100, synth.KITTI_H + 2, n_tracks).astype(np.float32)
        u1 = (u0 + rng.normal(0, 3, n_tracks)).astype(np.float32)
        v1 = (v0 + rng.normal(0, 2, n_tracks)).astype(np.float32)
        rec = np.zeros(n_tracks, dtype=[("id", "<u8"), ("u0", "<f4"), ("v0", "<f4"), ("u1", "<f4"), ("v1", "<f4"), ("pad", "<f4"), ("pad2", "<f4")])
        rec["id"], rec["u0"], rec["v0"], rec["u1"], rec["v1"] = ids, u0, v0, u1, v1
        (tmp_path / f"cloud_{k}.bin").write_bytes(cloud.tobytes())
        (tmp_path / f"tracks_{k}.bin").write_bytes(rec.tobytes())
        (tmp_path / f"inl_{k}.bin").write_bytes(inl.tobytes())
        frames.append((cloud, (coeffs, inl), ids.copy(), u0, v0, u1, v1))
    r = subprocess.run([str(TRACKLET_DEMO), str(tmp_path), str(n_frames), "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    known, ref_last = set(), None
    for k, (cloud, (coeffs, inl), fids, u0, v0, u1, v1) in enumerate(frames):
        coeffs = coeffs.copy()
        coeffs[3] = np.float32(1.73)
        ref = make_oracle(P)
        ref.set_cloud(cloud)
        ref.set_ground_plane(coeffs, inl)
        is_new = np.array([int(i) not in known for i in fids])
        e_cur, e_last, _, _ = oracle.tracklets_depth(ref, ref_last, u0, v0, u1, v1, is_new, n_threads=8)
        out = np.frombuffer((tmp_path / f"out_{k}.bin").read_bytes(),
                            dtype=[("d0", "<f4"), ("d1", "<f4"), ("len", "<i4")])
        assert out.shape[0] == n_tracks
        assert np.allclose(out["d0"], e_cur, rtol=0, atol=1e-4)
        assert np.allclose(out["d1"][is_new], e_last[is_new], rtol=0, atol=1e-4)
        # a new tracklet holds two features; a continued one grows by one per frame (TidyUpTracklets drops the rest)
        assert (out["len"][is_new] == 2).all() and (out["len"][~is_new] >= 2).all()
        if k == 0:
            assert (out["d1"] == -1).all()  # no previous cloud (tracklet_depth_module.cpp:93-96)
        assert f"frame {k} tracks {n_tracks} stored {n_tracks}" in r.stdout
        known = set(int(i) for i in fids)
        ref_last = ref


@pytest.mark.gpu
def test_cpp_tracklet_module_foreign_plane_that_fails(tmp_path):
    """A caller's own GroundPlane subclass estimates itself on the CPU inside process(); on frame 2 it throws
    ExceptionPclInvalid (tracklet_depth_module.cpp:318-347): the current frame's depths are -1, the new tracks' features
    on the PREVIOUS frame are still answered from its resident slot, and frame 3 has no previous cloud any more."""
    from oracle import oracle
    from helpers import make_oracle
    P = capi.params_c0()
    rng = np.random.default_rng(18)
    n_tracks, n_frames = 1200, 4
    ids = np.arange(n_tracks, dtype=np.uint64)
    next_id = n_tracks
    frames = []
    for k in range(n_frames):
        cloud = synth.make_cloud(synth.HDL64_KITTI, seed=23, frame=2 * k, stride_floats=8)
        coeffs, inl = synth.make_ground_plane(cloud)
        if k > 0:
            repl = rng.choice(n_tracks, n_tracks // 5, replace=False)
            ids = ids.copy()
            ids[repl] = np.arange(next_id, next_id + repl.size, dtype=np.uint64)
            next_id += repl.size
        u0 = rng.uniform(0, synth.KITTI_W, n_tracks).astype(np.float32)
        v0 = rng.uniform(100, synth.KITTI_H, n_tracks).astype(np.float32)
        u1 = (u0 + rng.normal(0, 3, n_tracks)).astype(np.float32)
        v1 = (v0 + rng.normal(0, 2, n_tracks)).astype(np.float32)
        rec = np.zeros(n_tracks, dtype=[("id", "<u8"), ("u0", "<f4"), ("v0", "<f4"), ("u1", "<f4"), ("v1", "<f4"), ("pad", "<f4"), ("pad2", "<f4")])
        rec["id"], rec["u0"], rec["v0"], rec["u1"], rec["v1"] = ids, u0, v0, u1, v1
        (tmp_path / f"cloud_{k}.bin").write_bytes(cloud.tobytes())
        (tmp_path / f"tracks_{k}.bin").write_bytes(rec.tobytes())
        (tmp_path / f"inl_{k}.bin").write_bytes(inl.tobytes())
        frames.append((cloud, (coeffs, inl), ids.copy(), u0, v0, u1, v1))
    r = subprocess.run([str(TRACKLET_DEMO), str(tmp_path), str(n_frames), "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    known, ref_last = set(), None
    for k, (cloud, (coeffs, inl), fids, u0, v0, u1, v1) in enumerate(frames):
        coeffs = coeffs.copy()
        coeffs[3] = np.float32(1.73)
        is_new = np.array([int(i) not in known for i in fids])
        out = np.frombuffer((tmp_path / f"out_{k}.bin").read_bytes(), dtype=[("d0", "<f4"), ("d1", "<f4"), ("len", "<i4")])
        if k == 2:
            assert (out["d0"] == -1).all()                       # current frame: invalid depths
            uv_old = np.stack([np.trunc(u1[is_new]).astype(np.float64), np.trunc(v1[is_new]).astype(np.float64)], axis=1)
            d_prev, _ = ref_last.calculate_depth(uv_old, 4)      # previous frame: answered from its resident slot
            assert np.allclose(out["d1"][is_new], d_prev.astype(np.float32), rtol=0, atol=1e-4) and (d_prev >= 0).sum() > 10
            ref_last = None                                      # cloud and plane are forgotten (:333-348)
        else:
            ref = make_oracle(P)
            ref.set_cloud(cloud)
            ref.set_ground_plane(coeffs, inl)
            e_cur, e_last, _, _ = oracle.tracklets_depth(ref, ref_last, u0, v0, u1, v1, is_new, n_threads=8)
            assert np.allclose(out["d0"], e_cur, rtol=0, atol=1e-4)
            assert np.allclose(out["d1"][is_new], e_last[is_new], rtol=0, atol=1e-4)
            if k == 3:
                assert (out["d1"][is_new] == -1).all()           # no previous cloud after the failed frame
            ref_last = ref
        known = set(int(i) for i in fids)
