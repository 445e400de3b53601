"""The synthetic generator is deterministic; the algorithmic byte model agrees with the oracle's neighbour counts."""
import numpy as np

from mono_lidar_depth_amd import capi, synth, traffic
from oracle import oracle


def test_generator_is_seeded_and_shaped():
    a = synth.make_cloud(synth.VLP16, seed=3, frame=2)
    b = synth.make_cloud(synth.VLP16, seed=3, frame=2)
    c = synth.make_cloud(synth.VLP16, seed=3, frame=3)
    assert a.shape == (16 * 1800, 4) and a.dtype == np.float32
    assert np.array_equal(a, b, equal_nan=True) and not np.array_equal(a, c, equal_nan=True)
    nan_frac = np.isnan(a[:, 0]).mean()
    assert 0.005 < nan_frac < 0.2
    a8 = synth.make_cloud(synth.VLP16, seed=3, frame=2, stride_floats=8)
    assert a8.shape == (16 * 1800, 8) and np.array_equal(a8[:, :3], a[:, :3], equal_nan=True)
    assert synth.HDL64.rings * synth.HDL64.azimuth_steps == 131072
    assert synth.DENSE128.rings * synth.DENSE128.azimuth_steps == 524288
    uv = synth.make_features(100, seed=1, integer=True)
    assert uv.shape == (100, 2) and np.array_equal(uv, np.floor(uv))
    coeffs, inl = synth.make_ground_plane(a, subsample=500)
    assert coeffs.tolist() == [0.0, 0.0, 1.0, 1.7300000190734863] and inl.size == 500 and np.all(np.diff(inl) > 0)


def test_traffic_model_counts_match_oracle_neighbours():
    P = capi.params_c0()
    cam = capi.MldCamera(synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV, synth.KITTI_W, synth.KITTI_H)
    cloud = synth.make_cloud(synth.Scanner(64, 1024, 2.0, -24.9), seed=2)
    coeffs, inl = synth.make_ground_plane(cloud)
    uv = synth.make_features(300, seed=2)
    ref = oracle.OracleDepthEstimator(P, cam, synth.T_CAM_LIDAR)
    ref.set_cloud(cloud)
    ref.set_ground_plane(coeffs, inl)
    d, t = ref.calculate_depth(uv)
    pm = ref.pixel_map()
    k1 = traffic.neighbour_counts(pm, uv, 3.0, 4.5)
    k2 = traffic.neighbour_counts(pm, uv, 6.0, 6.75)
    traces = [ref.trace_feature(*p) for p in uv]
    assert k1.tolist() == [len(tr["nb_idx"]) for tr in traces]
    for i, tr in enumerate(traces):
        if tr["reached_road"]:
            assert k2[i] == len(tr["road_idx"])
    fb = traffic.frame_bytes(P, cam.width, cam.height, cloud.shape[0], ref.nvis, pm, uv, t)
    assert fb["fallback_features"] == sum(tr["reached_road"] for tr in traces)
    assert fb["P1"] == 70 and fb["P2"] == 182
    assert fb["project_bytes"] == 16 * cloud.shape[0] + 4 * cam.width * cam.height + 28 * ref.nvis


def test_committed_counter_profile_prices_every_leg():
    """profiles/traffic.json is what bench.py prices its rooflines on: the config-2 kernels at the top level and one entry
    per counter-profiled leg (config 2 at k = 7, config 3 near the returns, config 5 at 256 sequences).  A summariser that
    drops the per-leg entries would silently turn those legs' `roofline` objects into null."""
    import json
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    tj = json.loads((root / "profiles" / "traffic.json").read_text())
    for k in ("k_project_scatter", "k_classify", "k_feature_fused"):
        assert tj[k]["hbm_bytes_per_launch"] > 0 and tj[k]["launch_s"] > 0
    assert set(tj["configs"]) >= {"2k", "3n", "5b256"}
    from bench_support import rooflines as bench
    kt = {"k_project_scatter": {"avg_ms": 0.5}, "k_classify": {"avg_ms": 0.1}, "k_feature_fused": {"avg_ms": 0.8},
          "k_feature_wave": {"avg_ms": 0.01}, "k_rs_batch": {"avg_ms": 0.0}}
    for key, frames in (("2k", 1024), ("3n", 256), ("5b256", 256), ("5b16", 16)):
        r = bench.config_roofline(key, kt, frames)
        assert r is not None and r["kernel"] == "k_feature_fused" and 0.0 < r["frac"] < 1.0, key
        assert abs(r["achieved"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
