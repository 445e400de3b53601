"""A slice of the randomised parity sweeps inside the driver-run suite (the committed seeds of
test_randomized_gpu.py missed two real defects that the sweeps then found).  The slice's first seed is derived from the
kernel / C-ABI sources (tests/sweeps.py: sha256), so every tree is checked on configurations no earlier tree has seen; the
seed of a failing configuration is in its test id (`MLD_SWEEP_BASE=<base> pytest -k <id>` reproduces it, as does
`profiles/tools/random_sweep.py <seed> 1 <route>`).

  single frames   300 seeds on the lane-per-feature route, 100 on each of the default / wave-only / dense routes
  batched         60 launch sets of 5 ragged frames through the shipped library's batched entry points
  estimated       60 production calls (plane estimated in the call: RANSAC / semantic in turn)
  tracklets       50 x (3 ragged sequences x 3 frames) through the batched tracklet layer
  dense cloud     24 seeds on the 128 x 4096 cloud (lists of up to 48 neighbours) on the dense and default routes
"""
import pytest

import sweeps

pytestmark = pytest.mark.gpu
BASE = sweeps.sweep_base()


def _route(monkeypatch, route):
    if route != "default":  # (as the feature_kernel_path fixture of conftest.py)
        monkeypatch.setenv("MLD_FORCE_WAVE_PATH" if route == "wave-only" else "MLD_FORCE_THREAD_PATH", "1")
        if route == "dense":
            monkeypatch.setenv("MLD_K1MAX", "48")
        from mono_lidar_depth_amd import capi
        monkeypatch.setattr(capi, "_lib", capi.load_ab())


@pytest.mark.parametrize("seed", [BASE + i for i in range(300)])
def test_sweep_single_frame_lane_per_feature_route(seed, monkeypatch):
    _route(monkeypatch, "fused")
    sweeps.check_single(seed)


@pytest.mark.parametrize("route", ["default", "wave-only", "dense"])
@pytest.mark.parametrize("seed", [BASE + 300 + i for i in range(100)])
def test_sweep_single_frame_other_routes(seed, route, monkeypatch):
    _route(monkeypatch, route)
    sweeps.check_single(seed)


@pytest.mark.parametrize("route", ["default", "dense"])
@pytest.mark.parametrize("seed", [BASE + 400 + i for i in range(12)])
def test_sweep_dense_cloud(seed, route, monkeypatch):
    _route(monkeypatch, route)
    sweeps.check_single(seed, dense128=True)


@pytest.mark.parametrize("seed", [BASE + 500 + i for i in range(60)])
def test_sweep_batched_entry_points(seed):
    sweeps.check_batch(seed, 5)


@pytest.mark.parametrize("seed", [BASE + 600 + i for i in range(60)])
def test_sweep_plane_estimated_in_the_call(seed):
    sweeps.check_estimate(seed)


@pytest.mark.parametrize("seed", [BASE + 700 + i for i in range(50)])
def test_sweep_tracklet_layer(seed):
    sweeps.check_tracklets(seed, 3, 2600)
