"""The lane-per-feature kernel's shared list budget (mld_set_list_budget): the narrow list is stored behind the lane's wide
list, both inside `total` entries of LDS.  Whatever the budget, the results are the oracle's - features whose lists do not fit
(wide > wide capacity, narrow > narrow capacity, or wide + narrow > total) are handed to the wave-cooperative kernel - on
a dense cloud where every one of these boundaries is crossed by hundreds of features (wide 0 ... 52, narrow 0 ... 22)."""
import numpy as np
import pytest
import torch

from mono_lidar_depth_amd import DepthEstimatorError, capi, synth

from helpers import assert_depth_parity, make_estimator, run_oracle

pytestmark = pytest.mark.gpu


def _mask(inl, n):
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return m.view(np.int32)


@pytest.fixture(scope="module")
def dense_frames():
    P = capi.params_c0()
    B, F = 2, 5000
    clouds = [synth.make_cloud(synth.DENSE128, seed=80 + b, frame=b) for b in range(B)]
    planes = [synth.make_ground_plane(c) for c in clouds]
    rng = np.random.default_rng(23)
    uvs = [np.floor(np.stack([rng.uniform(0, synth.KITTI_W, F), rng.uniform(90, synth.KITTI_H, F)], axis=1)) for _ in range(B)]
    ref = [run_oracle(P, clouds[b], uvs[b], planes[b], n_threads=8)[1] for b in range(B)]
    return P, clouds, planes, uvs, ref


# (capacities, budget): the default context (32 / 24 within 40), budgets at and between the bounds, the dense setting
@pytest.mark.parametrize("capacity,budget", [(None, None), ((32, 24), 32), ((32, 24), 56), ((48, 24), 48), ((48, 24), 52),
                                             ((48, 24), 60), ((48, 24), 0), ((64, 32), 64), ((16, 8), 17)])
@pytest.mark.parametrize("shared", [False, True])
def test_any_list_budget_gives_the_oracles_results(dense_frames, capacity, budget, shared):
    P, clouds, planes, uvs, ref = dense_frames
    dev = torch.device("cuda:0")
    B, F = len(clouds), uvs[0].shape[0]
    est = make_estimator(P, max_frames=B, max_features=F)
    if capacity is not None:
        est.setListCapacity(*capacity)
        est.setListBudget(budget)
    if shared:
        est.setSharedGpu(1)   # (LDS padded to two wavefronts per SIMD; the DENSE 1 instantiation beyond 32 / 24)
    d_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    d_masks = [torch.from_numpy(_mask(p[1], c.shape[0])).to(dev) for p, c in zip(planes, clouds)]
    d_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
    d_depth = [torch.full((F,), float("nan"), dtype=torch.float64, device=dev) for _ in range(B)]
    d_type = [torch.full((F,), -77, dtype=torch.int32, device=dev) for _ in range(B)]
    batch = est.prepareBatch(d_clouds, d_uvs, d_depth, d_type, np.stack([p[0] for p in planes]), d_masks, stride_bytes=16)
    for _ in range(2):   # (twice: the second pass runs on the queues / tags the first left)
        est.runBatch(batch)
    est.synchronize()
    for b in range(B):
        assert_depth_parity(d_depth[b].cpu().numpy(), d_type[b].cpu().numpy(), *ref[b])
    est.close()


def test_list_budget_argument_bounds():
    est = make_estimator(capi.params_c0())
    for bad in (31, 57, -1):      # a new context: capacities 32 / 24 -> 32 <= total <= 56
        with pytest.raises(DepthEstimatorError) as ei:
            est.setListBudget(bad)
        assert ei.value.code == capi.MLD_ERR_INVALID_ARG and "list budget" in str(ei.value)
    est.setListBudget(32)
    est.setListBudget(56)
    est.setListBudget(0)           # = wide + narrow
    est.setListCapacity(48, 24)    # resets the budget to 72
    with pytest.raises(DepthEstimatorError):
        est.setListBudget(47)
    est.setListBudget(72)
    est.close()


def test_path_counts_report_what_the_budget_hands_over(dense_frames):
    """mld_get_path_counts: features queued for the lane-per-feature kernel / features the wave-cooperative kernel worked on
    in the slot's last batched call - the feedback for choosing capacities.  On the dense cloud a tighter budget hands more
    features over, and the results stay the oracle's (test above); nothing is counted before the first batched call."""
    P, clouds, planes, uvs, ref = dense_frames
    dev = torch.device("cuda:0")
    B, F = len(clouds), uvs[0].shape[0]
    counts = {}
    for budget in (72, 56, 48):
        est = make_estimator(P, max_frames=B, max_features=F)
        est.setListCapacity(48, 24)
        est.setListBudget(budget)
        assert est.pathCounts(0) == (0, 0)
        d_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
        d_masks = [torch.from_numpy(_mask(p[1], c.shape[0])).to(dev) for p, c in zip(planes, clouds)]
        d_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
        d_depth = [torch.empty((F,), dtype=torch.float64, device=dev) for _ in range(B)]
        d_type = [torch.empty((F,), dtype=torch.int32, device=dev) for _ in range(B)]
        batch = est.prepareBatch(d_clouds, d_uvs, d_depth, d_type, np.stack([p[0] for p in planes]), d_masks, stride_bytes=16)
        est.runBatch(batch)
        est.synchronize()
        counts[budget] = [est.pathCounts(b) for b in range(B)]
        for b in range(B):
            lane, handed = counts[budget][b]
            dead = int((ref[b][1] == 2).sum())   # RadiusSearchInsufficientPoints: settled by the classification itself
            assert 0 < lane <= F and 0 <= handed <= F
            assert lane + handed >= F - dead   # every feature with neighbours went to (at least) one of the two kernels
        with pytest.raises(DepthEstimatorError):
            est.pathCounts(B)   # no such slot
        est.close()
    for b in range(B):
        assert counts[48][b][1] > counts[56][b][1] > counts[72][b][1]
