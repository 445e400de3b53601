"""Dense clouds (BASELINE config 5: 128 beams x 4096) at batch size through the DENSE instantiation of the
lane-per-feature kernel (list capacities 48 / 24, mld_set_list_capacity): neighbour lists of 2 ... 21 points, i.e. every
tier of the in-register max-spanning-triangle search (PlaneEstimationCalcMaxSpanningTriangle.cpp:37-100) and the
count-class order of the live queue - against the oracle, bit-exact on the triangle path."""
import numpy as np
import pytest
import torch

from mono_lidar_depth_amd import capi, synth

from helpers import assert_depth_parity, kitti_camera, make_estimator, run_oracle

pytestmark = pytest.mark.gpu


def _mask(inl, n):
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return m.view(np.int32)


@pytest.mark.parametrize("capacity", [(48, 24), (32, 24), (64, 32)])
@pytest.mark.parametrize("integer", [True, False])
def test_dense_batches_equal_the_oracle(capacity, integer):
    P = capi.params_c0()
    dev = torch.device("cuda:0")
    B, F = 3, 6000
    est = make_estimator(P, max_frames=B, max_features=F)
    est.setListCapacity(*capacity)
    clouds = [synth.make_cloud(synth.DENSE128, seed=60 + b, frame=b) for b in range(B)]
    planes = [synth.make_ground_plane(c) for c in clouds]
    rng = np.random.default_rng(17)
    uvs = []
    for b in range(B):
        uv = np.stack([rng.uniform(0, synth.KITTI_W, F), rng.uniform(90, synth.KITTI_H, F)], axis=1)
        uvs.append(np.floor(uv) if integer else uv)
    d_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    d_masks = [torch.from_numpy(_mask(p[1], c.shape[0])).to(dev) for p, c in zip(planes, clouds)]
    d_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
    d_depth = [torch.full((F,), float("nan"), dtype=torch.float64, device=dev) for _ in range(B)]
    d_type = [torch.full((F,), -77, dtype=torch.int32, device=dev) for _ in range(B)]
    batch = est.prepareBatch(d_clouds, d_uvs, d_depth, d_type, np.stack([p[0] for p in planes]), d_masks, stride_bytes=16)
    est.runBatch(batch)
    est.synchronize()
    seen = set()
    for b in range(B):
        _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], planes[b], n_threads=8)
        assert_depth_parity(d_depth[b].cpu().numpy(), d_type[b].cpu().numpy(), d0, t0)
        seen |= set(int(x) for x in np.unique(t0))
    assert {1, 2, 16} <= seen and len(seen) >= 8   # every live branch of the path appears on a dense cloud
    est.close()
