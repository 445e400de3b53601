"""Features scattered around the image positions of LiDAR returns (the second variant of BASELINE config 3 in bench.py):
on a sparse 16-ring cloud uniformly random features almost never see a neighbour; these always do, so every path behind
the neighbour search runs - collinear triangles, planarity / orthogonality rejections, threshold treatment modes, road
fallback - through all three kernel routes, one frame per call and batched."""
import numpy as np
import pytest

from mono_lidar_depth_amd import GroundPlane, capi, synth

from helpers import assert_depth_parity, make_estimator, run_oracle

pytestmark = pytest.mark.gpu

MODES = [("c0_dispose", {}),
         ("adjust_relative", dict(treshold_depth_mode=1, treshold_depth_local_mode=1, treshold_depth_local_valuetype=1)),
         ("adjust_absolute", dict(treshold_depth_mode=1, treshold_depth_local_mode=1, treshold_depth_local_valuetype=0)),
         ("no_planar_check_orth", dict(do_check_triangleplanar_condition=0, viewray_plane_orthoganality_treshold=0.3)),
         ("road_triangle", dict(plane_estimator_use_triangle_maximation=1))]


@pytest.mark.parametrize("name,kw", MODES)
@pytest.mark.parametrize("scanner", ["VLP16", "HDL64"])
def test_near_return_features_one_frame(name, kw, scanner, feature_kernel_path):
    P = capi.params_c0().replace(**kw)
    sc = getattr(synth, scanner)
    cloud = synth.make_cloud(sc, seed=31, frame=2)
    uv = synth.make_features_near_points(cloud, 3000, seed=31)
    plane = synth.make_ground_plane(cloud)
    est = make_estimator(P)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P, cloud, uv, plane, n_threads=4)
    assert_depth_parity(d, t, d0, t0)
    # the variant does what it is for: hardly any feature is left without neighbours, several result types occur
    assert (t0 == 2).mean() < 0.2 and np.unique(t0).size >= 4


def test_near_return_features_batched(feature_kernel_path):
    import torch
    P = capi.params_c0().replace(treshold_depth_mode=1, treshold_depth_local_mode=1, treshold_depth_local_valuetype=1)
    B, F = 6, 2500
    dev = torch.device("cuda:0")
    est = make_estimator(P, max_frames=B, max_features=F)
    scanners = [synth.VLP16, synth.HDL64, synth.HDL64_KITTI]
    clouds = [synth.make_cloud(scanners[b % 3], seed=40 + b, frame=b) for b in range(B)]
    uvs = [synth.make_features_near_points(clouds[b], F, seed=400 + b) for b in range(B)]
    planes = [synth.make_ground_plane(c) for c in clouds]
    d_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
    d_uv = [torch.from_numpy(u).to(dev) for u in uvs]
    masks = []
    for c, (co, inl) in zip(clouds, planes):
        m = np.zeros((c.shape[0] + 31) // 32, dtype=np.uint32)
        np.bitwise_or.at(m, inl >> 5, np.uint32(1) << (inl & 31).astype(np.uint32))
        masks.append(torch.from_numpy(m.view(np.int32)).to(dev))
    depth = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(B)]
    types = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(B)]
    batch = est.prepareBatch(d_clouds, d_uv, depth, types, np.stack([p[0] for p in planes]), masks)
    est.runBatch(batch)
    est.synchronize()
    for b in range(B):
        _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], planes[b], n_threads=4)
        assert_depth_parity(depth[b].cpu().numpy(), types[b].cpu().numpy(), d0, t0)
