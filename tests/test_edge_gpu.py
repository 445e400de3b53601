"""Edge cases of the C-ABI path beyond the reference's own tests: non-finite features, tiny and odd-sized images, no
tracks, tag wrap-around in the batched path, repeated re-initialisation."""
import ctypes as C

import numpy as np
import pytest

from mono_lidar_depth_amd import NO_PLANE, CameraPinhole, GroundPlane, TrackletDepthModule, capi, synth

from helpers import assert_depth_parity, kitti_camera, make_estimator, run_oracle

pytestmark = pytest.mark.gpu


def test_non_finite_feature_coordinates():
    """NaN / inf pixel coordinates: the reference casts them to int (undefined); here and in the oracle such a feature
    has an empty window (RadiusSearchInsufficientPoints) and never disturbs its neighbours in the wavefront."""
    P = capi.params_c0()
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=51, frame=0)
    plane = synth.make_ground_plane(cloud)
    uv = synth.make_features(640, seed=51)
    bad = np.array([[np.nan, 100.0], [300.0, np.nan], [np.inf, 50.0], [-np.inf, 60.0], [200.0, np.inf], [np.nan, np.nan]])
    pos = [0, 63, 64, 129, 300, 639]
    uv[pos] = bad
    est = make_estimator(P)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P, cloud, uv, plane)
    assert (t[pos] == 2).all() and (d[pos] == -1).all()
    assert_depth_parity(d, t, d0, t0)


@pytest.mark.parametrize("W,H", [(40, 8), (33, 33), (64, 1), (1, 64), (97, 131)])
def test_tiny_and_odd_images(W, H):
    """Image sizes around the 32-pixel word / 4-row group boundaries of the occupancy bitmap."""
    P = capi.params_c0().replace(pixelarea_search_witdh=4, pixelarea_search_height=4)
    cam = CameraPinhole(W, H, 0.6 * max(W, H), W / 2.0, H / 2.0)
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=52, frame=0)
    plane = synth.make_ground_plane(cloud)
    uv = synth.make_features(500, seed=52, width=W, height=H)
    est = make_estimator(P, camera=cam)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    ref, (d0, t0) = run_oracle(P, cloud, uv, plane, camera=cam)
    assert_depth_parity(d, t, d0, t0)
    assert np.array_equal(est.getPixelMap(), ref.pixel_map())
    assert np.array_equal(est.getPointIndex(), ref.point_index())


def test_tracklets_without_tracks():
    P = capi.params_c0()
    mod = TrackletDepthModule(P, kitti_camera(), synth.T_CAM_LIDAR)
    cloud = synth.make_cloud(synth.VLP16, seed=53)
    plane = synth.make_ground_plane(cloud)
    empty = np.zeros(0, dtype=np.float32)
    for _ in range(2):
        d_cur, d_last, is_new = mod.process(cloud, np.zeros(0, dtype=np.int64), empty, empty, empty, empty, GroundPlane(*plane))
        assert d_cur.shape == (0,) and d_last.shape == (0,) and is_new.shape == (0,)
    # and a frame with tracks afterwards still works (the previous slot is valid)
    ids = np.arange(50)
    u = np.linspace(100, 1100, 50).astype(np.float32)
    v = np.full(50, 250, dtype=np.float32)
    d_cur, d_last, is_new = mod.process(cloud, ids, u, v, u, v, GroundPlane(*plane))
    assert is_new.all() and np.isfinite(d_cur).all() and np.isfinite(d_last).all()


def test_tag_wraparound_in_the_batched_path():
    """More than 127 (kMaxTag) batched setInputCloud calls: the per-batch tag (passed as a kernel argument) wraps and every
    slot's map is re-zeroed; results after the wrap equal the oracle."""
    import torch
    P = capi.params_c0()
    B, F = 8, 200
    dev = torch.device("cuda:0")
    sc = synth.Scanner(16, 360, 2.0, -24.9)
    est = make_estimator(P, max_frames=B)
    sets = []
    for k in range(2):
        clouds = [synth.make_cloud(sc, seed=60 + b, frame=k) for b in range(B)]
        planes = [synth.make_ground_plane(c) for c in clouds]
        sets.append((clouds, planes, [torch.from_numpy(c).to(dev) for c in clouds],
                     [torch.from_numpy(p[1]).to(dev) for p in planes]))
    uvs = [synth.make_features(F, seed=70 + b) for b in range(B)]
    t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
    t_depth = [torch.empty(F, dtype=torch.float64, device=dev) for _ in range(B)]
    t_type = [torch.empty(F, dtype=torch.int32, device=dev) for _ in range(B)]
    torch.cuda.synchronize()
    for it in range(262):
        clouds, planes, t_clouds, t_inl = sets[it % 2]
        est.setInputClouds(t_clouds, 16)
        if it in (0, 1, 254, 255, 256, 261):
            for b in range(B):
                est.setGroundPlane(GroundPlane(planes[b][0], t_inl[b]), slot=b)
            est.CalculateDepths(t_uvs, t_depth, t_type)
            est.synchronize()
            for b in range(B):
                _, (d0, t0) = run_oracle(P, clouds[b], uvs[b], planes[b])
                assert_depth_parity(t_depth[b].cpu().numpy(), t_type[b].cpu().numpy(), d0, t0)


def test_reinitialise_with_other_parameters_and_camera():
    """Initialize() again on the same object (the reference allows it): the old context is destroyed, the new
    calibration is used."""
    P1 = capi.params_c0()
    P2 = capi.params_c0().replace(pixelarea_search_witdh=10, pixelarea_search_height=4, do_use_histogram_segmentation=0)
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=54, frame=0)
    plane = synth.make_ground_plane(cloud)
    est = make_estimator(P1)
    uv = synth.make_features(400, seed=54)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P1, cloud, uv, plane)
    assert_depth_parity(d, t, d0, t0)
    cam2 = CameraPinhole(800, 300, 500.0, 400.0, 150.0)
    est.InitConfig(P2)
    est.Initialize(cam2, synth.T_CAM_LIDAR)
    uv2 = synth.make_features(400, seed=55, width=800, height=300)
    d, t = est.CalculateDepth(cloud, uv2, GroundPlane(*plane))
    _, (d0, t0) = run_oracle(P2, cloud, uv2, plane, camera=cam2)
    assert_depth_parity(d, t, d0, t0)


def test_many_features_in_one_call():
    """300 000 features in one CalculateDepth (150x a frame's worth): spot-checked against the oracle, and the
    per-feature result must not depend on how many features share the call."""
    P = capi.params_c0()
    cloud = synth.make_cloud(synth.HDL64_KITTI, seed=56, frame=0)
    plane = synth.make_ground_plane(cloud)
    uv = synth.make_features(300000, seed=56)
    est = make_estimator(P)
    d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
    sel = np.r_[0:700, 150000:150700, 299300:300000]
    _, (d0, t0) = run_oracle(P, cloud, uv[sel], plane, n_threads=8)
    assert_depth_parity(d[sel], t[sel], d0, t0)
    # the same features in a small call: calls of up to 16 384 features run on the wave-cooperative kernel (one feature
    # per wavefront), larger ones on the lane-per-feature kernel.  Result types and main-path depths are identical bit for
    # bit; the road estimator's sums are associated differently in the two kernels (both within 1e-4 m of the oracle).
    d2, t2 = est.CalculateDepth(uv[sel])
    assert np.array_equal(t2, t[sel])
    road = t2 == 16
    assert np.array_equal(d2[~road], d[sel][~road])
    assert np.abs(d2[road] - d[sel][road]).max() < 1e-9


def _batch_against_oracle(P, cam, est, clouds, planes, uvs):
    import torch
    dev = torch.device("cuda:0")
    B, F = len(clouds), uvs[0].shape[0]
    masks = []
    for p, c in zip(planes, clouds):
        m = np.zeros((c.shape[0] + 31) // 32, dtype=np.uint32)
        np.bitwise_or.at(m, p[1] >> 5, (np.uint32(1) << (p[1] & 31).astype(np.uint32)))
        masks.append(torch.from_numpy(m.view(np.int32)).to(dev))
    d = [torch.full((F,), float("nan"), dtype=torch.float64, device=dev) for _ in range(B)]
    t = [torch.full((F,), -77, dtype=torch.int32, device=dev) for _ in range(B)]
    b = est.prepareBatch([torch.from_numpy(c).to(dev) for c in clouds], [torch.from_numpy(u).to(dev) for u in uvs], d, t,
                         np.stack([p[0] for p in planes]), masks)
    est.runBatch(b)
    est.synchronize()
    for i in range(B):
        _, (d0, t0) = run_oracle(P, clouds[i], uvs[i], planes[i], camera=cam)
        assert_depth_parity(d[i].cpu().numpy(), t[i].cpu().numpy(), d0, t0)
    return t


def test_classification_reads_a_large_image_bitmap_in_place():
    """An image whose occupancy bitmap (345 KB) does not fit the LDS of a classification block: k_classify reads the
    bitmap where it lies (16-byte loads of four rows, as the feature kernel's scan) - batches against the oracle."""
    W, H = 2600, 1000
    P = capi.params_c0()
    cam = CameraPinhole(W, H, 1500.0, W / 2.0 + 3.5, H / 2.0 - 7.25)
    clouds = [synth.make_cloud(synth.DENSE128, seed=57, frame=f) for f in range(3)]  # (dense: neighbours at this resolution)
    planes = [synth.make_ground_plane(c) for c in clouds]
    rng = np.random.default_rng(57)
    uvs = [np.stack([rng.uniform(-5, W + 5, 3000), rng.uniform(H * 0.4, H + 5, 3000)], axis=1) for _ in clouds]
    est = make_estimator(P, camera=cam, max_frames=3, max_features=3000)
    t = _batch_against_oracle(P, cam, est, clouds, planes, uvs)
    seen = set(int(x) for x in np.unique(np.concatenate([x.cpu().numpy() for x in t])))
    assert 2 in seen and len(seen) >= 4, seen   # dead features and several live outcomes
    est.close()


def test_classification_in_place_equals_staged(monkeypatch):
    """The same in-place classification forced for a KITTI-size image (test build: MLD_CLASSIFY_STAGED=0): every class
    boundary of k_classify - dead, live, long lists, windows clipped by the image border - against the oracle."""
    monkeypatch.setenv("MLD_CLASSIFY_STAGED", "0")
    monkeypatch.setattr(capi, "_lib", capi.load_ab())
    P = capi.params_c0()
    cam = kitti_camera()
    clouds = [synth.make_cloud(sc, seed=58, frame=f) for f, sc in enumerate((synth.HDL64_KITTI, synth.DENSE128, synth.VLP16))]
    planes = [synth.make_ground_plane(c) for c in clouds]
    rng = np.random.default_rng(58)
    uvs = [np.stack([rng.uniform(-3, synth.KITTI_W + 3, 4000), rng.uniform(-3, synth.KITTI_H + 3, 4000)], axis=1) for _ in clouds]
    est = make_estimator(P, max_frames=3, max_features=4000)
    _batch_against_oracle(P, cam, est, clouds, planes, uvs)
    est.close()
