"""BASELINE config 5 through the tracklet API: 128x4096 cloud, 10 000 tracks (10 % new per frame), three frames.

The GPU module keeps the previous frame's slot (no re-projection); the oracle restates TrackletDepthModule's
marshalling on the CPU with the previous cloud re-projected as the reference does.  Results must be identical.
"""
import numpy as np
import pytest

from mono_lidar_depth_amd import GroundPlane, TrackletDepthModule, capi, synth
from oracle import oracle

from helpers import kitti_camera, make_oracle

pytestmark = pytest.mark.gpu


def _tracks(rng, ids_prev, n_tracks, new_frac, W, H):
    n_new = int(n_tracks * new_frac) if ids_prev is not None else n_tracks
    keep = n_tracks - n_new
    start = (ids_prev.max() + 1) if ids_prev is not None else 0
    ids = np.concatenate([rng.choice(ids_prev, keep, replace=False) if keep else np.zeros(0, np.int64),
                          np.arange(start, start + n_new)]).astype(np.int64)
    rng.shuffle(ids)
    u0 = rng.uniform(-2, W + 2, n_tracks).astype(np.float32)
    v0 = rng.uniform(100, H + 2, n_tracks).astype(np.float32)
    u1 = (u0 + rng.normal(0, 3, n_tracks)).astype(np.float32)
    v1 = (v0 + rng.normal(0, 2, n_tracks)).astype(np.float32)
    return ids, u0, v0, u1, v1


@pytest.mark.parametrize("scanner,n_tracks", [(synth.DENSE128, 10000), (synth.VLP16, 3000)])
def test_tracklet_module_sequence(scanner, n_tracks):
    P = capi.params_c0()
    cam = kitti_camera()
    mod = TrackletDepthModule(P, cam, synth.T_CAM_LIDAR)
    rng = np.random.default_rng(3)
    ids_prev = None
    ref_last = None
    known = set()
    for frame in range(3):
        cloud = synth.make_cloud(scanner, seed=14, frame=frame * 2)
        coeffs, inl = synth.make_ground_plane(cloud)
        ids, u0, v0, u1, v1 = _tracks(rng, ids_prev, n_tracks, 0.10, cam.width, cam.height)
        d_cur, d_last, is_new = mod.process(cloud, ids, u0, v0, u1, v1, GroundPlane(coeffs, inl))

        ref_cur = make_oracle(P)
        ref_cur.set_cloud(cloud)
        ref_cur.set_ground_plane(coeffs, inl)
        exp_new = np.array([int(i) not in known for i in ids])
        assert np.array_equal(is_new, exp_new)
        e_cur, e_last, et_cur, et_last = oracle.tracklets_depth(ref_cur, ref_last, u0, v0, u1, v1, exp_new, n_threads=8)
        t_cur, t_last = mod.last_types
        assert np.array_equal(t_cur, et_cur)
        assert np.allclose(d_cur, e_cur, rtol=0, atol=1e-4, equal_nan=True)
        assert np.array_equal(t_last[is_new], et_last[is_new])
        assert np.allclose(d_last[is_new], e_last[is_new], rtol=0, atol=1e-4, equal_nan=True)
        assert np.isnan(d_last[~is_new]).all()
        if frame == 0:
            assert (d_last[is_new] == -1).all()  # no previous cloud (:93-96)
        else:
            assert is_new.sum() == int(n_tracks * 0.10)
        # bookkeeping: only the tracks of this frame survive (TidyUpTracklets), new ones hold two features
        known = set(int(i) for i in ids)
        assert set(mod.known_ids()) == known
        i_new = int(np.nonzero(is_new)[0][0])
        h = mod.tracklet(int(ids[i_new]))
        assert len(h) == 2 and h[0][:2] == (int(u0[i_new]), int(v0[i_new])) and h[1][:2] == (int(u1[i_new]), int(v1[i_new]))
        ids_prev, ref_last = ids, ref_cur


def test_tracklet_module_with_semantic_image():
    """The 4-argument process overload (tracklet_depth_module.cpp:261-284): per-frame SemanticPlane from a label image,
    the previous frame's semantic plane kept with its slot."""
    P = capi.params_c0()
    cam = kitti_camera()
    mod = TrackletDepthModule(P, cam, synth.T_CAM_LIDAR)
    rng = np.random.default_rng(5)
    ids_prev, ref_last, known = None, None, set()
    for frame in range(3):
        cloud = synth.make_cloud(synth.HDL64_KITTI, seed=15, frame=frame * 2)
        img = synth.make_label_image(cloud)
        ids, u0, v0, u1, v1 = _tracks(rng, ids_prev, 2000, 0.10, cam.width, cam.height)
        d_cur, d_last, is_new = mod.process(cloud, ids, u0, v0, u1, v1, None, img=img)
        ref_cur = make_oracle(P)
        ref_cur.set_cloud(cloud)
        ref_cur.estimate_semantic_plane(img, (6, 7, 8, 9), P.ransac_plane_refinement_treshold)
        exp_new = np.array([int(i) not in known for i in ids])
        e_cur, e_last, et_cur, et_last = oracle.tracklets_depth(ref_cur, ref_last, u0, v0, u1, v1, exp_new, n_threads=8)
        t_cur, t_last = mod.last_types
        assert np.array_equal(t_cur, et_cur)
        assert np.allclose(d_cur, e_cur, rtol=0, atol=1e-4, equal_nan=True)
        assert np.array_equal(t_last[is_new], et_last[is_new])
        assert np.allclose(d_last[is_new], e_last[is_new], rtol=0, atol=1e-4, equal_nan=True)
        assert (t_cur == 16).sum() > 20
        known = set(int(i) for i in ids)
        ids_prev, ref_last = ids, ref_cur


def test_tracklet_batch_of_sequences_equals_the_oracle_per_sequence():
    """mld_tracklets_depths_device: eight sequences (different scanners, cloud sizes and track counts), three frames each,
    both slots of every sequence - the current frame's features and the new tracks' previous features on the
    previous frame's resident slot - against the per-sequence CPU restatement of TrackletDepthModule."""
    import torch
    from mono_lidar_depth_amd import TrackletBatch
    P = capi.params_c0()
    cam = kitti_camera()
    dev = torch.device("cuda:0")
    scanners = [synth.HDL64_KITTI, synth.VLP16, synth.DENSE128, synth.HDL64_KITTI, synth.VLP16, synth.HDL64_KITTI,
                synth.VLP16, synth.HDL64_KITTI]
    n_tracks = [2500, 1800, 6000, 1000, 3000, 2048, 64, 3333]
    S = len(scanners)
    tb = TrackletBatch(P, cam, synth.T_CAM_LIDAR, S, max(n_tracks), list_capacity=(48, 24))  # the dense-cloud setting
    rng = np.random.default_rng(11)
    ids_prev = [None] * S
    ref_last = [None] * S
    known = [set() for _ in range(S)]

    def mask_of(inl, n):
        m = np.zeros((n + 31) // 32, dtype=np.uint32)
        np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
        return torch.from_numpy(m.view(np.int32)).to(dev)

    for frame in range(3):
        host = []
        for s in range(S):
            cloud = synth.make_cloud(scanners[s], seed=40 + s, frame=frame * 2)
            coeffs, inl = synth.make_ground_plane(cloud)
            ids, u0, v0, u1, v1 = _tracks(rng, ids_prev[s], n_tracks[s], 0.10, cam.width, cam.height)
            is_new = np.array([int(i) not in known[s] for i in ids])
            host.append((cloud, coeffs, inl, ids, u0, v0, u1, v1, is_new))
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        d_cur = [torch.empty(n, dtype=torch.float32, device=dev) for n in n_tracks]
        d_last = [torch.full((n,), float("nan"), dtype=torch.float32, device=dev) for n in n_tracks]
        t_cur = [torch.empty(n, dtype=torch.int32, device=dev) for n in n_tracks]
        t_last = [torch.zeros(n, dtype=torch.int32, device=dev) for n in n_tracks]
        tb.frame([to(h[0]) for h in host], np.stack([h[1] for h in host]), [mask_of(h[2], h[0].shape[0]) for h in host],
                 [to(h[4]) for h in host], [to(h[5]) for h in host], [to(h[6]) for h in host], [to(h[7]) for h in host],
                 [to(h[8].astype(np.uint8)) for h in host], d_cur, d_last, t_cur, t_last)
        tb.est.synchronize()
        for s in range(S):
            cloud, coeffs, inl, ids, u0, v0, u1, v1, is_new = host[s]
            ref_cur = make_oracle(P)
            ref_cur.set_cloud(cloud)
            ref_cur.set_ground_plane(coeffs, inl)
            e_cur, e_last, et_cur, et_last = oracle.tracklets_depth(ref_cur, ref_last[s], u0, v0, u1, v1, is_new, n_threads=8)
            dc, dl = d_cur[s].cpu().numpy(), d_last[s].cpu().numpy()
            tc, tl = t_cur[s].cpu().numpy(), t_last[s].cpu().numpy()
            assert np.array_equal(tc, et_cur), (frame, s)
            assert np.allclose(dc, e_cur, rtol=0, atol=1e-4, equal_nan=True), (frame, s)
            assert np.array_equal(tl[is_new], et_last[is_new]), (frame, s)
            assert np.allclose(dl[is_new], e_last[is_new], rtol=0, atol=1e-4, equal_nan=True), (frame, s)
            assert np.isnan(dl[~is_new]).all()
            if frame == 0:
                assert (dl[is_new] == -1).all()
            known[s] = set(int(i) for i in ids)
            ids_prev[s], ref_last[s] = ids, ref_cur
    tb.close()


def test_tracklet_batch_far_ahead_of_the_device_equals_step_by_step():
    """The step's descriptor tables leave through a ring of pinned generations (mld_api.hip upload_small); a host that
    queues 45 steps without waiting laps that ring three times.  Every step's outputs (own arrays per step) must equal
    those of the same steps submitted one at a time."""
    import torch
    from mono_lidar_depth_amd import TrackletBatch
    P = capi.params_c0()
    cam = kitti_camera()
    dev = torch.device("cuda:0")
    S, n_tracks, steps = 3, 1500, 45
    rng = np.random.default_rng(23)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731

    def mask_of(inl, n):
        m = np.zeros((n + 31) // 32, dtype=np.uint32)
        np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
        return torch.from_numpy(m.view(np.int32)).to(dev)

    frames = []  # three distinct frames of every sequence, cycled
    for k in range(3):
        per = []
        for s in range(S):
            cloud = synth.make_cloud([synth.HDL64_KITTI, synth.VLP16, synth.DENSE128][s], seed=70 + s, frame=2 * k)
            coeffs, inl = synth.make_ground_plane(cloud)
            _, u0, v0, u1, v1 = _tracks(rng, None, n_tracks, 0.10, cam.width, cam.height)
            per.append((to(cloud), coeffs, mask_of(inl, cloud.shape[0]), to(u0), to(v0), to(u1), to(v1),
                        to((rng.random(n_tracks) < 0.1).astype(np.uint8))))
        frames.append(per)

    def run_all(wait_every_step):
        tb = TrackletBatch(P, cam, synth.T_CAM_LIDAR, S, n_tracks, list_capacity=(48, 24))
        outs, preps = [], []
        for it in range(steps):
            per = frames[it % 3]
            o = ([torch.empty(n_tracks, dtype=torch.float32, device=dev) for _ in range(S)],
                 [torch.full((n_tracks,), float("nan"), dtype=torch.float32, device=dev) for _ in range(S)],
                 [torch.empty(n_tracks, dtype=torch.int32, device=dev) for _ in range(S)],
                 [torch.zeros(n_tracks, dtype=torch.int32, device=dev) for _ in range(S)])
            outs.append(o)
            preps.append(tb.prepare([p[0] for p in per], np.stack([p[1] for p in per]), [p[2] for p in per],
                                    [p[3] for p in per], [p[4] for p in per], [p[5] for p in per], [p[6] for p in per],
                                    [p[7] for p in per], *o))
        torch.cuda.synchronize()
        tb.est._after_torch(frames[0][0][0])
        for it in range(steps):
            tb.run(preps[it])
            if wait_every_step:
                tb.est.synchronize()
        tb.est.synchronize()
        res = [[t.cpu().numpy() for group in o for t in group] for o in outs]
        tb.close()
        return res

    ahead, stepwise = run_all(False), run_all(True)
    for it in range(steps):
        for a, b in zip(ahead[it], stepwise[it]):
            assert np.array_equal(a, b, equal_nan=True), it
    assert (ahead[-1][0] > 0).sum() > 100  # (depths were found)
