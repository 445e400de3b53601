"""TrackletDepthModule::process's GPU side as ONE C call (mld_tracklets_frame): upload of the new cloud, its ground plane
(estimated inside the call when it is not segmented yet - the reference builds a fresh one every frame,
tracklet_depth_module.cpp:269-284), the frame's only projection, the feature marshalling, both CalculateDepth calls (the
previous frame from its resident slot) and the float32 scatter.  Against the oracle, and against the two-call route
(setInputCloud + mld_tracklets_depth) it replaces."""
import ctypes as C

import numpy as np
import pytest

from mono_lidar_depth_amd import GroundPlane, RansacPlane, TrackletDepthModule, capi, synth
from oracle import oracle

from helpers import kitti_camera, make_oracle
from test_tracklets_gpu import _tracks

pytestmark = pytest.mark.gpu
LABELS = (6, 7, 8, 9)


def _sequence(plane_kind, one_call, frames=4, n_tracks=2500, scanner=synth.HDL64_KITTI, seed=21):
    P = capi.params_c0()
    cam = kitti_camera()
    mod = TrackletDepthModule(P, cam, synth.T_CAM_LIDAR)
    mod.one_call = one_call
    rng = np.random.default_rng(seed)
    ids_prev, ref_last, known = None, None, set()
    out = []
    for frame in range(frames):
        cloud = synth.make_cloud(scanner, seed=seed, frame=frame * 2)
        ids, u0, v0, u1, v1 = _tracks(rng, ids_prev, n_tracks, 0.15, cam.width, cam.height)
        ref_cur = make_oracle(P)
        ref_cur.set_cloud(cloud)
        if plane_kind == "semantic":
            img = synth.make_label_image(cloud)
            d_cur, d_last, is_new = mod.process(cloud, ids, u0, v0, u1, v1, None, img=img)
            ref_cur.estimate_semantic_plane(img, LABELS, P.ransac_plane_refinement_treshold)
        elif plane_kind == "ransac":
            gp = RansacPlane(seed=100 + frame)
            d_cur, d_last, is_new = mod.process(cloud, ids, u0, v0, u1, v1, gp)
            c0, inl0 = ref_cur.estimate_ground_plane(100 + frame)
            assert gp.isSegmented() and np.array_equal(gp.getModelCoeffs(), c0)
            assert np.array_equal(gp.getInlinersIndex(), inl0)
        else:
            coeffs, inl = synth.make_ground_plane(cloud)
            d_cur, d_last, is_new = mod.process(cloud, ids, u0, v0, u1, v1, GroundPlane(coeffs, inl))
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
        out.append((d_cur.copy(), d_last.copy(), t_cur.copy(), t_last.copy()))
        known = set(int(i) for i in ids)
        ids_prev, ref_last = ids, ref_cur
    return out


@pytest.mark.parametrize("plane_kind", ["supplied", "ransac", "semantic"])
def test_one_call_process_equals_the_oracle_and_the_two_call_route(plane_kind, feature_kernel_path):
    a = _sequence(plane_kind, one_call=True)
    b = _sequence(plane_kind, one_call=False)
    for (dc, dl, tc, tl), (dc2, dl2, tc2, tl2) in zip(a, b):
        assert np.array_equal(tc, tc2) and np.array_equal(tl, tl2)
        assert np.allclose(dc, dc2, rtol=0, atol=1e-4, equal_nan=True) and np.allclose(dl, dl2, rtol=0, atol=1e-4, equal_nan=True)


def test_failed_estimation_still_answers_the_previous_frame():
    """tracklet_depth_module.cpp:318-347: ExceptionPclInvalid on the current cloud leaves its depths at -1; the features of
    new tracks on the PREVIOUS frame are still answered.  Both for a NaN cloud and for one of fewer than three points."""
    P = capi.params_c0()
    cam = kitti_camera()
    mod = TrackletDepthModule(P, cam, synth.T_CAM_LIDAR)
    est, lib = mod.estimator, mod.estimator._lib
    rng = np.random.default_rng(7)
    cloud0 = synth.make_cloud(synth.HDL64_KITTI, seed=33, frame=0)
    ids, u0, v0, u1, v1 = _tracks(rng, None, 1500, 0.2, cam.width, cam.height)
    mod.process(cloud0, ids, u0, v0, u1, v1, RansacPlane(seed=4))   # frame 0 on slot 0
    ref_last = make_oracle(P)
    ref_last.set_cloud(cloud0)
    ref_last.estimate_ground_plane(4)
    for bad in (np.full((700, 4), np.nan, np.float32), np.array([[1, 0, -1.7, 0], [2, 0, -1.7, 0]], np.float32)):
        n = 900
        un = rng.uniform(0, cam.width, n).astype(np.float32)
        vn = rng.uniform(100, cam.height, n).astype(np.float32)
        uo = (un + 1).astype(np.float32)
        vo = (vn + 1).astype(np.float32)
        is_new = (rng.random(n) < 0.3).astype(np.uint8)
        d_cur = np.full(n, 7.0, np.float32)
        d_last = np.full(n, np.nan, np.float32)
        t_cur = np.zeros(n, np.int32)
        t_last = np.zeros(n, np.int32)
        nn = C.c_int64(0)
        req = capi.MldPlaneRequest()
        req.kind, req.seed = capi.MLD_PLANE_RANSAC, 1
        res = capi.MldPlaneResult()
        rc = lib.mld_tracklets_frame(est._ctx, 1, 0, bad.ctypes.data, bad.shape[0], 16, C.byref(req), None, None, 0,
                                     un.ctypes.data, vn.ctypes.data, uo.ctypes.data, vo.ctypes.data, is_new.ctypes.data, n,
                                     d_cur.ctypes.data, d_last.ctypes.data, t_cur.ctypes.data, t_last.ctypes.data,
                                     C.byref(nn), C.byref(res))
        assert rc == capi.MLD_ERR_CLOUD_TOO_SMALL and res.status == 1
        assert (d_cur == -1).all() and nn.value == int(is_new.sum())
        nw = is_new.astype(bool)
        uv_old = np.stack([np.trunc(uo[nw]).astype(np.float64), np.trunc(vo[nw]).astype(np.float64)], axis=1)
        d0, t0 = ref_last.calculate_depth(uv_old, 4)
        assert np.array_equal(t_last[nw], t0)
        assert np.allclose(d_last[nw], d0.astype(np.float32), rtol=0, atol=1e-4)
        assert np.isnan(d_last[~nw]).all()


def test_no_tracks_and_no_previous_frame():
    P = capi.params_c0()
    cam = kitti_camera()
    mod = TrackletDepthModule(P, cam, synth.T_CAM_LIDAR)
    cloud = synth.make_cloud(synth.VLP16, seed=2, frame=0)
    e = np.zeros(0, np.float32)
    d_cur, d_last, is_new = mod.process(cloud, np.zeros(0, np.int64), e, e, e, e, RansacPlane(seed=2))
    assert d_cur.size == 0 and d_last.size == 0
    ids = np.arange(50)
    u = np.linspace(100, 1100, 50).astype(np.float32)
    v = np.full(50, 250, np.float32)
    d_cur, d_last, is_new = mod.process(cloud, ids, u, v, u, v, RansacPlane(seed=3))
    assert is_new.all() and np.isfinite(d_cur).all() and np.isfinite(d_last).all()
