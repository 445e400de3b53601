"""Randomised parity sweeps as functions (one configuration per call; AssertionError on a mismatch), shared by
tests/test_sweep_gpu.py (a slice of seeds that changes whenever the kernel sources change) and the command-line tools under
profiles/tools/random_sweep*.py (arbitrary seed ranges).  TEST INFRASTRUCTURE: the oracle is the checker.

Each configuration = a random parameter set, camera, mounting and scanner from tests/test_randomized_gpu.py's generator.
  single     one frame per call (route: default / fused / wave-only / dense - the kernel routes of tests/conftest.py)
  batch      a launch set of B ragged frames through the shipped library's batched entry points
  estimate   the production call: plane ESTIMATED inside CalculateDepth (seeded RANSAC / semantic label image)
  tracklets  S ragged sequences x 3 frames through the batched tracklet layer (float32 outputs)
"""
import hashlib
import os
from pathlib import Path

import numpy as np

from mono_lidar_depth_amd import ExceptionPclInvalid, GroundPlane, RansacPlane, SemanticPlane, synth

from helpers import assert_depth_parity, make_estimator, make_oracle, run_oracle
from test_randomized_gpu import _random_setup

ROOT = Path(__file__).resolve().parent.parent
LABELS = (6, 7, 8, 9)


def sweep_base() -> int:
    """First seed of this tree's slice: derived from the kernel / C-ABI sources (sha256), so that every change of the code
    under test is checked on configurations no earlier tree has seen.  (`git rev-parse HEAD` would serve, but `.git/` does
    not travel to the GPU box.)  MLD_SWEEP_BASE overrides it (reproducing a failure: the seed is in the test id)."""
    env = os.environ.get("MLD_SWEEP_BASE")
    if env:
        return int(env)
    h = hashlib.sha256()
    src = ROOT / "mono_lidar_depth_amd" / "csrc"
    for f in sorted(src.glob("*")):
        if f.suffix in (".hip", ".h", ".cpp"):
            h.update(f.read_bytes())
    return 100000 + int.from_bytes(h.digest()[:4], "little") % 900000 * 1000  # (slices of 1000 seeds, disjoint per tree)


def mask_of(inl, n, dev):
    import torch
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return torch.from_numpy(m.view(np.int32)).to(dev)


def check_single(seed, dense128=False):
    """One frame per call on the library / route the caller selected.  Returns (max |depth - oracle|, oracle types)."""
    P, cam, T, scanner, kw = _random_setup(seed)
    nfeat = 900
    if dense128:  # the 128 x 4096 cloud of BASELINE config 5: lists of up to 48 neighbours
        scanner, nfeat = synth.DENSE128, 4000
    cloud = synth.make_cloud(scanner, seed=200 + seed, frame=seed % 5)
    uv = synth.make_features(nfeat, seed=300 + seed, width=cam.width, height=cam.height)
    plane = synth.make_ground_plane(cloud)
    est = make_estimator(P, camera=cam, T=T)
    try:
        d, t = est.CalculateDepth(cloud, uv, GroundPlane(*plane))
        ref, (d0, t0) = run_oracle(P, cloud, uv, plane, camera=cam, T=T)
        diff = assert_depth_parity(d, t, d0, t0, exact_main=not P.do_use_PCA)
        assert np.array_equal(est.getPointIndex(), ref.point_index()), "_pointIndex differs"
        assert np.array_equal(est.getPixelMap(), ref.pixel_map()), "pixel map differs"
    finally:
        est.close()
    return float(diff.max(initial=0.0)), t0


def check_batch(seed, B=5):
    """A launch set of B frames (different frames of the scanner, ragged feature counts), planes known at projection time.
    Returns (max |depth - oracle|, frames checked)."""
    import torch
    dev = torch.device("cuda:0")
    P, cam, T, scanner, kw = _random_setup(seed)
    clouds = [synth.make_cloud(scanner, seed=200 + seed, frame=(seed + b) % 7) for b in range(B)]
    uvs = [synth.make_features(600 + 97 * b, seed=300 + seed + 1000 * b, width=cam.width, height=cam.height) for b in range(B)]
    planes = [synth.make_ground_plane(c) for c in clouds]
    est = make_estimator(P, camera=cam, T=T, max_frames=B, max_features=max(u.shape[0] for u in uvs))
    worst = 0.0
    try:
        t_clouds = [torch.from_numpy(c).to(dev) for c in clouds]
        t_uvs = [torch.from_numpy(u).to(dev) for u in uvs]
        t_masks = [mask_of(p[1], c.shape[0], dev) for p, c in zip(planes, clouds)]
        t_depth = [torch.full((u.shape[0],), 7.0, dtype=torch.float64, device=dev) for u in uvs]
        t_type = [torch.full((u.shape[0],), -7, dtype=torch.int32, device=dev) for u in uvs]
        torch.cuda.synchronize()
        batch = est.prepareBatch(t_clouds, t_uvs, t_depth, t_type, np.stack([p[0] for p in planes]), t_masks, stride_bytes=16)
        est.runBatch(batch)  # (setInputClouds with the planes known at projection time + CalculateDepths)
        est.synchronize()
        for b in range(B):
            _, (d0, ty0) = run_oracle(P, clouds[b], uvs[b], planes[b], camera=cam, T=T)
            diff = assert_depth_parity(t_depth[b].cpu().numpy(), t_type[b].cpu().numpy(), d0, ty0, exact_main=not P.do_use_PCA)
            worst = max(worst, float(diff.max(initial=0.0)))
    finally:
        est.close()
    return worst, B


def check_estimate(seed):
    """The plane estimated inside the call: coefficients and inlier sets bit-equal to the oracle's estimate for the same
    request, then the depths.  Returns (max |depth - oracle|, kind) with kind in {"ransac", "semantic", "invalid"}."""
    P, cam, T, scanner, kw = _random_setup(seed)
    rng = np.random.default_rng(77000 + seed)
    P = P.replace(do_use_ransac_plane=1,
                  ransac_plane_distance_treshold=float(rng.choice([0.05, 0.1, 0.3])),
                  ransac_plane_refinement_treshold=float(rng.choice([0.05, 0.15, 0.4])),
                  ransac_plane_max_iterations=int(rng.choice([20, 200, 1000])),
                  ransac_plane_use_refinement=int(rng.random() < 0.8),
                  ransac_plane_min_z=float(rng.choice([-1001.0, -3.0])), ransac_plane_max_z=float(rng.choice([1000.0, -0.5])))
    cloud = synth.make_cloud(scanner, seed=200 + seed, frame=seed % 5)
    uv = synth.make_features(700, seed=300 + seed, width=cam.width, height=cam.height)
    semantic = bool(seed & 1)
    est = make_estimator(P, camera=cam, T=T)
    ref = make_oracle(P, camera=cam, T=T)
    ref.set_cloud(cloud)
    try:
        if semantic:
            # (rendered with the KITTI-like camera of the synthetic scenes at this camera's size: for a random camera the
            #  labels do not line up with the projection - any image is a valid request, both sides read the same one)
            img = synth.make_label_image(cloud, width=cam.width, height=cam.height)
            gp = SemanticPlane(img, LABELS, float(P.ransac_plane_refinement_treshold))
        else:
            gp = RansacPlane(seed=seed + 1)
        try:
            d, t = est.CalculateDepth(cloud, uv, gp)
            failed = False
        except ExceptionPclInvalid:
            failed = True
        try:
            c0, inl0 = (ref.estimate_semantic_plane(img, LABELS, float(P.ransac_plane_refinement_treshold)) if semantic
                        else ref.estimate_ground_plane(seed + 1))
            ref_failed = False
        except Exception:  # noqa: BLE001
            ref_failed = True
        assert failed == ref_failed, f"estimation outcome differs: hip failed={failed}, oracle failed={ref_failed}"
        if failed:
            return 0.0, "invalid"
        assert np.array_equal(gp.getModelCoeffs(), c0), "plane coefficients differ"
        assert np.array_equal(gp.getInlinersIndex(), inl0), "inlier sets differ"
        d0, ty0 = ref.calculate_depth(uv)
        diff = assert_depth_parity(d, t, d0, ty0, exact_main=not P.do_use_PCA)
    finally:
        est.close()
    return float(diff.max(initial=0.0)), "semantic" if semantic else "ransac"


def check_tracklets(seed, S=3, NT=2600):
    """S ragged sequences x 3 frames through mld_tracklets_depths_device (two banks of slots, the previous frame's slot
    resident, feature groups when the sequences are few; list capacities default / 48-24 by seed parity) against the CPU
    restatement of TrackletDepthModule::process.  Returns (max |d - oracle| in float32 outputs, sequence-frames checked)."""
    import torch
    from mono_lidar_depth_amd import TrackletBatch
    from oracle import oracle
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    P, cam, T, scanner, kw = _random_setup(seed)
    if P.do_use_PCA:   # (the comparison below is written for the 1e-4 m paths; PCA configurations are swept by check_single)
        P = P.replace(do_use_PCA=0)
    rng = np.random.default_rng(88000 + seed)
    n_tracks = [NT - 311 * s for s in range(S)]  # ragged
    tb = TrackletBatch(P, cam, T, S, max(n_tracks), list_capacity=(48, 24) if (seed & 1) else None)
    ref_last = [None] * S
    worst, checks = 0.0, 0
    try:
        for frame in range(3):
            host = []
            for s in range(S):
                cloud = synth.make_cloud(scanner, seed=200 + seed + 17 * s, frame=2 * frame)
                coeffs, inl = synth.make_ground_plane(cloud)
                n = n_tracks[s]
                u0 = rng.integers(-2, cam.width + 2, n).astype(np.float32)
                v0 = rng.integers(cam.height // 4, cam.height + 2, n).astype(np.float32)
                u1 = (u0 + rng.integers(-4, 5, n)).astype(np.float32)
                v1 = (v0 + rng.integers(-3, 4, n)).astype(np.float32)
                is_new = (rng.random(n) < (1.0 if frame == 0 else 0.12))
                host.append((cloud, coeffs, inl, u0, v0, u1, v1, is_new))
            d_cur = [torch.empty(n, dtype=torch.float32, device=dev) for n in n_tracks]
            d_last = [torch.full((n,), float("nan"), dtype=torch.float32, device=dev) for n in n_tracks]
            t_cur = [torch.empty(n, dtype=torch.int32, device=dev) for n in n_tracks]
            t_last = [torch.zeros(n, dtype=torch.int32, device=dev) for n in n_tracks]
            tb.frame([to(h[0]) for h in host], np.stack([h[1] for h in host]), [mask_of(h[2], h[0].shape[0], dev) for h in host],
                     [to(h[3]) for h in host], [to(h[4]) for h in host], [to(h[5]) for h in host], [to(h[6]) for h in host],
                     [to(h[7].astype(np.uint8)) for h in host], d_cur, d_last, t_cur, t_last)
            tb.est.synchronize()
            for s in range(S):
                cloud, coeffs, inl, u0, v0, u1, v1, is_new = host[s]
                ref_cur = make_oracle(P, camera=cam, T=T)
                ref_cur.set_cloud(cloud)
                ref_cur.set_ground_plane(coeffs, inl)
                e_cur, e_last, et_cur, et_last = oracle.tracklets_depth(ref_cur, ref_last[s], u0, v0, u1, v1, is_new, n_threads=8)
                dc, dl = d_cur[s].cpu().numpy(), d_last[s].cpu().numpy()
                tc, tl = t_cur[s].cpu().numpy(), t_last[s].cpu().numpy()
                assert np.array_equal(tc, et_cur), f"current types differ (frame {frame}, sequence {s})"
                assert np.array_equal(tl[is_new], et_last[is_new]), f"previous types differ (frame {frame}, sequence {s})"
                nan_c, nan_e = np.isnan(dc), np.isnan(e_cur)
                assert np.array_equal(nan_c, nan_e)
                dd = np.abs(np.where(nan_c, 0, dc).astype(np.float64) - np.where(nan_e, 0, e_cur))
                dm = float(dd.max(initial=0.0))
                dl2 = np.abs(np.nan_to_num(dl[is_new]).astype(np.float64) - np.nan_to_num(e_last[is_new]))
                dm = max(dm, float(dl2.max(initial=0.0)))
                # (outputs are float32 as FeaturePoint.d: equal up to the float32 rounding of a value within 1e-4 m)
                tol = 1e-4 + 1.2e-7 * float(np.nanmax(np.abs(np.where(nan_e, 0, e_cur)), initial=1.0))
                assert dm <= tol, f"max |d - oracle| = {dm:.3e} m (frame {frame}, sequence {s})"
                assert np.isnan(dl[~is_new]).all()
                checks += 1
                worst = max(worst, dm)
                ref_last[s] = ref_cur
    finally:
        tb.close()
    return worst, checks
