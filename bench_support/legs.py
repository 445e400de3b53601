"""Secondary legs of the benchmark (everything beside the headline): PCIe-inclusive latency / streaming, the plane-estimated
schedule and the other single-GPU BASELINE configs.  Run by `bench_support/run_legs.py`; results go to the detail file."""
from __future__ import annotations

import json
import os
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

from .resident import Resident, kernel_times, mask_words, timed_resident
from .rooflines import HBM_PEAK_GBS, config_roofline, design_bytes_project


def latency_leg(P, cam, T, clouds, planes, uvs, n_frames, device=0):
    """One frame per call through the host-pointer entry points (the reference's ROS usage): H2D of the cloud, the
    plane's inlier list and the features, kernels, D2H of depths/types, synchronise.  PCIe-inclusive; reported beside
    the resident-throughput `value`, never as it.  Three variants of the same call:
      supplied            the GroundPlane handed in is segmented (mld_calculate_depth_frame)
      estimated.ransac    a fresh RansacPlane per frame - setInputCloud estimates it (DepthEstimator.cpp:275-283)
      estimated.semantic  a fresh SemanticPlane per frame - what TrackletDepthModule::process does
                          (tracklet_depth_module.cpp:269-284); both through mld_calculate_depth_frame_estimate
    each with the median / p99 of the un-instrumented call and a phase breakdown (hipEvents + host clock,
    mld_frame_timing) from a second, instrumented pass."""
    from mono_lidar_depth_amd import DepthEstimator, GroundPlane, RansacPlane, SemanticPlane, synth
    from oracle import oracle
    est = DepthEstimator(device=device, max_points=clouds[0].shape[0], max_features=uvs[0].shape[0])
    est.InitConfig(P)
    est.Initialize(cam, T)
    labels = (6, 7, 8, 9)
    thr = float(P.ransac_plane_refinement_treshold)
    imgs = [synth.make_label_image(c) for c in clouds[:4]]

    def plane_for(kind, it):
        i = it % len(clouds)
        if kind == "supplied":
            return GroundPlane(*planes[i])
        if kind == "ransac":
            return RansacPlane(seed=it + 1)
        return SemanticPlane(imgs[i % len(imgs)], labels, thr)

    def run(kind, clouds=clouds):
        ts, host = [], []
        last = None
        for it in range(n_frames + 10):
            i = it % len(clouds) if kind != "semantic" else it % len(imgs)
            gp = plane_for(kind, it)
            t0 = time.perf_counter()
            d, t = est.CalculateDepth(clouds[i], uvs[i], gp)
            ts.append(time.perf_counter() - t0)
            host.append(est.frameTiming())  # (host-clock phases of the un-instrumented call)
            last = (it, i, d, t)
        ts = np.array(ts[10:]) * 1e3
        host_keys = ("pre_us", "copycall_us", "api_us", "wait_us", "total_us")
        host_med = {k: float(np.median([h[k] for h in host[10:]])) for k in host_keys}
        host_med["wrapper_us"] = float(np.median(ts) * 1e3 - host_med["total_us"])  # Python mirror around the C call
        # where the slowest calls lose their time: the host phases of the calls at or above the 99th percentile
        slow = np.nonzero(ts >= np.percentile(ts, 99))[0]
        host_tail = {k: float(np.mean([host[10 + j][k] for j in slow])) for k in host_keys}
        host_tail["wrapper_us"] = float(np.mean(ts[slow]) * 1e3 - host_tail["total_us"])
        host_tail["calls"] = int(len(slow))
        # phase breakdown: the same call with the phase events on
        est.timingEnable(True)
        ph = []
        for it in range(min(60, n_frames)):
            i = it % len(clouds) if kind != "semantic" else it % len(imgs)
            est.CalculateDepth(clouds[i], uvs[i], plane_for(kind, it))
            ph.append(est.frameTiming())
        est.timingEnable(False)
        breakdown = {k: float(np.median([p[k] for p in ph[5:]])) for k in ph[0]} if len(ph) > 5 else None
        out = {"frames": int(n_frames), "ms_per_frame_median": float(np.median(ts)),
               "ms_per_frame_p99": float(np.percentile(ts, 99)),
               "associations_per_s": float(uvs[0].shape[0] / np.median(ts) * 1e3),
               "breakdown_us_median": breakdown, "host_us_median": host_med, "host_us_p99_calls": host_tail}
        if kind != "supplied":  # the last frame against the oracle with the restatement's plane for the same request
            it, i, d, t = last
            ref = oracle.OracleDepthEstimator(P, cam.as_struct(), T)
            ref.set_cloud(clouds[i])
            if kind == "ransac":
                ref.estimate_ground_plane(it + 1)
            else:
                ref.estimate_semantic_plane(imgs[i % len(imgs)], labels, thr)
            d0, t0_ = ref.calculate_depth(uvs[i], 8)
            out["verified"] = bool(np.array_equal(t, t0_) and np.allclose(d, d0, rtol=0, atol=1e-4, equal_nan=True))
        return out

    # the same frames as 32-byte pcl::PointXYZI records (x,y,z,pad | intensity,pad,pad,pad) - the layout the reference's
    # caller hands over (DepthEstimator.h:62-63): twice the bytes cross PCIe for the same points
    def as_pcl(c):
        out = np.zeros((c.shape[0], 8), dtype=np.float32)
        out[:, :3] = c[:, :3]
        out[:, 4] = c[:, 3]
        return out
    clouds32 = [as_pcl(c) for c in clouds]

    def with32(kind):
        leg = run(kind)
        leg["stride_bytes"] = 16
        leg["stride32"] = {**run(kind, clouds32), "stride_bytes": 32,
                           "cloud_bytes": int(clouds32[0].nbytes)}
        return leg

    sup = with32("supplied")
    # the same call with the clouds in PINNED host memory (a caller that allocates its cloud buffers with hipHostMalloc /
    # hipHostRegister): the copy no longer blocks the calling thread and runs at the DMA rate - informational, the
    # reference's nodelets hand over pageable memory
    pinned = None
    try:
        import torch
        pc = [torch.from_numpy(c).pin_memory().numpy() for c in clouds]
        pinned = run("supplied", pc)
        del pc
        # Why a pinned source buys nothing here: the call's critical path is the cloud's H2D DMA on the GPU timeline
        # (breakdown h2d_us) either way.  A pageable source blocks the caller inside the copy call while the runtime stages
        # it (host copycall_us ~ the DMA time); a pinned one returns at once (copycall_us ~ 3) and the caller waits the same
        # time in the final synchronise instead (wait_us) - meanwhile its helper thread's staging of the small inputs, hidden
        # behind the blocking copy in the pageable case, shows up in api_us.
        try:
            pinned["explanation"] = {
                "h2d_us_pageable": sup["breakdown_us_median"]["h2d_us"], "h2d_us_pinned": pinned["breakdown_us_median"]["h2d_us"],
                "copycall_us_pageable": sup["host_us_median"]["copycall_us"], "copycall_us_pinned": pinned["host_us_median"]["copycall_us"],
                "wait_us_pageable": sup["host_us_median"]["wait_us"], "wait_us_pinned": pinned["host_us_median"]["wait_us"],
                "note": "same GPU-side H2D time; the host's blocking moves from the copy call to the final synchronise"}
        except (KeyError, TypeError):
            pass
    except Exception as e:  # noqa: BLE001
        pinned = {"error": str(e)}
    res = {
        "path": "host pointers, one frame per call: setInputCloud (H2D 2.1 MB) + ground plane (inlier list H2D) + "
                "CalculateDepth (uv H2D, kernels, depth/type D2H, sync)",
        **sup,
        "breakdown_keys": "breakdown_us_median (instrumented pass: hipEvents on the context's stream cost the call tens of "
                          "microseconds): h2d = cloud copy, plane = plane estimation kernels, kernels = projection + feature "
                          "kernel, d2h = result hand-over, gpu = first to last event.  host_us_median (the un-instrumented "
                          "calls of the median above): pre = host time before the cloud copy is submitted, copycall = host "
                          "time inside the cloud's hipMemcpyAsync (pageable source), api = entry to last enqueue, wait = "
                          "final synchronise, total = the C call, wrapper = the Python mirror around it",
    }
    res["pinned_source"] = pinned
    # TrackletDepthModule::process (tracklet_depth_module.cpp:261-396), the ROS callback itself: a new cloud, a fresh
    # SemanticPlane from the frame's label image and 2000 tracks (10 % new: their previous features are answered on the
    # resident previous frame) per call.  C-ABI time only (the Python mirror's tracklet bookkeeping is not the product):
    # ONE call (mld_tracklets_frame) against the route it replaces (setInputCloud + mld_tracklets_depth: two calls, two
    # synchronisations, nine small copies).
    if P.do_use_ransac_plane:
        from mono_lidar_depth_amd import TrackletDepthModule
        rng = np.random.default_rng(11)
        n_tr = uvs[0].shape[0]
        proc = {}
        for name, one in (("one_call", True), ("two_calls", False)):
            mod = TrackletDepthModule(P, cam, T, device=device, keep_history=False)
            mod.one_call = one
            ids = np.arange(n_tr, dtype=np.int64)
            nxt = n_tr
            ts = []
            for src_name, src in (("stride16", clouds), ("stride32", clouds32)) if one else (("stride16", clouds),):
                ts = []
                for it in range(min(n_frames, 120) + 10):
                    i = it % len(imgs)
                    fresh = rng.choice(n_tr, n_tr // 10, replace=False)  # a tenth of the tracks are replaced by new ones
                    ids = ids.copy()
                    ids[fresh] = np.arange(nxt, nxt + fresh.size)
                    nxt += fresh.size
                    u0 = uvs[i][:, 0].astype(np.float32)
                    v0 = uvs[i][:, 1].astype(np.float32)
                    mod.process(src[i], ids, u0, v0, u0 + 1.0, v0 + 1.0, None, img=imgs[i])
                    ts.append(mod.last_abi_seconds)
                ts = np.array(ts[10:]) * 1e3
                leg = {"ms_per_frame_median": float(np.median(ts)), "ms_per_frame_p99": float(np.percentile(ts, 99)),
                       "frames": int(ts.size)}
                if src_name == "stride16":
                    proc[name] = leg
                else:
                    proc[name]["stride32"] = leg
            mod.estimator.close()
        proc["path"] = ("TrackletDepthModule::process per frame: host cloud + fresh SemanticPlane (label image) + "
                        f"{n_tr} tracks, 10 % new; C-ABI calls only")
        proc["tracks"] = n_tr
        res["process"] = proc
    if P.do_use_ransac_plane:
        res["estimated"] = {
            "path": "the same call with a GroundPlane that is not segmented yet (the reference's production call): plane "
                    "estimated on the GPU ahead of the projection, one C call, one synchronisation "
                    "(mld_calculate_depth_frame_estimate)",
            "ransac": with32("ransac"), "semantic": with32("semantic")}
    est.close()
    return res



def streaming_leg(P, cam, T, clouds, planes, uvs, device, frames_per_batch, n_batches, stride_floats=4, pack_threads=0):
    """Frames streamed from pinned host memory: double-buffered H2D copies on a copy stream overlapped with the kernels
    on the context's stream, results copied back.  PCIe-inclusive THROUGHPUT (the latency leg is the unpipelined
    counterpart); reported beside `value`, never as it.
    pack_threads > 0: the frames start as 32-byte pcl::PointXYZI records in ordinary (pageable) host memory, as the
    reference's caller holds them; `pack_threads` host threads stage them into the pinned batch as packed 16-byte records
    (mld_pack_points_host) inside the timed pipeline - the staging copy a driver makes anyway, at half the PCIe bytes."""
    import torch
    from mono_lidar_depth_amd import DepthEstimator, capi
    dev = torch.device("cuda", device)
    S, N, F = frames_per_batch, clouds[0].shape[0], uvs[0].shape[0]
    U = len(clouds)
    words = (N + 31) // 32
    # pinned host batch (what a driver thread would fill from the sensor queue) and two device buffer sets
    SF = int(stride_floats)  # 4: packed xyzi; 8: pcl::PointXYZI records (x,y,z,pad | intensity,pad,pad,pad)
    h_cloud = torch.zeros((S, N, SF), dtype=torch.float32).pin_memory()
    h_mask = torch.empty((S, words), dtype=torch.int32).pin_memory()
    h_uv = torch.empty((S, F, 2), dtype=torch.float64).pin_memory()
    coeffs = np.empty((S, 4), dtype=np.float32)
    for b in range(S):
        h_cloud[b, :, :3] = torch.from_numpy(clouds[b % U][:, :3])
        h_cloud[b, :, 4 if SF == 8 else 3] = torch.from_numpy(clouds[b % U][:, 3])
        h_mask[b] = torch.from_numpy(mask_words(planes[b % U][1], N))
        h_uv[b] = torch.from_numpy(uvs[b % len(uvs)])
        coeffs[b] = planes[b % U][0]
    h_depth = [torch.empty((S, F), dtype=torch.float64).pin_memory() for _ in range(2)]
    h_type = [torch.empty((S, F), dtype=torch.int32).pin_memory() for _ in range(2)]
    pool = src32 = None
    h_stage = [h_cloud, h_cloud]
    if pack_threads > 0:
        assert SF == 4
        from concurrent.futures import ThreadPoolExecutor
        lib = capi.load()
        src32 = []
        for c in clouds:
            a = np.zeros((N, 8), dtype=np.float32)
            a[:, :3] = c[:, :3]
            a[:, 4] = c[:, 3]
            src32.append(a)
        h_stage = [h_cloud, torch.zeros((S, N, 4), dtype=torch.float32).pin_memory()]  # (one pinned batch per buffer set)
        pool = ThreadPoolExecutor(max_workers=int(pack_threads))
        row_bytes = N * 16

        def pack_batch(k):
            base = h_stage[k].data_ptr()
            futs = [pool.submit(lib.mld_pack_points_host, base + b * row_bytes, src32[b % U].ctypes.data, N, 32, 1)
                    for b in range(S)]
            assert all(f.result() == 0 for f in futs)
    est = DepthEstimator(device=device, max_frames=S)
    est.InitConfig(P)
    est.Initialize(cam, T)
    compute = torch.cuda.ExternalStream(est.stream, device=dev)
    copy_in = torch.cuda.Stream(device=dev)
    copy_out = torch.cuda.Stream(device=dev)
    bufs, batches = [], []
    for _ in range(2):
        d = {"cloud": torch.empty((S, N, SF), dtype=torch.float32, device=dev),
             "mask": torch.empty((S, words), dtype=torch.int32, device=dev),
             "uv": torch.empty((S, F, 2), dtype=torch.float64, device=dev),
             "depth": torch.empty((S, F), dtype=torch.float64, device=dev),
             "type": torch.empty((S, F), dtype=torch.int32, device=dev)}
        bufs.append(d)
        batches.append(est.prepareBatch([d["cloud"][b] for b in range(S)], [d["uv"][b] for b in range(S)],
                                        [d["depth"][b] for b in range(S)], [d["type"][b] for b in range(S)], coeffs,
                                        [d["mask"][b] for b in range(S)], stride_bytes=4 * SF))
    torch.cuda.synchronize()
    copied = [None, None]
    done = [None, None]

    def submit(i):
        k = i % 2
        if pool is not None:
            if copied[k] is not None:
                copied[k].synchronize()  # the previous upload from this pinned batch has left it
            pack_batch(k)
        with torch.cuda.stream(copy_in):
            if done[k] is not None:
                copy_in.wait_event(done[k])  # the buffer set is free once its previous results are on the host
            bufs[k]["cloud"].copy_(h_stage[k], non_blocking=True)
            bufs[k]["mask"].copy_(h_mask, non_blocking=True)
            bufs[k]["uv"].copy_(h_uv, non_blocking=True)
            copied[k] = copy_in.record_event()
        compute.wait_event(copied[k])
        est.runBatch(batches[k])
        ev = torch.cuda.Event()
        ev.record(compute)
        # torch's allocators only ever see torch-owned streams (the context's stream is used for event traffic alone)
        copy_out.wait_event(ev)
        with torch.cuda.stream(copy_out):
            h_depth[k].copy_(bufs[k]["depth"], non_blocking=True)
            h_type[k].copy_(bufs[k]["type"], non_blocking=True)
            done[k] = copy_out.record_event()

    for i in range(2):
        submit(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_batches):
        submit(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    packed_ok = None
    if pool is not None:
        pool.shutdown()
        packed_ok = bool(np.array_equal(h_stage[1][S - 1].numpy(), clouds[(S - 1) % U], equal_nan=True))  # (no-return points are NaN)
    est.close()
    frames = S * n_batches
    h2d = frames * (N * 4 * SF + words * 4 + F * 16)
    return {
        "path": ("32-byte pcl::PointXYZI records in pageable host memory, staged by %d host threads into pinned 16-byte "
                 "batches (mld_pack_points_host) inside the pipeline, " % pack_threads if pool is not None else
                 "pinned host batches, ") + "double-buffered H2D on a copy stream overlapped with the kernels, "
                "depths/types copied back",
        **({"pack_threads": int(pack_threads), "packed_equals_source": packed_ok} if pool is not None else {}),
        "frames_per_batch": S, "batches": n_batches, "frames": frames, "stride_bytes": 4 * SF,
        "frames_per_s": frames / el,
        "associations_per_s": frames * F / el,
        "ms_per_frame": 1e3 * el / frames,
        "h2d_GBps": h2d / el / 1e9,
    }



def config2_k_leg(P, cam, T, device, B, F, steps=20, contexts=2, shared_mode=1):
    """BASELINE config 2 "at its stated neighbour count" (k = 7): the same 64x2048 clouds, parameters and schedule as the
    headline, but every feature sits on a LiDAR return whose search window (6 x 9, parameters.yaml:14,17;
    NeighborFinderPixel.cpp:67-88) holds at least six returns - no feature is settled by the classification alone, every
    one runs the neighbour gather, the histogram and a plane fit.  (The headline's uniformly random features see 2.3
    neighbours on average and 42 % of them none.)  Reported beside the headline, never as it; every frame of every
    output set is checked against the oracle."""
    from mono_lidar_depth_amd import capi, synth, traffic
    res = Resident(P, cam, T, synth.HDL64, B, min(16, B), F, 2, device, contexts=contexts, shared_mode=shared_mode,
                   near_points="k")
    loops, kt = timed_resident(res, steps, 3, True, 2, repeats=3, min_timed_s=0.3)
    el = float(np.median(loops))
    ok, rep = res.verify(-1)
    hist = np.zeros(capi.MLD_RESULT_TYPE_COUNT, dtype=np.int64)
    for t_set in res.out_type:
        hist += res.ests[0].resultHistogram(t_set.reshape(-1))
    last = res.last_context()
    stats = [traffic.frame_bytes(P, cam.width, cam.height, res.N, last.getVisibleCount(b), last.getPixelMap(b), res.uvs_h[b],
                                 res.all_type[b].cpu().numpy()) for b in range(0, B, max(1, B // 4))][:4]
    out = {
        "workload": (f"BASELINE config 2 at k = 7: 64x2048 cloud x {F} features/frame on returns whose 6 x 9 search window "
                     f"holds >= 6 returns, {B} device-resident frames per step, {contexts} contexts alternating"),
        "value": B * F * steps / el, "unit": "feature-depth associations/s", "ms_per_step": 1e3 * el / steps,
        "ms_per_frame": 1e3 * el / steps / B, "frames_per_step": B,
        "k1_mean": float(np.mean([s_["k1_mean"] for s_ in stats])),
        "k2_mean_fallback": float(np.mean([s_["k2_mean_fallback"] for s_ in stats])),
        "fallback_features_per_frame": float(np.mean([s_["fallback_features"] for s_ in stats])),
        "result_types": {capi.RESULT_TYPE_NAMES[i]: int(c) for i, c in enumerate(hist) if c},
        "success_fraction": float((hist[1] + hist[16]) / max(1, hist.sum())),
        "dead_features": int(hist[2]),
        "kernels_ms_per_launch": {k: v["avg_ms"] for k, v in kt.items()},
        "verified": ok, "frames_checked": rep["frames_checked"], "max_abs_depth_diff_m": rep["max_abs_depth_diff_m"],
        "poison_left": rep["poison_left"],
        "roofline": config_roofline("2k", kt, res.S),
    }
    res.close()
    return out


def config3_leg(cam, T, device, B, steps=8, only_near=False):
    """BASELINE config 3: VLP-16 sparse cloud (16x1800), 5000 features/frame, device-resident, C0 parameters plus the
    threshold-treatment sweep of SURVEY.md §8(d) (Dispose/Adjust x absolute/relative)."""
    from mono_lidar_depth_amd import capi, synth
    P0 = capi.params_c0()
    out = {"workload": f"BASELINE config 3: VLP-16 16x1800 cloud x 5000 features/frame, {B} device-resident frames per "
                       "step, plane known at projection", "modes": {}}
    def run_mode(kw, near):
        res = Resident(P0.replace(**kw), cam, T, synth.VLP16, B, 8, 5000, 3, device, near_points=near)
        loops, kt = timed_resident(res, steps, 2, True, 2)
        el = loops[0]
        ok, rep = res.verify(-1)  # every frame (the oracle sets each distinct cloud once)
        hist = np.zeros(capi.MLD_RESULT_TYPE_COUNT, dtype=np.int64)
        for b in range(0, B, max(1, B // 16)):
            hist += res.ests[0].resultHistogram(res.all_type[b])
        m = {
            "roofline": config_roofline("3n" if near else "3", kt, B),
            "associations_per_s": B * 5000 * steps / el, "ms_per_frame": 1e3 * el / steps / B, "verified": ok,
            "frames_checked": rep["frames_checked"],
            "max_abs_depth_diff_m": rep["max_abs_depth_diff_m"],
            "kernels_ms_per_launch": {k: v["avg_ms"] for k, v in kt.items()},
            "success_fraction": float((hist[1] + hist[16]) / max(1, hist.sum())),
        }
        if near:
            m["result_types"] = {capi.RESULT_TYPE_NAMES[i]: int(c) for i, c in enumerate(hist) if c}
        return res, kt, m

    if only_near:  # (counter passes: the one leg, nothing else in the trace)
        res, kt, m = run_mode({}, True)
        res.close()
        return {"workload": out["workload"], "near_returns": {"modes": {"c0_dispose": m}}, "verified": m["verified"]}
    for name, kw in (("c0_dispose", {}),
                     ("adjust_relative", dict(treshold_depth_mode=1, treshold_depth_local_mode=1,
                                              treshold_depth_local_valuetype=1)),
                     ("adjust_absolute", dict(treshold_depth_mode=1, treshold_depth_local_mode=1,
                                              treshold_depth_local_valuetype=0))):
        res, kt, out["modes"][name] = run_mode(kw, False)
        if name == "c0_dispose":
            db = [design_bytes_project(res.clouds_h[u], cam, T, res.planes_h[u][1]) for u in range(2)]
            pb = float(np.mean([d["bytes"] for d in db])) * B
            pms = kt["k_project_scatter"]["avg_ms"]
            out["roofline_project"] = {"design_bytes_per_launch": pb, "kernel_ms": pms,
                                       "frac": pb / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS if pms > 0 else None}
        res.close()
    # Uniformly random features almost never see a neighbour on a 16-ring cloud (rings ~25 px apart, window 9 px high): the
    # workload above is what BASELINE specifies, but it mostly measures the classification.  Second variant: the same
    # clouds with the features scattered around the image positions of the returns, so that every path behind the
    # neighbour search runs (collinear triangles, planarity / orthogonality rejections, thresholds, road fallback).
    out["near_returns"] = {"workload": "the same clouds, 5000 features per frame within a few pixels of LiDAR returns",
                           "modes": {}}
    for name, kw in (("c0_dispose", {}),
                     ("adjust_relative", dict(treshold_depth_mode=1, treshold_depth_local_mode=1,
                                              treshold_depth_local_valuetype=1))):
        res, kt, out["near_returns"]["modes"][name] = run_mode(kw, True)
        res.close()
    out["near_returns"]["verified"] = all(m["verified"] for m in out["near_returns"]["modes"].values())
    out["verified"] = all(m["verified"] for m in out["modes"].values()) and out["near_returns"]["verified"]
    return out


def config5_leg(cam, T, device, n_frames):
    """BASELINE config 5: 128x4096 dense cloud (524 288 points), 10 000 tracks/frame through the tracklet API with
    device pointers (mld_tracklets_depth_device); the previous frame's slot stays resident (no re-projection), 10 % of
    the tracks are new every frame.  One frame per call sequence, as a tracker delivers them."""
    import ctypes as C
    import torch
    from mono_lidar_depth_amd import DepthEstimator, GroundPlane, capi, synth
    from oracle import oracle
    dev = torch.device("cuda", device)
    P = capi.params_c0()
    U, n_tracks = 4, 10000
    clouds_h = [synth.make_cloud(synth.DENSE128, seed=5, frame=f) for f in range(U)]
    planes_h = [synth.make_ground_plane(c) for c in clouds_h]
    N = clouds_h[0].shape[0]
    d_clouds = [torch.from_numpy(c).to(dev) for c in clouds_h]
    d_masks = [torch.from_numpy(mask_words(p[1], N)).to(dev) for p in planes_h]
    rng = np.random.default_rng(5)
    K = 8  # distinct track sets
    sets = []
    for k in range(K):
        u0 = rng.integers(0, cam.width, n_tracks).astype(np.float32)
        v0 = rng.integers(100, cam.height, n_tracks).astype(np.float32)
        u1 = (u0 + rng.integers(-3, 4, n_tracks)).astype(np.float32)
        v1 = (v0 + rng.integers(-2, 3, n_tracks)).astype(np.float32)
        new = np.zeros(n_tracks, dtype=np.uint8)
        new[rng.choice(n_tracks, n_tracks // 10, replace=False)] = 1
        sets.append(tuple(torch.from_numpy(a).to(dev) for a in (u0, v0, u1, v1, new)) + ((u0, v0, u1, v1, new),))
    d_cur = torch.empty(n_tracks, dtype=torch.float32, device=dev)
    d_last = torch.zeros(n_tracks, dtype=torch.float32, device=dev)
    t_cur = torch.empty(n_tracks, dtype=torch.int32, device=dev)
    t_last = torch.zeros(n_tracks, dtype=torch.int32, device=dev)
    est = DepthEstimator(device=device, max_frames=2, max_features=n_tracks)
    est.InitConfig(P)
    est.Initialize(cam, T)
    lib, ctx = est._lib, est._ctx
    ptrs = (C.c_void_p * 1)()
    cnt = (C.c_int64 * 1)(N)
    mptr = (C.c_void_p * 1)()

    def frame(it, slot_cur, have_last):
        i = it % U
        s = sets[it % K]
        est._check(lib.mld_set_cloud_device(ctx, slot_cur, d_clouds[i].data_ptr(), N, 16))
        co = (C.c_float * 4)(*[float(x) for x in planes_h[i][0]])
        est._check(lib.mld_set_ground_plane_mask_device(ctx, slot_cur, co, d_masks[i].data_ptr()))
        est._check(lib.mld_tracklets_depth_device(ctx, slot_cur, (1 - slot_cur) if have_last else -1,
                                                  s[0].data_ptr(), s[1].data_ptr(), s[2].data_ptr(), s[3].data_ptr(),
                                                  s[4].data_ptr(), n_tracks, d_cur.data_ptr(), d_last.data_ptr(),
                                                  t_cur.data_ptr(), t_last.data_ptr(), None))

    slot = 0
    est.timingEnable(True)  # (the timers' events are created during the warm-up, not inside the timed loop)
    for it in range(6):
        frame(it, slot, it > 0)
        slot = 1 - slot
    est.synchronize()
    est.timingReset()
    t0 = time.perf_counter()
    for it in range(6, 6 + n_frames):
        frame(it, slot, True)
        slot = 1 - slot
    est.synchronize()
    el = time.perf_counter() - t0
    kt = kernel_times(est)
    est.timingEnable(False)
    # check the last frame's current-slot depths against the oracle (integer-pixel features, float32 depths)
    it = 6 + n_frames - 1
    i, s = it % U, sets[it % K][5]
    ref = oracle.OracleDepthEstimator(P, cam.as_struct(), T)
    ref.set_cloud(clouds_h[i])
    ref.set_ground_plane(*planes_h[i])
    uv = np.stack([np.trunc(s[0]).astype(np.float64), np.trunc(s[1]).astype(np.float64)], axis=1)
    d0, t0_ = ref.calculate_depth(uv, 8)
    ok = bool(np.array_equal(t_cur.cpu().numpy(), t0_) and
              np.allclose(d_cur.cpu().numpy(), d0.astype(np.float32), rtol=0, atol=1e-4, equal_nan=True))
    est.close()
    assoc = n_tracks + n_tracks // 10
    db = design_bytes_project(clouds_h[0], cam, T, planes_h[0][1])
    pms = kt["k_project_scatter"]["avg_ms"]
    return {
        "workload": f"BASELINE config 5: 128x4096 cloud ({N} points), {n_tracks} tracks/frame (10 % new) through "
                    "mld_tracklets_depth_device, one frame per call sequence, previous frame's slot resident",
        "frames": n_frames, "ms_per_frame": 1e3 * el / n_frames, "associations_per_s": assoc * n_frames / el,
        "kernels_ms_per_launch": {k: v["avg_ms"] for k, v in kt.items()},
        "launches_per_frame": {k: v["launches"] / n_frames for k, v in kt.items()},
        "roofline_project": {"design_bytes_per_launch": db["bytes"], "kernel_ms": pms,
                             "frac": db["bytes"] / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS if pms > 0 else None},
        "verified": ok,
    }


def config5_batched_leg(cam, T, device, S, steps=6, two_contexts=False):
    """BASELINE config 5 at batch size: the current frames of S independent sequences (128x4096 cloud, 10 000 tracks, 10 %
    new) per step through mld_set_clouds_planes_range_device + mld_tracklets_depths_device; every sequence's previous
    frame stays resident in the other bank of slots.  Distinct HBM per slot; checked per sequence against the oracle."""
    import torch
    from mono_lidar_depth_amd import TrackletBatch, capi, synth
    from oracle import oracle
    dev = torch.device("cuda", device)
    P = capi.params_c0()
    U, n_tracks, K = 4, 10000, 8
    clouds_h = [synth.make_cloud(synth.DENSE128, seed=5, frame=f) for f in range(U)]
    planes_h = [synth.make_ground_plane(c) for c in clouds_h]
    N = clouds_h[0].shape[0]
    words = (N + 31) // 32
    all_clouds = torch.empty((2, S, N, 4), dtype=torch.float32, device=dev)  # two banks of S slots
    all_masks = torch.empty((2, S, words), dtype=torch.int32, device=dev)
    d_unique = [torch.from_numpy(c).to(dev) for c in clouds_h]
    m_unique = [torch.from_numpy(mask_words(p[1], N)).to(dev) for p in planes_h]
    for b in range(2):
        for q in range(S):
            all_clouds[b, q].copy_(d_unique[(b + 2 * q) % U])
            all_masks[b, q].copy_(m_unique[(b + 2 * q) % U])
    del d_unique, m_unique
    rng = np.random.default_rng(5)
    sets_h, sets_d = [], []
    for k in range(K):
        u0 = rng.integers(0, cam.width, n_tracks).astype(np.float32)
        v0 = rng.integers(100, cam.height, n_tracks).astype(np.float32)
        u1 = (u0 + rng.integers(-3, 4, n_tracks)).astype(np.float32)
        v1 = (v0 + rng.integers(-2, 3, n_tracks)).astype(np.float32)
        new = np.zeros(n_tracks, dtype=np.uint8)
        new[rng.choice(n_tracks, n_tracks // 10, replace=False)] = 1
        sets_h.append((u0, v0, u1, v1, new))
        sets_d.append(tuple(torch.from_numpy(a).to(dev) for a in (u0, v0, u1, v1, new)))
    d_cur = torch.empty((S, n_tracks), dtype=torch.float32, device=dev)
    d_last = torch.zeros((S, n_tracks), dtype=torch.float32, device=dev)
    t_cur = torch.empty((S, n_tracks), dtype=torch.int32, device=dev)
    t_last = torch.zeros((S, n_tracks), dtype=torch.int32, device=dev)
    # (dense cloud: 16 neighbours in the road window on average, 48 at most: list capacities 48 / 24, include/mld.h)
    tb = TrackletBatch(P, cam, T, S, n_tracks, device=device, list_capacity=(48, 24))
    rows = lambda t: [t[q] for q in range(S)]  # noqa: E731

    def prepared(tbx, dc, dl, tc, tl):
        out = []
        for b in range(2):  # one prepared frame per bank
            pick = lambda j: [sets_d[(b + q) % K][j] for q in range(S)]  # noqa: E731
            out.append(tbx.prepare(rows(all_clouds[b]), np.stack([planes_h[(b + 2 * q) % U][0] for q in range(S)]),
                                   rows(all_masks[b]), pick(0), pick(1), pick(2), pick(3), pick(4), rows(dc), rows(dl),
                                   rows(tc), rows(tl)))
        return out

    prep = prepared(tb, d_cur, d_last, t_cur, t_last)
    torch.cuda.synchronize()
    # (warm-up WITH the kernel timers on and as long as the timed loop: the hipEvents they record exist afterwards)
    tb.est.timingEnable(True)
    warm = steps + (steps & 1) + 1  # (odd: the timed loop starts on the other bank)
    for it in range(warm):
        tb.run(prep[it % 2])
    tb.est.synchronize()
    tb.est.timingReset()
    t0 = time.perf_counter()
    for it in range(warm, warm + steps):
        tb.run(prep[it % 2])
    tb.est.synchronize()
    el = time.perf_counter() - t0
    kt = kernel_times(tb.est)
    tb.est.timingEnable(False)
    last_b = (warm + steps - 1) % 2  # data set of the last frame the context processed
    # Two contexts in turn (a second set of S sequences - here the same resident clouds and tracks, own frame slots and
    # outputs): each step still is the current frames of S sequences, but its projection runs beside the other set's feature
    # kernel - the schedule of the config-2 bench, with the 168-register dense instantiation of the feature kernel (DENSE 1).
    two = None
    if two_contexts:
        d2c, d2l = torch.empty_like(d_cur), torch.zeros_like(d_last)
        t2c, t2l = torch.empty_like(t_cur), torch.zeros_like(t_last)
        tb2 = TrackletBatch(P, cam, T, S, n_tracks, device=device, list_capacity=(48, 24))
        for x in (tb, tb2):
            x.est.setSharedGpu(1)
        prep2 = prepared(tb2, d2c, d2l, t2c, t2l)
        pair = [(tb, prep), (tb2, prep2)]
        torch.cuda.synchronize()
        two = {}
        for ho in ("classify",):
            n2 = 2 * steps
            for x in (tb, tb2):
                x.est.timingEnable(True)
            for it in range(n2):  # (a whole repetition's worth)
                x, pr = pair[it % 2]
                x.run(pr[(it // 2) % 2], pair[(it + 1) % 2][0], ho)
            reps, kts, submit = [], [], []
            for _ in range(5):
                for x in (tb, tb2):
                    x.est.synchronize()
                    x.est.timingReset()
                t0 = time.perf_counter()
                for it in range(n2, 2 * n2):
                    x, pr = pair[it % 2]
                    x.run(pr[(it // 2) % 2], pair[(it + 1) % 2][0], ho)
                submit.append(time.perf_counter() - t0)
                for x in (tb, tb2):
                    x.est.synchronize()
                reps.append(time.perf_counter() - t0)
                kts.append(kernel_times([tb.est, tb2.est]))
            el2 = float(np.median(reps))
            kt2 = kts[int(np.argsort(reps)[len(reps) // 2])]
            for x in (tb, tb2):
                x.est.timingEnable(False)
            two[ho] = {"ms_per_step": 1e3 * el2 / n2, "associations_per_s": (n_tracks + n_tracks // 10) * S * n2 / el2,
                       "ms_per_step_runs": [1e3 * r / n2 for r in reps],
                       "submit_ms_per_step_runs": [1e3 * r / n2 for r in submit],
                       "kernels_ms_per_launch_runs": [{k: round(v["avg_ms"], 4) for k, v in kt_.items()} for kt_ in kts],
                       "kernels_ms_per_launch": {k: v["avg_ms"] for k, v in kt2.items()}}
        last_b = ((2 * n2 - 2) // 2) % 2  # (the first context's last step in this phase)
        tb.est.setSharedGpu(0)
        two["second_context_equals_first"] = bool(torch.equal(t2c, t_cur) and torch.equal(d2c, d_cur))
        tb2.close()
    # the last step's bank against the oracle: EVERY sequence, both slots (the sequences cycle through a few distinct
    # (current cloud, previous cloud, track set) combinations, each of which the oracle computes once)
    b = last_b
    ok = True
    expect = {}
    bad_seq = []
    hc, hl, htc, htl = d_cur.cpu().numpy(), d_last.cpu().numpy(), t_cur.cpu().numpy(), t_last.cpu().numpy()
    for q in range(S):
        i, j, k = (b + 2 * q) % U, ((1 - b) + 2 * q) % U, (b + q) % K
        if (i, j, k) not in expect:
            ref = oracle.OracleDepthEstimator(P, cam.as_struct(), T)
            ref.set_cloud(clouds_h[i])
            ref.set_ground_plane(*planes_h[i])
            ref_l = oracle.OracleDepthEstimator(P, cam.as_struct(), T)
            ref_l.set_cloud(clouds_h[j])
            ref_l.set_ground_plane(*planes_h[j])
            u0, v0, u1, v1, new = sets_h[k]
            expect[(i, j, k)] = oracle.tracklets_depth(ref, ref_l, u0, v0, u1, v1, new.astype(bool), n_threads=8) + (new.astype(bool),)
        e_cur, e_last, et_cur, et_last, nw = expect[(i, j, k)]
        good = bool(np.array_equal(htc[q], et_cur) and np.allclose(hc[q], e_cur, rtol=0, atol=1e-4, equal_nan=True) and
                    np.array_equal(htl[q][nw], et_last[nw]) and np.allclose(hl[q][nw], e_last[nw], rtol=0, atol=1e-4, equal_nan=True))
        if not good:
            bad_seq.append(q)
    ok = not bad_seq
    tb.close()
    assoc = (n_tracks + n_tracks // 10) * S
    db = design_bytes_project(clouds_h[0], cam, T, planes_h[0][1])
    pms = kt["k_project_scatter"]["avg_ms"]
    return {"sequences": S, "ms_per_step": 1e3 * el / steps, "ms_per_frame": 1e3 * el / steps / S,
            "associations_per_s": assoc * steps / el,
            "kernels_ms_per_launch": {k: v["avg_ms"] for k, v in kt.items()},
            "roofline": config_roofline(f"5b{S}", kt, S),
            "roofline_project": {"design_bytes_per_launch": db["bytes"] * S, "kernel_ms": pms,
                                 "frac": db["bytes"] * S / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS if pms > 0 else None},
            "two_contexts": two,
            "sequences_checked": S, "distinct_oracle_cases": len(expect), "mismatching_sequences": bad_seq[:32],
            "verified": ok and (two is None or two["second_context_equals_first"])}



def estimated_leg(P, cam, T, device, B, F, steps=10, contexts=2, shared_mode=1, schedule="halves"):
    """The reference's default call at batch size: the plane of every frame ESTIMATED on the GPU (seeded RANSAC, batched,
    no host round trip: mld_set_clouds_estimate_planes_device) instead of supplied; EVERY frame of the leg's output set is
    re-checked against the oracle with the restatement's own estimate for the frame's seed."""
    from mono_lidar_depth_amd import synth
    from oracle import oracle
    res = Resident(P, cam, T, synth.HDL64, B, min(16, B), F, 0, device, contexts=contexts, shared_mode=shared_mode)
    res.est_schedule = schedule
    alt = schedule == "alternate" and res.whole and len(res.ests) > 1
    if alt:
        res.est_S = res.S
    else:
        for e in res.ests:
            e.setSharedGpu(False)
    n_e = max(2, steps)
    loops_e, kt_e = timed_resident(res, n_e, 2, True, 2, estimated=True)
    el_e = loops_e[0]
    est_set = ((res.k - 1) % len(res.batches)) if alt else 0  # output set of the last step of this leg
    poison_e = res.poison_left(sets=[est_set])  # (halves: every frame goes into the first output set)
    ok_e = poison_e["type_minus77"] == 0
    bad_e = []
    dg_all, tg_all = res.out_depth[est_set].cpu().numpy(), res.out_type[est_set].cpu().numpy()
    ref = oracle.OracleDepthEstimator(P, cam.as_struct(), T)
    t_or = time.perf_counter()
    # (grouped by cloud: the oracle's serial stage A runs once per distinct cloud, the estimate ~1 ms per frame)
    for u in range(res.U):
        ref.set_cloud(res.clouds_h[u])
        for fr in range(u, B, res.U):
            ref.estimate_ground_plane((fr % res.est_S) + 1)
            d0, t0 = ref.calculate_depth(res.uvs_h[fr], 8)
            if not (np.array_equal(tg_all[fr], t0) and np.allclose(dg_all[fr], d0, rtol=0, atol=1e-4, equal_nan=True)):
                bad_e.append(fr)
    ok_e = ok_e and not bad_e
    out = {"plane": "estimated", "value": B * F * n_e / el_e, "ms_per_step": 1e3 * el_e / n_e,
           "kernels_ms_per_launch": {k: v["avg_ms"] for k, v in kt_e.items()},
           "ransac_us_per_frame": 1e3 * kt_e.get("k_rs_batch", {}).get("avg_ms", 0.0) / res.est_S,
           "frame_slots_per_launch": res.est_S, "schedule": schedule, "verified": bool(ok_e),
           "frames_checked": B, "all_frames": True, "mismatching_frames": bad_e[:64],
           "oracle_seconds": time.perf_counter() - t_or, "poison_left": poison_e}
    res.close()
    return out
