#!/usr/bin/env python3
"""Benchmark of the DepthEstimator hot path on MI355X (contract: see the task's bench.py section).

One "step" = one pass of the hot path (setInputCloud with its ground plane + CalculateDepth) over one batch of
`--frames-per-step` synthetic frames of BASELINE.json config 2 (64x2048 cloud, 2000 features/frame, C0 parameters), all
inputs resident in HBM before the timed region.  Metric: feature-depth associations per second = features x frames /
wall time (every submitted feature counts).

Multi-GPU: one process per GPU, each rank owns its own sequence (weak scaling); the only collective is the RCCL
broadcast of the calibration block.  `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run as a
child process, before this process touches the GPU) unless it already runs under a launcher (WORLD_SIZE set).

The JSON line also carries
  verified      sampled slots of the timed batch compared with the CPU oracle after the timed region (exit 1 if not)
  roofline      PHYSICAL: HBM bytes the PMC counters saw for the dominant kernel (committed profile, profiles/traffic.json,
                scaled to this run's launch size) / its hipEvent-measured duration here, vs 8 TB/s; beside it the scalars
                frac_exclusive, whole_step_frac_of_peak, whole_step_compulsory_frac, whole_step_hbm_busy_frac, gather_frac
                and, marked as such, SURVEY.md §8(d)'s formula figures (formula_*: they charge a map clear and a
                camera-frame copy that are never performed and may exceed the peak)
  cpu_baseline  the restated reference CPU path (oracle/, kind "port") timed on this host, rank 0, N=1 only
  latency / streaming / configs["2"|"3"|"5"]   PCIe-inclusive legs (16- and 32-byte cloud records) and the other single-GPU
                BASELINE configs (config 2 at its stated neighbour count, config 3, config 5), each with its own roofline
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # same table: what a float4 copy reaches on this part (79 % of the spec)
K_PROJECT, K_FUSED, K_WAVE, K_RANSAC, K_CLASSIFY = 0, 1, 3, 4, 5  # mld_kernel_time_ms ids (include/mld.h)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps per loop (0.15 s of GPU time at the default)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=1024, help="resident frames processed per step")
    ap.add_argument("--slots", type=int, default=0,
                    help="frame slots of the context (0 = frames-per-step); a step runs frames-per-step/slots launch sets")
    ap.add_argument("--no-estimated", action="store_true", help="skip the plane-estimated leg")
    ap.add_argument("--est-schedule", choices=["halves", "alternate"], default="halves",
                    help="plane-estimated leg with two contexts: every context takes half of a step's frames, side by side "
                         "(halves), or whole steps alternate as in the supplied-plane schedule (alternate)")
    ap.add_argument("--no-exclusive", action="store_true",
                    help="skip the pass that times each kernel alone (profiling: keeps the kernel statistics of a trace "
                         "to the launches of the timed schedule)")
    ap.add_argument("--contexts", type=int, default=2,
                    help="contexts (HIP streams) the frames of a step are dealt to in turn; with 2 the projection of one "
                         "context runs beside the feature kernels of the other (mld_order_after / mld_set_shared_gpu); "
                         "1 = everything on one stream, one kernel at a time")
    ap.add_argument("--shared-mode", type=int, default=1,
                    help="mld_set_shared_gpu argument of the alternating contexts: 1 = on; + 256 * n = n feature-kernel "
                         "wavefronts per CU instead of 8")
    ap.add_argument("--handover", choices=("classify", "projection"), default="classify",
                    help="two contexts: the next context's projection is released behind this context's classification "
                         "kernel (mld_order_after_classify; default) or at the end of its projection (mld_order_after)")
    ap.add_argument("--pair", action="store_true",
                    help="two contexts: run both contexts' projections on one shared stream (mld_pair_contexts) instead of "
                         "handing the projection over with mld_order_after (one event across streams).  Measured: the "
                         "back-to-back projections starve k_classify of wave slots and the step gets longer (0.79 / 0.90 "
                         "against 0.77 ms); kept as an option of the library, not the bench default")
    ap.add_argument("--unique-frames", type=int, default=16, help="distinct synthetic clouds generated per rank")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--latency-frames", type=int, default=1000,
                    help="frames of the one-frame-per-call host-pointer leg (PCIe-inclusive latency; 0 = skip)")
    ap.add_argument("--streaming-batches", type=int, default=24,
                    help="batches of the pipelined host->device leg (PCIe-inclusive throughput; 0 = skip)")
    ap.add_argument("--streaming-frames", type=int, default=64, help="frames per batch of the streaming leg")
    ap.add_argument("--config-frames", type=int, default=256,
                    help="resident frames of the config-3 leg / sequence length of the config-5 leg (0 = skip both)")
    ap.add_argument("--only-config", type=int, default=0, choices=[0, 2, 3, 5],
                    help="run only that BASELINE config's leg and print its object (for per-config rocprofv3 runs)")
    ap.add_argument("--leg", default="",
                    help="with --only-config: run just that leg's counter-profiled part ('near' for config 3: the c0_dispose "
                         "near-returns mode; a batch size such as '256' for config 5: that batched leg alone) - what the "
                         "rocprofv3 --pmc passes of profiles/run_profile.sh trace")
    ap.add_argument("--verify-slots", type=int, default=-1,
                    help="frames of the timed batch checked against the oracle: -1 (default) = EVERY frame of every "
                         "context's output set (a few seconds: the oracle sets each distinct cloud once); n > 0 = n frames "
                         "spread over the output sets; 0 = no check")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed loop of --steps steps runs at least this many times back to back; value / "
                         "ms_per_step are the median loop, ms_per_step_min / _max the spread")
    ap.add_argument("--min-timed-seconds", type=float, default=0.5,
                    help="the timed loop is repeated beyond --repeats until this much time has been timed in all (at "
                         "most 64 loops): a 20-step loop is 15 ms, less than the GPU clocks take to settle")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-every", type=int, default=4,
                    help="record hipEvents around the kernels of every n-th timed step (event records cost ~6 us each)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n: int) -> int:
    """`--gpus N` without a launcher: start N ranks (one per GPU) as a child torch.distributed.run and relay their
    output.  Nothing in this process has initialised the GPU (torch is imported in the workers only)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------ legs
def cpu_baseline(P, cam_struct, T, clouds, planes, uvs, seconds):
    """The restated reference CPU path on this host's cores (OpenMP over features, DepthEstimator.cpp:455)."""
    from oracle import oracle
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ref = oracle.OracleDepthEstimator(P, cam_struct, T)
    ref.set_cloud(clouds[0])
    ref.set_ground_plane(*planes[0])
    # The reference runs its feature loop with OpenMP's default team (all cores).  2000 features are little work
    # per thread, so the fastest team size is probed first (~0.5 s each) and used for the measurement.
    probe = {}
    for nt in sorted({1, 2, 4, 8, 16, 32, 64, avail}):
        if nt > avail:
            continue
        ref.calculate_depth(uvs[0], nt)  # team start-up
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 0.4:
            ref.calculate_depth(uvs[reps % len(uvs)], nt)
            reps += 1
        probe[nt] = (time.perf_counter() - t0) / reps
    cores = min(probe, key=probe.get)
    seconds = max(1.0, seconds - 0.5 * len(probe))
    frames, t_a, t_b = 0, 0.0, 0.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        i = frames % len(clouds)
        ta = time.perf_counter()
        ref.set_cloud(clouds[i])
        ref.set_ground_plane(*planes[i])
        tb = time.perf_counter()
        ref.calculate_depth(uvs[i], cores)
        tc = time.perf_counter()
        t_a += tb - ta
        t_b += tc - tb
        frames += 1
    el = time.perf_counter() - t0
    F = uvs[0].shape[0]
    return {
        "value": F * frames / el,
        "unit": "feature-depth associations/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"{frames} frames of the same workload in {el:.1f} s; stage A (setInputCloud, serial) "
                   f"{1e3 * t_a / frames:.2f} ms/frame, stage B (feature loop, {cores} OpenMP threads = fastest of "
                   f"{sorted(probe)} probed on {avail} available cores) {1e3 * t_b / frames:.2f} ms/frame"),
        "ms_per_frame": 1e3 * el / frames,
        # the same path with a single-thread feature loop (stage A is serial in the reference anyway)
        "one_thread": {"value": F / (t_a / frames + probe[1]), "ms_per_frame": 1e3 * (t_a / frames + probe[1])},
        "stage_a_ms": 1e3 * t_a / frames,
        "stage_b_ms": 1e3 * t_b / frames,
    }


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


def mask_words(inl, n):
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return m.view(np.int32)


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


class Resident:
    """B device-resident frames (U distinct clouds, B distinct feature sets) and a context with S frame slots."""

    def __init__(self, P, cam, T, scanner, B, U, F, seq, device, integer_uv=False, slots=0, contexts=1, shared_mode=1,
                 pair=False, near_points=False):
        import torch
        from mono_lidar_depth_amd import DepthEstimator, synth
        dev = torch.device("cuda", device)
        self.P, self.cam, self.T, self.B, self.F = P, cam, T, B, F
        self.shared_mode = shared_mode
        self.handover = "classify"
        self.clouds_h = [synth.make_cloud(scanner, seed=seq, frame=f) for f in range(U)]
        self.planes_h = [synth.make_ground_plane(c) for c in self.clouds_h]
        if near_points == "k":  # features whose search window holds >= 6 returns (config 2 at its stated neighbour count)
            per = (B + U - 1) // U  # one draw per distinct cloud, cut into its frames' feature sets
            pools = [synth.make_features_k_neighbours(self.clouds_h[u], F * per, seed=seq * 100000 + u, min_neighbours=6,
                                                      window=(P.pixelarea_search_witdh, P.pixelarea_search_height))
                     for u in range(U)]
            self.uvs_h = [np.ascontiguousarray(pools[b % U][(b // U) * F:(b // U + 1) * F]) for b in range(B)]
        elif near_points:  # features around the image positions of the frame's own returns (config 3, second variant)
            self.uvs_h = [synth.make_features_near_points(self.clouds_h[b % U], F, seed=seq * 100000 + b) for b in range(B)]
        else:
            self.uvs_h = [synth.make_features(F, seed=seq * 100000 + b, integer=integer_uv) for b in range(B)]
        self.N = N = self.clouds_h[0].shape[0]
        self.U = U
        words = (N + 31) // 32
        # distinct HBM per slot, carved out of one allocation per kind (large, contiguous mappings)
        self.all_clouds = torch.empty((B, N, 4), dtype=torch.float32, device=dev)
        self.all_masks = torch.empty((B, words), dtype=torch.int32, device=dev)
        self.all_uvs = torch.empty((B, F, 2), dtype=torch.float64, device=dev)
        # results: in the default schedule both contexts process the same resident frames in turn, so each context gets
        # its own output set (nothing is written twice concurrently, and the check below sees both); with --slots the
        # launch sets are disjoint rows of one set
        n_out = max(1, contexts) if slots <= 0 else 1
        self.out_depth = [torch.empty((B, F), dtype=torch.float64, device=dev) for _ in range(n_out)]
        self.out_type = [torch.empty((B, F), dtype=torch.int32, device=dev) for _ in range(n_out)]
        self.all_depth, self.all_type = self.out_depth[0], self.out_type[0]
        d_unique = [torch.from_numpy(c).to(dev) for c in self.clouds_h]
        m_unique = [torch.from_numpy(mask_words(p[1], N)).to(dev) for p in self.planes_h]
        for b in range(B):
            self.all_clouds[b].copy_(d_unique[b % U])
            self.all_masks[b].copy_(m_unique[b % U])
            self.all_uvs[b].copy_(torch.from_numpy(self.uvs_h[b]))
        del d_unique, m_unique
        self.coeffs = np.stack([self.planes_h[b % U][0] for b in range(B)])
        torch.cuda.synchronize()
        NC = max(1, contexts)
        # Default: every context holds the whole step (S = B frame slots) and consecutive steps alternate between the
        # contexts, so the projection of step k+1 runs beside the feature kernels of step k.  With --slots S a step is
        # cut into launch sets of S frames that are dealt to the contexts in turn.
        self.whole = slots <= 0
        self.S = S = B if self.whole else slots
        assert B % S == 0 and (self.whole or (B // S) % NC == 0), "--frames-per-step must be a multiple of --slots x --contexts"
        self.k = 0
        self.est_batches = None
        self.est_S = S if (not self.whole or NC == 1) else B // NC  # frame slots per launch of the plane-estimated leg
        self.ests = []
        for _ in range(NC):
            e = DepthEstimator(device=device, max_frames=S, max_features=F)  # queues allocated up front
            e.InitConfig(P)
            e.Initialize(cam, T)
            if NC > 1:
                e.setSharedGpu(shared_mode)
            self.ests.append(e)
        if NC == 2 and pair:
            self.ests[0].pairWith(self.ests[1])  # projections back to back on one stream
        # a step walks the B resident frames in launch sets of S frame slots, dealt round-robin to the contexts (one HIP
        # stream each); the slots' pixel maps are reused from one launch set to the next
        rows = lambda t, i: [t[b] for b in range(i, i + S)]  # noqa: E731
        prep = lambda e, i, o: (e, e.prepareBatch(  # noqa: E731
            rows(self.all_clouds, i), rows(self.all_uvs, i), rows(self.out_depth[o], i), rows(self.out_type[o], i),
            self.coeffs[i:i + S], rows(self.all_masks, i), stride_bytes=16))
        if self.whole:
            # the same resident frames, one descriptor set and one output set per context
            self.batches = [prep(e, 0, c) for c, e in enumerate(self.ests)]
        else:
            self.batches = [prep(self.ests[(i // S) % NC], i, 0) for i in range(0, B, S)]

    def poison(self):
        """Results that are not rewritten by the timed region cannot pass the check."""
        for d, t in zip(self.out_depth, self.out_type):
            d.fill_(float("nan"))
            t.fill_(-77)

    def run_step(self):
        # contexts in turn; the next context's projection is released by the end of this one's, so it streams its
        # clouds beside this context's feature kernels
        nb = len(self.batches)
        if self.whole:
            e, b = self.batches[self.k % nb]
            e.runBatchBeside(b, self.batches[(self.k + 1) % nb][0], self.handover)
            self.k += 1
            return
        for i, (e, b) in enumerate(self.batches):
            e.runBatchBeside(b, self.batches[(i + 1) % nb][0], self.handover)

    def last_context(self):
        """The context whose slots hold the most recent launch set."""
        if self.whole:
            return self.batches[(self.k - 1) % len(self.batches)][0]
        return self.batches[-1][0]

    def run_exclusive(self, n):
        """n passes of context 0's first launch set with nothing else on the GPU (kernel durations when each kernel has
        the chip to itself; the timed region of the bench overlaps two contexts)."""
        e, b = self.batches[0]
        for _ in range(n):
            e.runBatch(b)
            e.synchronize()

    def run_step_estimated(self):
        """The same pass with the ground plane of every frame ESTIMATED on the GPU (the reference's default call:
        the GroundPlane handed to setInputCloud is not segmented yet) instead of supplied."""
        import ctypes as C
        todo = self.batches
        if self.whole and len(self.ests) > 1 and getattr(self, "est_schedule", "halves") == "alternate":
            # whole steps in turn, as run_step: the next context's estimation + projection released by the end of this one's
            nb = len(self.batches)
            e, b = self.batches[self.k % nb]
            nxt = self.batches[(self.k + 1) % nb][0]
            n = b["n"]
            if "seeds" not in b:
                b["seeds"] = (C.c_uint32 * n)(*range(1, n + 1))
            e._check(e._lib.mld_set_clouds_estimate_planes_device(e._ctx, n, b["cloud_ptrs"], b["cloud_n"], b["stride"],
                                                                  b["seeds"]))
            nxt.orderAfter(e)
            e._check(e._lib.mld_calculate_depths_device(e._ctx, n, b["uv_ptrs"], b["F"], b["depth_ptrs"], b["type_ptrs"]))
            self.k += 1
            return
        if self.whole and len(self.ests) > 1:
            # (k_rs_batch - a 1024-thread block and 150 KB of LDS per frame - fits neither beside the feature kernels nor
            # beside a projection, so alternating whole steps gains nothing here: every context takes its share of the
            # step's frames, side by side)
            if self.est_batches is None:
                NC, Sh = len(self.ests), self.B // len(self.ests)
                rows = lambda t, i: [t[b] for b in range(i, i + Sh)]  # noqa: E731
                self.est_batches = [(e, e.prepareBatch(rows(self.all_clouds, k * Sh), rows(self.all_uvs, k * Sh),
                                                       rows(self.all_depth, k * Sh), rows(self.all_type, k * Sh),
                                                       self.coeffs[k * Sh:(k + 1) * Sh], rows(self.all_masks, k * Sh),
                                                       stride_bytes=16)) for k, e in enumerate(self.ests)]
            todo = self.est_batches
        for e, b in todo:
            n = b["n"]
            if "seeds" not in b:
                b["seeds"] = (C.c_uint32 * n)(*range(1, n + 1))
            e._check(e._lib.mld_set_clouds_estimate_planes_device(e._ctx, n, b["cloud_ptrs"], b["cloud_n"], b["stride"],
                                                                  b["seeds"]))
            e._check(e._lib.mld_calculate_depths_device(e._ctx, n, b["uv_ptrs"], b["F"], b["depth_ptrs"], b["type_ptrs"]))

    def sync(self):
        for e in self.ests:
            e.synchronize()

    def close(self):
        for e in reversed(self.ests):  # a pair's borrower before the owner of the projection stream
            e.close()

    def poison_left(self, sets=None):
        """Device-side scan of the output sets (all, or the listed ones) for entries the timed region did not rewrite
        (poison(): type -77, depth NaN).  A result type is written with every depth, so a surviving -77 is a feature
        nobody processed."""
        import torch
        left = {"type_minus77": 0, "nan_depth": 0}
        for o, (d, t) in enumerate(zip(self.out_depth, self.out_type)):
            if sets is not None and o not in sets:
                continue
            left["type_minus77"] += int((t == -77).sum().item())
            left["nan_depth"] += int(torch.isnan(d).sum().item())
        return left

    def verify(self, n_slots=-1):
        """The timed batch against the CPU oracle (checker only, outside every timed region): result types identical,
        depths bit-exact on the main path and within 1e-4 m on the road path.  n_slots < 0: every frame of every output
        set (each context's set holds the last step that context ran); n_slots > 0: that many frames spread over the sets.
        The frames are grouped by their cloud, so the oracle's serial stage A runs once per distinct cloud.
        Returns (ok, report)."""
        from oracle import oracle
        ref = oracle.OracleDepthEstimator(self.P, self.cam.as_struct(), self.T)
        n_out = len(self.out_depth)
        worst, bad, picks_all = 0.0, [], []
        for o in range(n_out):
            if n_slots < 0:
                picks = list(range(self.B))
            else:
                per = max(1, (n_slots + n_out - 1) // n_out)
                lo = (self.B * o) // (2 * n_out) if n_out > 1 else 0  # different frames per output set
                picks = sorted({int(x) for x in np.linspace(lo, self.B - 1, per)})
            picks_all.append(picks)
        host = [(d.cpu().numpy(), t.cpu().numpy()) for d, t in zip(self.out_depth, self.out_type)]
        t_begin = time.perf_counter()
        n_checked = 0
        for u in range(self.U):
            todo = [(o, fr) for o in range(n_out) for fr in picks_all[o] if fr % self.U == u]
            if not todo:
                continue
            ref.set_cloud(self.clouds_h[u])
            ref.set_ground_plane(*self.planes_h[u])
            cache = {}
            for o, fr in todo:
                if fr not in cache:
                    cache[fr] = ref.calculate_depth(self.uvs_h[fr], 8)
                d0, t0 = cache[fr]
                d, t = host[o][0][fr], host[o][1][fr]
                same_t = np.array_equal(t, t0)
                diff = np.abs(np.nan_to_num(d, nan=-7.0) - np.nan_to_num(d0, nan=-7.0))
                main = t0 != 16
                ok = same_t and diff.max(initial=0.0) <= 1e-4 and np.array_equal(d[main], d0[main], equal_nan=True)
                worst = max(worst, float(diff.max(initial=0.0)))
                n_checked += 1
                if not ok:
                    bad.append([o, fr])
        left = self.poison_left()
        ok_all = (not bad) and left["type_minus77"] == 0
        rep = {"frames_checked": n_checked, "frames_per_output_set": [len(p) for p in picks_all], "output_sets": n_out,
               "all_frames": bool(n_slots < 0), "max_abs_depth_diff_m": worst, "mismatching_frames": bad[:64],
               "mismatching_count": len(bad), "poison_left": left, "oracle_seconds": time.perf_counter() - t_begin}
        if n_slots >= 0:
            rep["frames"] = [[o, fr] for o in range(n_out) for fr in picks_all[o]]
        return ok_all, rep


def design_bytes_project(cloud, cam, T, inl):
    """Bytes k_project_scatter has to move for one frame in THIS design: the cloud once (16 B/point), one 4-byte map
    entry per point that lands in the image in front of the camera, the occupancy words those points set, and the
    inlier-mask words read for them.  (No map clear, no camera-frame copy: DESIGN.md §2.)"""
    xyz = cloud[:, :3].astype(np.float64)
    p = xyz @ T[:, :3].T + T[:, 3]
    with np.errstate(invalid="ignore", divide="ignore"):
        u = (cam.focal_length * p[:, 0] + cam.principal_point_x * p[:, 2]) / p[:, 2]
        v = (cam.focal_length * p[:, 1] + cam.principal_point_y * p[:, 2]) / p[:, 2]
        vis = (p[:, 2] > 0) & (u > 0) & (u < cam.width) & (v > 0) & (v < cam.height)
    idx = np.nonzero(vis)[0]
    words = np.unique((u[idx].astype(np.int64) >> 5) * 100000 + v[idx].astype(np.int64)).size
    mwords = np.unique(idx >> 5).size
    return {"bytes": 16 * cloud.shape[0] + 4 * idx.size + 4 * words + 4 * mwords, "n_front_in_image": int(idx.size),
            "bitmap_words": int(words)}


def pmc_traffic(kernel, frames_per_launch):
    """HBM bytes per launch of `kernel` and the launch time they were measured with (the kernel alone on the GPU), from
    the committed rocprofv3 PMC passes (profiles/traffic.json, written by profiles/summarize.py: FETCH_SIZE/WRITE_SIZE
    in separate --pmc runs, gfx950 correction applied), scaled from the profile's frames per launch to this run's."""
    try:
        t = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        scale = float(frames_per_launch) / float(t["frames_per_launch"])  # traffic is proportional to the frames
        return float(t[kernel]["hbm_bytes_per_launch"]) * scale, float(t[kernel]["launch_s"]) * scale, t.get("source", "")
    except Exception:  # noqa: BLE001
        return None


def config_roofline(key, kt, frames_per_launch):
    """Physical roofline of one BASELINE-config leg: its dominant kernel (longest average launch, hipEvents of THIS run)
    priced on the HBM bytes the PMC counters saw for that kernel in the committed profile of the same leg
    (profiles/traffic.json["configs"][key]: FETCH_SIZE, gfx950-corrected for the projection's wide loads, + WRITE_SIZE,
    separate --pmc passes), scaled to this run's frames per launch.  None where no counter profile is committed."""
    tj = traffic_profile_json() or {}
    prof = (tj.get("configs") or {}).get(key)
    if prof is None and key.startswith("5b"):  # config 5 at another batch size: the S = 256 profile, scaled by the slots
        key = "5b256"
        prof = (tj.get("configs") or {}).get(key)
    times = {k: v["avg_ms"] for k, v in kt.items() if v.get("avg_ms", 0.0) > 0}
    if not prof or not times:
        return None
    scale = float(frames_per_launch) / float(prof["frames_per_launch"])
    per = {}
    for k, t_ms in times.items():
        if k in prof:
            nb = float(prof[k]["hbm_bytes_per_launch"]) * scale
            per[k] = {"kernel_ms": t_ms, "traffic": nb, "achieved": nb / (t_ms * 1e-3) / 1e9,
                      "frac": nb / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "profile_launch_ms": float(prof[k]["launch_s"]) * 1e3 * scale}
    dominant = max(times, key=times.get)
    if dominant not in per:
        return None
    total = sum(v["traffic"] for v in per.values())
    t_all = sum(times.values())
    return {"bound": "hbm", "kernel": dominant, "kernel_ms": times[dominant], "achieved": per[dominant]["achieved"],
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per[dominant]["frac"], "traffic": per[dominant]["traffic"],
            "all_kernels_frac": total / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS,  # counter bytes of the leg's kernels / their summed launch times
            "kernels": per, "bytes_source": f"profiles/traffic.json configs.{key} ({prof.get('source', '')})"}


def traffic_profile_json():
    try:
        return json.loads((ROOT / "profiles" / "traffic.json").read_text())
    except Exception:  # noqa: BLE001
        return None


def kernel_times(ests):
    """Average launch duration per kernel over the timed launches of all the given contexts."""
    ests = ests if isinstance(ests, (list, tuple)) else [ests]
    out = {}
    for name, k in (("k_project_scatter", K_PROJECT), ("k_classify", K_CLASSIFY), ("k_feature_fused", K_FUSED),
                    ("k_feature_wave", K_WAVE), ("k_rs_batch", K_RANSAC)):
        tot, n = 0.0, 0
        for e in ests:
            ms, m = e.kernelTimeMs(k)
            tot += ms * m
            n += m
        if n or k != K_RANSAC:
            out[name] = {"avg_ms": tot / n if n else 0.0, "launches": n}
    return out


def timed_resident(res, steps, warmup, timing, timing_every, barrier=lambda: None, estimated=False, repeats=1,
                   reduce_max=lambda x: x, min_timed_s=0.0):
    """At least `repeats` timed loops of `steps` steps each (more until `min_timed_s` seconds have been timed, at most
    64), every loop bracketed by barrier + synchronize on both sides.  Returns (per-loop elapsed seconds, each the max
    over ranks; kernel times averaged over the sampled steps of all loops)."""
    import torch
    step = res.run_step_estimated if estimated else res.run_step
    for _ in range(warmup):
        step()
    res.sync()
    res.poison()
    if timing:
        for e in res.ests:
            e.timingEnable(True)
            e.timingReset()
            e.timingEnable(False)
    loops, local = [], []
    res.local_loops = local
    # (the loop count follows the max-over-ranks times, which every rank holds: all ranks run the same number)
    while len(loops) < max(1, repeats) or (sum(loops) < min_timed_s and len(loops) < 64):
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(steps):
            if timing:
                for e in res.ests:
                    e.timingEnable(it % max(1, timing_every) == 0)  # sampled steps of the timed region
            step()
        res.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        local.append(el)
        loops.append(reduce_max(el))
        barrier()
    kt = kernel_times(res.ests) if timing else {}
    for e in res.ests:
        e.timingEnable(False)
    return loops, kt


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


# ------------------------------------------------------------------------------------------------ worker
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # No cyclic garbage collection while legs are timed (a full collection of this process's heap is milliseconds on the
    # submitting thread, a step of the short legs 0.2-0.4 ms); reference counting frees everything the legs allocate.
    # (Not what made the S = 64 two-context leg of rounds 4 and 5 bimodal - that was a DMA queue being set up inside a
    # hipMemcpyAsync of the step's descriptors, LAB.md 5.16; `submit_ms_per_step_runs` is the host-side check for either.)
    gc.disable()
    # Host placement first - before torch is imported, before the first GPU call and before any pinned buffer exists: the
    # rank's threads and its staging memory belong on the NUMA node of ITS GPU (sysfs only; nothing is re-executed).
    affinity = None
    if world > 1 and os.environ.get("MLD_BENCH_AFFINITY", "1") != "0":
        from mono_lidar_depth_amd.sharding import bind_to_gpu_numa_node
        affinity = bind_to_gpu_numa_node(local_rank)

    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one process per GPU; RCCL ("nccl" on ROCm) carries the calibration broadcast and the scalar reductions.
    # MLD_BENCH_BACKEND=gloo is a functional-test hook: it lets several ranks share one GPU box (collectives on CPU
    # tensors) so that the N>1 code path can be exercised where only one GPU is visible.
    backend = os.environ.get("MLD_BENCH_BACKEND", "nccl")
    gpu_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    coll_dev = dev if backend == "nccl" else None
    if affinity is not None:
        # the sysfs guess against what the runtime says this rank's device is (HIP order need not be PCI order)
        from mono_lidar_depth_amd.sharding import rebind_if_device_differs
        pr = torch.cuda.get_device_properties(gpu_index)
        pci = None
        if all(hasattr(pr, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        affinity = rebind_if_device_differs(affinity, pci)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from mono_lidar_depth_amd import CameraPinhole, capi, sharding, synth, traffic

    # ---- calibration: rank 0 owns it, everyone receives it over RCCL -------------------------------------
    if rank == 0:
        P = capi.params_c0()
        cam_struct = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV).as_struct()
        T = synth.T_CAM_LIDAR
    else:
        P = cam_struct = T = None
    P, cam_struct, T = sharding.broadcast_calibration(P, cam_struct, T, device=coll_dev)
    cam = CameraPinhole(cam_struct.width, cam_struct.height, cam_struct.focal_length, cam_struct.principal_point_x,
                        cam_struct.principal_point_y)

    if args.only_config:
        if args.only_config == 2:
            leg = config2_k_leg(P, cam, T, gpu_index, args.frames_per_step, args.features, contexts=args.contexts,
                                shared_mode=args.shared_mode)
        elif args.only_config == 3:
            leg = config3_leg(cam, T, gpu_index, args.config_frames, only_near=args.leg == "near")
        elif args.leg:
            S5 = int(args.leg.rstrip("t"))  # ("256t": with the two-context schedule as well)
            leg = {"workload": "BASELINE config 5, batched leg only",
                   "batched": {str(S5): config5_batched_leg(cam, T, gpu_index, S5, two_contexts=args.leg.endswith("t"))}}
            leg["verified"] = leg["batched"][str(S5)]["verified"]
        else:
            leg = config5_leg(cam, T, gpu_index, min(args.config_frames, 200))
            leg["batched"] = {str(S5): config5_batched_leg(cam, T, gpu_index, S5, two_contexts=True) for S5 in (16, 64, 256)}
            leg["verified"] = bool(leg["verified"] and all(v["verified"] for v in leg["batched"].values()))
        print(json.dumps({"config": str(args.only_config), **leg}), flush=True)
        sys.exit(0 if leg.get("verified") else 1)

    # ---- this rank's sequence, resident in HBM ------------------------------------------------------------
    B, F = args.frames_per_step, args.features
    U = max(1, min(args.unique_frames, B))
    seq = sharding.assign_sequences(world, world)[rank][0]  # one sequence per rank (config 4 layout)
    res = Resident(P, cam, T, synth.HDL64, B, U, F, seq, gpu_index, slots=args.slots, contexts=args.contexts,
                   shared_mode=args.shared_mode, pair=args.pair)
    res.handover = args.handover
    S, N = res.S, res.N

    def barrier():
        if world > 1:
            dist.barrier()

    timing = not args.no_kernel_timing
    loops, kt = timed_resident(res, args.steps, args.warmup, timing, args.timing_every, barrier, repeats=args.repeats,
                               reduce_max=lambda x: sharding.max_over_ranks(x, device=coll_dev),
                               min_timed_s=args.min_timed_seconds)
    elapsed = float(np.median(loops))  # every loop: exactly --steps steps, max over ranks
    loop_start = "idle GPU (synchronised): exactly --steps steps per loop, first projection and last feature kernels unpartnered"
    elapsed_local = float(np.median(res.local_loops)) if getattr(res, "local_loops", None) else elapsed
    units = sharding.sum_over_ranks(float(B * F * args.steps), device=coll_dev)

    # ---- the timed batch against the oracle (every rank checks its own sequence) ---------------------------
    verified, vrep = res.verify(args.verify_slots) if args.verify_slots != 0 else (None, {})
    n_bad = sharding.sum_over_ranks(0.0 if verified in (True, None) else 1.0, device=coll_dev)
    per_rank_value = sharding.gather_over_ranks(B * F * args.steps / elapsed_local, device=coll_dev)
    numa_nodes = sharding.gather_over_ranks(float(affinity["numa_node"]) if affinity and affinity.get("numa_node") is not None
                                            else -1.0, device=coll_dev)
    pinned_ranks = sharding.sum_over_ranks(1.0 if affinity and affinity.get("applied") else 0.0, device=coll_dev)
    distributed = None
    streaming_ranks = None
    if world > 1:
        distributed = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                       "ranks_verified": int(world - n_bad) if args.verify_slots != 0 else 0,
                       "resident_associations_per_s_per_rank": {"min": min(per_rank_value), "max": max(per_rank_value)},
                       # host placement of the ranks (sharding.bind_to_gpu_numa_node, before the first GPU call)
                       "affinity": {"rank0": affinity, "numa_node_per_rank": [int(x) for x in numa_nodes],
                                    "ranks_repinned": int(pinned_ranks)}}
        if args.streaming_batches > 0:
            # BASELINE config 4 as written: every rank STREAMS its own sequence from pinned host memory (PCIe-inclusive)
            barrier()
            st = streaming_leg(P, cam, T, res.clouds_h, res.planes_h, res.uvs_h, gpu_index, args.streaming_frames,
                               args.streaming_batches)
            fps = sharding.gather_over_ranks(st["frames_per_s"], device=coll_dev)
            slowest = sharding.max_over_ranks(st["frames"] / st["frames_per_s"], device=coll_dev)
            streaming_ranks = {**st, "ranks": world, "frames_per_s": st["frames"] * world / slowest,
                               "associations_per_s": st["frames"] * world * F / slowest,
                               "frames_per_s_per_rank": {"min": min(fps), "max": max(fps)},
                               "aggregate": "frames of all ranks / the slowest rank's elapsed time"}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        sys.exit(0 if verified in (True, None) else 1)
    if args.verify_slots != 0:
        verified = n_bad == 0

    est = res.ests[0]
    # ---- the same kernels with the GPU to themselves (outside the timed region): with two contexts the timed region
    # runs a projection beside the other context's feature kernels, so a kernel's launch duration there is not its
    # speed; these are the durations one launch set takes alone
    kt_x = {}
    if timing and len(res.ests) > 1 and not args.no_exclusive:
        res.run_exclusive(1)
        est.timingEnable(True)
        est.timingReset()
        res.run_exclusive(4)
        kt_x = kernel_times(est)
        est.timingEnable(False)
    # ---- byte counts (sampled slots, outside the timed region) --------------------------------------------
    type_hist = np.zeros(capi.MLD_RESULT_TYPE_COUNT, dtype=np.int64)
    for t_set in res.out_type:  # every context's output set (a poisoned -77 entry would fall outside the histogram)
        type_hist += est.resultHistogram(t_set.reshape(-1))
    type_hist_sets = len(res.out_type)
    stat_slots = list(range(0, S, max(1, S // 4)))[:4]
    stats, design = [], []
    last_est = res.last_context()
    for b in stat_slots:
        fr = B - S + b  # the slots of the last launch set's context hold the last sub-batch of the step
        stats.append(traffic.frame_bytes(P, cam.width, cam.height, N, last_est.getVisibleCount(b), last_est.getPixelMap(b),
                                         res.uvs_h[fr], res.all_type[fr].cpu().numpy()))
        design.append(design_bytes_project(res.clouds_h[fr % U], cam, T, res.planes_h[fr % U][1]))
    formula_project = float(np.mean([s["project_bytes"] for s in stats])) * S  # per launch: S frames
    formula_feature = float(np.mean([s["feature_bytes"] for s in stats])) * S
    design_project = float(np.mean([d["bytes"] for d in design])) * S

    def ms(k):
        return kt.get(k, {}).get("avg_ms", 0.0)

    def gbps(nbytes, t_ms):
        return (nbytes / (t_ms * 1e-3)) / 1e9 if t_ms > 0 else 0.0

    # Both long kernels on their roofs; the line's top-level roofline is the one that takes longer in the timed schedule.
    #   k_project_scatter  HBM: the bytes THIS design moves (cloud once + map / bitmap / mask words) over its launch time
    #   k_feature_fused    HBM by SURVEY 8(d)'s per-feature formula (algorithmic bytes), and - what actually bounds it - a
    #                      gather roof: the cache-line requests its divergent loads make (PMC profile) against the line
    #                      rates random gathers reach on this part (profiles/tools/randgather.hip, same profile session)
    tj = traffic_profile_json()
    gather = None
    fj = (tj or {}).get("k_feature_fused", {})
    ceil = (tj or {}).get("gather_ceilings")
    if ceil and fj.get("tcp_tcc_read_req") and ms("k_feature_fused") > 0:
        scale = float(S) / float(tj["frames_per_launch"])
        l2_req = fj["tcp_tcc_read_req"] * scale           # L1 misses: lines requested from L2
        hbm_req = fj.get("tcc_ea_rdreq", 0.0) * scale      # of those, lines L2 had to fetch from memory (64 B each)
        t_s = ms("k_feature_fused") * 1e-3
        # time the requests take at the measured random-gather rates (L2-resident lines / memory-resident lines)
        t_floor = max(l2_req - hbm_req, 0.0) / (ceil["l2_Glines_s"] * 1e9) + hbm_req / (ceil["hbm_Glines_s"] * 1e9)
        gather = {"requests_per_launch": l2_req, "memory_fetches_per_launch": hbm_req,
                  "l1_accesses_per_launch": fj.get("tcp_total_cache_accesses", 0.0) * scale,
                  "achieved_Glines_s": l2_req / t_s / 1e9,
                  "ceiling_Glines_s": l2_req / t_floor / 1e9 if t_floor > 0 else None,
                  "frac": t_floor / t_s,
                  "ceilings": ceil,
                  "model": "lines requested from L2 (TCP_TCC_READ_REQ) priced at the random-gather rate of L2-resident lines, "
                           "the share fetched from memory (TCC_EA0_RDREQ) at the rate of memory-resident lines; frac = "
                           "that service time / the launch time measured in this run",
                  "source": (tj or {}).get("source", "")}

    def pmc_bytes(name):
        t = pmc_traffic(name, S)
        return t[0] if t else None

    def x_ms(name):
        return kt_x.get(name, {}).get("avg_ms", 0.0)

    def frac_of(nbytes, t_ms):
        return gbps(nbytes, t_ms) / HBM_PEAK_GBS if (nbytes and t_ms > 0) else None

    # Per kernel: the HBM bytes the PMC counters saw for a launch (committed profile: FETCH_SIZE with the gfx950 correction
    # + WRITE_SIZE, separate --pmc passes; scaled to this run's frames per launch) over the launch duration measured in
    # THIS run with hipEvents - the physical figure.  SURVEY 8(d)'s per-unit formula and the bytes this design has to move
    # by construction ride along as formula_* / design_* fields.
    def kernel_entry(name, design_bytes, formula_bytes):
        cb = pmc_bytes(name)
        e = {**kt.get(name, {}), "bound": "hbm", "kernel": name, "kernel_ms": ms(name), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "traffic": cb, "achieved": gbps(cb, ms(name)) if cb else None, "frac": frac_of(cb, ms(name)),
             "frac_exclusive": frac_of(cb, x_ms(name)), "exclusive_kernel_ms": x_ms(name) or None,
             "bytes_source": "PMC counters of the committed profile (profiles/traffic.json), scaled to this run's frames per launch"}
        if design_bytes is not None:
            e.update({"design_bytes_per_launch": design_bytes, "design_GBps": gbps(design_bytes, ms(name)),
                      "design_frac": frac_of(design_bytes, ms(name)), "design_frac_exclusive": frac_of(design_bytes, x_ms(name))})
        if formula_bytes is not None:
            e.update({"formula_bytes_per_launch": formula_bytes, "formula_GBps": gbps(formula_bytes, ms(name)),
                      "formula_frac": frac_of(formula_bytes, ms(name))})
        return e

    entries = {
        "k_project_scatter": kernel_entry("k_project_scatter", design_project, formula_project),
        "k_feature_fused": kernel_entry("k_feature_fused", None, formula_feature),
    }
    entries["k_project_scatter"]["design_model"] = "16 B/point + 4 B per map entry + occupancy and inlier-mask words touched"
    entries["k_feature_fused"]["gather"] = gather
    # The dominant kernel.  With two contexts the projection of one step and the feature kernel of the previous one run
    # side by side for (nearly) the whole step and their launch durations differ by a few per cent, in either direction
    # from box to box: among the kernels within 15 % of the longest launch the one that moves the most HBM bytes is taken -
    # the projection, which also sits on the chain that sets the step period (projection -> classification -> hand-over).
    longest = max(ms(k) for k in entries)
    near = [k for k in entries if ms(k) >= 0.85 * longest] or list(entries)
    dominant = max(near, key=lambda k: (entries[k]["traffic"] or 0.0, ms(k)))
    dom = entries[dominant]
    if dom["frac"] is None:  # no committed counter profile: the design bytes (projection) are the only physical count at hand
        fb = design_project if dominant == "k_project_scatter" else None
        dom = {**dom, "traffic": None, "achieved": gbps(fb, ms(dominant)) if fb else 0.0, "frac": frac_of(fb, ms(dominant)) or 0.0,
               "bytes_source": "design bytes (no profiles/traffic.json)"}
    pmc = pmc_traffic(dominant, S)
    # The step as a whole, stated physically: the HBM bytes the counters saw for its kernels (committed profile) and, as
    # the floor, the compulsory bytes (every cloud read once), over the measured step.
    step_s = elapsed / args.steps
    sets_per_step = B // S
    counter_bytes = None
    if tj:
        scale = float(S) / float(tj["frames_per_launch"])
        counter_bytes = sum(float(tj[k]["hbm_bytes_per_launch"]) for k in
                            ("k_project_scatter", "k_classify", "k_feature_fused", "k_feature_wave") if k in tj) * scale * sets_per_step
    compulsory = 16.0 * N * B
    # HBM time of the step: the streamed bytes (projection, classification) at the rate the projection reaches alone, plus
    # the feature kernel's fetches from memory - random 64-byte lines - at the random-line rate of the device
    # (profiles/tools/randgather.hip, same session as the counters).  A pure device-to-device copy beside one context slows
    # its feature kernel exactly as the other context's projection does (profiles/tools/interference.py, LAB.md 4.25):
    # memory, not wave slots or registers, is what the two contexts share.
    hbm_busy = None
    try:
        if tj:
            scale = float(S) / float(tj["frames_per_launch"]) * sets_per_step
            stream_rate = float(tj["k_project_scatter"]["hbm_bytes_per_launch"]) / float(tj["k_project_scatter"]["launch_s"])
            streamed = (float(tj["k_project_scatter"]["hbm_bytes_per_launch"]) + float(tj["k_classify"]["hbm_bytes_per_launch"])) * scale
            lines = float(tj["k_feature_fused"]["tcc_ea_rdreq"]) * scale
            line_rate = float(tj["gather_ceilings"]["hbm_Glines_s"]) * 1e9
            busy_s = streamed / stream_rate + lines / line_rate
            hbm_busy = {"streamed_bytes": streamed, "stream_rate_GBps": stream_rate / 1e9, "random_lines": lines,
                        "random_line_rate_Glines_s": line_rate / 1e9, "busy_ms": 1e3 * busy_s, "frac_of_step": busy_s / step_s}
    except (KeyError, TypeError, ZeroDivisionError):
        hbm_busy = None
    formula_note = ("SURVEY 8(d)'s per-unit formula charges a per-frame clear of the 1.86 MB pixel map and a 28 B camera-frame "
                    "copy per visible point that this design never performs (tagged map keys, neighbours re-derived from the "
                    "raw point; maps / _pointIndex / depths equal the oracle's: saved work, not skipped work), and 4 B per window "
                    "cell per feature where the kernels scan a 63 KB occupancy bitmap instead - so formula bytes over measured "
                    "time may exceed the HBM peak; they are kept as formula_* fields and are no bound for this design")
    roofline = {
        # ---- the contract's fields, physical: counter bytes of the dominant kernel / its hipEvent time in this run
        "bound": "hbm", "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
        "traffic": dom["traffic"],
        "kernel": dominant, "kernel_ms": ms(dominant), "bytes_source": dom["bytes_source"],
        "dominant_by": ("most HBM bytes among the kernels within 15 % of the longest average launch of the timed schedule "
                        "(hipEvents, this run)"),
        # ---- scalars beside it
        "frac_exclusive": dom.get("frac_exclusive"),           # the same bytes over a launch that has the GPU to itself
        "exclusive_kernel_ms": dom.get("exclusive_kernel_ms"),
        "whole_step_frac_of_peak": (counter_bytes / step_s / 1e9 / HBM_PEAK_GBS) if counter_bytes else None,
        "whole_step_compulsory_frac": compulsory / step_s / 1e9 / HBM_PEAK_GBS,
        "whole_step_hbm_busy_frac": hbm_busy["frac_of_step"] if hbm_busy else None,
        "gather_frac": gather["frac"] if gather else None,     # k_feature_fused: service time of its gathers / its launch
        "feature_kernel_frac": entries["k_feature_fused"]["frac"],
        "design_frac": dom.get("design_frac"),
        "formula_frac": dom.get("formula_frac"), "formula_GBps": dom.get("formula_GBps"),
        "formula_bytes_per_launch": dom.get("formula_bytes_per_launch"),
        "whole_step_formula_GBps": ((formula_project + formula_feature) * (B // S) * args.steps / elapsed) / 1e9,
        "whole_step_formula_frac": ((formula_project + formula_feature) * (B // S) * args.steps / elapsed) / 1e9 / HBM_PEAK_GBS,
        "formula_note": formula_note,
        # two contexts: the projection of one runs beside the feature kernels of the other during the timed region, so
        # `frac` (priced on the launch duration measured THERE, as the contract asks) understates what a kernel does
        # with the chip to itself; frac_exclusive / `exclusive` price the same bytes on a launch that runs alone
        "concurrent": (f"{len(res.ests)} contexts: k_project_scatter of one (step k+1) beside k_classify / k_feature_fused "
                       "/ k_feature_wave of the other (step k)" if len(res.ests) > 1 else None),
        "exclusive": ({"kernel_ms": x_ms(dominant), "achieved": gbps(dom["traffic"] or 0.0, x_ms(dominant)),
                       "frac": dom.get("frac_exclusive") or frac_of(design_project, x_ms(dominant)) or 0.0,
                       "kernels_ms": {k: v.get("avg_ms", 0.0) for k, v in kt_x.items()}}
                      if x_ms(dominant) > 0 else None),
        # the committed PMC profile of the same launch size, priced with ITS OWN kernel time (reproducible from
        # profiles/: traffic.json and the kernel-stats summary it names)
        "traffic_profile": ({"hbm_bytes_per_launch": pmc[0], "launch_ms": pmc[1] * 1e3,
                             "frac": pmc[0] / pmc[1] / 1e9 / HBM_PEAK_GBS, "source": pmc[2]} if pmc else None),
        "kernels": {
            "k_project_scatter": entries["k_project_scatter"],
            "k_classify": {**kt.get("k_classify", {}), "traffic": pmc_bytes("k_classify")},
            "k_feature_fused": entries["k_feature_fused"],
            "k_feature_wave": kt.get("k_feature_wave", {}),
        },
        "whole_step_design_GBps": (design_project * (B // S) * args.steps / elapsed) / 1e9,
    }
    roofline["whole_step"] = {
        "hbm_busy": hbm_busy,
        "counter_bytes": counter_bytes,
        "compulsory_bytes": compulsory,
        "step_ms": 1e3 * step_s,
        "frac_of_peak": (counter_bytes / step_s / 1e9 / HBM_PEAK_GBS) if counter_bytes else None,
        "frac_of_copy_rate": (counter_bytes / step_s / 1e9 / HBM_COPY_GBS) if counter_bytes else None,
        "compulsory_frac_of_peak": compulsory / step_s / 1e9 / HBM_PEAK_GBS,
        "compulsory_frac_of_copy_rate": compulsory / step_s / 1e9 / HBM_COPY_GBS,
        "copy_rate_GBps": HBM_COPY_GBS,
        "counter_source": (tj or {}).get("source", None),
        "formula_note": formula_note,
    }

    cpu = latency = streaming = estimated = None
    configs = {}
    if world == 1 and P.do_use_ransac_plane and not args.no_estimated:
        # the reference's default call: the plane of every frame estimated on the GPU (seeded RANSAC, batched, no host
        # round trip) instead of supplied; checked against the restatement's estimate for three frames
        from oracle import oracle
        res.est_schedule = args.est_schedule
        alt = args.est_schedule == "alternate" and res.whole and len(res.ests) > 1
        if alt:
            res.est_S = res.S
        else:
            for e in res.ests:
                e.setSharedGpu(False)
        loops_e, kt_e = timed_resident(res, max(2, args.steps // 2), 2, timing, 2, estimated=True)
        el_e = loops_e[0]
        est_set = ((res.k - 1) % len(res.batches)) if alt else 0  # output set of the last step of this leg
        for e in res.ests:
            e.setSharedGpu(res.shared_mode if len(res.ests) > 1 else 0)
        # EVERY frame of the leg's output set against the oracle with the restatement's own estimate for the frame's seed
        # (grouped by cloud: the oracle's serial stage A runs once per distinct cloud, the estimate ~1 ms per frame)
        poison_e = res.poison_left(sets=[est_set])  # (halves: every frame goes into the first output set)
        ok_e = poison_e["type_minus77"] == 0
        bad_e = []
        dg_all, tg_all = res.out_depth[est_set].cpu().numpy(), res.out_type[est_set].cpu().numpy()
        ref = oracle.OracleDepthEstimator(P, cam_struct, T)
        t_or = time.perf_counter()
        for u in range(U):
            ref.set_cloud(res.clouds_h[u])
            for fr in range(u, B, U):
                ref.estimate_ground_plane((fr % res.est_S) + 1)
                d0, t0 = ref.calculate_depth(res.uvs_h[fr], 8)
                if not (np.array_equal(tg_all[fr], t0) and np.allclose(dg_all[fr], d0, rtol=0, atol=1e-4, equal_nan=True)):
                    bad_e.append(fr)
        ok_e = ok_e and not bad_e
        est_frames = list(range(B))
        est_oracle_s = time.perf_counter() - t_or
        n_e = max(2, args.steps // 2)
        estimated = {"plane": "estimated", "value": B * F * n_e / el_e, "ms_per_step": 1e3 * el_e / n_e,
                     "kernels_ms_per_launch": {k: v["avg_ms"] for k, v in kt_e.items()},
                     "ransac_us_per_frame": 1e3 * kt_e.get("k_rs_batch", {}).get("avg_ms", 0.0) / res.est_S,
                     "frame_slots_per_launch": res.est_S, "schedule": args.est_schedule, "verified": ok_e,
                     "frames_checked": len(est_frames), "all_frames": True, "mismatching_frames": bad_e[:64],
                     "oracle_seconds": est_oracle_s, "poison_left": poison_e}
    if world == 1:
        # (the one-frame latency legs first: the CPU baseline keeps sixteen OpenMP threads busy for seconds, and calls
        # timed right after it are 10 % slower and noisier)
        if args.latency_frames > 0:
            latency = latency_leg(P, cam, T, res.clouds_h, res.planes_h, res.uvs_h, args.latency_frames, gpu_index)
        if args.streaming_batches > 0:
            streaming = streaming_leg(P, cam, T, res.clouds_h, res.planes_h, res.uvs_h, gpu_index, args.streaming_frames,
                                      args.streaming_batches)
            streaming["stride32"] = streaming_leg(P, cam, T, res.clouds_h, res.planes_h, res.uvs_h, gpu_index,
                                                  args.streaming_frames, args.streaming_batches, stride_floats=8)
            # the reference caller's records, repacked by host threads while they are staged (half the PCIe bytes)
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            streaming["stride32_packed"] = streaming_leg(P, cam, T, res.clouds_h, res.planes_h, res.uvs_h, gpu_index,
                                                         args.streaming_frames, args.streaming_batches,
                                                         pack_threads=max(1, min(32, cores // 2)))
        if args.cpu_seconds > 0:
            cpu = cpu_baseline(P, cam_struct, T, res.clouds_h, res.planes_h, res.uvs_h, args.cpu_seconds)
    clouds_kept = None  # noqa: F841
    res.close()
    del res
    torch.cuda.empty_cache()
    if world == 1 and args.config_frames > 0:
        configs["2"] = {"near_returns": config2_k_leg(P, cam, T, gpu_index, min(B, 1024), F, contexts=args.contexts,
                                                      shared_mode=args.shared_mode)}
        configs["2"]["verified"] = configs["2"]["near_returns"]["verified"]
        torch.cuda.empty_cache()
        configs["3"] = config3_leg(cam, T, gpu_index, args.config_frames)
        configs["5"] = config5_leg(cam, T, gpu_index, min(args.config_frames, 200))
        configs["5"]["batched"] = {str(S5): config5_batched_leg(cam, T, gpu_index, S5, two_contexts=True) for S5 in (16, 64, 256)}
        configs["5"]["verified"] = bool(configs["5"]["verified"] and
                                        all(v["verified"] for v in configs["5"]["batched"].values()))

    value = units / elapsed
    out = {
        "metric": "feature-depth associations/sec",
        "value": value,
        "unit": "feature-depth associations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_step_min": 1e3 * min(loops) / args.steps,
        "ms_per_step_max": 1e3 * max(loops) / args.steps,
        "timed_loops": {"repeats": len(loops), "ms_per_step": [1e3 * x / args.steps for x in loops],
                        "start": loop_start},
        "ms_per_frame": 1e3 * elapsed / args.steps / (B * world),  # whole job: all ranks' frames
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "verified": verified,
        "verification": vrep,
        "config": {
            "workload": (f"BASELINE config 2: 64x2048 cloud ({N} points) x {F} features/frame, C0 parameters, "
                         f"{B} device-resident frames per step per GPU in launch sets of {S}, ground plane known at "
                         "projection (setInputCloud(cloud, plane))"),
            "frames_per_step": B,
            "frame_slots_per_launch": S,
            "contexts": args.contexts,
            "schedule": (("consecutive steps alternate between the contexts" if args.slots <= 0 else
                          "the launch sets of a step alternate between the contexts") +
                         ": the projection of one runs beside the feature kernels of the other" +
                         (" (released behind the classification kernel: mld_order_after_classify)"
                          if args.handover == "classify" and not args.pair else "")
                         if args.contexts > 1 else "one stream, one kernel at a time"),
            "handover": args.handover if args.contexts > 1 and not args.pair else None,
            "features_per_frame": F,
            "points_per_frame": N,
            "sequences": world,
            "parallelism": f"sequence-per-gpu x{world}",
        },
        "result_types": {capi.RESULT_TYPE_NAMES[i]: int(c) for i, c in enumerate(type_hist) if c},
        "result_types_output_sets": type_hist_sets,
        "success_fraction": float((type_hist[1] + type_hist[16]) / max(1, type_hist.sum())),
        "frame_stats": {**{k: float(np.mean([s[k] for s in stats])) for k in
                           ("n_visible", "k1_mean", "k2_mean_fallback", "fallback_features")},
                        "n_front_in_image": float(np.mean([d["n_front_in_image"] for d in design]))},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "plane_estimated": estimated,
        "latency": latency,
        "streaming": streaming if world == 1 else streaming_ranks,
        "distributed": distributed,
        "configs": configs,
    }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    bad_cfg = any(c.get("verified") is False for c in configs.values()) or bool(estimated and not estimated["verified"])
    if latency and latency.get("estimated"):
        bad_cfg = bad_cfg or not all(latency["estimated"][k]["verified"] for k in ("ransac", "semantic"))
    sys.exit(1 if (verified is False or bad_cfg) else 0)


if __name__ == "__main__":
    main()
