#!/usr/bin/env python3
"""Benchmark of the DepthEstimator hot path on MI355X (contract: the task's bench.py section).

One "step" = one pass of the hot path (setInputCloud with its ground plane + CalculateDepth) over one batch of
`--frames-per-step` synthetic frames of BASELINE.json config 2 (64x2048 cloud, 2000 features/frame, C0 parameters), all
inputs resident in HBM before the timed region.  Metric: feature-depth associations per second = features x frames /
wall time (every submitted feature counts).

stdout carries exactly ONE line: a compact JSON object (< 4 KB; tests/test_bench_gpu.py asserts <= 8192 bytes) with the
contract's keys, `verified` (every frame of the timed batch re-computed by the CPU oracle; exit 1 if not), a physical
`roofline` and `cpu_baseline`.  Everything else - per-kernel rooflines, per-loop times, the verification report and the
SECONDARY LEGS (plane-estimated schedule, PCIe-inclusive latency / streaming, BASELINE configs 2 at k = 7, 3 and 5; run by
`bench_support/run_legs.py` in a child process with a timeout, so a failing leg cannot take the headline line down) - goes
to the detail file (`--detail`, default gpurun_out/bench_detail.json); the compact line names it and carries the legs'
verified flags.  The exit code follows the HEADLINE's `verified` only; leg failures are reported in `legs`.

Multi-GPU: one process per GPU, each rank owns its own sequence (weak scaling); the only collective is the RCCL
broadcast of the calibration block.  `python bench.py --gpus N` starts the N ranks itself (torch.distributed.run as a
child process, before this process touches the GPU) unless it already runs under a launcher (WORLD_SIZE set).  With the
real `nccl` backend a launch wider than the visible GPUs fails at once (exit 3, message naming the rank): ranks are never
wrapped onto one GPU.
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

EXIT_TOO_FEW_GPUS = 3
MAX_LINE_BYTES = 8192


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps per loop (0.15 s of GPU time at the default)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames-per-step", type=int, default=1024, help="resident frames processed per step")
    ap.add_argument("--slots", type=int, default=0,
                    help="frame slots of the context (0 = frames-per-step); a step runs frames-per-step/slots launch sets")
    ap.add_argument("--no-exclusive", action="store_true",
                    help="skip the pass that times each kernel alone (profiling: keeps the kernel statistics of a trace "
                         "to the launches of the timed schedule)")
    ap.add_argument("--contexts", type=int, default=2,
                    help="contexts (HIP streams) consecutive steps are dealt to in turn; with 2 the projection of one "
                         "context runs beside the feature kernels of the other (mld_order_after_classify / "
                         "mld_set_shared_gpu); 1 = everything on one stream, one kernel at a time")
    ap.add_argument("--shared-mode", type=int, default=1,
                    help="mld_set_shared_gpu argument of the alternating contexts: 1 = on; + 256 * n = n feature-kernel "
                         "wavefronts per CU instead of 10")
    ap.add_argument("--handover", choices=("classify", "projection"), default="classify",
                    help="two contexts: the next context's projection is released behind this context's classification "
                         "kernel (mld_order_after_classify; default) or at the end of its projection (mld_order_after)")
    ap.add_argument("--pair", action="store_true",
                    help="two contexts: both projections on one shared stream (mld_pair_contexts); an option of the "
                         "library, slower than the default (LAB.md)")
    ap.add_argument("--unique-frames", type=int, default=16, help="distinct synthetic clouds generated per rank")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--verify-slots", type=int, default=-1,
                    help="frames of the timed batch checked against the oracle: -1 (default) = EVERY frame of every "
                         "context's output set; n > 0 = n frames spread over the output sets; 0 = no check")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed loop of --steps steps runs at least this many times back to back; value / "
                         "ms_per_step are the median loop, ms_per_step_min / _max the spread")
    ap.add_argument("--min-timed-seconds", type=float, default=0.5,
                    help="the timed loop is repeated beyond --repeats until this much time has been timed in all (at "
                         "most 64 loops): a 20-step loop is 15 ms, less than the GPU clocks take to settle")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-every", type=int, default=4,
                    help="record hipEvents around the kernels of every n-th timed step (event records cost ~6 us each)")
    ap.add_argument("--detail", default="", help="detail file (default: gpurun_out/bench_detail.json under the repo root)")
    ap.add_argument("--legs", default="all",
                    help="secondary legs run after the headline (child process, results in the detail file): 'all', 'none' or a "
                         "comma list of estimated,latency,streaming,c2k,c3,c5 (bench_support/run_legs.py)")
    ap.add_argument("--legs-timeout", type=float, default=600.0, help="seconds the secondary legs may take in all")
    # sizes of the secondary legs (0 drops the leg)
    ap.add_argument("--latency-frames", type=int, default=1000)
    ap.add_argument("--streaming-batches", type=int, default=24)
    ap.add_argument("--streaming-frames", type=int, default=64)
    ap.add_argument("--config-frames", type=int, default=256)
    ap.add_argument("--no-estimated", action="store_true", help="drop the plane-estimated leg")
    ap.add_argument("--est-schedule", choices=["halves", "alternate"], default="halves")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def visible_gpus() -> int:
    """GPUs this process could use (torch.cuda.device_count() does not initialise the GPU on this image)."""
    import torch
    return int(torch.cuda.device_count())


def launch_ranks(n: int) -> int:
    """`--gpus N` without a launcher: start N ranks (one per GPU) as a child torch.distributed.run and relay their
    output.  Nothing in this process has initialised the GPU (torch is imported in the workers only; the device COUNT is
    read first so that a launch wider than the node fails before any rank starts)."""
    import socket
    if os.environ.get("MLD_BENCH_BACKEND", "nccl") == "nccl":
        have = visible_gpus()
        if have < n:
            log(f"bench.py: --gpus {n} over nccl (RCCL) needs {n} visible GPUs, this node shows {have}: not started "
                "(ranks are never wrapped onto one GPU; MLD_BENCH_BACKEND=gloo is the functional-test hook for that)")
            return EXIT_TOO_FEW_GPUS
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------ cpu baseline
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
        # (<= 128 characters: what the driver keeps of a string)
        "sample": (f"{frames} frames of the workload in {el:.1f} s: stage A serial {1e3 * t_a / frames:.2f} + stage B "
                   f"{1e3 * t_b / frames:.2f} ms/frame on {cores} of {avail} cores"),
        "sample_detail": (f"stage A = setInputCloud (serial in the reference), stage B = feature loop with {cores} OpenMP "
                          f"threads = fastest of {sorted(probe)} probed on {avail} available cores"),
        "ms_per_frame": 1e3 * el / frames,
        # the same path with a single-thread feature loop (stage A is serial in the reference anyway)
        "one_thread": {"value": F / (t_a / frames + probe[1]), "ms_per_frame": 1e3 * (t_a / frames + probe[1])},
        "stage_a_ms": 1e3 * t_a / frames,
        "stage_b_ms": 1e3 * t_b / frames,
    }



# ------------------------------------------------------------------------------------------------ secondary legs
def legs_to_run(args) -> str:
    want = ["estimated", "latency", "streaming", "c2k", "c3", "c5"] if args.legs == "all" else \
        [x for x in args.legs.split(",") if x and x != "none"]
    drop = set()
    if args.no_estimated:
        drop.add("estimated")
    if args.latency_frames <= 0:
        drop.add("latency")
    if args.streaming_batches <= 0:
        drop.add("streaming")
    if args.config_frames <= 0:
        drop.update(("c2k", "c3", "c5"))
    return ",".join(x for x in want if x not in drop)


def run_secondary_legs(args, gpu_index, legs, out_path):
    """The secondary legs in a CHILD process (started, not exec'ed; this process keeps its GPU context but has released its
    buffers): whatever happens there - exception, crash, hang past the timeout - the headline line is still printed."""
    cmd = [sys.executable, str(ROOT / "bench_support" / "run_legs.py"), "--legs", legs, "--out", str(out_path),
           "--device", str(gpu_index), "--frames-per-step", str(args.frames_per_step), "--features", str(args.features),
           "--unique-frames", str(args.unique_frames), "--contexts", str(args.contexts), "--shared-mode", str(args.shared_mode),
           "--est-steps", str(max(2, args.steps // 2)), "--est-schedule", args.est_schedule,
           "--latency-frames", str(args.latency_frames), "--streaming-batches", str(args.streaming_batches),
           "--streaming-frames", str(args.streaming_frames), "--config-frames", str(args.config_frames)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    rep = {"requested": legs.split(","), "verified": {}, "errors": {}}
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True, timeout=args.legs_timeout,
                           cwd=str(ROOT))
        rep["rc"] = r.returncode
        detail = None
        try:
            detail = json.loads(Path(out_path).read_text())
        except Exception:  # noqa: BLE001
            lines = [ln for ln in (r.stdout or "").splitlines() if ln.startswith("{")]
            detail = json.loads(lines[-1]) if lines else None
        if detail is None:
            rep["errors"]["run_legs"] = f"no detail object (rc {r.returncode})"
        else:
            rep["verified"] = detail.get("verified", {})
            rep["errors"] = {k: str(v)[:160] for k, v in detail.get("errors", {}).items()}
        rep["seconds"] = round(time.perf_counter() - t0, 1)
        return rep, detail
    except subprocess.TimeoutExpired:
        rep["errors"]["run_legs"] = f"timed out after {args.legs_timeout:.0f} s"
    except Exception as e:  # noqa: BLE001
        rep["errors"]["run_legs"] = f"{type(e).__name__}: {e}"
    rep["seconds"] = round(time.perf_counter() - t0, 1)
    return rep, None


def brief(detail):
    """A few scalars of the secondary legs for the compact line (None where a leg did not run)."""
    if not detail:
        return None
    out = {}

    def get(d, *path):
        for p in path:
            if not isinstance(d, dict) or p not in d:
                return None
            d = d[p]
        return d
    pe = detail.get("plane_estimated")
    if pe:
        out["plane_estimated_ms_per_step"] = pe.get("ms_per_step")
    la = detail.get("latency")
    if la:
        out["frame_call_ms"] = {"supplied": la.get("ms_per_frame_median"),
                                "ransac": get(la, "estimated", "ransac", "ms_per_frame_median"),
                                "semantic": get(la, "estimated", "semantic", "ms_per_frame_median"),
                                "process": get(la, "process", "one_call", "ms_per_frame_median")}
    st = detail.get("streaming")
    if st:
        out["streaming_frames_per_s"] = {"pinned16": st.get("frames_per_s"), "pinned32": get(st, "stride32", "frames_per_s"),
                                         "repacked32": get(st, "stride32_packed", "frames_per_s")}
    cf = detail.get("configs") or {}
    c2 = get(cf, "2", "near_returns")
    if c2:
        out["config2_k7"] = {"value": c2.get("value"), "ms_per_step": c2.get("ms_per_step"),
                             "kernel": get(c2, "roofline", "kernel"), "frac": get(c2, "roofline", "frac")}
    c3 = cf.get("3")
    if c3:
        out["config3"] = {"value": get(c3, "modes", "c0_dispose", "associations_per_s"),
                          "near_returns_value": get(c3, "near_returns", "modes", "c0_dispose", "associations_per_s"),
                          "kernel": get(c3, "near_returns", "modes", "c0_dispose", "roofline", "kernel"),
                          "frac": get(c3, "near_returns", "modes", "c0_dispose", "roofline", "frac")}
    c5 = cf.get("5")
    if c5:
        b = c5.get("batched") or {}
        out["config5"] = {"frame_call_ms": c5.get("ms_per_frame"),
                          "batched_value": {s: get(v, "associations_per_s") for s, v in b.items()},
                          "batched_ms_per_step": {s: get(v, "ms_per_step") for s, v in b.items()},
                          "kernel": get(b, "256", "roofline", "kernel"), "frac": get(b, "256", "roofline", "frac")}
    return out


# ------------------------------------------------------------------------------------------------ worker
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # stdout carries the contract line and nothing else: whatever a library prints there from now on (gloo's connection
    # notes, a runtime warning) goes to stderr; the line itself is written to the saved descriptor at the very end
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # No cyclic garbage collection while anything is timed (a full collection of this process's heap is milliseconds on the
    # submitting thread, a step 0.7 ms); reference counting frees everything the bench allocates.
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
    n_dev = int(torch.cuda.device_count())
    if n_dev < 1:
        log(f"bench.py rank {rank}: no GPU visible - the HIP path has no CPU fallback")
        sys.exit(EXIT_TOO_FEW_GPUS)
    if backend == "nccl" and world > 1 and (local_world > n_dev or local_rank >= n_dev):
        # a mis-sized launch must not print an N-rank number measured on fewer GPUs
        log(f"bench.py rank {rank} (local rank {local_rank}): {local_world} ranks on this node but only {n_dev} visible "
            f"GPU(s) - one process per GPU over RCCL needs {local_world}; refusing to share a GPU between ranks")
        sys.exit(EXIT_TOO_FEW_GPUS)
    gpu_index = local_rank % n_dev  # (the modulo only ever applies under the gloo test hook)
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    coll_dev = dev if backend == "nccl" else None
    pr = torch.cuda.get_device_properties(gpu_index)
    pci = None
    if all(hasattr(pr, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    if affinity is not None:
        # the sysfs guess against what the runtime says this rank's device is (HIP order need not be PCI order)
        from mono_lidar_depth_amd.sharding import rebind_if_device_differs
        affinity = rebind_if_device_differs(affinity, pci)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    from bench_support.resident import Resident, kernel_times, timed_resident
    from bench_support.rooflines import design_bytes_project, headline_roofline
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
    elapsed_local = float(np.median(res.local_loops)) if getattr(res, "local_loops", None) else elapsed
    units = sharding.sum_over_ranks(float(B * F * args.steps), device=coll_dev)

    # ---- the timed batch against the oracle (every rank checks its own sequence) ---------------------------
    verified, vrep = res.verify(args.verify_slots) if args.verify_slots != 0 else (None, {})
    n_bad = sharding.sum_over_ranks(0.0 if verified in (True, None) else 1.0, device=coll_dev)
    per_rank_value = sharding.gather_over_ranks(B * F * args.steps / elapsed_local, device=coll_dev)
    numa_nodes = sharding.gather_over_ranks(float(affinity["numa_node"]) if affinity and affinity.get("numa_node") is not None
                                            else -1.0, device=coll_dev)
    pinned_ranks = sharding.sum_over_ranks(1.0 if affinity and affinity.get("applied") else 0.0, device=coll_dev)
    # which physical GPU every rank sits on (PCI address as a number; the device index where the runtime gives none)
    gpu_ident = float(int(pci.replace(":", "").replace(".", ""), 16)) if pci else float(gpu_index)
    gpu_idents = sharding.gather_over_ranks(gpu_ident, device=coll_dev)
    distributed = None
    streaming_ranks = None
    if world > 1:
        distributed = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                       "gpus_distinct": len(set(gpu_idents)),
                       "ranks_verified": int(world - n_bad) if args.verify_slots != 0 else 0,
                       "resident_associations_per_s_per_rank": {"min": min(per_rank_value), "max": max(per_rank_value)},
                       # host placement of the ranks (sharding.bind_to_gpu_numa_node, before the first GPU call)
                       "affinity": {"numa_node_per_rank": [int(x) for x in numa_nodes], "ranks_repinned": int(pinned_ranks)}}
        if args.streaming_batches > 0 and args.legs != "none":
            # BASELINE config 4 as written: every rank STREAMS its own sequence from pinned host memory (PCIe-inclusive)
            try:
                from bench_support.legs import streaming_leg
                barrier()
                st = streaming_leg(P, cam, T, res.clouds_h, res.planes_h, res.uvs_h, gpu_index, args.streaming_frames,
                                   args.streaming_batches)
                fps = sharding.gather_over_ranks(st["frames_per_s"], device=coll_dev)
                slowest = sharding.max_over_ranks(st["frames"] / st["frames_per_s"], device=coll_dev)
                streaming_ranks = {**st, "ranks": world, "frames_per_s": st["frames"] * world / slowest,
                                   "associations_per_s": st["frames"] * world * F / slowest,
                                   "frames_per_s_per_rank": {"min": min(fps), "max": max(fps)},
                                   "aggregate": "frames of all ranks / the slowest rank's elapsed time"}
            except Exception as e:  # noqa: BLE001  (collectives inside: every rank fails or none, by construction of the leg)
                streaming_ranks = {"error": f"{type(e).__name__}: {e}"}
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
    roofline, roofline_detail = headline_roofline(kt, kt_x, S, B, N, args.steps, elapsed, len(res.ests), design_project,
                                                  formula_project, formula_feature)

    cpu = None
    if world == 1 and args.cpu_seconds > 0:
        try:
            cpu = cpu_baseline(P, cam_struct, T, res.clouds_h, res.planes_h, res.uvs_h, args.cpu_seconds)
        except Exception as e:  # noqa: BLE001
            cpu = {"error": f"{type(e).__name__}: {e}"}
    res.close()
    del res
    torch.cuda.empty_cache()

    value = units / elapsed
    line = {
        "metric": "feature-depth associations/sec",
        "value": value,
        "unit": "feature-depth associations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "ms_per_frame": 1e3 * elapsed / args.steps / (B * world),  # whole job: all ranks' frames
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "verified": verified,
        "config": {
            "workload": f"BASELINE config 2: 64x2048 cloud x {F} features/frame, C0 parameters, {B} HBM-resident frames/step/GPU",
            "frames_per_step": B, "frame_slots_per_launch": S, "features_per_frame": F, "points_per_frame": N,
            "contexts": args.contexts, "sequences": world, "parallelism": f"sequence-per-gpu x{world}",
        },
        "roofline": roofline,
        "cpu_baseline": ({k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample", "ms_per_frame") if k in cpu}
                         if cpu and "error" not in cpu else cpu),
        "ms_per_step_min": 1e3 * min(loops) / args.steps,
        "ms_per_step_max": 1e3 * max(loops) / args.steps,
        "timed_loops": len(loops),
        "frames_checked": vrep.get("frames_checked"),
        "max_abs_depth_diff_m": vrep.get("max_abs_depth_diff_m"),
        "success_fraction": float((type_hist[1] + type_hist[16]) / max(1, type_hist.sum())),
    }
    if distributed:
        line["distributed"] = {k: distributed[k] for k in ("backend", "world_size", "gpus_distinct", "ranks_verified")}
        if streaming_ranks and "frames_per_s" in streaming_ranks:
            line["distributed"]["streaming_frames_per_s"] = streaming_ranks["frames_per_s"]

    detail = {
        **{k: v for k, v in line.items() if k not in ("roofline",)},
        "roofline": roofline_detail,
        "cpu_baseline": cpu,
        "timed_loops": {"repeats": len(loops), "ms_per_step": [1e3 * x / args.steps for x in loops],
                        "start": "idle GPU (synchronised): exactly --steps steps per loop"},
        "verification": vrep,
        "config": {**line["config"],
                   "schedule": (("consecutive steps alternate between the contexts" if args.slots <= 0 else
                                 "the launch sets of a step alternate between the contexts") +
                                ": the projection of one runs beside the feature kernels of the other" +
                                (" (released behind the classification kernel: mld_order_after_classify)"
                                 if args.handover == "classify" and not args.pair else "")
                                if args.contexts > 1 else "one stream, one kernel at a time"),
                   "handover": args.handover if args.contexts > 1 and not args.pair else None},
        "result_types": {capi.RESULT_TYPE_NAMES[i]: int(c) for i, c in enumerate(type_hist) if c},
        "result_types_output_sets": len(vrep.get("frames_per_output_set", [])) or None,
        "frame_stats": {**{k: float(np.mean([s[k] for s in stats])) for k in
                           ("n_visible", "k1_mean", "k2_mean_fallback", "fallback_features")},
                        "n_front_in_image": float(np.mean([d["n_front_in_image"] for d in design]))},
        "distributed": ({**distributed, "affinity": {**distributed["affinity"], "rank0": affinity}} if distributed else None),
        "streaming": streaming_ranks,
    }
    detail_path = Path(args.detail) if args.detail else ROOT / "gpurun_out" / "bench_detail.json"
    try:  # (a read-only checkout: the detail goes to the temporary directory instead)
        detail_path.parent.mkdir(parents=True, exist_ok=True)
        if not os.access(detail_path.parent, os.W_OK):
            raise OSError("not writable")
    except OSError:
        import tempfile
        detail_path = Path(tempfile.gettempdir()) / "mld_bench_detail.json"

    def write_detail():
        try:
            detail_path.parent.mkdir(parents=True, exist_ok=True)
            detail_path.write_text(json.dumps(detail) + "\n")
            return True
        except OSError as e:
            log(f"bench.py: detail file not written: {e}")
            return False

    write_detail()  # (the headline's detail is on disk before any secondary leg starts)
    legs = legs_to_run(args) if world == 1 else ""
    if legs:
        legs_path = detail_path.with_name(detail_path.stem + "_legs.json")
        rep, legs_detail = run_secondary_legs(args, gpu_index, legs, legs_path)
        line["legs"] = {**rep, "brief": brief(legs_detail)}
        if legs_detail:
            for k in ("plane_estimated", "latency", "streaming", "configs"):
                detail[k] = legs_detail.get(k)
        detail["legs"] = rep
        detail["errors"] = rep["errors"]
    line["detail"] = str(detail_path.relative_to(ROOT)) if detail_path.is_relative_to(ROOT) else str(detail_path)
    if not write_detail():
        line["detail"] = None

    text = json.dumps(line)
    if len(text) > MAX_LINE_BYTES:  # (cannot happen with the fields above; a guard, not a mechanism)
        line.pop("legs", None)
        text = json.dumps(line)
    log(f"bench.py: detail in {detail_path}")
    os.write(line_fd, (text + "\n").encode())
    if world > 1:
        dist.destroy_process_group()
    sys.exit(1 if verified is False else 0)


if __name__ == "__main__":
    main()
