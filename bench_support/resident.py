"""Device-resident frames of one rank + the timed loop (used by bench.py and by the secondary legs)."""
from __future__ import annotations

import json
import os
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

K_PROJECT, K_FUSED, K_WAVE, K_RANSAC, K_CLASSIFY = 0, 1, 3, 4, 5  # mld_kernel_time_ms ids (include/mld.h)


def mask_words(inl, n):
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return m.view(np.int32)



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

