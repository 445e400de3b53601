#!/usr/bin/env python3
"""Parity sweep of the batched tracklet layer (mld_tracklets_depths_device: two banks of slots, the previous frame's slot
resident, feature groups when the sequences are few) over random configurations (the generator of
tests/test_randomized_gpu.py): S sequences x 3 frames per configuration against the CPU restatement of
TrackletDepthModule::process (TEST TOOL: the oracle is the checker).
usage: random_sweep_tracklets.py first_seed n_seeds [sequences=3] [tracks=2600]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from mono_lidar_depth_amd import TrackletBatch, synth  # noqa: E402
from helpers import make_oracle  # noqa: E402
from oracle import oracle  # noqa: E402
from test_randomized_gpu import _random_setup  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
NT = int(sys.argv[4]) if len(sys.argv) > 4 else 2600
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731


def mask_of(inl, n):
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return torch.from_numpy(m.view(np.int32)).to(dev)


t0 = time.perf_counter()
bad, worst, worst_seed, checks = [], 0.0, -1, 0
for seed in range(first, first + count):
    P, cam, T, scanner, kw = _random_setup(seed)
    if P.do_use_PCA:   # (the tracklet comparison below is written for the 1e-4 m paths; PCA configurations are swept elsewhere)
        P = P.replace(do_use_PCA=0)
    rng = np.random.default_rng(88000 + seed)
    n_tracks = [NT - 311 * s for s in range(S)]  # ragged
    tb = TrackletBatch(P, cam, T, S, max(n_tracks), list_capacity=(48, 24) if (seed & 1) else None)
    ref_last = [None] * S
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
            tb.frame([to(h[0]) for h in host], np.stack([h[1] for h in host]), [mask_of(h[2], h[0].shape[0]) for h in host],
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
                if dm > worst:
                    worst, worst_seed = dm, seed
                ref_last[s] = ref_cur
    except AssertionError as e:  # noqa: PERF203
        bad.append((seed, str(e)[:200]))
    finally:
        tb.close()
print(f"tracklet sweep ({S} sequences x 3 frames, {NT} tracks, list capacities default / 48-24 by seed parity): seeds {first} .. "
      f"{first + count - 1}: {count - len(bad)} of {count} configurations ({checks} sequence-frames) equal to the oracle in "
      f"{time.perf_counter() - t0:.0f} s; max |d - oracle| = {worst:.3e} m (seed {worst_seed}; float32 outputs)")
for s, why in bad[:20]:
    print("MISMATCH seed", s, why)
sys.exit(1 if bad else 0)
