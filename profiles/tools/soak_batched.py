#!/usr/bin/env python3
"""Soak of the batched tracklet steps (descriptor tables through the pinned upload ring, two contexts handing over behind
the classification): thousands of steps queued without waiting, three distinct frame sets in rotation with their own
output arrays; at the end and at every K-th synchronisation the outputs of every set are compared with what the FIRST
pass over that set produced (bit for bit), and the first pass itself with the oracle (TEST TOOL: the oracle is the
checker).  usage: soak_batched.py [steps] [sync_every]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
from mono_lidar_depth_amd import CameraPinhole, TrackletBatch, capi, synth  # noqa: E402
from helpers import make_oracle  # noqa: E402
from oracle import oracle  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 997
dev = torch.device("cuda:0")
P = capi.params_c0()
cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV)
S, NT = 4, 1500
rng = np.random.default_rng(5)
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731


def mask_of(inl, n):
    m = np.zeros((n + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
    return torch.from_numpy(m.view(np.int32)).to(dev)


scanners = [synth.HDL64_KITTI, synth.VLP16, synth.DENSE128, synth.HDL64]
host = []   # host[k][s] = (cloud, coeffs, inl, u0, v0, u1, v1, is_new)
for k in range(3):
    per = []
    for s in range(S):
        cloud = synth.make_cloud(scanners[s], seed=90 + s, frame=2 * k)
        coeffs, inl = synth.make_ground_plane(cloud)
        u0 = rng.integers(0, cam.width, NT).astype(np.float32)
        v0 = rng.integers(0, cam.height, NT).astype(np.float32)
        u1 = np.clip(u0 + rng.integers(-3, 4, NT), 0, cam.width - 1).astype(np.float32)
        v1 = np.clip(v0 + rng.integers(-3, 4, NT), 0, cam.height - 1).astype(np.float32)
        per.append((cloud, coeffs, inl, u0, v0, u1, v1, (rng.random(NT) < 0.1)))
    host.append(per)


def context():
    tb = TrackletBatch(P, cam, synth.T_CAM_LIDAR, S, NT, list_capacity=(48, 24))
    outs, preps = [], []
    for k in range(3):
        per = host[k]
        o = ([torch.empty(NT, dtype=torch.float32, device=dev) for _ in range(S)],
             [torch.full((NT,), float("nan"), dtype=torch.float32, device=dev) for _ in range(S)],
             [torch.empty(NT, dtype=torch.int32, device=dev) for _ in range(S)],
             [torch.zeros(NT, dtype=torch.int32, device=dev) for _ in range(S)])
        outs.append(o)
        preps.append(tb.prepare([to(p[0]) for p in per], np.stack([p[1] for p in per]),
                                [mask_of(p[2], p[0].shape[0]) for p in per], [to(p[3]) for p in per], [to(p[4]) for p in per],
                                [to(p[5]) for p in per], [to(p[6]) for p in per], [to(p[7].astype(np.uint8)) for p in per], *o))
    return tb, outs, preps


def snapshot(outs):
    return [[t.cpu().numpy().copy() for g in o for t in g] for o in outs]


a, outs_a, preps_a = context()
b, outs_b, preps_b = context()
for x in (a, b):
    x.est.setSharedGpu(1)
torch.cuda.synchronize()
a.est._after_torch(outs_a[0][0][0])
b.est._after_torch(outs_b[0][0][0])
pair = [(a, preps_a), (b, preps_b)]
# first pass (3 + 3 steps of each context: every set has been the current AND has had its predecessor as the previous frame)
for it in range(12):
    x, pr = pair[it % 2]
    x.run(pr[(it // 2) % 3], pair[(it + 1) % 2][0], "classify")
for x in (a, b):
    x.est.synchronize()
ref_a, ref_b = snapshot(outs_a), snapshot(outs_b)
for k in range(3):
    for r, q in zip(ref_a[k], ref_b[k]):
        assert np.array_equal(r, q, equal_nan=True), "the two contexts disagree"
# the first pass against the oracle: set k's current frame, with set k-1 as the previous frame
bad = 0
for k in range(3):
    for s in range(S):
        cloud, coeffs, inl, u0, v0, u1, v1, new = host[k][s]
        pc, pco, pin = host[(k - 1) % 3][s][:3]
        cur, last = make_oracle(P), make_oracle(P)
        cur.set_cloud(cloud)
        cur.set_ground_plane(coeffs, inl)
        last.set_cloud(pc)
        last.set_ground_plane(pco, pin)
        e_cur, e_last, et_cur, et_last = oracle.tracklets_depth(cur, last, u0, v0, u1, v1, new, n_threads=8)
        dc, dl, tc, tl = ref_a[k][s], ref_a[k][S + s], ref_a[k][2 * S + s], ref_a[k][3 * S + s]
        ok = (np.array_equal(tc, et_cur) and np.allclose(dc, e_cur, rtol=0, atol=1e-4, equal_nan=True)
              and np.array_equal(tl[new], et_last[new]) and np.allclose(dl[new], e_last[new], rtol=0, atol=1e-4, equal_nan=True))
        bad += 0 if ok else 1
print(f"first pass against the oracle: {3 * S - bad} of {3 * S} (set, sequence) pairs equal")
t0 = time.perf_counter()
mism = checks = 0
for it in range(12, 12 + STEPS):
    x, pr = pair[it % 2]
    x.run(pr[(it // 2) % 3], pair[(it + 1) % 2][0], "classify")
    if (it % EVERY) == 0 or it == 12 + STEPS - 1:
        for x in (a, b):
            x.est.synchronize()
        for ref, outs in ((ref_a, outs_a), (ref_b, outs_b)):
            now = snapshot(outs)
            for k in range(3):
                checks += 1
                if not all(np.array_equal(r, q, equal_nan=True) for r, q in zip(ref[k], now[k])):
                    mism += 1
el = time.perf_counter() - t0
print(f"soak: {STEPS} steps of {S} sequences x {NT} tracks in {el:.1f} s ({1e3 * el / STEPS:.3f} ms per step incl. checks), "
      f"{checks} output-set comparisons, mismatches: {mism}, oracle mismatches: {bad}")
a.close()
b.close()
sys.exit(1 if (mism or bad) else 0)
