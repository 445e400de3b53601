"""Pipelined use of one context: double-buffered host->device copies on a copy stream, the kernels on the context's
stream (mld_get_stream), results copied back on a third stream — every batch must equal the oracle (no buffer is
overwritten before its consumer has run, no result is read before it is written)."""
import numpy as np
import pytest

from mono_lidar_depth_amd import capi, synth

from helpers import assert_depth_parity, make_estimator, run_oracle

pytestmark = pytest.mark.gpu


def test_double_buffered_batches_match_oracle():
    import torch
    P = capi.params_c0()
    S, F, n_batches = 4, 600, 6
    dev = torch.device("cuda:0")
    sc = synth.HDL64_KITTI
    N = synth.make_cloud(sc, seed=0).shape[0]
    words = (N + 31) // 32
    frames = []
    for i in range(n_batches * S):
        cloud = synth.make_cloud(sc, seed=400 + i % 5, frame=i)
        uv = synth.make_features(F, seed=500 + i)
        coeffs, inl = synth.make_ground_plane(cloud)
        m = np.zeros(words, dtype=np.uint32)
        np.bitwise_or.at(m, inl >> 5, (np.uint32(1) << (inl & 31).astype(np.uint32)))
        frames.append((cloud, uv, (coeffs, inl), m.view(np.int32)))
    est = make_estimator(P, max_frames=S)
    compute = torch.cuda.ExternalStream(est.stream, device=dev)
    copy_in, copy_out = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    bufs, batches = [], []
    coeffs = np.stack([frames[0][2][0]] * S)
    for _ in range(2):
        d = {"cloud": torch.empty((S, N, 4), dtype=torch.float32, device=dev),
             "mask": torch.empty((S, words), dtype=torch.int32, device=dev),
             "uv": torch.empty((S, F, 2), dtype=torch.float64, device=dev),
             "depth": torch.empty((S, F), dtype=torch.float64, device=dev),
             "type": torch.empty((S, F), dtype=torch.int32, device=dev)}
        bufs.append(d)
        batches.append(est.prepareBatch([d["cloud"][b] for b in range(S)], [d["uv"][b] for b in range(S)],
                                        [d["depth"][b] for b in range(S)], [d["type"][b] for b in range(S)], coeffs,
                                        [d["mask"][b] for b in range(S)]))
    h_in = [(torch.empty((S, N, 4), dtype=torch.float32).pin_memory(), torch.empty((S, words), dtype=torch.int32).pin_memory(),
             torch.empty((S, F, 2), dtype=torch.float64).pin_memory()) for _ in range(n_batches)]
    h_out = [(torch.empty((S, F), dtype=torch.float64).pin_memory(), torch.empty((S, F), dtype=torch.int32).pin_memory())
             for _ in range(n_batches)]
    for i in range(n_batches):
        for b in range(S):
            cloud, uv, _, mask = frames[i * S + b]
            h_in[i][0][b] = torch.from_numpy(cloud)
            h_in[i][1][b] = torch.from_numpy(mask)
            h_in[i][2][b] = torch.from_numpy(uv)
    done = [None, None]
    torch.cuda.synchronize()
    for i in range(n_batches):
        k = i % 2
        with torch.cuda.stream(copy_in):
            if done[k] is not None:
                copy_in.wait_event(done[k])
            bufs[k]["cloud"].copy_(h_in[i][0], non_blocking=True)
            bufs[k]["mask"].copy_(h_in[i][1], non_blocking=True)
            bufs[k]["uv"].copy_(h_in[i][2], non_blocking=True)
            copied = copy_in.record_event()
        compute.wait_event(copied)
        est.runBatch(batches[k])
        ev = torch.cuda.Event()
        ev.record(compute)
        copy_out.wait_event(ev)
        with torch.cuda.stream(copy_out):
            h_out[i][0].copy_(bufs[k]["depth"], non_blocking=True)
            h_out[i][1].copy_(bufs[k]["type"], non_blocking=True)
            done[k] = copy_out.record_event()
    torch.cuda.synchronize()
    for i in range(n_batches):
        for b in range(S):
            cloud, uv, plane, _ = frames[i * S + b]
            _, (d0, t0) = run_oracle(P, cloud, uv, plane)
            assert_depth_parity(h_out[i][0][b].numpy(), h_out[i][1][b].numpy(), d0, t0)
