"""Multi-GPU layout of the path: independent sequences shard across ranks.

A frame's result depends only on (cloud, features, calibration, parameters, ground plane), so sequences are
assigned statically to GPUs and there is NO data-path collective.  The only exchange is one broadcast of the
calibration block (camera intrinsics, T_cam_lidar, parameter struct; < 1 KB) from rank 0 at start-up — RCCL when
the process group's backend is "nccl", gloo in the CPU tests — plus the max-over-ranks of the elapsed time that
the benchmark contract asks for.  One process per GPU (torch.distributed); inside a sequence the GPUs are
replicas only.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from .capi import MldCamera, MldParams

_CALIB_BYTES = C.sizeof(MldParams) + C.sizeof(MldCamera) + 12 * 8


def assign_sequences(n_sequences: int, world_size: int) -> List[List[int]]:
    """Static assignment sequence s -> rank s mod world_size."""
    out: List[List[int]] = [[] for _ in range(world_size)]
    for s in range(n_sequences):
        out[s % world_size].append(s)
    return out


def pack_calibration(params: MldParams, camera: MldCamera, T_cam_lidar) -> np.ndarray:
    T = np.ascontiguousarray(np.asarray(T_cam_lidar, dtype=np.float64)[:3, :4])
    buf = bytes(params) + bytes(camera) + T.tobytes()
    assert len(buf) == _CALIB_BYTES
    return np.frombuffer(buf, dtype=np.uint8).copy()


def unpack_calibration(blob: np.ndarray) -> Tuple[MldParams, MldCamera, np.ndarray]:
    raw = np.ascontiguousarray(blob, dtype=np.uint8).tobytes()
    assert len(raw) == _CALIB_BYTES
    p = MldParams.from_buffer_copy(raw[:C.sizeof(MldParams)])
    off = C.sizeof(MldParams)
    cam = MldCamera.from_buffer_copy(raw[off:off + C.sizeof(MldCamera)])
    off += C.sizeof(MldCamera)
    T = np.frombuffer(raw[off:], dtype=np.float64).reshape(3, 4).copy()
    return p, cam, T


def broadcast_calibration(params, camera, T_cam_lidar, device=None, src: int = 0):
    """Rank `src` supplies the calibration; every rank returns the same (params, camera, T).

    Non-source ranks may pass None for the three inputs.  Without an initialised process group this is the
    identity (single-GPU use).
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return params, camera, np.asarray(T_cam_lidar, dtype=np.float64)[:3, :4]
    if dist.get_rank() == src:
        blob = torch.from_numpy(pack_calibration(params, camera, T_cam_lidar))
    else:
        blob = torch.zeros(_CALIB_BYTES, dtype=torch.uint8)
    if device is not None:
        blob = blob.to(device)
    dist.broadcast(blob, src=src)
    return unpack_calibration(blob.cpu().numpy())


def max_over_ranks(seconds: float, device=None) -> float:
    """Elapsed time of the slowest rank (benchmark contract)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value: float, device=None):
    """The value of every rank, in rank order (per-rank figures of the benchmark's multi-GPU line)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
