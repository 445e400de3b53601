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
from typing import List, Optional, Tuple

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


# ---------------------------------------------------------------------------------------------- host placement
def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_numa_nodes(sysfs_root: str = "/sys") -> List[Tuple[str, int]]:
    """(PCI address, NUMA node) of every AMD GPU with a render node, in PCI-address order - the order the HIP runtime
    enumerates them in when no *_VISIBLE_DEVICES variable re-orders it.  Reads sysfs only: no GPU call."""
    import glob
    import os
    found = {}
    for node in glob.glob(os.path.join(sysfs_root, "class/drm/renderD*")):
        dev = os.path.realpath(os.path.join(node, "device"))
        try:
            with open(os.path.join(dev, "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(dev, "numa_node")) as f:
                numa = int(f.read().strip())
        except (OSError, ValueError):
            continue
        found[os.path.basename(dev)] = numa
    return sorted(found.items())


def rebind_if_device_differs(info: dict, device_pci: Optional[str], sysfs_root: str = "/sys") -> dict:
    """After the GPU is initialised: `device_pci` is the PCI address ("0000:c1:00.0") the runtime reports for this rank's
    device.  bind_to_gpu_numa_node guessed the device from sysfs order; when the guess named another GPU, the process is
    re-pinned to the right node's cores (sched_setaffinity only - nothing is re-executed; staging buffers are allocated
    later).  Returns `info` with `pci_device`, `pci_matches` and, after a correction, the new node."""
    import os
    info = dict(info or {})
    info["pci_device"] = device_pci
    if not device_pci or not info.get("pci"):
        info["pci_matches"] = None
        return info
    info["pci_matches"] = device_pci.lower() == str(info["pci"]).lower()
    if info["pci_matches"]:
        return info
    nodes = dict((k.lower(), v) for k, v in gpu_numa_nodes(sysfs_root))
    numa = nodes.get(device_pci.lower())
    if numa is None or numa < 0:
        return {**info, "reason": "device not found in sysfs / no NUMA node: affinity left as guessed"}
    try:
        with open(os.path.join(sysfs_root, f"devices/system/node/node{numa}/cpulist")) as f:
            want = set(_parse_cpulist(f.read()))
        # never wider than the mask the process was started under (cgroup / taskset / launcher restriction)
        allowed = info.get("initial_cpus")
        if allowed is not None:
            want &= set(allowed)
        want = sorted(want)
        if not want:
            return {**info, "reason": "none of the right node's cores is in the process's initial affinity mask: left as guessed"}
        os.sched_setaffinity(0, want)
    except OSError as e:
        return {**info, "reason": f"re-bind failed: {e}"}
    return {**info, "pci": device_pci, "numa_node": numa, "applied": True, "cpus": len(want), "cpu_first": want[0],
            "cpu_last": want[-1], "corrected_after_init": True}


def bind_to_gpu_numa_node(local_rank: int, sysfs_root: str = "/sys") -> dict:
    """Pins the calling process to the CPU cores of the NUMA node its GPU hangs off (config 4: every rank streams its
    sequence from pinned host memory, 8 x ~50 GB/s on one node - a rank whose staging buffers and copy threads sit on
    the other socket pays the inter-socket link on every frame).  Call it in the worker BEFORE the first GPU call and
    before any pinned allocation: memory is then first-touched, and the runtime's helper threads are created, on that
    node.  Never re-executes anything.  Returns what it did (reported as `distributed.affinity` by bench.py)."""
    import os
    info = {"local_rank": int(local_rank), "applied": False, "numa_node": None, "pci": None, "cpus": None}
    try:
        before = sorted(os.sched_getaffinity(0))
    except AttributeError:  # not Linux
        return info
    info["cpus"] = len(before)
    info["initial_cpus"] = before  # (the mask the process was started under: a later re-bind stays inside it)
    gpus = gpu_numa_nodes(sysfs_root)
    visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    index = int(local_rank)
    if visible:
        try:
            ids = [int(x) for x in visible.split(",") if x.strip() != ""]
            index = ids[index % len(ids)] if ids else index
        except ValueError:
            return {**info, "reason": "non-numeric *_VISIBLE_DEVICES"}
    if not gpus:
        return {**info, "reason": "no AMD render node in sysfs"}
    pci, numa = gpus[index % len(gpus)]
    info.update({"pci": pci, "numa_node": numa})
    if numa < 0:
        return {**info, "reason": "the GPU reports no NUMA node (single-node host)"}
    try:
        with open(os.path.join(sysfs_root, f"devices/system/node/node{numa}/cpulist")) as f:
            node_cpus = set(_parse_cpulist(f.read()))
    except OSError:
        return {**info, "reason": f"node{numa}/cpulist unreadable"}
    want = sorted(node_cpus & set(before))
    if not want:
        return {**info, "reason": "none of the node's cores is in this process's affinity mask"}
    os.sched_setaffinity(0, want)
    return {**info, "applied": want != before, "cpus": len(want), "cpu_first": want[0], "cpu_last": want[-1]}
