"""The calibration broadcast and the scalar reductions of the multi-GPU layout through RCCL itself (backend "nccl" on ROCm),
with the one GPU of the test box: a process group of ONE rank.  No exchange between GPUs happens (there is one), but
everything else does - RCCL is loaded, the communicator is created on the rank's device (`device_id=`), the uint8
calibration block and the float64 scalars go through ncclBroadcast / ncclAllReduce / ncclAllGather on device tensors and come
back intact.  (The N > 1 path: world-size-2 gloo tests on CPU, `bench.py --gpus 2 / 8` over the gloo hook; an 8-GPU node has
never been available.)"""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def test_calibration_broadcast_over_rccl_single_rank():
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        import torch
        import torch.distributed as dist
        from mono_lidar_depth_amd import CameraPinhole, capi, sharding, synth
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group(backend="nccl", device_id=dev)
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        P = capi.params_c0().replace(pixelarea_search_witdh=7, treshold_depth_max=77)
        cam = CameraPinhole(synth.KITTI_W, synth.KITTI_H, synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV).as_struct()
        # (world size 1 short-circuits in sharding.*: call the collectives the N > 1 path uses, on device tensors)
        blob = torch.from_numpy(sharding.pack_calibration(P, cam, synth.T_CAM_LIDAR)).to(dev)
        sent = blob.clone()
        dist.broadcast(blob, src=0)
        P2, cam2, T2 = sharding.unpack_calibration(blob.cpu().numpy())
        assert torch.equal(blob, sent) and bytes(P2) == bytes(P) and bytes(cam2) == bytes(cam)
        assert np.array_equal(T2, np.asarray(synth.T_CAM_LIDAR)[:3, :4])
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = [torch.zeros_like(t)]
        dist.all_gather(out, t)
        assert float(out[0].item()) == 1.25
        dist.barrier()
        dist.destroy_process_group()
        print("rccl ok")
    """ % str(ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29633",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_numa_pinning_against_the_real_sysfs_of_the_gpu_box():
    """`bind_to_gpu_numa_node` + `rebind_if_device_differs` against the box's own /sys and the PCI address the runtime
    reports (the CPU suite only has a fake sysfs): whatever the box looks like - GPUs without a NUMA node, more GPUs in sysfs
    than are visible, a restricted affinity mask -, the rank ends inside the mask it was started under, with at least one
    core, and when the device is found in sysfs with a NUMA node, pinned to THAT node's cores."""
    code = textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        from mono_lidar_depth_amd import sharding
        start = sorted(os.sched_getaffinity(0))
        info = sharding.bind_to_gpu_numa_node(0)                 # before any GPU call, as bench.py's worker does
        guessed = sorted(os.sched_getaffinity(0))
        import torch
        pr = torch.cuda.get_device_properties(0)
        pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        info = sharding.rebind_if_device_differs(info, pci)
        final = sorted(os.sched_getaffinity(0))
        nodes = dict((k.lower(), v) for k, v in sharding.gpu_numa_nodes())
        out = {"info": {k: v for k, v in info.items() if k != "initial_cpus"}, "start": len(start), "final": len(final),
               "inside": set(final) <= set(start), "sysfs_gpus": len(nodes), "device_in_sysfs": pci.lower() in nodes,
               "device_node": nodes.get(pci.lower())}
        if out["device_in_sysfs"] and nodes[pci.lower()] >= 0:
            with open(f"/sys/devices/system/node/node{nodes[pci.lower()]}/cpulist") as f:
                node_cpus = set(sharding._parse_cpulist(f.read()))
            out["node_cpus"] = len(node_cpus)
            out["on_node"] = (set(final) <= node_cpus) if (node_cpus & set(start)) else None
        torch.zeros(8, device="cuda").sum().item()                # the GPU still works from the pinned process
        print("NUMA " + json.dumps(out))
    """ % str(ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    import json
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("NUMA ")][-1][5:])
    print(out)
    assert out["inside"] and out["final"] >= 1, out
    info = out["info"]
    assert info["pci_device"] and info["local_rank"] == 0, out
    if out["device_in_sysfs"]:
        # the guess either named the device or the re-bind corrected it (or had a stated reason not to)
        assert info["pci_matches"] or info.get("corrected_after_init") or info.get("reason"), out
        if out.get("on_node") is not None and not info.get("reason"):
            assert out["on_node"], out
            assert info["numa_node"] == out["device_node"], out
    else:
        assert info["pci_matches"] in (False, None), out
