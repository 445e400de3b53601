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
