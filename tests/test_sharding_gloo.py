"""world_size-2 gloo tests of the multi-GPU layout: calibration broadcast, sequence assignment, timing reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mono_lidar_depth_amd import capi, sharding, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 0:
            P = capi.params_c0().replace(treshold_depth_max=77, histogram_segmentation_bin_witdh=0.25)
            cam = capi.MldCamera(synth.KITTI_F, synth.KITTI_CU, synth.KITTI_CV, synth.KITTI_W, synth.KITTI_H)
            T = synth.T_CAM_LIDAR
        else:
            P = cam = T = None
        P, cam, T = sharding.broadcast_calibration(P, cam, T)
        seqs = sharding.assign_sequences(8, world)[rank]
        slow = sharding.max_over_ranks(1.0 + rank)
        total = sharding.sum_over_ranks(100.0 * (rank + 1))
        q.put((rank, bytes(P), bytes(cam), T.tolist(), seqs, slow, total))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, p0, c0, t0, s0, slow0, tot0), (r1, p1, c1, t1, s1, slow1, tot1) = res
    assert p0 == p1 and c0 == c1 and t0 == t1  # every rank holds rank 0's calibration
    P = capi.MldParams.from_buffer_copy(p1)
    assert P.treshold_depth_max == 77 and P.histogram_segmentation_bin_witdh == 0.25
    assert np.array_equal(np.array(t1), synth.T_CAM_LIDAR)
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5, 7]  # sequence s -> rank s mod world
    assert slow0 == slow1 == 2.0 and tot0 == tot1 == 300.0


def test_pack_roundtrip_and_identity_without_process_group():
    P = capi.params_default()
    cam = capi.MldCamera(600.0, 50.0, 50.0, 100, 100)
    blob = sharding.pack_calibration(P, cam, synth.T_CAM_LIDAR)
    assert blob.nbytes < 1024
    P2, cam2, T2 = sharding.unpack_calibration(blob)
    assert bytes(P2) == bytes(P) and bytes(cam2) == bytes(cam) and np.array_equal(T2, synth.T_CAM_LIDAR)
    P3, cam3, T3 = sharding.broadcast_calibration(P, cam, synth.T_CAM_LIDAR)
    assert P3 is P and cam3 is cam
    assert sharding.max_over_ranks(3.5) == 3.5 and sharding.sum_over_ranks(2.0) == 2.0
    assert sharding.assign_sequences(3, 8) == [[0], [1], [2], [], [], [], [], []]


def _fake_sysfs(root, gpus, nodes):
    """gpus: [(pci, vendor, numa)], nodes: {node: cpulist} - the sysfs entries bind_to_gpu_numa_node reads."""
    for i, (pci, vendor, numa) in enumerate(gpus):
        dev = root / "devices" / "pci0000:00" / pci
        dev.mkdir(parents=True)
        (dev / "vendor").write_text(vendor + "\n")
        (dev / "numa_node").write_text(f"{numa}\n")
        node = root / "class" / "drm" / f"renderD{128 + i}"
        node.mkdir(parents=True)
        (node / "device").symlink_to(dev)
    for n, cpus in nodes.items():
        d = root / "devices" / "system" / "node" / f"node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")


def test_rank_is_pinned_to_the_numa_node_of_its_gpu(tmp_path, monkeypatch):
    """Config 4 readiness: rank r's process is confined to the cores of GPU r's NUMA node before any GPU call
    (sysfs only).  GPUs are taken in PCI-address order; a foreign render node (not AMD) is skipped."""
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        import pytest
        pytest.skip("needs two cores in the affinity mask")
    lo, hi = have[:len(have) // 2], have[len(have) // 2:]
    as_list = lambda cs: ",".join(str(c) for c in cs)  # noqa: E731
    _fake_sysfs(tmp_path, [("0000:85:00.0", "0x1002", 1), ("0000:05:00.0", "0x1002", 0), ("0000:03:00.0", "0x10de", 0),
                           ("0000:c5:00.0", "0x1002", -1)], {0: as_list(lo), 1: as_list(hi)})
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert sharding.gpu_numa_nodes(str(tmp_path)) == [("0000:05:00.0", 0), ("0000:85:00.0", 1), ("0000:c5:00.0", -1)]
    try:
        a = sharding.bind_to_gpu_numa_node(0, str(tmp_path))
        assert a["numa_node"] == 0 and a["pci"] == "0000:05:00.0" and sorted(os.sched_getaffinity(0)) == lo and a["applied"]
        os.sched_setaffinity(0, have)
        b = sharding.bind_to_gpu_numa_node(1, str(tmp_path))
        assert b["numa_node"] == 1 and sorted(os.sched_getaffinity(0)) == hi and b["cpus"] == len(hi)
        os.sched_setaffinity(0, have)
        c = sharding.bind_to_gpu_numa_node(2, str(tmp_path))   # a GPU without a NUMA node: nothing changes
        assert c["numa_node"] == -1 and not c["applied"] and sorted(os.sched_getaffinity(0)) == have
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")       # local rank 0 is device 1 of the PCI order
        d = sharding.bind_to_gpu_numa_node(0, str(tmp_path))
        assert d["pci"] == "0000:85:00.0" and sorted(os.sched_getaffinity(0)) == hi
        # after GPU initialisation the runtime names the rank's device: a wrong guess is corrected, a right one kept
        monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
        os.sched_setaffinity(0, have)
        e = sharding.bind_to_gpu_numa_node(0, str(tmp_path))
        same = sharding.rebind_if_device_differs(e, "0000:05:00.0", str(tmp_path))
        assert same["pci_matches"] is True and sorted(os.sched_getaffinity(0)) == lo and "corrected_after_init" not in same
        moved = sharding.rebind_if_device_differs(e, "0000:85:00.0", str(tmp_path))
        assert moved["pci_matches"] is False and moved["corrected_after_init"] and moved["numa_node"] == 1
        assert sorted(os.sched_getaffinity(0)) == hi
        unknown = sharding.rebind_if_device_differs(e, None, str(tmp_path))
        assert unknown["pci_matches"] is None
        # a rank started under a restricted mask (cgroup / taskset) is never widened past it by the correction: started on
        # the lower half only, the right node's cores (upper half) are all outside -> affinity left as guessed, reason given
        os.sched_setaffinity(0, lo)
        f = sharding.bind_to_gpu_numa_node(0, str(tmp_path))
        assert f["initial_cpus"] == lo
        kept = sharding.rebind_if_device_differs(f, "0000:85:00.0", str(tmp_path))
        assert "corrected_after_init" not in kept and "initial affinity mask" in kept["reason"]
        assert sorted(os.sched_getaffinity(0)) == lo
    finally:
        os.sched_setaffinity(0, have)
    assert sharding.bind_to_gpu_numa_node(0, str(tmp_path / "nothing"))["applied"] is False
