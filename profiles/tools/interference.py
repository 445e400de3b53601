#!/usr/bin/env python3
"""What slows the feature kernels down beside a projection: memory traffic or the projection's wavefronts?
One context runs its step (projection, classification, feature kernels; shared-GPU mode as in the two-context schedule)
(a) alone, (b) beside a pure device-to-device copy stream on another HIP stream (HBM traffic, next to no ALU work, few
registers), (c) beside a register-only spin kernel of torch (ALU work, no memory traffic).  Per-kernel hipEvent times."""
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import resident  # noqa: E402

w = resident.build(B=1024)
est, b = w["est"], w["batch"]
est.setSharedGpu(True)
dev = torch.device("cuda:0")
NAMES = {0: "k_project_scatter", 5: "k_classify", 1: "k_feature_fused", 3: "k_feature_wave"}


def measure(label, steps=6):
    for _ in range(2):
        est.runBatch(b)
    est.synchronize()
    est.timingEnable(True)
    est.timingReset()
    t0 = time.perf_counter()
    for _ in range(steps):
        est.runBatch(b)
    est.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{label:34s} step {dt * 1e3:7.3f} ms  " + "  ".join(f"{n} {est.kernelTimeMs(k)[0] * 1e3:6.1f}" for k, n in NAMES.items()), flush=True)
    est.timingEnable(False)


measure("alone")
side = torch.cuda.Stream(device=dev)
src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)  # 1 GiB
dst = torch.empty_like(src)
stop = False


def background(kind):
    with torch.cuda.stream(side):
        while not stop:
            if kind == "copy":
                for _ in range(8):
                    dst.copy_(src, non_blocking=True)
            else:
                x = src[: 64 * 1024 * 1024]
                for _ in range(8):
                    torch.sin_(x)  # (memory-light relative to its ALU work? no: still streams; see "copy" for the contrast)
            side.synchronize()


for kind in ("copy",):
    stop = False
    th = threading.Thread(target=background, args=(kind,))
    th.start()
    time.sleep(0.2)
    t0 = time.perf_counter()
    measure(f"beside a {kind} stream")
    stop = True
    th.join()
    torch.cuda.synchronize()
# rate of the copy stream alone
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(side):
    for _ in range(8):
        dst.copy_(src, non_blocking=True)
side.synchronize()
dt = time.perf_counter() - t0
print(f"copy stream alone: {8 * 2 * src.numel() * 4 / dt / 1e12:.2f} TB/s (read + write)")
