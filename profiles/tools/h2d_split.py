#!/usr/bin/env python3
"""How long does a 2.1 MB / 4.2 MB / 8.4 MB host-to-device copy take on this box as ONE copy and split over 2 / 4 HIP
streams (pinned source; would several DMA engines in parallel shorten the one-frame call's critical path?), and what a
GPU kernel reading the pinned buffer in place (zero-copy) reaches.  GPU-timeline times (events), median of 200."""
import time

import numpy as np
import torch

dev = torch.device("cuda:0")


def timed(fn, n=200):
    ts = []
    for _ in range(n + 20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts[20:]))


for mb in (2.097152, 4.194304, 8.388608):
    n = int(mb * 1e6) // 4
    src = torch.empty(n, dtype=torch.float32).pin_memory()
    src.uniform_()
    pag = torch.empty(n, dtype=torch.float32)
    pag.copy_(src)
    dst = torch.empty(n, dtype=torch.float32, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    out = {}
    out["pinned, 1 copy"] = timed(lambda: dst.copy_(src, non_blocking=True))
    out["pageable, 1 copy"] = timed(lambda: dst.copy_(pag, non_blocking=True))
    for k in (2, 4):
        step = n // k

        def split(k=k, step=step):
            for i in range(k):
                with torch.cuda.stream(streams[i]):
                    dst[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step], non_blocking=True)
        out[f"pinned, {k} streams"] = timed(split)
    # zero-copy: a kernel reads the pinned host buffer directly (torch: a device-side copy kernel from mapped memory)
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        dptr = ctypes.c_void_p()
        rc = hip.hipHostGetDevicePointer(ctypes.byref(dptr), ctypes.c_void_p(src.data_ptr()), 0)
        if rc == 0:
            from torch.utils import dlpack  # noqa: F401
            # view the mapped pointer as a CUDA tensor through __cuda_array_interface__
            class _M:
                pass
            m = _M()
            m.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (dptr.value, False), "version": 2}
            mapped = torch.as_tensor(m, device=dev)
            out["zero-copy kernel read"] = timed(lambda: dst.copy_(mapped))
    except Exception as e:  # noqa: BLE001
        out["zero-copy kernel read"] = f"n/a ({e})"
    print(f"{mb:.1f} MB: " + "; ".join(f"{k} {v:.1f} us ({mb * 1e6 / (v * 1e-6) / 1e9:.1f} GB/s)" if isinstance(v, float) else f"{k} {v}"
                                     for k, v in out.items()))
