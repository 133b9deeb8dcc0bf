#!/usr/bin/env python
"""Developer tool (GPU box): host-buffer entry points, one device vs the same device listed twice / four times
(zn_compress_multi / zn_decompress_multi) — prices the range split and the plane-major assembly on the host.  With one
physical GPU the ranges run one after the other (the device's lock), so this is overhead, not speed-up."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi   # noqa: E402

lib = _capi.lib()
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(gib * (1 << 30))
x = (torch.randn(n // 2, generator=torch.Generator().manual_seed(1)) * 0.02).to(torch.bfloat16).view(torch.uint8).numpy()
hdr = bytes(32)
C = 256 * 1024


def best(f, reps=3):
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); t = min(t, time.perf_counter() - t0)
    return t, r


ref = None
for devs in ([0], [0, 0], [0, 0, 0, 0]):
    if len(devs) == 1:
        tc, frame = best(lambda: lib.compress(hdr, x, 2, 1, 10, C, 0.95))
        td, back = best(lambda: lib.decompress(memoryview(frame)[32:], 2, 1, 10, C, n))
    else:
        tc, frame = best(lambda: lib.compress_multi(hdr, x, 2, 1, 10, C, 0.95, devs))
        td, back = best(lambda: lib.decompress_multi(memoryview(frame)[32:], 2, 1, 10, C, n, devs))
    if ref is None:
        ref = bytes(frame)
    same = bytes(frame) == ref and np.array_equal(np.frombuffer(back, dtype=np.uint8), x)
    print(f"{gib:g} GiB bf16, host buffers, devices {devs}: compress {tc * 1e3:7.1f} ms {n / tc / 1e9:6.1f} GB/s   decompress {td * 1e3:7.1f} ms {n / td / 1e9:6.1f} GB/s   identical={same}", flush=True)
