#!/usr/bin/env python
"""Developer tool (GPU box, under rocprofv3 --kernel-trace): compress one tensor of <MiB> repeatedly, so that the
per-kernel durations show whether the emit kernel's re-read of a slab that the stats kernel has just read is served
by the Infinity Cache.  Usage: python scripts/slab_probe.py <MiB> [reps]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec   # noqa: E402

mib = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _capi.lib()
n = mib << 20
g = torch.Generator(device="cuda"); g.manual_seed(5)
x = (torch.randn(n // 2, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
flat = codec.flat_bytes(x)
for _ in range(reps):
    body = codec.compress_device(lib, flat, 2, 1, 10, 256 * 1024, 0.95)
torch.cuda.synchronize()
print(mib, "MiB ratio", body.numel() / n)
