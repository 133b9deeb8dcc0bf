#!/usr/bin/env python
"""Developer tool (GPU box): bf16 decode time by size and wide-kernel mode (zn_set_decode_wide: 1 auto, 0 never, 3 two waves per stream, 2 four)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec   # noqa: E402
lib = _capi.ZnLib(os.environ["ZN_LIB"]) if os.environ.get("ZN_LIB") else _capi.lib()      # (ZN_LIB: a library variant built for an A/B)
C = 256 * 1024
for mib in (64, 96, 128, 192, 256, 384, 512, 1024):
    n = mib << 20
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = (torch.randn(n // 2, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, 2, 1, 10, C, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    row = []
    for mode, nm in ((1, "auto"), (0, "fused"), (3, "wide2"), (2, "wide4")):
        lib.set_decode_wide(mode)
        for _ in range(12): codec.decompress_device(lib, body, 2, 1, 10, C, n, out=out, check=False)
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): codec.decompress_device(lib, body, 2, 1, 10, C, n, out=out, check=False)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
        assert torch.equal(out, flat)
        row.append(f"{nm} {best * 1e6:7.1f} us {n / best / 1e9:6.0f} GB/s")
    lib.set_decode_wide(1)
    print(f"{mib:5d} MiB ({n // C:5d} chunks): " + " | ".join(row), flush=True)
