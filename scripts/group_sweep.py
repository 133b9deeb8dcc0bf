#!/usr/bin/env python
"""Developer tool (GPU box): decode time by chunks-per-workgroup (zn_set_decode_group) and tensor size."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec   # noqa: E402
lib = _capi.ZnLib(os.environ["ZN_LIB"]) if os.environ.get("ZN_LIB") else _capi.lib()      # (ZN_LIB: a library variant built for an A/B)
C = 256 * 1024
for mib in (tuple(int(a) for a in sys.argv[1:]) or (64, 128, 256, 512, 1024, 2048, 4096)):
    n = mib << 20
    x = torch.empty(n // 2, dtype=torch.bfloat16, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for off in range(0, x.numel(), 1 << 27):
        x[off:off + (1 << 27)] = (torch.randn(min(1 << 27, x.numel() - off), generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, 2, 1, 10, C, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    row = []
    for grp in ("", "1", "2", "3", "4"):
        lib.set_decode_group(int(grp) if grp else 0)
        for _ in range(12): codec.decompress_device(lib, body, 2, 1, 10, C, n, out=out, check=False)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): codec.decompress_device(lib, body, 2, 1, 10, C, n, out=out, check=False)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        assert torch.equal(out, flat)
        row.append(f"{'auto' if not grp else 'g' + grp} {best * 1e3:7.3f} ms {n / best / 1e9:6.0f} GB/s")
    print(f"{mib:5d} MiB: " + " | ".join(row), flush=True)
    del x, flat, body, out
    torch.cuda.empty_cache()
