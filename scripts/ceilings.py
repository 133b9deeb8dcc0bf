#!/usr/bin/env python
"""Developer tool (GPU box): what the memory system delivers to plain copies, next to the decode kernel —
the practical ceiling the HBM roofline fraction should be read against.  4 GiB buffers, best of 5 x 10."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec   # noqa: E402


def best_of(fn, reps=5, inner=10):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(inner):
            fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / inner)
    return best


def main():
    n = 4 << 30
    lib = _capi.lib()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.empty(n // 2, dtype=torch.bfloat16, device="cuda")
    step = 1 << 27
    for off in range(0, x.numel(), step):
        x[off:off + step] = (torch.randn(step, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    t = best_of(lambda: out.copy_(flat))
    print(f"torch copy_ (D2D, 4 GiB)           {t * 1e3:7.3f} ms   read+write {2 * n / t / 1e9:6.0f} GB/s")
    t = best_of(lambda: out.fill_(7))
    print(f"torch fill_ (write only, 4 GiB)    {t * 1e3:7.3f} ms   write      {n / t / 1e9:6.0f} GB/s")
    v = flat.view(torch.int32)
    t = best_of(lambda: v.sum())
    print(f"torch sum (read only, 4 GiB int32) {t * 1e3:7.3f} ms   read       {n / t / 1e9:6.0f} GB/s")
    for label, thr in (("all planes raw (threshold 0)", 0.0), ("weights (exponent plane huff0)", 0.95)):
        body = codec.compress_device(lib, flat, 2, 1, 10, 256 * 1024, thr).clone()
        codec.decompress_device(lib, body, 2, 1, 10, 256 * 1024, n, out=out)
        assert torch.equal(out, flat)
        t = best_of(lambda: codec.decompress_device(lib, body, 2, 1, 10, 256 * 1024, n, out=out, check=False))
        c = body.numel() - 9 * 2 * (n // (256 * 1024))
        print(f"decode, {label:31s} {t * 1e3:7.3f} ms   payload in + tensor out {(c + n) / t / 1e9:6.0f} GB/s   ({n / t / 1e9:.0f} GB/s uncompressed)")
        del body


if __name__ == "__main__":
    main()
