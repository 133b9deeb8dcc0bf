#!/usr/bin/env python
"""Developer tool (GPU box): decode of tensors with / without a partial last chunk, library variants interleaved.
Usage: python scripts/ab_ragged_decode.py libA.so libB.so ..."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd._capi import ZnLib
from zipnn_amd import codec


def main():
    libs = [(os.path.basename(p), ZnLib(p)) for p in sys.argv[1:]]
    dev = torch.device("cuda:0")
    for n in ((4 << 30), (4 << 30) + 200 * 1024, (3 << 29), (3 << 29) + 200 * 1024, (1 << 30), (1 << 30) + 200 * 1024, (100 << 20), (100 << 20) + 250 * 1024 + 2, (65 << 20) + 250 * 1024, (8 << 20), (8 << 20) + 3000, (8 << 20) + 200 * 1024):
        g = torch.Generator(device=dev); g.manual_seed(1)
        x = (torch.randn(n // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16)
        flat = codec.flat_bytes(x)
        body = codec.compress_device(libs[0][1], flat, 2, 1, 10, 262144, 0.95).clone()
        out = torch.empty(flat.numel(), dtype=torch.uint8, device=dev)
        best = {k: 1e9 for k, _ in libs}; kern = {}
        for k, L in libs:
            out.zero_(); codec.decompress_device(L, body, 2, 1, 10, 262144, flat.numel(), out=out)
            assert torch.equal(out, flat), k
            kern[k] = L.last_kernels()
        for _ in range(5):
            for k, L in libs:
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20): codec.decompress_device(L, body, 2, 1, 10, 262144, flat.numel(), out=out, check=False)
                torch.cuda.synchronize(); best[k] = min(best[k], (time.perf_counter() - t0) / 20)
        print(f"{flat.numel():12d} bytes  " + "   ".join(f"{k}: {best[k] * 1e6:7.1f} us [{kern[k]}]" for k, _ in libs), flush=True)


if __name__ == "__main__":
    main()
