#!/usr/bin/env python
"""Developer tool (GPU box): the fused delta (XOR) path against `torch.bitwise_xor` + the plain path.
Usage: python scripts/bench_delta.py [GiB [lib.so]]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec

C = 256 * 1024


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    libpath = sys.argv[2] if len(sys.argv) > 2 else None
    n = int(gib * (1 << 30)) // C * C
    dev = torch.device("cuda:0")
    lib = _capi.ZnLib(libpath) if libpath else _capi.lib()
    g = torch.Generator(device=dev).manual_seed(5)
    base = (torch.randn(n // 2, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    x = base.clone()
    idx = torch.randint(0, x.numel(), (x.numel() // 50,), device=dev, generator=g)
    x[idx] = (x[idx].float() * 1.01).to(torch.bfloat16)
    fx, fb = codec.flat_bytes(x), codec.flat_bytes(base)
    body = codec.compress_device(lib, fx, 2, 1, 10, C, 0.95, delta=fb).clone()
    plain = codec.compress_device(lib, fx, 2, 1, 10, C, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    codec.decompress_device(lib, body, 2, 1, 10, C, n, out=out, delta=fb)
    assert torch.equal(out, fx)
    print(f"{gib:g} GiB bf16, 2 % of the elements changed by 1 %: delta ratio {body.numel() / n:.4f}, plain ratio {plain.numel() / n:.4f}")
    rows = [
        ("decode  plain                  ", lambda: codec.decompress_device(lib, plain, 2, 1, 10, C, n, out=out, check=False)),
        ("decode  delta fused            ", lambda: codec.decompress_device(lib, body, 2, 1, 10, C, n, out=out, check=False, delta=fb)),
        ("decode  delta + torch xor pass ", lambda: (codec.decompress_device(lib, body, 2, 1, 10, C, n, out=out, check=False), torch.bitwise_xor(out, fb, out=out))),
        ("compress plain                 ", lambda: codec.compress_device(lib, fx, 2, 1, 10, C, 0.95)),
        ("compress delta fused           ", lambda: codec.compress_device(lib, fx, 2, 1, 10, C, 0.95, delta=fb)),
        ("compress torch xor pass + plain", lambda: codec.compress_device(lib, torch.bitwise_xor(fx, fb), 2, 1, 10, C, 0.95)),
    ]
    for name, fn in rows:
        t = timed(fn)
        print(f"  {name}  {t * 1e3:7.3f} ms   {n / t / 1e9:7.0f} GB/s")


if __name__ == "__main__":
    main()
