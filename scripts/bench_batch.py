#!/usr/bin/env python
"""Developer bench for SURVEY §8(f1): decode a model-shaped set of tensors per tensor vs in one batched call.
Synthetic weights (N(0,0.02)), Llama-3-8B layer shapes in bf16 and GPT-2 shapes in fp32; device-resident."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi, codec

C = 256 * 1024


def shapes_llama(layers):
    s = []
    for _ in range(layers):
        s += [(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336), (4096,), (4096,)]
    return s


def shapes_gpt2():
    s = [(50257, 768), (1024, 768)]
    for _ in range(12):
        s += [(768,), (768,), (768, 2304), (2304,), (768, 768), (768,), (768,), (768,), (768, 3072), (3072,), (3072, 768), (768,)]
    return s + [(768,), (768,)]


def run(name, shapes, dtype, P, rot, bm):
    lib = _capi.lib(); dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(3)
    items, flats = [], []
    for sh in shapes:
        n = 1
        for d in sh: n *= d
        x = (torch.randn(n, generator=g, device=dev) * 0.02).to(dtype)
        flat = codec.flat_bytes(x)
        body = codec.compress_device(lib, flat, P, rot, bm, C, 0.95).clone()
        items.append((body, P, rot, bm, C, flat.numel())); flats.append(flat)
    total = sum(f.numel() for f in flats)
    outs = codec.decompress_device_batch(lib, items)
    ok = all(torch.equal(o, f) for o, f in zip(outs, flats))
    res = {}
    def per_tensor():
        for (b, p, r, m, c, n), o in zip(items, outs):
            codec.decompress_device(lib, b, p, r, m, c, n, out=o, check=False)
    def batched():
        codec.decompress_device_batch(lib, items, check=False)
    for nm, fn in (("per-tensor", per_tensor), ("batched", batched)):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): fn()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 3)
        res[nm] = best
    print(f"{name}: {len(shapes)} tensors, {total / 2**30:.2f} GiB, ok={ok}  per-tensor {res['per-tensor'] * 1e3:.2f} ms "
          f"({total / res['per-tensor'] / 1e9:.0f} GB/s)   batched {res['batched'] * 1e3:.2f} ms ({total / res['batched'] / 1e9:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    run("llama-3-8B layers x8, bf16", shapes_llama(8), torch.bfloat16, 2, 1, 10)
    run("gpt2 (124M), fp32", shapes_gpt2(), torch.float32, 4, 1, 220)


def run_compress(name, shapes, dtype, P, rot, bm):
    lib = _capi.lib(); dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(3)
    flats = []
    for sh in shapes:
        n = 1
        for d in sh: n *= d
        flats.append(codec.flat_bytes((torch.randn(n, generator=g, device=dev) * 0.02).to(dtype)))
    total = sum(f.numel() for f in flats)
    res = {}
    def loop():
        return [codec.compress_device(lib, f, P, rot, bm, C, 0.95) for f in flats]
    def batch():
        return codec.compress_device_batch(lib, [(f, P, rot, bm, C, 0.95) for f in flats])
    a, b = loop(), batch()
    ok = all(torch.equal(x, y) for x, y in zip(a, b))
    for nm, fn in (("per-tensor", loop), ("batched", batch)):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res[nm] = best
    print(f"compress {name}: {len(shapes)} tensors, {total / 2**30:.2f} GiB, same bytes={ok}  per-tensor {res['per-tensor'] * 1e3:.2f} ms "
          f"({total / res['per-tensor'] / 1e9:.0f} GB/s)   batched {res['batched'] * 1e3:.2f} ms ({total / res['batched'] / 1e9:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    run_compress("llama-3-8B layers x8, bf16", shapes_llama(8), torch.bfloat16, 2, 1, 10)
    run_compress("gpt2 (124M), fp32", shapes_gpt2(), torch.float32, 4, 1, 220)
