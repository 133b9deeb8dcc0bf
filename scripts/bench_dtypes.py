#!/usr/bin/env python
"""Developer tool: device-resident compress / decompress timing for the other dtypes of BASELINE.json
configs[2] (fp16, fp32 at 1 GiB; fp8 beside them).  Not the headline bench (bench.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi, codec


def make(kind, n_bytes, dev):
    g = torch.Generator(device=dev); g.manual_seed(1234)
    if kind == "bf16":
        return (torch.randn(n_bytes // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16), 2, 1, 10, 256 * 1024
    if kind == "fp16":
        return (torch.randn(n_bytes // 2, generator=g, device=dev) * 0.02).to(torch.float16), 2, 0, 10, 256 * 1024
    if kind == "fp32":
        return torch.randn(n_bytes // 4, generator=g, device=dev) * 0.02, 4, 1, 220, 256 * 1024
    if kind == "fp8":
        x = (torch.randn(n_bytes, generator=g, device=dev) * 0.02).to(torch.float8_e4m3fn)
        return x.view(torch.uint8), 1, 1, 10, 128 * 1024
    if kind == "rand16":   # uniform random bits: every plane is rejected by huff0 (ratio 1.0001) — the pure split / merge path
        return torch.randint(0, 65536, (n_bytes // 2,), generator=g, device=dev, dtype=torch.int32).to(torch.int16).view(torch.bfloat16), 2, 1, 10, 256 * 1024
    raise ValueError(kind)


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    lib = _capi.lib(); dev = torch.device("cuda:0")
    n = int(gib * (1 << 30))
    for kind in ("bf16", "fp16", "fp32", "fp8", "rand16"):
        x, P, rot, bm, chunk = make(kind, n, dev)
        flat = codec.flat_bytes(x)
        body = codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95).clone()
        out = torch.empty(n, dtype=torch.uint8, device=dev)
        codec.decompress_device(lib, body, P, rot, bm, chunk, n, out=out)
        ok = torch.equal(out, flat); fused = lib.last_fused_chunks()
        res = {}
        for name, fn in (("decompress", lambda: codec.decompress_device(lib, body, P, rot, bm, chunk, n, out=out, check=False)),
                         ("compress", lambda: codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95))):
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): fn()
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 5)
            res[name] = best
        print(f"{kind:5s} {gib} GiB  ratio {body.numel() / n:.4f}  fused {fused}/{n // chunk}  ok={ok}  "
              f"decompress {res['decompress'] * 1e3:.3f} ms {n / res['decompress'] / 1e9:.0f} GB/s   "
              f"compress {res['compress'] * 1e3:.3f} ms {n / res['compress'] / 1e9:.0f} GB/s", flush=True)
        del x, flat, body, out


if __name__ == "__main__":
    main()
