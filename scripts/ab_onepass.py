#!/usr/bin/env python
"""Developer tool (GPU box): the one-pass encoder against the four-kernel encoder — same bodies, time per call, interleaved.
Usage: python scripts/ab_onepass.py [GiB of bf16, default 4]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from zipnn_amd import _capi, codec  # noqa: E402


def main():
    lib = _capi.lib()
    dev = torch.device("cuda:0")
    big = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    cases = [("bf16", int(big * (1 << 30)), torch.bfloat16, 2, 1, 10, 256 * 1024), ("bf16 2GiB", 2 << 30, torch.bfloat16, 2, 1, 10, 256 * 1024),
             ("bf16 1GiB", 1 << 30, torch.bfloat16, 2, 1, 10, 256 * 1024), ("bf16 256MiB", 256 << 20, torch.bfloat16, 2, 1, 10, 256 * 1024),
             ("bf16 100MiB+250KB", (100 << 20) + 250_000, torch.bfloat16, 2, 1, 10, 256 * 1024),
             ("fp16 4GiB", 4 << 30, torch.float16, 2, 0, 10, 256 * 1024), ("fp32 4GiB", 4 << 30, torch.float32, 4, 1, 220, 256 * 1024),
             ("fp8 2GiB", 2 << 30, torch.float8_e4m3fn, 1, 0, 10, 128 * 1024)]
    for name, n, dt, P, rot, bm, chunk in cases:
        es = torch.empty(0, dtype=dt).element_size()
        x = bench.make_tensor(n // es * es, dev, 1234, dt)
        flat = codec.flat_bytes(x)
        cap = lib.compress_bound(flat.numel(), P, chunk, 0)
        bodies = {}
        buf = {m: torch.empty(cap, dtype=torch.uint8, device=dev) for m in (1, 0)}
        best = {1: 1e9, 0: 1e9}
        for rnd in range(4):
            for m in (1, 0):
                lib.set_encode_onepass(2 if m else 0)
                b = codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95, body=buf[m])
                if rnd == 0:
                    bodies[m] = b; kern = lib.last_kernels()
                    print(f"   [{m}] {kern}")
                torch.cuda.synchronize(); t0 = time.perf_counter()
                reps = 5 if n >= (1 << 30) else 20
                for _ in range(reps):
                    codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95, body=buf[m])
                torch.cuda.synchronize(); best[m] = min(best[m], (time.perf_counter() - t0) / reps)
        lib.set_encode_onepass(1)
        same = bodies[1].numel() == bodies[0].numel() and torch.equal(bodies[1], bodies[0])
        out = codec.decompress_device(lib, bodies[1], P, rot, bm, chunk, flat.numel())
        rt = torch.equal(out, flat)
        algo = flat.numel() + bodies[1].numel()
        print(f"{name:18s} {flat.numel() >> 20:5d} MiB  one-pass {best[1] * 1e3:8.3f} ms ({algo / best[1] / 8e12:.3f} of 8 TB/s)   four-kernel {best[0] * 1e3:8.3f} ms ({algo / best[0] / 8e12:.3f})"
              f"   bodies equal: {same}   round trip: {rt}", flush=True)
        del x, flat, buf, bodies, out


if __name__ == "__main__":
    main()
