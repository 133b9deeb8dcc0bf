#!/usr/bin/env python
"""Developer tool: A/B decode-kernel build variants on the GPU box.  Each variant is a set of -D flags;
the library is rebuilt into a temp .so and the 4 GiB bf16 decode is timed (best of 3 x 10 launches)."""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec
from zipnn_amd.build import hipcc_path, sources

VARIANTS = {
    "base": [],
    "ent_stats": ["-DZN_E_NTLOAD_STATS"],
    "ent_emit": ["-DZN_E_NTLOAD_EMIT"],
    "ent_both": ["-DZN_E_NTLOAD_STATS", "-DZN_E_NTLOAD_EMIT"],
    "dmax6": ["-DZN_F_DMAX=6"],
    "d20": ["-DZN_F_DELTA0=20"],
    "d24": ["-DZN_F_DELTA0=24"],
    "rb6": ["-DZN_F_RBMAX=6"],
    "rb4": ["-DZN_F_RBMAX=4"],
    "w3": ["-DZN_F_WAVES_PER_SIMD=3"],
    "d16": ["-DZN_F_DELTA0=16"],
    "d32": ["-DZN_F_DELTA0=32"],
    "r3k": ["-DZN_F_RING_BYTES=3072u", "-DZN_F_DCONST=3"],
    "r5k_w3": ["-DZN_F_RING_BYTES=5120u", "-DZN_F_DMAX=6", "-DZN_F_DCONST=6", "-DZN_F_WAVES_PER_SIMD=3"],
}



def main():
    names = sys.argv[1:] or list(VARIANTS)
    n = 4 << 30
    torch.manual_seed(1)
    x = torch.empty(n // 2, dtype=torch.bfloat16, device="cuda")
    for off in range(0, x.numel(), 1 << 27):
        x[off:off + (1 << 27)] = (torch.randn(1 << 27, device="cuda") * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = None
    for name in names:
        so = os.path.join(ROOT, "zipnn_amd", f"libzipnn_hip_ab_{name}.so")
        r = subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so] + VARIANTS[name] + sources(),
                           capture_output=True, text=True)
        if r.returncode:
            print(name, "BUILD FAILED", r.stderr[-400:]); continue
        lib = _capi.ZnLib(so)
        if body is None:
            body = codec.compress_device(lib, flat, 2, 1, 10, 256 * 1024, 0.95).clone()
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        codec.decompress_device(lib, body, 2, 1, 10, 256 * 1024, n, out=out)
        ok = torch.equal(out, flat)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                codec.decompress_device(lib, body, 2, 1, 10, 256 * 1024, n, out=out, check=False)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        fused = lib.last_fused_chunks()
        cb = codec.compress_device(lib, flat, 2, 1, 10, 256 * 1024, 0.95)
        ok = ok and torch.equal(cb, body)
        bestc = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                codec.compress_device(lib, flat, 2, 1, 10, 256 * 1024, 0.95)
            torch.cuda.synchronize(); bestc = min(bestc, (time.perf_counter() - t0) / 5)
        print(f"{name:14s} ok={ok} fused={fused} decode {best * 1e3:.3f} ms  {n / best / 1e9:.0f} GB/s   compress {bestc * 1e3:.3f} ms  {n / bestc / 1e9:.0f} GB/s", flush=True)
        lib.release_workspace()
        os.remove(so)

if __name__ == "__main__":
    main()
