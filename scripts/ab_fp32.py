#!/usr/bin/env python
"""Developer tool: A/B decode-kernel build variants on fp32 (4 planes) and fp8 (1 plane), 1 GiB each."""
import os, subprocess, sys, time
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec
from zipnn_amd.build import hipcc_path, sources
VARIANTS = {"base": [], "dmax6": ["-DZN_F_DMAX=6"], "dmax6_c6": ["-DZN_F_DMAX=6", "-DZN_F_DCONST=6"], "dmax5": ["-DZN_F_DMAX=5"]}
def main():
    names = sys.argv[1:] or list(VARIANTS)
    n = 1 << 30
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x32 = codec.flat_bytes(torch.randn(n // 4, generator=g, device="cuda") * 0.02)
    x8 = codec.flat_bytes((torch.randn(n, generator=g, device="cuda") * 0.02).to(torch.float8_e4m3fn).view(torch.uint8))
    for name in names:
        so = os.path.join(ROOT, "zipnn_amd", f"libzipnn_hip_ab_{name}.so")
        r = subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so] + VARIANTS[name] + sources(), capture_output=True, text=True)
        if r.returncode: print(name, "BUILD FAILED", r.stderr[-300:]); continue
        lib = _capi.ZnLib(so)
        res = []
        for flat, P, bm, ch in ((x32, 4, 220, 262144), (x8, 1, 10, 131072)):
            body = codec.compress_device(lib, flat, P, 1, bm, ch, 0.95).clone()
            out = torch.empty(n, dtype=torch.uint8, device="cuda")
            codec.decompress_device(lib, body, P, 1, bm, ch, n, out=out)
            ok = torch.equal(out, flat); best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): codec.decompress_device(lib, body, P, 1, bm, ch, n, out=out, check=False)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
            res.append(f"P={P} ok={ok} {best * 1e3:.3f} ms {n / best / 1e9:.0f} GB/s")
        print(f"{name:8s} " + "   ".join(res), flush=True)
        lib.release_workspace(); os.remove(so)
if __name__ == "__main__":
    main()
