#!/usr/bin/env python
"""Developer tool: time fp32 / bf16 decode with the kernels of given commits (git archive of zipnn_amd/csrc + the
current Python host side when the C ABI allows; used to bisect a regression)."""
import os, subprocess, sys, tempfile, time, ctypes
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from zipnn_amd.build import hipcc_path
def main():
    n = 1 << 30
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x32 = (torch.randn(n // 4, generator=g, device="cuda") * 0.02).view(torch.uint8).reshape(-1)
    for tag in sys.argv[1:]:
        d = os.path.join(ROOT, "scripts", "_bisect", "src_" + tag)
        srcs = sorted(os.path.join(d, "zipnn_amd", "csrc", f) for f in os.listdir(os.path.join(d, "zipnn_amd", "csrc")) if f.endswith(".hip"))
        so = os.path.join(ROOT, "zipnn_amd", f"libzipnn_hip_ab_{tag}.so")
        r = subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so] + srcs, capture_output=True, text=True)
        if r.returncode: print(tag, "BUILD FAILED", r.stderr[-300:]); continue
        L = ctypes.CDLL(so)
        sz = ctypes.c_size_t; vp = ctypes.c_void_p; ci = ctypes.c_int
        L.zn_compress_bound.restype = sz; L.zn_compress_bound.argtypes = [sz, ci, sz, sz]
        L.zn_compress_dev.argtypes = [vp, sz, ci, ci, ci, sz, ctypes.c_float, vp, sz, ctypes.POINTER(sz), vp]
        L.zn_decompress_dev.argtypes = [vp, sz, ci, ci, ci, sz, sz, vp, vp, ci]
        cap = L.zn_compress_bound(n, 4, 262144, 0)
        body = torch.empty(cap, dtype=torch.uint8, device="cuda"); ln = sz(0)
        assert L.zn_compress_dev(x32.data_ptr(), n, 4, 1, 220, 262144, 0.95, body.data_ptr(), cap, ctypes.byref(ln), None) == 0
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        assert L.zn_decompress_dev(body.data_ptr(), ln.value, 4, 1, 220, 262144, n, out.data_ptr(), st, 1) == 0
        ok = torch.equal(out, x32); best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): L.zn_decompress_dev(body.data_ptr(), ln.value, 4, 1, 220, 262144, n, out.data_ptr(), st, 0)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        print(f"{tag:10s} fp32 ok={ok} decode {best * 1e3:.3f} ms {n / best / 1e9:.0f} GB/s", flush=True)
        os.remove(so)
if __name__ == "__main__":
    main()
