#!/usr/bin/env python
"""Developer tool (GPU box): time the decode of prebuilt library variants against each other, interleaved
(A B A B ...) so that clock drift hits both.  Usage: python scripts/ab_libs.py libA.so libB.so ...
(the variants are built here, in the container, e.g. from `git archive <commit> zipnn_amd/csrc`)."""
import ctypes, os, sys, time
import torch


def load(path):
    L = ctypes.CDLL(path)
    sz, vp, ci = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
    L.zn_compress_bound.restype = sz; L.zn_compress_bound.argtypes = [sz, ci, sz, sz]
    L.zn_compress_dev.argtypes = [vp, sz, ci, ci, ci, sz, ctypes.c_float, vp, sz, ctypes.POINTER(sz), vp]
    L.zn_decompress_dev.argtypes = [vp, sz, ci, ci, ci, sz, sz, vp, vp, ci]
    return L


def main():
    paths = sys.argv[1:]
    libs = [(os.path.basename(p), load(p)) for p in paths]
    C0 = 262144
    cases = [("bf16 4GiB", 4 << 30, 2, 1, 10, torch.bfloat16), ("fp32 1GiB", 1 << 30, 4, 1, 220, torch.float32), ("fp16 1GiB", 1 << 30, 2, 0, 10, torch.float16), ("fp8 1GiB", 1 << 30, 1, 0, 10, torch.float8_e4m3fn)]
    st = torch.cuda.current_stream().cuda_stream
    for name, n, P, rot, bm, dt in cases:
        C = C0 if P > 1 else 131072
        es = torch.empty(0, dtype=dt).element_size()
        x = torch.empty(n // es, dtype=dt, device="cuda")
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        step = 1 << 27
        for off in range(0, x.numel(), step):
            x[off:off + step] = (torch.randn(min(step, x.numel() - off), generator=g, device="cuda") * 0.02).to(dt)
        flat = x.view(torch.uint8).reshape(-1)
        L0 = libs[0][1]
        cap = L0.zn_compress_bound(n, P, C, 0)
        body = torch.empty(cap, dtype=torch.uint8, device="cuda"); ln = ctypes.c_size_t(0)
        assert L0.zn_compress_dev(flat.data_ptr(), n, P, rot, bm, C, 0.95, body.data_ptr(), cap, ctypes.byref(ln), None) == 0
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        best = {k: 1e9 for k, _ in libs}; bestc = {k: 1e9 for k, _ in libs}
        for k, L in libs:
            assert L.zn_decompress_dev(body.data_ptr(), ln.value, P, rot, bm, C, n, out.data_ptr(), st, 1) == 0
            ok_ = torch.equal(out, flat); print("   roundtrip", k, ok_)
        body2 = torch.empty(cap, dtype=torch.uint8, device="cuda"); ln2 = ctypes.c_size_t(0)
        for rnd in range(4):
            for k, L in libs:
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    L.zn_decompress_dev(body.data_ptr(), ln.value, P, rot, bm, C, n, out.data_ptr(), st, 0)
                torch.cuda.synchronize(); best[k] = min(best[k], (time.perf_counter() - t0) / 10)
                t0 = time.perf_counter()
                for _ in range(4):
                    L.zn_compress_dev(flat.data_ptr(), n, P, rot, bm, C, 0.95, body2.data_ptr(), cap, ctypes.byref(ln2), st)
                torch.cuda.synchronize(); bestc[k] = min(bestc[k], (time.perf_counter() - t0) / 4)
        for k, _ in libs:
            print(f"{name:10s} {k:34s} decode {best[k] * 1e3:.3f} ms {n / best[k] / 1e9:6.0f} GB/s   compress {bestc[k] * 1e3:.3f} ms {n / bestc[k] / 1e9:6.0f} GB/s", flush=True)
        del x, flat, body, out, body2


if __name__ == "__main__":
    main()
