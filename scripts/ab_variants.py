#!/usr/bin/env python
"""Developer tool: A/B build variants of the library on the GPU box with one gpurun call.

    python scripts/ab_variants.py --build        (in the build container: compiles every variant into
                                                  zipnn_amd/libzipnn_hip_ab_<name>.so — git-ignored, but
                                                  shipped to the GPU box with the snapshot)
    python scripts/ab_variants.py [name ...]     (on the GPU box: parity check of every variant, then interleaved
                                                  timing A B C ... A B C ... so that clock drift hits all of them)

A variant is a set of -D flags over the current sources, or the sources of an older commit (`git archive`).
The ZN_F_ABLATE variants repeat one phase of the fused decode kernel with unchanged results: the time they add
is the price of that phase on the device, undisturbed by timers.
"""
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "zipnn_amd")

# name -> (commit or None, [flags])
VARIANTS = {
    "new": (None, []),                              # the sources in the tree
    "c1": ("HEAD", []),                             # the last commit
    # the kernels of earlier rounds (their -D switches are gone from the tree: round 6 made the settled ones constants; a variant of an old switch is built from
    # the commit that still had it, e.g. ("741f196", ["-DZN_F_RB2=4"]))
    "r01": ("fb215d2", []), "r02": ("8959d1d", []), "r03": ("3c0f9d7", []), "r04a": ("5b359c5", []), "r05": ("741f196", []), "r06g": ("61a3fa5", []),
    # compiler scheduling options (same sources)
    "ilp": (None, ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
    "iter": (None, ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]),
    "maxocc": (None, ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]),
    "nopost": (None, ["-mllvm", "-enable-post-misched=0"]),
    "o2": (None, ["-O2"]),
    # round 6: profiles/r06_glds_stream_tile.patch (the next stream tile by LDS-DMA) is applied to a scratch copy of the tree by hand; the two libraries it was
    # measured with were "new" with and without it
}


def so_path(name):
    return os.path.join(PKG, f"libzipnn_hip_ab_{name}.so")


def build_one(name):
    from zipnn_amd.build import hipcc_path
    commit, flags = VARIANTS[name]
    src_dir = os.path.join(PKG, "csrc")
    tmp = None
    if commit:
        tmp = tempfile.mkdtemp(prefix="zn_ab_")
        subprocess.run(f"git -C {ROOT} archive {commit} zipnn_amd/csrc include | tar -x -C {tmp}", shell=True, check=True)
        src_dir = os.path.join(tmp, "zipnn_amd", "csrc")
    srcs = sorted(os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith(".hip"))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DZN_DEV_BUILD", "-o", so_path(name)] + flags + srcs
    r = subprocess.run(cmd, capture_output=True, text=True)
    return name, r.returncode, r.stderr[-400:]


def build(names):
    with ThreadPoolExecutor(max_workers=4) as ex:
        for name, rc, err in ex.map(build_one, names):
            print(name, "ok" if rc == 0 else "BUILD FAILED " + err, flush=True)


def load(path):
    L = ctypes.CDLL(path)
    sz, vp, ci = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
    L.zn_compress_bound.restype = sz; L.zn_compress_bound.argtypes = [sz, ci, sz, sz]
    L.zn_compress_dev.argtypes = [vp, sz, ci, ci, ci, sz, ctypes.c_float, vp, sz, ctypes.POINTER(sz), vp]
    L.zn_decompress_dev.argtypes = [vp, sz, ci, ci, ci, sz, sz, vp, vp, ci]
    return L


ALLD = None


def run(names):
    import torch
    libs = [(n, load(so_path(n))) for n in names if os.path.exists(so_path(n))]
    print("variants:", [n for n, _ in libs], flush=True)
    C = 262144
    f8 = getattr(torch, "float8_e4m3fn", None)
    cases = [("bf16 4GiB", 4 << 30, 2, 1, 10, torch.bfloat16, None),
             ("fp16 1GiB", 1 << 30, 2, 0, 10, torch.float16, ALLD),
             ("fp32 1GiB", 1 << 30, 4, 1, 220, torch.float32, ALLD),
             ("bf16 256MiB", 256 << 20, 2, 1, 10, torch.bfloat16, ALLD)]
    if f8 is not None:
        cases.append(("fp8 1GiB", 1 << 30, 1, 0, 10, f8, ALLD))
    st = torch.cuda.current_stream().cuda_stream
    results = {}
    for name, n, P, rot, bm, dt, only in cases:
        use = list(libs)
        if not use:
            continue
        es = torch.empty(0, dtype=dt).element_size()
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        x = torch.empty(n // es, dtype=dt, device="cuda")
        step = 1 << 27
        for off in range(0, x.numel(), step):
            x[off:off + step] = (torch.randn(min(step, x.numel() - off), generator=g, device="cuda") * 0.02).to(dt)
        flat = x.view(torch.uint8).reshape(-1)
        chunk = C if P > 1 else C // 2
        L0 = use[0][1]
        cap = L0.zn_compress_bound(n, P, chunk, 0)
        body = torch.empty(cap, dtype=torch.uint8, device="cuda"); ln = ctypes.c_size_t(0)
        assert L0.zn_compress_dev(flat.data_ptr(), n, P, rot, bm, chunk, 0.95, body.data_ptr(), cap, ctypes.byref(ln), None) == 0
        out = torch.empty(n, dtype=torch.uint8, device="cuda")
        ok = {}
        for k, L in use:
            out.zero_()
            rc = L.zn_decompress_dev(body.data_ptr(), ln.value, P, rot, bm, chunk, n, out.data_ptr(), st, 1)
            ok[k] = (rc == 0) and bool(torch.equal(out, flat))
        best = {k: 1e9 for k, _ in use}
        for rnd in range(5):
            for k, L in use:
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    L.zn_decompress_dev(body.data_ptr(), ln.value, P, rot, bm, chunk, n, out.data_ptr(), st, 0)
                torch.cuda.synchronize(); best[k] = min(best[k], (time.perf_counter() - t0) / 10)
        # compress: same frame body as the first variant's, then interleaved timing (one length read-back per call)
        bestc = {k: 1e9 for k, _ in use}; okc = {}
        body2 = torch.empty(cap, dtype=torch.uint8, device="cuda"); ln2 = ctypes.c_size_t(0)
        for k, L in use:
            body2.zero_()
            rc = L.zn_compress_dev(flat.data_ptr(), n, P, rot, bm, chunk, 0.95, body2.data_ptr(), cap, ctypes.byref(ln2), st)
            okc[k] = (rc == 0) and ln2.value == ln.value and bool(torch.equal(body2[:ln.value], body[:ln.value]))
        for rnd in range(4):
            for k, L in use:
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    L.zn_compress_dev(flat.data_ptr(), n, P, rot, bm, chunk, 0.95, body2.data_ptr(), cap, ctypes.byref(ln2), st)
                torch.cuda.synchronize(); bestc[k] = min(bestc[k], (time.perf_counter() - t0) / 5)
        for k, _ in use:
            print(f"{name:12s} {k:10s} exact={ok[k]}  decode {best[k] * 1e3:7.3f} ms {n / best[k] / 1e9:7.0f} GB/s  ratio {ln.value / n:.4f}"
                  f"   compress same={okc[k]} {bestc[k] * 1e3:7.3f} ms {n / bestc[k] / 1e9:6.0f} GB/s", flush=True)
            results[f"{name}/{k}"] = {"ms": best[k] * 1e3, "GBps": n / best[k] / 1e9, "exact": ok[k],
                                      "compress_ms": bestc[k] * 1e3, "compress_same": okc[k]}
        del body2
        del x, flat, body, out
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ab_variants.json"), "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--build":
        build(args[1:] or list(VARIANTS))
    else:
        run(args or list(VARIANTS))
