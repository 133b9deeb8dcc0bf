#!/usr/bin/env python
"""Developer tool: build libzipnn_hip_prof.so (-DZN_PHASE_TIMERS) and print where wave 0 of the
fused decode kernel spends its shader-clock cycles, per chunk.  Runs on the GPU box:
    python scripts/phase_profile.py [GiB] [bf16|fp16|fp32|fp8] [-Dflag ...]
(`--build-only` compiles the library and exits: done in the build container, the .so travels with the snapshot.)
"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec   # noqa: E402
from zipnn_amd.build import hipcc_path, sources   # noqa: E402

NAMES = {0: "metadata", 1: "tree description (serial)", 2: "LUT fill", 3: "flush rows", 4: "stage tile", 5: "sync run-in",
         6: "decode pass", 7: "fix-up passes", 8: "scan/shuffles", 9: "compaction / write pass"}


GEOM = {"bf16": (torch.bfloat16, 2, 1, 10, 256 * 1024), "fp16": (torch.float16, 2, 0, 10, 256 * 1024),
        "fp32": (torch.float32, 4, 1, 220, 256 * 1024), "fp8": (getattr(torch, "float8_e4m3fn", None), 1, 0, 10, 128 * 1024)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    gib = float(args[0]) if args else 1.0
    kind = args[1] if len(args) > 1 else "bf16"
    so = os.path.join(ROOT, "zipnn_amd", "libzipnn_hip_prof.so")
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    if extra or "--build-only" in sys.argv or not os.path.exists(so):
        subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DZN_PHASE_TIMERS", "-DZN_DEV_BUILD",
                        "-o", so] + extra + sources(), check=True)
    if "--build-only" in sys.argv:
        return
    lib = _capi.ZnLib(so)
    lib.set_decode_wide(0)            # (the timers are the fused kernel's)
    raw = ctypes.CDLL(so)
    dt, P, rot, bm, chunk = GEOM[kind]
    n = int(gib * (1 << 30)) // chunk * chunk
    torch.manual_seed(1)
    x = (torch.randn(n // torch.empty(0, dtype=dt).element_size(), device="cuda") * 0.02).to(dt)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    acc = (ctypes.c_ulonglong * 64)()
    codec.decompress_device(lib, body, P, rot, bm, chunk, n, out=out)
    raw.zn_debug_phase_read(acc, 1)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    codec.decompress_device(lib, body, P, rot, bm, chunk, n, out=out, check=False)
    t1.record(); torch.cuda.synchronize()
    raw.zn_debug_phase_read(acc, 1)
    assert torch.equal(out, flat)
    chunks = acc[19] or 1
    tot = sum(acc[i] for i in range(10))
    print(f"{kind} {gib} GiB, {chunks} chunks of {chunk >> 10} KiB, ratio {body.numel() / n:.4f}, decode {t0.elapsed_time(t1):.3f} ms (with timers)")
    for i in range(10):
        print(f"  {NAMES[i]:28s} {acc[i] / chunks:10.0f} cyc/chunk  {100.0 * acc[i] / tot:5.1f} %")
    print(f"  total                        {tot / chunks:10.0f} cyc/chunk")
    for i, nm in () if not any("ZN_PHASE_TIMERS_SUB" in a or a == "--sub" for a in sys.argv) else ((10, "readNCount"), (11, "FSE decode table"), (12, "FSE state chain"), (13, "stage+weights total"), (14, "weight statistics"), (15, "canonical order")):
        print(f"    tree description / {nm:22s} {acc[i] / chunks:10.0f} cyc/chunk")
    print(f"  (in flush) wait for fetched rows {acc[10] / chunks:10.0f} cyc/chunk")
    print(f"  wait for slowest wave at chunk start {acc[21] / chunks:10.0f} cyc/chunk")
    print(f"  tiles/chunk(wave0) {acc[18] / chunks:.2f}  fix-up iterations/tile {acc[16] / max(acc[18], 1):.3f}  "
          f"mismatching lanes/tile {acc[17] / max(acc[18], 1):.3f}  tiles in the looping form {acc[22] / max(acc[18], 1):.3f}")
    # encoder statistics kernel
    raw.zn_debug_phase_read_enc(acc, 1)
    codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95)
    raw.zn_debug_phase_read_enc(acc, 1)
    ch = acc[19] or 1
    jobs = acc[18] or 1
    print(f"encode: stats kernel (per chunk, {ch} chunks) / tables kernel (per table, {jobs} tables)")
    for i, nm, d in ((0, "stats: zero + 4 quarter histograms", ch), (1, "stats: decisions + counts out", ch), (2, "tables: counts + rank sort", jobs),
                     (5, "tables: tree + code lengths (serial)", jobs), (9, "tables: values + weights (parallel)", jobs), (6, "tables: tree description (wave, uniform chain)", jobs), (8, "tables: stream sizes", jobs), (3, "tables: descriptor out", jobs)):
        print(f"  {nm:36s} {acc[i] / d:10.0f} cyc")

if __name__ == "__main__":
    main()
