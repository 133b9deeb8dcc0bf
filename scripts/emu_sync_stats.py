#!/usr/bin/env python
"""Developer tool (CPU): build SIMT-emulator variants of the kernels with -D flags and count, per dtype, how the fused decoder's
tiles went — tiles, tiles in the looping form, fix-up iterations — from the emulated build's path counters.
    python scripts/emu_sync_stats.py name:-Dflag,-Dflag ...      (name "base" = no flags)"""
import ctypes, os, subprocess, sys, glob
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zipnn_amd._capi import ZnLib
import oracle_lib as O


def build(name, flags):
    so = f"/tmp/libzn_emu_{name}.so"
    srcs = []
    for f in sorted(glob.glob(os.path.join(ROOT, "zipnn_amd", "csrc", "*.hip"))):
        srcs += ["-x", "c++", f]
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-w", "-I" + os.path.join(ROOT, "tests", "simt"), "-DZN_SIMT_EMUL=1"] + flags + srcs + ["-o", so], check=True)
    return so


def data(kind, n):
    g = torch.Generator().manual_seed(11)
    if kind == "bf16": return (torch.randn(n // 2, generator=g) * 0.02).to(torch.bfloat16).view(torch.uint8).numpy().tobytes(), 2, 1, 10, 256 * 1024
    if kind == "fp16": return (torch.randn(n // 2, generator=g) * 0.02).to(torch.float16).view(torch.uint8).numpy().tobytes(), 2, 0, 10, 256 * 1024
    if kind == "fp32": return (torch.randn(n // 4, generator=g) * 0.02).view(torch.uint8).numpy().tobytes(), 4, 1, 220, 256 * 1024
    if kind == "fp8": return (torch.randn(n, generator=g) * 0.02).to(torch.float8_e4m3fn).view(torch.uint8).numpy().tobytes(), 1, 0, 10, 128 * 1024
    raise ValueError(kind)


def main():
    specs = sys.argv[1:] or ["base"]
    kinds = os.environ.get("KINDS", "bf16 fp16 fp8").split()
    n = int(os.environ.get("MIB", "2")) << 20
    for spec in specs:
        name, _, fl = spec.partition(":")
        so = build(name, [f for f in fl.split(",") if f])
        lib = ZnLib(so); raw = ctypes.CDLL(so)
        for kind in kinds:
            d, P, rot, bm, chunk = data(kind, n)
            frame = O.compress_frame(bytes(32), d, P, rot, bm, chunk, threads=4)
            c = (ctypes.c_ulonglong * 8)()
            raw.zn_debug_tile_counters(c, 1)
            out = lib.decompress(frame[32:], P, rot, bm, chunk, len(d))
            raw.zn_debug_tile_counters(c, 1)
            assert bytes(out) == d
            t = max(c[0], 1)
            print(f"{name:10s} {kind:5s} tiles {c[0]:6d}  looping-form {c[1] / t:6.3f}  fix-up iterations/tile {c[2] / t:6.3f}  lane-group tiles {c[3] / t:6.3f}", flush=True)


if __name__ == "__main__":
    main()
