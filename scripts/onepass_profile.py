#!/usr/bin/env python
"""Developer tool (GPU box): where wave 0 of the one-pass encoder spends its shader-clock cycles, per chunk (libzipnn_hip_prof.so,
-DZN_PHASE_TIMERS; `--build-only` compiles it in the build container).  python scripts/onepass_profile.py [GiB] [bf16|fp16|fp32|fp8]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zipnn_amd import _capi, codec   # noqa: E402
from zipnn_amd.build import build_extension   # noqa: E402
from scripts.phase_profile import GEOM   # noqa: E402


def main():
    so = os.path.join(ROOT, "zipnn_amd", "libzipnn_hip_prof.so")
    if "--build-only" in sys.argv or not os.path.exists(so):
        build_extension(force=True, defines=["ZN_PHASE_TIMERS", "ZN_DEV_BUILD"], out=so)
    if "--build-only" in sys.argv:
        return
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    gib = float(args[0]) if args else 4.0
    kind = args[1] if len(args) > 1 else "bf16"
    lib = _capi.ZnLib(so); raw = ctypes.CDLL(so)
    dt, P, rot, bm, chunk = GEOM[kind]
    n = int(gib * (1 << 30)) // chunk * chunk
    torch.manual_seed(1)
    x = (torch.randn(n // torch.empty(0, dtype=dt).element_size(), device="cuda") * 0.02).to(dt)
    flat = codec.flat_bytes(x)
    acc = (ctypes.c_ulonglong * 64)()
    codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95)
    raw.zn_debug_phase_read_enc(acc, 1)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95); t1.record(); torch.cuda.synchronize()
    raw.zn_debug_phase_read_enc(acc, 1)
    ch = acc[17] or 1
    print(f"{kind} {gib} GiB: one-pass encoder, {ch} chunks, call {t0.elapsed_time(t1):.3f} ms (with timers); wave 0, cycles per chunk:")
    rows = ((10, "ticket + segment"), (11, "histograms (read 1)"), (12, "decisions, counts to LDS"), (2, "table: counts + rank sort"), (5, "table: tree + code lengths"),
            (9, "table: values + weights"), (6, "table: tree description"), (8, "table: stream sizes"), (13, "table: rest + barrier"), (14, "look-back"), (15, "emit (read 2, pack, write)"))
    tot = sum(acc[i] for i, _ in rows)
    for i, nm in rows:
        print(f"  {nm:30s} {acc[i] / ch:10.0f}  {100.0 * acc[i] / tot:5.1f} %")
    print(f"  total                          {tot / ch:10.0f}   = {tot / ch / 2400:.1f} us at 2.4 GHz")


if __name__ == "__main__":
    main()
