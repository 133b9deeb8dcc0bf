#!/usr/bin/env python
"""Developer tool (GPU box): start and end time of every chunk workgroup of a fused decode launch (a library VARIANT: `git apply scripts/wg_times.patch`, build_extension(defines=["ZN_WG_TIMES"],
out="zipnn_amd/libzipnn_hip_ab_wgt.so"), `git checkout zipnn_amd/csrc` — the patch records s_memrealtime per workgroup and is not part of the product): how the rounds of workgroups look, how long the chip drains at the end."""
import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi, codec
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zipnn_amd", "libzipnn_hip_ab_wgt.so")
lib = _capi.ZnLib(so); raw = ctypes.CDLL(so)
C = 256 * 1024
for gib in (4.0, 1.0):
    n = int(gib * (1 << 30))
    x = torch.empty(n // 2, dtype=torch.bfloat16, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for off in range(0, x.numel(), 1 << 27):
        x[off:off + (1 << 27)] = (torch.randn(min(1 << 27, x.numel() - off), generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, 2, 1, 10, C, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(5): codec.decompress_device(lib, body, 2, 1, 10, C, n, out=out, check=False)
    torch.cuda.synchronize()
    nwg = (n // C + 3) // 4
    buf = (ctypes.c_ulonglong * (2 * 8192))()
    assert raw.zn_debug_wg_times(buf, 2 * 8192) == 0
    t = np.array(buf[:2 * nwg], dtype=np.float64).reshape(nwg, 2) / 100.0      # us
    t0 = t[:, 0].min(); st = t[:, 0] - t0; en = t[:, 1] - t0; dur = en - st
    T = en.max()
    print(f"{gib} GiB: {nwg} workgroups of 4 chunks; kernel span {T:.1f} us; workgroup duration mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f} us")
    print("  starts: first round", f"{np.sort(st)[:1024].max():.1f} us;", "round boundaries (start time of workgroup 1024, 2048, 3072):", [round(float(np.sort(st)[k]), 1) for k in (1024, 2048, 3072) if k < nwg])
    last = np.sort(en)
    print("  ends: percentiles of (kernel end - workgroup end) for the LAST 1024 workgroups:", {p: round(float(T - np.percentile(last[-1024:], p)), 1) for p in (0, 10, 25, 50, 75, 90, 100)})
    busy = dur.sum() / (1024 * T)
    print(f"  slot occupancy over the span: {busy:.3f}  (idle slot-time {1024 * T - dur.sum():.0f} us of {1024 * T:.0f})")
    del x, flat, body, out
