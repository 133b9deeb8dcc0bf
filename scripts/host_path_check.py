#!/usr/bin/env python
"""PCIe-inclusive rate of the host-buffer entry points — zn_compress / zn_decompress through raw ctypes, i.e. exactly what the INTEGRATION stub
(tests/ref_binding/zipnn_core.py) and ZipNN().compress(bytes) reach — with FRESH caller buffers (allocated inside the timed region, released
inside it too: what a caller that does not recycle buffers pays) and WARM ones (recycled), one-shot and pipelined.
    python scripts/host_path_check.py [GiB]          (ZIPNN_AMD_HOST_DIRECT=0 for the staged-only path of rounds 1-5; ZN_HOST_PIPE_TRACE=1 for the phases)"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi
lib = _capi.lib(); L = lib._L
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(gib * (1 << 30)) // 262144 * 262144
x = (torch.randn(n // 2, device="cuda") * 0.02).to(torch.bfloat16).cpu().view(torch.uint8).numpy()
hdr = np.zeros(32, dtype=np.uint8)
cap = L.zn_compress_bound(n, 2, 262144, 32)
sz = ctypes.c_size_t(0)


def compress_into(out):
    rc = L.zn_compress(hdr.ctypes.data, 32, x.ctypes.data, n, 2, 1, 10, 262144, ctypes.c_float(0.95), 0, out.ctypes.data, cap, ctypes.byref(sz))
    assert rc == 0, rc
    return sz.value


def decompress_into(frame, flen, out):
    rc = L.zn_decompress(frame.ctypes.data + 32, flen - 32, 2, 1, 10, 262144, n, 0, out.ctypes.data)
    assert rc == 0, rc


frame = np.empty(cap, dtype=np.uint8); flen = compress_into(frame)
back = np.empty(n, dtype=np.uint8); decompress_into(frame, flen, back)
assert np.array_equal(back, x)
res = {}
modes = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--direct=")] or [4, 7]
for mode, slices in [(m, sl) for m in modes for sl in ((0, 1) if m == 4 else (0,))]:
    lib.set_host_direct(mode)
    lib.set_host_slices(slices)
    tag = ("staged" if mode == 4 else f"direct={mode}") + (", one shot" if slices == 1 else ", automatic slices" if slices == 0 else f", {slices} slices")
    for name in ("compress", "decompress"):
        warm = fresh = fresh_kept = 1e9
        for _ in range(3):
            if name == "compress":
                t0 = time.perf_counter(); compress_into(frame); warm = min(warm, time.perf_counter() - t0)
                t0 = time.perf_counter(); o = np.empty(cap, dtype=np.uint8); compress_into(o); t1 = time.perf_counter(); del o; t2 = time.perf_counter()
            else:
                t0 = time.perf_counter(); decompress_into(frame, flen, back); warm = min(warm, time.perf_counter() - t0)
                t0 = time.perf_counter(); o = np.empty(n, dtype=np.uint8); decompress_into(frame, flen, o); t1 = time.perf_counter(); del o; t2 = time.perf_counter()
            fresh_kept = min(fresh_kept, t1 - t0); fresh = min(fresh, t2 - t0)
        res[(name, tag)] = (warm, fresh_kept, fresh)
        print(f"host-buffer {name:10s} [{tag:26s}] {gib:g} GiB bf16: warm buffers {warm * 1e3:6.1f} ms = {n / warm / 1e9:5.1f} GB/s | fresh result buffer {fresh_kept * 1e3:6.1f} ms = {n / fresh_kept / 1e9:5.1f} GB/s"
              f" | fresh, allocated AND freed inside {fresh * 1e3:6.1f} ms = {n / fresh / 1e9:5.1f} GB/s", flush=True)
lib.set_host_slices(0); lib.set_host_direct(4)
# results from the library's pinned arena (zn_host_alloc; what _capi.ZnLib.compress / decompress and the INTEGRATION stub hand out from 8 MiB up): a FRESH block per call,
# allocated and released inside the timed region (the arena recycles it)
for name in ("compress", "decompress"):
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        o = lib.host_buffer(cap if name == "compress" else n)
        compress_into(o) if name == "compress" else decompress_into(frame, flen, o)
        if name == "decompress":
            ok = o[0] == x[0] and o[-1] == x[-1]
        del o
        best = min(best, time.perf_counter() - t0)
    print(f"host-buffer {name:10s} [result from zn_host_alloc   ] {gib:g} GiB bf16: a fresh arena block per call, allocated and released inside {best * 1e3:6.1f} ms = {n / best / 1e9:5.1f} GB/s", flush=True)
o = lib.host_buffer(n); decompress_into(frame, flen, o); assert np.array_equal(o, x); del o
if "--streaming" in sys.argv:
    from zipnn_amd import ZipNN
    raw = x[: 256 << 20].tobytes()
    bc = bd = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); blob = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 20).compress(raw); t1 = time.perf_counter()
        b2 = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 20).decompress(blob); t2 = time.perf_counter()
        bc = min(bc, t1 - t0); bd = min(bd, t2 - t1)
        assert bytes(b2) == raw
        del b2
    print(f"streaming 256 MiB in 1 MiB frames (one batched call each way): compress {bc * 1e3:.0f} ms = {len(raw) / bc / 1e9:.2f} GB/s, decompress {bd * 1e3:.0f} ms = {len(raw) / bd / 1e9:.2f} GB/s")
