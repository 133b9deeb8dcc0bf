#!/usr/bin/env python
"""PCIe-inclusive rate of the host-buffer entry points (zn_compress / zn_decompress with pageable host memory)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi
lib = _capi.lib()
n = 1 << 30
x = (torch.randn(n // 2, device="cuda") * 0.02).to(torch.bfloat16).cpu().view(torch.uint8).numpy()
hdr = bytes(32)
frame = lib.compress(hdr, x, 2, 1, 10, 262144, 0.95)
back = lib.decompress(memoryview(frame)[32:], 2, 1, 10, 262144, n)
assert bytes(back[:4096]) == x[:4096].tobytes() and len(back) == n
for slices, name, fn in [(sl, nm, f) for sl in (1, 0, 4, 8, 16) for nm, f in (("compress", lambda: lib.compress(hdr, x, 2, 1, 10, 262144, 0.95)), ("decompress", lambda: lib.decompress(memoryview(frame)[32:], 2, 1, 10, 262144, n)))]:
    lib.set_host_slices(slices)
    name = f"{name} [{'one shot' if slices == 1 else 'pipelined, automatic slices' if slices == 0 else f'pipelined, {slices} slices'}]"
    best = best_free = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter()      # the call, result kept alive
        del r; t2 = time.perf_counter()                                    # … and with the result released (munmap of its pages)
        best = min(best, t1 - t0); best_free = min(best_free, t2 - t0)
    print(f"host-buffer {name}: 1 GiB bf16 in {best * 1e3:.1f} ms = {n / best / 1e9:.1f} GB/s (pageable host memory, PCIe both ways; "
          f"{best_free * 1e3:.1f} ms with the result buffer freed again)")

lib.set_host_slices(0)
# streaming `.znn` blob (1 MiB frames): batched compress and decompress, best of 3 (the first call of a process also
# pays for the pinned bounce buffers and the allocator's first 256 MiB blocks)
from zipnn_amd import ZipNN
raw = x[: 256 << 20].tobytes()
bc = bd = 1e9
for _ in range(3):
    t0 = time.perf_counter(); blob = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 20).compress(raw); t1 = time.perf_counter()
    back = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 20).decompress(blob); t2 = time.perf_counter()
    bc = min(bc, t1 - t0); bd = min(bd, t2 - t1)
    assert bytes(back) == raw
    del back
print(f"streaming 256 MiB in 1 MiB frames (one batched call each way): compress {bc * 1e3:.0f} ms = {len(raw) / bc / 1e9:.2f} GB/s, "
      f"decompress {bd * 1e3:.0f} ms = {len(raw) / bd / 1e9:.2f} GB/s")
