"""Developer probe (GPU box): wall time of ZipNN().compress / decompress through the Python API by input size — bytes in, bytes out (host buffers, PCIe inside) —
beside the reference's C core (oracle/_ref, 16 threads) on the same frame body.  TEST-SIDE use of oracle/: a probe, not product."""
import os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from zipnn_amd import ZipNN
import oracle_lib as O
z = ZipNN(input_format="byte", bytearray_dtype="bfloat16")
ref = O.ref_core()
for n in (4096, 65536, 1 << 20, 8 << 20, 64 << 20, 256 << 20):
    r = np.random.default_rng(1)
    x = (r.standard_normal(n // 2).astype(np.float32) * 0.02)
    b = torch.from_numpy(x).to(torch.bfloat16).view(torch.uint8).numpy().tobytes()
    f = z.compress(b); back = z.decompress(f); assert bytes(back) == b
    reps = 30 if n <= (8 << 20) else 5
    tc = td = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): f = z.compress(b)
        tc = min(tc, (time.perf_counter() - t0) / reps)
        t0 = time.perf_counter()
        for _ in range(reps): back = z.decompress(f)
        td = min(td, (time.perf_counter() - t0) / reps)
    line = f"{n:10d} B  GPU path: compress {tc * 1e6:9.1f} us ({n / tc / 1e9:6.2f} GB/s)  decompress {td * 1e6:9.1f} us ({n / td / 1e9:6.2f} GB/s)"
    if ref is not None:
        hdr = bytes(f[:32]); rc = rd = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps): fr = O.ref_compress_frame(hdr, b, 2, 1, 10, 262144, 0.95, threads=16)
            rc = min(rc, (time.perf_counter() - t0) / reps)
            t0 = time.perf_counter()
            for _ in range(reps): br = O.ref_decompress_body(bytes(fr)[32:], 2, 1, 10, 262144, n, threads=16)
            rd = min(rd, (time.perf_counter() - t0) / reps)
        line += f"   | reference core, 16 threads: compress {rc * 1e6:9.1f} us  decompress {rd * 1e6:9.1f} us"
    print(line, flush=True)
