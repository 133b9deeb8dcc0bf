import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi
lib = _capi.lib()
n = 1 << 30
x = (torch.randn(n // 2, device="cuda") * 0.02).to(torch.bfloat16).cpu().view(torch.uint8).numpy()
hdr = bytes(32)
for sl in (1, 4):
    lib.set_host_slices(sl)
    frame = lib.compress(hdr, x, 2, 1, 10, 262144, 0.95)
    back = lib.decompress(memoryview(frame)[32:], 2, 1, 10, 262144, n)
    os.environ["ZN_HOST_PIPE_TRACE"] = "1"
    print(f"---- slices {sl}: compress", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); frame = lib.compress(hdr, x, 2, 1, 10, 262144, 0.95); t1 = time.perf_counter()
    print(f"---- {1e3 * (t1 - t0):.1f} ms; decompress", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); back = lib.decompress(memoryview(frame)[32:], 2, 1, 10, 262144, n); t1 = time.perf_counter()
    print(f"---- {1e3 * (t1 - t0):.1f} ms", file=sys.stderr, flush=True)
    os.environ.pop("ZN_HOST_PIPE_TRACE")
