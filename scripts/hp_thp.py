import ctypes, os, sys, time, mmap
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi
lib = _capi.lib()
libc = ctypes.CDLL("libc.so.6", use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
print("THP enabled:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
n = 1 << 30
t = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for mode in ("plain", "MADV_HUGEPAGE", "MADV_POPULATE_WRITE", "plain"):
    best = 1e9
    for _ in range(3):
        out = np.empty(n, dtype=np.uint8)
        addr = out.ctypes.data
        t0 = time.perf_counter()
        a = (addr + (2 << 20) - 1) & ~((2 << 20) - 1)
        if mode == "MADV_HUGEPAGE":
            r = libc.madvise(a, n - (a - addr) & ~((2 << 20) - 1), 14)
        elif mode == "MADV_POPULATE_WRITE":
            r = libc.madvise(addr & ~4095, n, 23)
        else:
            r = 0
        lib.copy_to_host(out, t.data_ptr())
        dt = time.perf_counter() - t0
        best = min(best, dt)
        del out
    print(f"D2H 1 GiB into a fresh numpy buffer, {mode:20s} madvise rc {r}: {best * 1e3:.1f} ms = {n / best / 1e9:.1f} GB/s", flush=True)
src = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); lib.copy_to_device(t.data_ptr(), src); best = min(best, time.perf_counter() - t0)
print(f"H2D 1 GiB from a touched numpy buffer: {best * 1e3:.1f} ms = {n / best / 1e9:.1f} GB/s")
out = np.empty(n, dtype=np.uint8); out[:] = 0
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); lib.copy_to_host(out, t.data_ptr()); best = min(best, time.perf_counter() - t0)
print(f"D2H 1 GiB into an already-touched buffer: {best * 1e3:.1f} ms = {n / best / 1e9:.1f} GB/s")
