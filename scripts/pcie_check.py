import time, torch
n = 1 << 30
h = torch.empty(n, dtype=torch.uint8); h.fill_(7)
hp = torch.empty(n, dtype=torch.uint8).pin_memory(); hp.fill_(7)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, src, dst in (("H2D pageable", h, d), ("H2D pinned", hp, d), ("D2H pageable", d, h), ("D2H pinned", d, hp)):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(src); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"{name}: {n / best / 1e9:.1f} GB/s")
t0 = time.perf_counter(); b = bytearray(n); t1 = time.perf_counter(); print(f"bytearray(1GiB): {(t1 - t0) * 1e3:.0f} ms")
import numpy as np
t0 = time.perf_counter(); a = np.empty(n, dtype=np.uint8); a[::4096] = 1; t1 = time.perf_counter(); print(f"np.empty + touch: {(t1 - t0) * 1e3:.0f} ms")
