#!/usr/bin/env python
"""Developer tool (GPU box): does a buffer that was just read come back faster the second time (Infinity Cache)?
Repeated device copies of one buffer, 16 MiB … 2 GiB; GB/s = (read + write) bytes / s."""
import time
import torch


def main():
    for mib in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
        n = mib << 20
        a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
        b = torch.empty_like(a)
        reps = max(10, min(400, (8 << 30) // n))
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                b.copy_(a)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
        print(f"{mib:5d} MiB  copy {best * 1e6:8.1f} us   {2 * n / best / 1e9:7.0f} GB/s", flush=True)
        del a, b


if __name__ == "__main__":
    main()
