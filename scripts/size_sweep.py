import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
for mib in (64, 128, 256, 512, 1024, 4096):
    n = mib << 20
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = (torch.randn(n // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    res = {}
    for name, fn in (("dec", lambda: codec.decompress_device(lib, body, 2, 1, 10, 262144, n, out=out, check=False)),
                     ("enc", lambda: codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95))):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): fn()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        res[name] = best
    print(f"{mib:5d} MiB  decode {res['dec']*1e3:7.3f} ms {n/res['dec']/1e9:6.0f} GB/s   compress {res['enc']*1e3:7.3f} ms {n/res['enc']/1e9:6.0f} GB/s", flush=True)
    del x, flat, body, out
