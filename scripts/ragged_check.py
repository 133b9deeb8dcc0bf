import sys, time, torch
sys.path.insert(0, "/root/repo")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
for n in ((1 << 30), (1 << 30) + 200 * 1024, (100 << 20) + 250 * 1024 + 2, (100 << 20)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = (torch.randn(n // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95).clone()
    out = torch.empty(flat.numel(), dtype=torch.uint8, device=dev)
    codec.decompress_device(lib, body, 2, 1, 10, 262144, flat.numel(), out=out)
    ok = torch.equal(out, flat)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): codec.decompress_device(lib, body, 2, 1, 10, 262144, flat.numel(), out=out, check=False)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
    print(f"{flat.numel():12d} bytes ok={ok} tail_planes={lib.last_tail_planes()} decode {best*1e3:.3f} ms {flat.numel()/best/1e9:.0f} GB/s  kernels={lib.last_kernels()}", flush=True)
