"""Developer tool (GPU box): tensors with / without a partial last chunk, both directions — the tail workgroups of the fused launches
(decode: round 1; encode: round 4) against the same tensor without its tail.  python scripts/ragged_check.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
for n in ((1 << 30), (1 << 30) + 200 * 1024, (100 << 20), (100 << 20) + 250 * 1024 + 2, (8 << 20), (8 << 20) + 3000):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = (torch.randn(n // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    buf = torch.empty(lib.compress_bound(flat.numel(), 2, 262144, 0), dtype=torch.uint8, device=dev)
    body = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95, body=buf).clone()
    ck = lib.last_kernels()
    out = torch.empty(flat.numel(), dtype=torch.uint8, device=dev)
    codec.decompress_device(lib, body, 2, 1, 10, 262144, flat.numel(), out=out)
    ok = torch.equal(out, flat)
    best = bestc = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): codec.decompress_device(lib, body, 2, 1, 10, 262144, flat.numel(), out=out, check=False)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95, body=buf)
        torch.cuda.synchronize(); bestc = min(bestc, (time.perf_counter() - t0) / 10)
    print(f"{flat.numel():12d} bytes ok={ok} decode {best*1e3:.3f} ms {flat.numel()/best/1e9:.0f} GB/s | compress {bestc*1e3:.3f} ms {flat.numel()/bestc/1e9:.0f} GB/s  kernels={ck}", flush=True)
