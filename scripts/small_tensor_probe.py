"""Developer probe: decode / encode time of ONE small tensor per call (a plugin's get_tensor without read-ahead), by size and dtype."""
import sys, time, torch
sys.path.insert(0, ".")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
for dt, P, rot, bm in ((torch.bfloat16, 2, 1, 10), (torch.float32, 4, 1, 220)):
    es = torch.empty(0, dtype=dt).element_size()
    for n in (1536, 6144, 16384, 65536, 200000, 262144, 300000, 1 << 20, (1 << 20) + 5000, 3538944, 4718592, 9437184):
        n -= n % es
        g = torch.Generator(device=dev); g.manual_seed(1)
        x = (torch.randn(n // es, generator=g, device=dev) * 0.02).to(dt)
        flat = codec.flat_bytes(x)
        body = codec.compress_device(lib, flat, P, rot, bm, 262144, 0.95).clone()
        out = torch.empty(n, dtype=torch.uint8, device=dev)
        codec.decompress_device(lib, body, P, rot, bm, 262144, n, out=out); ok = torch.equal(out, flat); k = lib.last_kernels()
        best = bestc = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): codec.decompress_device(lib, body, P, rot, bm, 262144, n, out=out, check=False)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): codec.compress_device(lib, flat, P, rot, bm, 262144, 0.95)
            torch.cuda.synchronize(); bestc = min(bestc, (time.perf_counter() - t0) / 10)
        print(f"{str(dt)[6:]:9s} {n:9d} B  decode {best * 1e6:6.1f} us  compress {bestc * 1e6:6.1f} us  ok={ok} ratio {body.numel() / n:.3f} [{k}]", flush=True)
