"""Developer probe: large ragged bf16 tensors (more than 2 048 chunks + a partial one), decode time."""
import os, sys, time, torch
sys.path.insert(0, ".")
from zipnn_amd import _capi, codec
dev = torch.device("cuda:0")
libs = [("tree", _capi.lib())] + [(os.path.basename(p), _capi.ZnLib(p)) for p in sys.argv[1:]]      # (further libraries to time beside the tree's: scripts/ab_variants.py --build …)
lib = libs[0][1]
for n in ((128 << 20) + 250000, (200 << 20) + 250000, (256 << 20) + 250000, (300 << 20) + 250000, (600 << 20) + 200000, (1 << 30) + 200000, (4 << 30) + 200000, 4 << 30):
    x = (torch.randn(n // 2, device=dev) * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    res = []
    for name, L in libs:
        out.zero_(); codec.decompress_device(L, body, 2, 1, 10, 262144, n, out=out); ok = torch.equal(out, flat); k = L.last_kernels()
        res.append([name, 1e9, ok, k])
    for _ in range(4):
        for r, (name, L) in zip(res, libs):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): codec.decompress_device(L, body, 2, 1, 10, 262144, n, out=out, check=False)
            torch.cuda.synchronize(); r[1] = min(r[1], (time.perf_counter() - t0) / 10)
    print(f"{n:12d} B  " + "   ".join(f"{name}: {t * 1e6:8.1f} us ok={ok} [{k}]" for name, t, ok, k in res), flush=True)
    del x, flat, body, out; torch.cuda.empty_cache()
