import ctypes, os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"scripts"))
import torch
from ab_variants import load
names=sys.argv[1:]
libs=[(n, load(os.path.join(ROOT,"zipnn_amd",f"libzipnn_hip_ab_{n}.so"))) for n in names]
st = torch.cuda.current_stream().cuda_stream
for mib in (64, 1024, 4096):
    n = mib<<20
    x = torch.empty(n//4, dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for off in range(0, x.numel(), 1<<27): x[off:off+(1<<27)] = torch.randn(min(1<<27, x.numel()-off), generator=g, device="cuda")*0.02
    flat = x.view(torch.uint8).reshape(-1)
    L0 = libs[0][1]; cap = L0.zn_compress_bound(n, 4, 262144, 0)
    body = torch.empty(cap, dtype=torch.uint8, device="cuda"); ln = ctypes.c_size_t(0)
    assert L0.zn_compress_dev(flat.data_ptr(), n, 4, 1, 220, 262144, 0.95, body.data_ptr(), cap, ctypes.byref(ln), None) == 0
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    best = {k: 1e9 for k,_ in libs}; ok = {}
    for k, L in libs:
        out.zero_(); rc = L.zn_decompress_dev(body.data_ptr(), ln.value, 4, 1, 220, 262144, n, out.data_ptr(), st, 1); ok[k] = rc == 0 and bool(torch.equal(out, flat))
    reps = 40 if mib <= 128 else 10
    for rnd in range(5):
        for k, L in libs:
            torch.cuda.synchronize(); t0=time.perf_counter()
            for _ in range(reps): L.zn_decompress_dev(body.data_ptr(), ln.value, 4, 1, 220, 262144, n, out.data_ptr(), st, 0)
            torch.cuda.synchronize(); best[k]=min(best[k], (time.perf_counter()-t0)/reps)
    print(mib, "MiB fp32:", "  ".join(f"{k}={best[k]*1e6:.1f}us ({n/best[k]/1e9:.0f} GB/s) ok={ok[k]}" for k,_ in libs), flush=True)
    del x, flat, body, out; torch.cuda.empty_cache()
