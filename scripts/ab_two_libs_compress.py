import ctypes, os, sys, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"scripts"))
import torch
from ab_variants import load
names=sys.argv[1:]
libs=[(n, load(os.path.join(ROOT,"zipnn_amd",f"libzipnn_hip_ab_{n}.so"))) for n in names]
st = torch.cuda.current_stream().cuda_stream
for mib in (16, 64, 256, 1024):
    n = mib<<20
    x = (torch.randn(n//2, device="cuda")*0.02).to(torch.bfloat16); flat = x.view(torch.uint8).reshape(-1)
    L0 = libs[0][1]; cap = L0.zn_compress_bound(n, 2, 262144, 0)
    body = torch.empty(cap, dtype=torch.uint8, device="cuda"); ln = ctypes.c_size_t(0)
    best = {k: 1e9 for k,_ in libs}; lens = {}
    for rnd in range(6):
        for k, L in libs:
            torch.cuda.synchronize(); t0=time.perf_counter()
            for _ in range(40): L.zn_compress_dev(flat.data_ptr(), n, 2, 1, 10, 262144, 0.95, body.data_ptr(), cap, ctypes.byref(ln), st)
            torch.cuda.synchronize(); best[k]=min(best[k], (time.perf_counter()-t0)/40); lens[k]=ln.value
    print(mib, "MiB compress:", "  ".join(f"{k}={best[k]*1e6:.1f}us len={lens[k]}" for k,_ in libs), flush=True)
