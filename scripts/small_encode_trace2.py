import sys, time, torch
sys.path.insert(0, ".")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
def tm(flat, body):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95, body=body)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
    return min(ts)
n = 6144
body = torch.empty(lib.compress_bound(n, 2, 262144, 0) + 64, dtype=torch.uint8, device=dev)
keep = []
for seed in range(12):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    x = (torch.randn(n // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x); keep.append(x)
    t1 = tm(flat, body)
    y = x.clone(); keep.append(y)                      # same data, another address
    t2 = tm(codec.flat_bytes(y), body)
    b = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95, body=body)
    # per-stage timing through events is not available from here: report the table's shape instead
    hi = flat.view(-1, 2)[:, 1]
    print(f"seed {seed:2d} addr%4096={flat.data_ptr() % 4096:4d} t={t1:6.1f} us  clone addr%4096={y.data_ptr() % 4096:4d} t={t2:6.1f} us  body {b.numel()}  distinct exponent bytes {hi.unique().numel()}", flush=True)
