"""Developer stress (GPU): random sizes and dtypes through compress, every tensor four times — the four bodies must be the same bytes (an encoder race shows as a
body that differs from run to run) and decode to the input."""
import sys, time, random, torch
sys.path.insert(0, ".")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
bad = calls = 0; forms = {}
f8 = getattr(torch, "float8_e4m3fn", None)
while time.time() < t_end:
    dt, P, rot, bm, chunk = random.choice([(torch.bfloat16, 2, 1, 10, 262144), (torch.float32, 4, 1, 220, 262144), (torch.float16, 2, 0, 10, 262144)] + ([(f8, 1, 0, 10, 131072)] if f8 else []))
    es = torch.empty(0, dtype=dt).element_size()
    full = random.choice([0, 1, 2, 7, 30, 100, 256, 400, 1100, 2100, 6200, 7000])
    tail = random.choice([0, 2, 510, 1026, 5000, 8200, 70000, 131070, 200000])
    n = full * chunk + (tail % chunk); n -= n % es
    if n == 0: continue
    x = (torch.randn(n // es, device=dev) * 0.02).to(dt)
    flat = codec.flat_bytes(x)
    first = None
    for rep in range(4):
        b = codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95).clone(); calls += 1
        if first is None: first = b
        elif b.numel() != first.numel() or not torch.equal(b, first):
            bad += 1; print("BODY DIFFERS", dt, n, lib.last_kernels(), flush=True)
    out = codec.decompress_device(lib, first, P, rot, bm, chunk, n)
    if not torch.equal(out, flat): bad += 1; print("ROUND TRIP", dt, n, flush=True)
    k = lib.last_kernels(); 
    del x, flat, first, out, b
print("calls", calls, "bad", bad)
