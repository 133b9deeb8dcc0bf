"""Developer stress (GPU): random ragged sizes through the automatic decode path (small-input forms with tail + merge workgroups in the launch, fused launch, large
calls), every output compared with the input, every call repeated — the tail -> merge hand-over is a cross-workgroup flag inside one launch and races differently every time."""
import sys, time, random, torch
sys.path.insert(0, ".")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)
bad = calls = 0; forms = {}
while time.time() < t_end:
    dt, P, rot, bm = random.choice([(torch.bfloat16, 2, 1, 10), (torch.float32, 4, 1, 220), (torch.float16, 2, 0, 10)])
    es = torch.empty(0, dtype=dt).element_size()
    full = random.choice([0, 1, 2, 7, 30, 100, 200, 240, 250, 256, 300, 400, 480, 500, 520, 1100, 2100])
    tail = random.choice([0, 2, 510, 1022, 1026, 2050, 5000, 8190, 8200, 70000, 131070, 200000, 262142])
    n = full * 262144 + tail; n -= n % es
    if n == 0: continue
    x = (torch.randn(n // es, device=dev) * 0.02).to(dt)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, P, rot, bm, 262144, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    for rep in range(8):
        out.fill_(0x5A)
        codec.decompress_device(lib, body, P, rot, bm, 262144, n, out=out, check=(rep == 0))
        calls += 1
        if not torch.equal(out, flat):
            bad += 1; print("MISMATCH", dt, n, full, tail, lib.last_kernels(), flush=True)
    k = lib.last_kernels(); forms[k] = forms.get(k, 0) + 1
    del x, flat, body, out
print("calls", calls, "bad", bad)
for k, v in sorted(forms.items(), key=lambda kv: -kv[1]): print(f"{v:5d}  {k}")
