#!/usr/bin/env python
"""Developer tool (GPU box): run zn_compress_dev of a 4 GiB bf16 tensor a few times with each variant library given on
the command line — meant to run under `rocprofv3 --kernel-trace --stats`, which then shows the per-kernel times of the
encoder (stats / tables / emit) per variant run.  Ablated variants produce wrong frames: timing only."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ab_variants as AB
name = sys.argv[1]
L = AB.load(AB.so_path(name))
n = 4 << 30
g = torch.Generator(device="cuda"); g.manual_seed(5)
x = torch.empty(n // 2, dtype=torch.bfloat16, device="cuda")
for off in range(0, x.numel(), 1 << 27):
    x[off:off + (1 << 27)] = (torch.randn(min(1 << 27, x.numel() - off), generator=g, device="cuda") * 0.02).to(torch.bfloat16)
flat = x.view(torch.uint8).reshape(-1)
cap = L.zn_compress_bound(n, 2, 262144, 0)
body = torch.empty(cap, dtype=torch.uint8, device="cuda"); ln = ctypes.c_size_t(0)
st = torch.cuda.current_stream().cuda_stream
for _ in range(12):
    L.zn_compress_dev(flat.data_ptr(), n, 2, 1, 10, 262144, 0.95, body.data_ptr(), cap, ctypes.byref(ln), st)
torch.cuda.synchronize()
print(name, "done", ln.value)
