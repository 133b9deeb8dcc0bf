"""Developer probe (under rocprofv3 --kernel-trace --stats): 40 automatic-mode decodes each of three ragged bf16 tensors — 8 MiB + 100 000 B (16-wave small-input
form), 100 MiB + 250 000 B (8-wave form) and 1 GiB + 200 000 B (fused launch) — so that the kernel statistics show the partial-chunk path's launches by name."""
import sys, torch
sys.path.insert(0, ".")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
for n in ((8 << 20) + 100000, (100 << 20) + 250000, (1 << 30) + 200000):
    x = (torch.randn(n // 2, device=dev) * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(40): codec.decompress_device(lib, body, 2, 1, 10, 262144, n, out=out, check=False)
    torch.cuda.synchronize(); assert torch.equal(out, flat)
    print(n, lib.last_kernels())
