"""Developer tool (GPU box, under rocprofv3 --kernel-trace --stats): compress + decompress of a bf16 tensor of argv[1] bytes, 20 calls each."""
import sys, torch
sys.path.insert(0, "/root/repo")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else (100 << 20) + 250 * 1024 + 2
g = torch.Generator(device=dev); g.manual_seed(1)
x = (torch.randn(n // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16)
flat = codec.flat_bytes(x)
buf = torch.empty(lib.compress_bound(flat.numel(), 2, 262144, 0), dtype=torch.uint8, device=dev)
body = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95, body=buf).clone()
out = torch.empty(flat.numel(), dtype=torch.uint8, device=dev)
for _ in range(20):
    codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95, body=buf)
    codec.decompress_device(lib, body, 2, 1, 10, 262144, flat.numel(), out=out, check=False)
torch.cuda.synchronize()
