import sys, time, torch
sys.path.insert(0, "/root/repo")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
n = (100 << 20) + 250 * 1024 + 2
g = torch.Generator(device=dev); g.manual_seed(1)
x = (torch.randn(n // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16)
flat = codec.flat_bytes(x)
body = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95).clone()
out = torch.empty(flat.numel(), dtype=torch.uint8, device=dev)
for _ in range(5): codec.decompress_device(lib, body, 2, 1, 10, 262144, flat.numel(), out=out, check=False)
torch.cuda.synchronize()
