"""Developer probe: a ragged mid-size tensor through the fused kernel (automatic) and through the wide forms (forced), with the round-6 tail workgroups."""
import sys, time, torch
sys.path.insert(0, ".")
from zipnn_amd import _capi, codec
lib = _capi.lib(); dev = torch.device("cuda:0")
for n in ((64 << 20) + 250000, (60 << 20) + 250000, (126 << 20) + 250000, (128 << 20) + 250000, (100 << 20) + 250 * 1024 + 2, (8 << 20) + 3000, (8 << 20) + 100000, (16 << 20) + 200000, (32 << 20) + 70000):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = (torch.randn(n // 2, generator=g, device=dev) * 0.02).to(torch.bfloat16)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, 2, 1, 10, 262144, 0.95).clone()
    out = torch.empty(flat.numel(), dtype=torch.uint8, device=dev)
    res = []
    for mode in (1, 2, 3):
        lib.set_decode_wide(mode)
        out.zero_(); codec.decompress_device(lib, body, 2, 1, 10, 262144, flat.numel(), out=out); ok = torch.equal(out, flat); k = lib.last_kernels()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): codec.decompress_device(lib, body, 2, 1, 10, 262144, flat.numel(), out=out, check=False)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        res.append(f"mode {mode}: {best * 1e6:6.1f} us ok={ok} [{k}]")
    lib.set_decode_wide(1)
    print(f"{flat.numel():12d} B  " + "   ".join(res), flush=True)
