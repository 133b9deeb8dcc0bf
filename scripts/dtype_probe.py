#!/usr/bin/env python
"""Developer tool (GPU box): a few compress + decompress calls of ONE dtype's 1 GiB tensor, for rocprofv3 to wrap —
kernel statistics or one --pmc set per run (scripts/gpu_pmc_dtypes.sh).  Usage: python scripts/dtype_probe.py fp8 [GiB] [calls]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi, codec
from scripts.bench_dtypes import make


def main():
    kind = sys.argv[1]; gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0; calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lib = _capi.lib(); dev = torch.device("cuda:0")
    if os.environ.get("ZN_WIDE_MODE"):
        lib.set_decode_wide(int(os.environ["ZN_WIDE_MODE"]))
    n = int(gib * (1 << 30))
    x, P, rot, bm, chunk = make(kind, n, dev)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95).clone()
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(calls):
        codec.decompress_device(lib, body, P, rot, bm, chunk, n, out=out, check=False)
        codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95)
    torch.cuda.synchronize()
    assert torch.equal(out, flat)
    print(kind, "bytes", n, "payload", body.numel() - 9 * P * (n // chunk), "kernels", lib.last_kernels())


if __name__ == "__main__":
    main()
