#!/usr/bin/env python
"""Developer tool (GPU): the small-input decoder (zn_decode_wide.hpp) against the fused kernel — same bytes, time per call by tensor
size and dtype, in the modes of zn_set_decode_wide (0 never / 1 automatic / 2 always).
    python scripts/wide_check.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from zipnn_amd import _capi, codec


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    lib = _capi.lib(); dev = torch.device("cuda", 0)
    rows = []
    for kind, P, rot, bm in (("bf16", 2, 1, 10), ("fp32", 4, 1, 220), ("fp16", 2, 0, 10), ("fp8", 1, 0, 10)):
        for mib in ((4, 16, 32, 64, 65, 128, 256) if kind == "bf16" else (16, 64)):
            n = (mib << 20) + (250_000 if mib == 65 else 0)
            g = torch.Generator(device="cpu").manual_seed(mib)
            if kind == "fp8":
                x = (torch.randn(n, generator=g) * 0.02).to(torch.float8_e4m3fn).to(dev)
            else:
                dt = {"bf16": torch.bfloat16, "fp32": torch.float32, "fp16": torch.float16}[kind]
                x = (torch.randn(n // dt.itemsize, generator=g) * 0.02).to(dt).to(dev)
            flat = x.view(torch.uint8).reshape(-1); n = flat.numel()
            body = codec.compress_device(lib, flat, P, rot, bm, B.CHUNK, B.THR).clone()
            row = {"kind": kind, "MiB": round(n / 2**20, 2)}
            for mode in (0, 1, 2, 3):
                lib.set_decode_wide(mode)
                dst = torch.zeros(n, dtype=torch.uint8, device=dev)
                codec.decompress_device(lib, body, P, rot, bm, B.CHUNK, n, out=dst, check=True)
                eq = bool(torch.equal(dst, flat))
                for _ in range(5):
                    codec.decompress_device(lib, body, P, rot, bm, B.CHUNK, n, out=dst, check=False)
                d = B.stats(B.time_events(lambda: codec.decompress_device(lib, body, P, rot, bm, B.CHUNK, n, out=dst, check=False), steps))
                row[f"mode{mode}"] = {"ms": round(d["avg"], 4), "min_ms": round(d["min"], 4), "GBps": round(n / d["avg"] / 1e6, 1), "exact": eq, "kernels": lib.last_kernels()}
            lib.set_decode_wide(1)
            rows.append(row)
            print(json.dumps(row), flush=True)
    return rows


if __name__ == "__main__":
    main()
