#!/usr/bin/env python
"""Developer tool (GPU): random tensors through the small-input decoder (zn_set_decode_wide 2) — sizes, dtypes, scales (code densities around
the rule that admits a chunk, staging-buffer fill, record slots), mixtures of chunk kinds, slow-sync codes — each decoded several times and
compared with the input; the frame is checked against the oracle for a sample.   python scripts/wide_stress.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zipnn_amd import _capi, codec

C = 256 * 1024


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = _capi.lib(); dev = torch.device("cuda", 0)
    r = np.random.default_rng(seed)
    bad = 0; wide_used = 0; t0 = time.time()
    for i in range(cases):
        kind = r.choice(["bf16", "bf16", "bf16", "fp32", "fp16", "fp8", "mix", "peaky"])
        K = int(r.choice([1, 2, 3, 5, 17, 64, 100, 255, 256]))
        tail = int(r.choice([0, 0, 10, 4096, 100000, C - 2]))
        scale = float(10 ** r.uniform(-4, 1))
        g = torch.Generator(device="cuda"); g.manual_seed(int(r.integers(1 << 30)))
        if kind in ("bf16", "mix", "peaky"):
            P, rot, bm, es, dt = 2, 1, 10, 2, torch.bfloat16
        elif kind == "fp32":
            P, rot, bm, es, dt = 4, 1, 220, 4, torch.float32
        elif kind == "fp16":
            P, rot, bm, es, dt = 2, 0, 10, 2, torch.float16
        else:
            P, rot, bm, es, dt = 1, 0, 10, 1, torch.float8_e4m3fn
        n = (K * C + tail) // es * es
        x = torch.randn(n // es, generator=g, device=dev) * scale
        if kind == "peaky":        # most values at one magnitude: a short dominant code, tiles that decode to many symbols
            m = torch.rand(n // es, generator=g, device=dev) < float(r.uniform(0.5, 0.97))
            x = torch.where(m, torch.full_like(x, scale), x)
        x = x.to(dt)
        flat = x.view(torch.uint8).reshape(-1).clone()
        if kind == "mix":          # some chunks incompressible, some constant
            for c in range(0, K, 3):
                flat[c * C:(c + 1) * C] = torch.randint(0, 256, (min(C, n - c * C),), generator=g, device=dev, dtype=torch.uint8)
            for c in range(1, K, 5):
                flat[c * C:(c + 1) * C] = 7
        n = flat.numel()
        body = codec.compress_device(lib, flat, P, rot, bm, C, 0.95).clone()
        outs = {}
        for mode in (2, 3, 0):
            lib.set_decode_wide(mode)
            for rep in range(3 if mode else 1):
                dst = torch.zeros(n, dtype=torch.uint8, device=dev)
                codec.decompress_device(lib, body, P, rot, bm, C, n, out=dst, check=True)
                if not torch.equal(dst, flat):
                    bad += 1
                    print("MISMATCH", i, kind, K, tail, scale, "mode", mode, "rep", rep, "first diff", int((dst != flat).nonzero()[0]), flush=True)
            if mode == 2:
                wide_used += lib.last_kernels().startswith("zn_k_decode_wide")
        lib.set_decode_wide(1)
        if i % 25 == 0:
            print(f"case {i}: {kind} K={K} tail={tail} scale={scale:.2g} ratio={body.numel() / n:.3f} ok so far, {time.time() - t0:.0f}s", flush=True)
    print(f"done: {cases} cases, {bad} mismatches, wide launch in {wide_used} of them")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
