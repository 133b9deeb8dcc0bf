#!/usr/bin/env python
"""Developer tool (GPU box): tensors whose chunk counts sit ON and AROUND the round boundaries of the fused decoder (1 024 workgroups a round, groups of 1..4 chunks:
K = 1023 .. 1025, 2047 .. 2049, … 5121), with and without a partial last chunk, every dtype — compressed by both encoders (bodies must be identical), decoded with the
automatic group size and with every forced one (bytes must equal the input), a sample compared with the oracle's frame.   python scripts/round_stress.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from zipnn_amd import _capi, codec

GEOM = {"bf16": (torch.bfloat16, 2, 1, 10, 256 * 1024), "fp16": (torch.float16, 2, 0, 10, 256 * 1024),
        "fp32": (torch.float32, 4, 1, 220, 256 * 1024), "fp8": (torch.float8_e4m3fn, 1, 0, 10, 128 * 1024)}


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rng = np.random.default_rng(seed)
    lib = _capi.lib(); dev = torch.device("cuda:0")
    try:
        import oracle_lib as O
    except Exception:
        O = None
    ks = [k + d for k in (1024, 2048, 3072, 4096, 5120) for d in (-1, 0, 1)] + [513, 1536, 2304, 3584, 6144, 6145]
    n_ok = 0
    for K in ks:
        kind = ("bf16", "fp16", "fp32", "fp8")[int(rng.integers(0, 4))] if K not in (6144, 6145) else "bf16"
        dt, P, rot, bm, chunk = GEOM[kind]
        tail = int(rng.integers(0, 3))            # 0: whole chunks, 1: a few bytes more, 2: most of a chunk more
        n = K * chunk + (0 if tail == 0 else (int(rng.integers(1, 64)) * 4 if tail == 1 else chunk - int(rng.integers(1, 5000)) * 4))
        es = torch.empty(0, dtype=dt).element_size()
        n -= n % es
        g = torch.Generator(device=dev); g.manual_seed(int(rng.integers(1, 1 << 30)))
        x = (torch.randn(n // es, generator=g, device=dev) * float(rng.choice([0.02, 0.5, 3.0]))).to(dt)
        flat = codec.flat_bytes(x)
        bodies = []
        for mode in (2, 0):
            lib.set_encode_onepass(mode)
            bodies.append(codec.compress_device(lib, flat, P, rot, bm, chunk, 0.95).clone())
        lib.set_encode_onepass(1)
        assert torch.equal(bodies[0], bodies[1]), (kind, K, tail, "one-pass and four-kernel bodies differ")
        body = bodies[0]
        for grp in (0, 1, 2, 3, 4):
            lib.set_decode_group(grp)
            out = torch.empty(n, dtype=torch.uint8, device=dev)
            codec.decompress_device(lib, body, P, rot, bm, chunk, n, out=out)
            assert torch.equal(out, flat), (kind, K, tail, "group", grp)
        lib.set_decode_group(0)
        if O is not None and K <= 1025:
            ref = O.compress_frame(bytes(range(32)), flat.cpu().numpy().tobytes(), P, rot, bm, chunk, threads=16)
            assert ref[32:] == body.cpu().numpy().tobytes(), (kind, K, tail, "frame differs from the oracle's")
        n_ok += 1
        print(f"ok {kind:5s} K = {K:5d} tail {tail}  n = {n}  ratio {body.numel() / n:.4f}  groups auto={lib.decode_group_for(-(-n // chunk))}", flush=True)
        del x, flat, bodies, body, out
    print(f"{n_ok} tensors: every encoder / group size agrees")


if __name__ == "__main__":
    main()
