"""Generate tests/golden/golden_v1.npz from the REFERENCE itself.

Runs only in the build container, where /root/reference exists: it imports the
reference's own Python package (`/root/reference/zipnn`) on top of oracle/_ref/
zipnn_core.so (the reference csrc/ compiled from where it lies, linked to the system
libzstd 1.4.8 huff0 — see oracle/Makefile) and records, for a set of small seeded
inputs, the exact frame `ZipNN(...).compress(x)` returns and the bytes `decompress`
gives back (as a sha256 — the input is recoverable by decoding the frame, and the
hash pins that decode).  The reference tree holds no golden vectors of its own,
so these fixtures are the compressed-bytes pin for the oracle and for the HIP path.

    python tests/golden/make_golden.py          # rewrites golden_v1.npz

Inputs follow the reference's tests (tests/simple_stress_tests.py:19-70,154-203,
205-264: rand*2-1 bf16 around chunk boundaries, urandom bytes, fp32 bytes,
half-constant/half-random fp16/bf16/fp8 matrices) plus weights-like N(0,0.02).
"""
import hashlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))   # zipnn_core (reference C ext)
sys.dont_write_bytecode = True                               # (nothing is written under /root/reference, not even __pycache__)
sys.path.insert(1, "/root/reference")                        # zipnn (reference Python)


def main():
    import contextlib
    from zipnn import ZipNN  # the reference package

    KB = 1024
    cases = []

    def add(name, make, ctor, kind):
        cases.append((name, make, ctor, kind))

    def bf16_u(n, seed):
        g = torch.Generator().manual_seed(seed)
        return (torch.rand(n, generator=g) * 2 - 1).to(torch.bfloat16)

    def bf16_n(n, seed):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(n, generator=g) * 0.02).to(torch.bfloat16)

    def half_const(dtype, seed):
        g = torch.Generator().manual_seed(seed)
        t = torch.ones(100, 100)
        t[50:] = torch.rand(50, 100, generator=g) * 2 - 1
        return t.to(dtype)

    # torch format, bf16, chunk boundary -1/0/+1 element and a partial tail
    for i, kib in enumerate((256, 257)):
        add(f"torch_bf16_u_{kib}k", lambda kib=kib, i=i: bf16_u(kib * KB // 2, 100 + i),
            dict(input_format="torch"), "torch")
    add("torch_bf16_n_2chunks_plus1", lambda: bf16_n(128 * KB + 513, 7), dict(input_format="torch"), "torch")
    add("torch_bf16_2d", lambda: bf16_n(100 * 333, 8).reshape(100, 333), dict(input_format="torch"), "torch")
    add("torch_fp16_halfconst", lambda: half_const(torch.float16, 9), dict(input_format="torch"), "torch")
    add("torch_bf16_halfconst", lambda: half_const(torch.bfloat16, 10), dict(input_format="torch"), "torch")
    add("torch_fp8e4m3_halfconst", lambda: half_const(torch.float8_e4m3fn, 11), dict(input_format="torch"), "torch")
    add("torch_fp8e5m2_n", lambda: (torch.randn(140 * KB, generator=torch.Generator().manual_seed(12)) * 0.5).to(torch.float8_e5m2),
        dict(input_format="torch"), "torch")
    add("torch_fp32_n_chunk64k", lambda: torch.randn(37 * KB + 3, generator=torch.Generator().manual_seed(13)) * 0.02,
        dict(input_format="torch", compression_chunk=64 * KB), "torch")
    add("torch_fp16_n_chunk64k", lambda: (torch.randn(65 * KB, generator=torch.Generator().manual_seed(14)) * 0.02).half(),
        dict(input_format="torch", compression_chunk=64 * KB), "torch")
    # byte format
    rng = np.random.default_rng(1234)
    add("byte_bf16_urandom_129k", lambda: rng.integers(0, 256, 129 * KB, dtype=np.uint8).tobytes(),
        dict(bytearray_dtype="bfloat16", compression_chunk=64 * KB), "byte")
    add("byte_bf16_odd_len", lambda: bf16_n(70001, 15).view(torch.uint8).numpy().tobytes()[:140001],
        dict(bytearray_dtype="bfloat16"), "byte")
    add("byte_bf16_len_mod4_eq2", lambda: bf16_n(70001, 16).view(torch.uint8).numpy().tobytes(),
        dict(bytearray_dtype="bfloat16"), "byte")
    add("byte_fp32_8k", lambda: (torch.rand(8 * KB, generator=torch.Generator().manual_seed(17)) * 2 - 1).numpy().tobytes(),
        dict(bytearray_dtype="float32"), "byte")
    add("byte_fp16_n", lambda: (torch.randn(20000, generator=torch.Generator().manual_seed(18)) * 0.02).half().numpy().tobytes(),
        dict(bytearray_dtype="float16"), "byte")
    add("byte_bf16_chunk64k", lambda: bf16_n(100000, 19).view(torch.uint8).numpy().tobytes(),
        dict(bytearray_dtype="bfloat16", compression_chunk=64 * KB), "byte")
    add("byte_bf16_zeros_ones", lambda: (b"\x00" * (600 * KB) + b"\x01" * (424 * KB)),
        dict(bytearray_dtype="bfloat16"), "byte")
    add("byte_bf16_streaming_10k", lambda: bf16_n(5 * KB, 20).view(torch.uint8).numpy().tobytes(),
        dict(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 19), "byte")
    add("byte_bf16_streaming_multi", lambda: bf16_n(150 * KB + 77, 21).view(torch.uint8).numpy().tobytes(),
        dict(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 17, compression_chunk=64 * KB), "byte")
    add("byte_fp8_n_129k", lambda: (torch.randn(129 * KB, generator=torch.Generator().manual_seed(22)) * 0.05)
        .to(torch.float8_e4m3fn).view(torch.uint8).numpy().tobytes(), dict(bytearray_dtype="float8_e4m3fn"), "byte")

    out = {}
    meta = []
    for name, make, ctor, kind in cases:
        x = make()
        if kind == "torch":
            raw = x.contiguous().view(torch.uint8).numpy().tobytes() if x.dtype != torch.uint8 else x.numpy().tobytes()
            src = x.clone()          # the reference rotates its input in place
        else:
            raw = bytes(x)
            src = bytearray(raw)
        with contextlib.redirect_stdout(io.StringIO()):
            frame = bytes(ZipNN(**ctor).compress(src))
            back = ZipNN(**ctor).decompress(frame)
        if kind == "torch":
            assert back.dtype == x.dtype and tuple(back.shape) == tuple(x.shape), name
            back_raw = back.contiguous().view(torch.uint8).numpy().tobytes()
        else:
            back_raw = bytes(back)
        assert back_raw == raw, f"reference round trip failed for {name}"
        out[name + ".frame"] = np.frombuffer(frame, dtype=np.uint8)
        meta.append(dict(name=name, kind=kind, ctor=ctor,
                         dtype=(str(x.dtype).replace("torch.", "") if kind == "torch" else ctor.get("bytearray_dtype")),
                         shape=(list(x.shape) if kind == "torch" else None),
                         in_len=len(raw), frame_len=len(frame),
                         in_sha256=hashlib.sha256(raw).hexdigest(), frame_sha256=hashlib.sha256(frame).hexdigest()))
        print(f"{name:32s} in={len(raw):8d} frame={len(frame):8d} ratio={len(frame) / max(len(raw), 1):.4f}")
    out["meta.json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
