"""Generate tests/golden/golden_delta_v1.npz from the REFERENCE itself: delta (XOR) frames.

Build container only (imports /root/reference/zipnn on top of oracle/_ref/zipnn_core.so, exactly as
make_golden.py does; see its header).  The reference's delta mode XORs the input with a second buffer of the same
length on the host before the core call and after it (zipnn/zipnn.py:625-640, 983-1004); its own tests cover it at
tests/simple_stress_tests.py:85-150,180-203.  Recorded per case: the frame `ZipNN(delta_compressed_type=…)
.compress(data, delta_second_data=base)` returns, the base bytes (a decoder needs them) and the sha256 of the data.

    python tests/golden/make_golden_delta.py          # rewrites golden_delta_v1.npz

Cases: a fine-tune-like pair of bf16 tensors (2 % of the elements differ: both planes compress), ragged length;
the reference test's own shape (random bytes, first half equal); streaming delta; float32 bytes streaming delta.
"""
import contextlib
import hashlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
sys.dont_write_bytecode = True
sys.path.insert(1, "/root/reference")


def main():
    from zipnn import ZipNN  # the reference package
    KB = 1024
    rng = np.random.default_rng(4321)
    rb = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()      # noqa: E731

    def finetune_pair(n, seed, frac=0.02):
        g = torch.Generator().manual_seed(seed)
        base = (torch.randn(n, generator=g) * 0.02).to(torch.bfloat16)
        new = base.clone()
        idx = torch.randperm(n, generator=g)[: int(n * frac)]
        new[idx] = (new[idx].float() * 1.01 + 1e-4).to(torch.bfloat16)
        tb = lambda t: t.view(torch.uint8).numpy().tobytes()              # noqa: E731
        return tb(new), tb(base)

    a, b, c = rb(10 * KB), rb(10 * KB), rb(10 * KB)
    fa, fb, fc = (rng.random(8 * KB).astype(np.float32) for _ in range(3))
    ft = finetune_pair(300 * KB + 77, 31)
    ft_s = finetune_pair(150 * KB + 5, 32)
    cases = [
        ("delta_byte_bf16_finetune_ragged", ft[0], ft[1], dict(bytearray_dtype="bfloat16", delta_compressed_type="byte")),
        ("delta_byte_reference_test_shape", a + b, a + c, dict(delta_compressed_type="byte")),
        ("delta_byte_streaming", a + b, a + c, dict(delta_compressed_type="byte", is_streaming=True)),
        ("delta_byte_bf16_finetune_streaming_multi", ft_s[0], ft_s[1],
         dict(bytearray_dtype="bfloat16", delta_compressed_type="byte", is_streaming=True, streaming_chunk=1 << 17, compression_chunk=64 * KB)),
        ("delta_byte_float32_streaming", np.concatenate([fa, fb]).tobytes(), np.concatenate([fa, fc]).tobytes(),
         dict(bytearray_dtype="float32", delta_compressed_type="byte", is_streaming=True)),
    ]
    out, meta = {}, []
    for name, data, base, ctor in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            frame = bytes(ZipNN(**ctor).compress(bytearray(data), delta_second_data=bytearray(base)))
            back = bytes(ZipNN(**ctor).decompress(frame, delta_second_data=bytearray(base)))
            plain = bytes(ZipNN(**{k: v for k, v in ctor.items() if k != "delta_compressed_type"}).compress(bytearray(data)))
        assert back == data, f"reference delta round trip failed for {name}"
        out[name + ".frame"] = np.frombuffer(frame, dtype=np.uint8)
        out[name + ".base"] = np.frombuffer(base, dtype=np.uint8)
        meta.append(dict(name=name, ctor=ctor, in_len=len(data), frame_len=len(frame), plain_frame_len=len(plain),
                         in_sha256=hashlib.sha256(data).hexdigest(), frame_sha256=hashlib.sha256(frame).hexdigest()))
        print(f"{name:44s} in={len(data):8d} frame={len(frame):8d} (without delta {len(plain):8d})")
    out["meta.json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "golden_delta_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
