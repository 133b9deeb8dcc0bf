"""Runs the STOCK reference Python package (`/root/reference/zipnn`) over whatever `zipnn_core` module is first on
sys.path — the reference's own compiled extension (oracle/_ref) or this repository's binding (tests/ref_binding/
zipnn_core.py → libzipnn) — on the cases of the reference's tests/simple_stress_tests.py (sizes trimmed so that the
SIMT-emulated kernels finish in seconds; inputs seeded instead of os.urandom so that two runs can be compared) and
prints one JSON object: sha256 of every frame and whether it decoded back to its input.
Called by tests/test_ref_binding.py (not a test module itself) in a subprocess per binding (the module name `zipnn_core` can only mean one thing
per process)."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np
import torch


def main():
    from zipnn import ZipNN      # the reference package
    import zipnn_core
    out = {"zipnn_core": os.path.abspath(zipnn_core.__file__)}
    sha = lambda b: hashlib.sha256(bytes(b)).hexdigest()      # noqa: E731
    rng = np.random.default_rng(123)
    rb = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()      # noqa: E731
    KB = 1024

    def case(name, z, data, back_eq, **kw):
        # (a real copy: the reference core rotates the caller's buffer IN PLACE — even an immutable bytes object — which is
        #  why its own tests copy first, simple_stress_tests.py:44; bytes(x) of a bytes object is x itself)
        keep = data.clone() if isinstance(data, torch.Tensor) else bytearray(data)
        frame = z.compress(data, **kw)
        back = z.decompress(frame, **kw)
        out[name] = {"frame": sha(frame), "len": len(frame), "roundtrip": bool(back_eq(keep, back))}

    teq = lambda a, b: torch.equal(a, b)                      # noqa: E731
    beq = lambda a, b: bytearray(a) == bytearray(b)           # noqa: E731
    g = torch.Generator().manual_seed(9)
    # simple_stress_tests.py:19-70 — bf16 torch tensors around the chunk boundary (255/256/257 K elements), random bytes
    for kb in (255, 256, 257):
        t = (torch.rand(kb * KB, generator=g) * 2 - 1).to(torch.bfloat16)
        case(f"torch_bf16_{kb}k", ZipNN(input_format="torch"), t, teq)
    case("bytes_255k", ZipNN(), rb(255 * KB), beq)
    w = (torch.randn(300 * KB, generator=g) * 0.02).to(torch.bfloat16)
    case("torch_bf16_weights", ZipNN(input_format="torch"), w, teq)
    case("torch_fp32", ZipNN(input_format="torch"), torch.randn(70 * KB, generator=g) * 0.02, teq)
    hc = torch.ones(100, 100); hc[50:] = torch.rand(50, 100, generator=g) * 2 - 1
    case("torch_fp16_half_const", ZipNN(input_format="torch"), hc.to(torch.float16), teq)       # :205-264
    # :72-83 streaming, several chunk sizes; and a blob of several frames
    for sc in (2 ** 19, 2 ** 20):
        case(f"streaming_{sc}", ZipNN(is_streaming=True, streaming_chunk=sc), rb(10 * KB), beq)
    case("streaming_multi_frame", ZipNN(is_streaming=True, streaming_chunk=2 ** 18),
         (torch.randn(400 * KB, generator=g) * 0.02).to(torch.bfloat16).view(torch.uint8).numpy().tobytes(), beq)
    # :85-112 delta (byte), streaming delta
    a, b, c = rb(10 * KB), rb(10 * KB), rb(10 * KB)
    case("delta_byte", ZipNN(delta_compressed_type="byte"), a + b, beq, delta_second_data=a + c)
    case("delta_byte_streaming", ZipNN(delta_compressed_type="byte", is_streaming=True), a + b, beq, delta_second_data=a + c)
    # :114-150 delta from file
    with tempfile.NamedTemporaryFile(delete=False) as f:
        f.write(a + c)
    try:
        case("delta_file", ZipNN(delta_compressed_type="file"), a + b, beq, delta_second_data=f.name)
    finally:
        os.unlink(f.name)
    # :152-203 float32 bytes, streaming float32, streaming delta float32
    fa, fb, fc = (rng.random(8 * KB).astype(np.float32) for _ in range(3))
    case("bytes_float32", ZipNN(bytearray_dtype="float32"), fa.tobytes(), beq)
    case("bytes_float32_streaming", ZipNN(bytearray_dtype="float32", is_streaming=True), fa.tobytes(), beq)
    case("bytes_float32_streaming_delta", ZipNN(bytearray_dtype="float32", is_streaming=True, delta_compressed_type="byte"),
         np.concatenate([fa, fb]).tobytes(), beq, delta_second_data=np.concatenate([fa, fc]).tobytes())
    print("RESULT " + json.dumps(out, sort_keys=True))


if __name__ == "__main__":
    main()
