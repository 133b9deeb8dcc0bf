"""CPU tests (-m "not gpu") that execute the PRODUCT's kernel sources under the SIMT
emulator (tests/simt) and compare with the oracle and the golden frames.  They debug
kernel and host logic where no GPU exists; the GPU parity tests proper are in
tests/test_gpu_parity.py and run the real libzipnn_hip.so."""
import numpy as np
import pytest
import torch

import golden_util as G
import oracle_lib as O
from test_oracle import gen_bytes

HDR = bytes(range(32))
C = 64 * 1024

CASES = [("bf16", 1, 2, 1, 10, C), ("bf16", 2, 2, 1, 10, C), ("bf16", 7, 2, 1, 10, C), ("bf16", 1002, 2, 1, 10, C),
         ("bf16", C, 2, 1, 10, C), ("bf16", C + 6, 2, 1, 10, C), ("fp16", 2 * C + 31, 2, 0, 10, C),
         ("const", 3 * C, 2, 1, 10, C), ("rand", C + 3, 2, 1, 10, C), ("fp32", C + 4, 4, 1, 220, C),
         ("fp32", 1000, 4, 1, 220, C), ("fp8", C + 1, 1, 1, 10, C), ("fp8", 5, 1, 1, 10, C),
         ("rand", 4096, 4, 1, 220, C), ("bf16", 0, 2, 1, 10, C), ("bf16", 256 * 1024 + 2, 2, 1, 10, 256 * 1024)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}-P{c[2]}")
def test_c_abi_matches_oracle(simt_lib, case):
    kind, nb, P, rot, bm, chunk = case
    d = gen_bytes(kind, nb, 3)
    ref = O.compress_frame(HDR, d, P, rot, bm, chunk)
    got = bytes(simt_lib.compress(HDR, d, P, rot, bm, chunk, 0.95))
    assert got == ref                                     # compressed bytes identical
    if nb:
        assert bytes(simt_lib.decompress(ref[32:], P, rot, bm, chunk, nb)) == d


def test_input_buffer_is_not_modified(simt_lib):
    d = bytearray(gen_bytes("bf16", 5000, 1))
    keep = bytes(d)
    simt_lib.compress(HDR, d, 2, 1, 10, C, 0.95)
    assert bytes(d) == keep      # the reference rotates its input in place; we must not


def test_bad_type_byte_and_corrupt_body(simt_lib):
    d = gen_bytes("bf16", 3 * C, 2)
    f = bytearray(O.compress_frame(HDR, d, 2, 1, 10, C))
    bad = bytearray(f); bad[32] = 7
    with pytest.raises(MemoryError):                      # reference: MemoryError("Compress Type is not correct…")
        simt_lib.decompress(bytes(bad[32:]), 2, 1, 10, C, len(d))
    bad = bytearray(f); bad[32 + 3 * 2 + 8 * 3 * 2 + 40] ^= 0x55   # inside plane 0 payload: still decodes (raw) …
    trunc = bytes(f[32:len(f) - 100])
    with pytest.raises(RuntimeError):                     # … but a truncated body must be rejected, not read OOB
        simt_lib.decompress(trunc, 2, 1, 10, C, len(d))
    with pytest.raises(ValueError):
        simt_lib.decompress(bytes(f[32:]), 3, 1, 10, C, len(d))


@pytest.mark.parametrize("name", [n for n in G.names() if "256k" not in n and "257k" not in n and "2chunks" not in n])
def test_zipnn_api_reproduces_golden(use_simt, name):
    """ZipNN(**ctor).decompress(golden) == input and .compress(input) == golden, byte for byte."""
    from zipnn_amd import ZipNN
    meta, blob = G.get(name)
    ctor = dict(meta["ctor"])
    back = ZipNN(**ctor).decompress(blob)
    if meta["kind"] == "torch":
        assert str(back.dtype) == "torch." + meta["dtype"] and list(back.shape) == meta["shape"]
        raw = back.contiguous().view(torch.uint8).numpy().tobytes()
        src = back.clone()
    else:
        raw = bytes(back)
        src = raw
    assert len(raw) == meta["in_len"] and G.sha(raw) == meta["in_sha256"]
    again = bytes(ZipNN(**ctor).compress(src))
    assert G.sha(again) == meta["frame_sha256"]
