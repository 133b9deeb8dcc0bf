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


@pytest.mark.parametrize("case", [("bf16", 1024 * 700 + 10, 2, 1, 10, 1024), ("fp32", 512 * 333 + 4, 4, 1, 220, 512),       # (scan blocks hold ~ 512 entries: three blocks each, every plane start inside one)
                                  ("rand", 256 * 2049, 1, 1, 10, 256)],
                         ids=lambda c: f"{c[0]}-K{c[1] // c[5]}-P{c[2]}")
def test_many_chunks_multi_block_size_scan(simt_lib, case):
    """Thousands of (plane, chunk) entries: the size scan runs over several workgroups and plane starts fall
    inside blocks; the frame must still be the oracle's, byte for byte."""
    kind, nb, P, rot, bm, chunk = case
    d = gen_bytes(kind, nb, 9)
    ref = O.compress_frame(HDR, d, P, rot, bm, chunk)
    assert bytes(simt_lib.compress(HDR, d, P, rot, bm, chunk, 0.95)) == ref
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


@pytest.mark.parametrize("name", G.delta_names())
def test_zipnn_api_reproduces_reference_written_delta_frames(use_simt, name):
    """ZipNN(delta_compressed_type="byte") on frames the REFERENCE wrote in its delta mode (tests/golden/make_golden_delta.py;
    reference zipnn/zipnn.py:625-640, 983-1004): decompress(frame, base) == data and compress(data, base) == frame, byte for byte —
    with the XOR fused into the kernels here instead of a host pass."""
    from zipnn_amd import ZipNN
    meta, blob, base = G.delta_get(name)
    ctor = dict(meta["ctor"])
    back = bytes(ZipNN(**ctor).decompress(blob, delta_second_data=base))
    assert len(back) == meta["in_len"] and G.sha(back) == meta["in_sha256"]
    assert G.sha(bytes(ZipNN(**ctor).compress(back, delta_second_data=base))) == meta["frame_sha256"]


FUSED = [("bf16", 3 * C, 2, 1, 10, C, 3), ("bf16", 2 * C + 100, 2, 1, 10, C, 2), ("fp16", 2 * C, 2, 0, 10, C, 2),
         ("fp32", 2 * C, 4, 1, 220, C, 2), ("fp8", 2 * C, 1, 1, 10, C, 2), ("const", 2 * C, 2, 1, 10, C, 2),
         ("rand", 2 * C, 2, 1, 10, C, 2), ("u11", 2 * C, 2, 1, 10, C, 2), ("skew", 2 * C, 1, 1, 10, C, 2),
         ("burst", 2 * C, 1, 1, 10, C, 2), ("burst16", 4 * C, 2, 0, 10, 2 * C, 2),
         ("bf16", 2 * 4096 * 3, 2, 1, 10, 4096, 6), ("bf16", 256 * 1024, 2, 1, 10, 256 * 1024, 1),
         # every plane Huffman-coded: one accumulate pass per further plane
         ("skew", 2 * C, 2, 0, 10, C, 2), ("skew", 2 * C, 2, 1, 10, C, 2), ("skew", 2 * C, 4, 1, 220, C, 2), ("skew", 65536, 4, 0, 220, 16384, 4)]


def _gen2(kind, nb, seed):
    if kind == "u11":
        g = torch.Generator().manual_seed(seed)
        return (torch.rand(nb // 2, generator=g) * 2 - 1).to(torch.bfloat16).view(torch.uint8).numpy().tobytes()
    if kind == "skew":   # one symbol > 50 %: 1-bit codes, smallest sub-blocks
        r = np.random.default_rng(seed)
        return r.choice(np.array([7, 9, 200, 31, 32, 33], dtype=np.uint8), nb, p=[0.6, 0.2, 0.1, 0.05, 0.03, 0.02]).tobytes()
    if kind == "burst16":  # the same in the even bytes, incompressible odd bytes (one Huffman plane + one raw plane)
        r = np.random.default_rng(seed)
        a = np.frombuffer(_gen2("burst", nb, seed), dtype=np.uint8).copy()
        a[0::2] = a[:nb // 2]; a[1::2] = r.integers(0, 256, nb // 2, dtype=np.uint8)
        return a.tobytes()
    if kind == "burst":  # long runs of a 1-bit symbol between incompressible stretches: tiles far denser than
        r = np.random.default_rng(seed)   # the stream average, so the staging buffer is flushed in lane groups
        a = r.integers(1, 251, nb, dtype=np.uint8)
        blk = np.arange(nb) // 4096
        a[(blk % 5) < 3] = 0
        return a.tobytes()
    return gen_bytes(kind, nb, seed)


@pytest.mark.parametrize("case", FUSED, ids=lambda c: f"{c[0]}-{c[1]}-P{c[2]}-c{c[5]}")
def test_fused_decode_path(simt_lib, case):
    """Full chunks must go through zn_k_decode_fused (further Huffman planes: extra passes) and give the input back."""
    kind, nb, P, rot, bm, chunk, want_fused = case
    d = _gen2(kind, nb, 11)
    frame = O.compress_frame(HDR, d, P, rot, bm, chunk)
    body = torch.frombuffer(bytearray(frame[32:]), dtype=torch.uint8)
    out = torch.empty(nb, dtype=torch.uint8)
    simt_lib.decompress_dev(body.data_ptr(), body.numel(), P, rot, bm, chunk, nb, out.data_ptr())
    assert out.numpy().tobytes() == d
    assert simt_lib.last_fused_chunks() == want_fused
    assert "zn_k_decode_fused" in simt_lib.last_kernels()


@pytest.mark.parametrize("group", [1, 2, 3, 4])
def test_fused_decode_chunk_groups(simt_lib, group, decode_group):
    """A workgroup decodes `group` consecutive chunks; groups may mix Huffman, raw-only, RLE and
    two-Huffman-plane (one extra accumulate pass) chunks, and the last group may be short."""
    decode_group(simt_lib, group)
    ch = 16384
    r = np.random.default_rng(5)
    parts = []
    for k in range(11):
        kind = ["bf16", "rand", "const", "u11", "skewpair", "bf16"][k % 6]
        if kind == "skewpair":   # both planes compressible → two Huffman planes → a second pass of the fused kernel
            parts.append(r.choice(np.array([1, 2, 3, 4], dtype=np.uint8), ch, p=[0.7, 0.1, 0.1, 0.1]).tobytes())
        else:
            parts.append(_gen2(kind, ch, 20 + k))
    d = b"".join(parts) + _gen2("bf16", 1000, 3)           # + a partial tail chunk
    frame = O.compress_frame(HDR, d, 2, 0, 10, ch)
    body = torch.frombuffer(bytearray(frame[32:]), dtype=torch.uint8)
    out = torch.empty(len(d), dtype=torch.uint8)
    simt_lib.decompress_dev(body.data_ptr(), body.numel(), 2, 0, 10, ch, len(d), out.data_ptr())
    assert out.numpy().tobytes() == d
    assert simt_lib.last_fused_chunks() == 11           # every full chunk, incl. the two "skewpair" ones (two passes)


def test_fused_detects_corrupt_stream(simt_lib):
    d = gen_bytes("bf16", 2 * C, 4)
    frame = bytearray(O.compress_frame(HDR, d, 2, 1, 10, C))
    p = G.parse_frame(bytes(frame) if frame[8] in (2, 3) else bytes(frame[:32]) + bytes(frame[32:])) if False else None
    K = 2
    meta = 32 + 9 * 2 * K
    plane0 = int.from_bytes(frame[32 + 2 * K + 8 * (K - 1): 32 + 2 * K + 8 * K], "little")
    # flip bits in the middle of the first Huffman block of plane 1
    frame[meta + plane0 + 5000] ^= 0xFF
    body = torch.frombuffer(bytearray(frame[32:]), dtype=torch.uint8)
    out = torch.empty(2 * C, dtype=torch.uint8)
    try:
        simt_lib.decompress_dev(body.data_ptr(), body.numel(), 2, 1, 10, C, 2 * C, out.data_ptr())
        same = out.numpy().tobytes() == d
    except RuntimeError:
        same = False
    assert not same    # either flagged corrupt or (self-synchronising code) different bytes; never a crash or hang


ENC = [("bf16", 3 * C + 100, 2, 1, 10, C), ("fp16", 2 * C, 2, 0, 10, C), ("fp32", 2 * C, 4, 1, 220, C), ("fp8", 2 * C, 1, 1, 10, C),
       ("u11", 2 * C, 2, 1, 10, C), ("skew", 2 * C, 1, 1, 10, C), ("skew", 2 * C, 2, 0, 10, C), ("rand", 2 * C, 4, 1, 220, C),
       ("const", 2 * C + 2, 2, 1, 10, C), ("bf16", 256 * 1024, 2, 1, 10, 256 * 1024), ("bf16", 3 * 16384, 2, 1, 10, 16384),
       ("bf16", 3 * 8192, 2, 1, 10, 8192),
       # every byte of a plane in ONE bin, in the largest chunks the fused encoders take: the 16-bit halves of the histogram's shared counters at their bound (4 096 / 8 192 counts a column)
       ("const", 2 * 256 * 1024, 2, 1, 10, 256 * 1024), ("const", 512 * 1024, 4, 1, 220, 512 * 1024), ("skew", 256 * 1024, 2, 0, 10, 256 * 1024)]


@pytest.mark.parametrize("onepass", [True, False], ids=["onepass", "four-kernel"])
@pytest.mark.parametrize("case", ENC, ids=lambda c: f"{c[0]}-{c[1]}-P{c[2]}-c{c[5]}")
def test_fused_encode_path_bit_exact(simt_lib, case, onepass):
    """Full chunks go through the one-pass encoder (zn_k_encode_onepass: histogram, table, look-back, emit by one workgroup per chunk; when its
    layout speculation fails — `const`: RLE planes — the four-kernel encoder redoes the call) or, with the knob off, through
    zn_k_encode_stats/_tables/_emit; either way the body must equal the oracle's."""
    kind, nb, P, rot, bm, chunk = case
    d = _gen2(kind, nb, 21)
    want = O.compress_frame(HDR, d, P, rot, bm, chunk)
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8)
    src16 = torch.empty(nb + 64, dtype=torch.uint8)
    off = (-src16.data_ptr()) % 16
    src16[off:off + nb] = src                                    # 16-byte aligned "device" buffer
    cap = simt_lib.compress_bound(nb, P, chunk, 0)
    body = torch.zeros(cap, dtype=torch.uint8)
    simt_lib.set_encode_onepass(onepass)
    try:
        used = simt_lib.compress_dev(src16.data_ptr() + off, nb, P, rot, bm, chunk, 0.95, body.data_ptr(), cap)
    finally:
        simt_lib.set_encode_onepass(1)
    assert body[:used].numpy().tobytes() == want[32:]
    k = simt_lib.last_kernels()
    if chunk % 16384 == 0 and chunk % (8192 * P) == 0 and chunk // P <= 128 * 1024:
        K = (nb + chunk - 1) // chunk
        mis = any(want[32 + i] != 0 for i in range((P - 1) * K))              # a plane in front of the last one is not stored raw: not what weights look like
        if onepass:
            assert "zn_k_encode_onepass" in k
            assert ("speculation failed" in k) == mis
        if not onepass or mis:
            assert "zn_k_encode_stats" in k and "zn_k_encode_emit" in k
        else:
            assert "zn_k_encode_emit" not in k.replace("zn_k_encode_emit+tail", "")


def test_onepass_encoder_look_back_across_many_chunks_and_batches(simt_lib):
    """The one-pass encoder's running sum (decoupled look-back, 64 predecessors per step): tensors of more chunks than one step covers, a batch
    whose tensors each restart the sum, a tensor with a partial last chunk (its ragged planes ride in the three fused launches), two calls in a
    row (the look-back words carry a generation tag and are never zeroed) — bodies equal the oracle's."""
    from zipnn_amd import codec
    c = 16384
    specs = [("bf16", 70 * c, 2, 1, 10, c), ("fp32", 5 * 2 * c + 4 * 77, 4, 1, 220, 2 * c), ("fp8", 66 * c + 5, 1, 1, 10, c), ("fp16", 3 * c, 2, 0, 10, c),
             ("bf16", c, 2, 1, 10, c), ("bf16", 3 * c - 2, 2, 1, 10, c)]            # (70 / 66 chunks: more than the 64 predecessors one look-back step covers)
    data = [_gen2(k, nb, 40 + i) for i, (k, nb, *_r) in enumerate(specs)]
    simt_lib.set_encode_onepass(True)
    for rep in range(1):
        items = [(torch.frombuffer(bytearray(d), dtype=torch.uint8), P, rot, bm, ch, 0.95) for d, (_, _, P, rot, bm, ch) in zip(data, specs)]
        bodies = codec.compress_device_batch(simt_lib, items)
        assert simt_lib.last_kernels().count("zn_k_encode_onepass") == 3        # one launch per plane count
        for b, d, (k, nb, P, rot, bm, ch) in zip(bodies, data, specs):
            assert b.numpy().tobytes() == O.compress_frame(HDR, d, P, rot, bm, ch)[32:], (k, nb)
    try:
        one = codec.compress_device(simt_lib, torch.frombuffer(bytearray(data[0]), dtype=torch.uint8), 2, 1, 10, c, 0.95)
    finally:
        simt_lib.set_encode_onepass(1)
    assert one.numpy().tobytes() == O.compress_frame(HDR, data[0], 2, 1, 10, c)[32:]
    # automatic mode (the default): only bf16-like calls of at least ZN_ONEPASS_MIN_CHUNKS full chunks take the one-pass kernel — not this one
    simt_lib.compress(HDR, data[0][:8 * c], 2, 1, 10, c, 0.95)
    assert "zn_k_encode_onepass" not in simt_lib.last_kernels() and "zn_k_encode_stats" in simt_lib.last_kernels()


def test_batched_decode_matches_per_tensor_decode(simt_lib):
    """zn_decompress_batch_dev: tensors of all plane counts, with tails, tiny and empty ones, in one call."""
    from zipnn_amd import codec
    specs = [("bf16", 3 * C + 10, 2, 1, 10, C), ("fp32", 2 * C + 4, 4, 1, 220, C), ("fp8", C + 1, 1, 1, 10, C),
             ("bf16", 7, 2, 1, 10, C), ("bf16", 0, 2, 1, 10, C), ("fp16", 5 * 16384, 2, 0, 10, 16384),
             ("rand", 4 * C, 2, 1, 10, C), ("const", 2 * C, 2, 1, 10, C), ("fp32", 1000, 4, 1, 220, C)]
    datas, items = [], []
    for i, (kind, nb, P, rot, bm, chunk) in enumerate(specs):
        d = gen_bytes(kind, nb, 30 + i)
        frame = O.compress_frame(HDR, d, P, rot, bm, chunk)
        datas.append(d)
        body = frame[32:]
        items.append((torch.frombuffer(bytearray(body), dtype=torch.uint8) if body else torch.empty(0, dtype=torch.uint8), P, rot, bm, chunk, nb))
    outs = codec.decompress_device_batch(simt_lib, items)
    for d, o in zip(datas, outs):
        assert o.numpy().tobytes() == d
    assert simt_lib.last_fused_chunks() == 3 + 2 + 1 + 5 + 4 + 2      # full chunks of the fused-eligible tensors


TAILS = [("bf16", C + C // 2 + 10, 2, 1, 10, C, 1), ("bf16", C // 2 + 3, 2, 1, 10, C, 1), ("fp32", 2 * C + C // 2 + 4, 4, 1, 220, C, 1),
         ("fp8", C + 20001, 1, 1, 10, C, 1), ("fp16", 3 * C - 2, 2, 0, 10, C, 1), ("skew", C + 30000, 2, 0, 10, C, 2),
         ("burst", 40001, 1, 1, 10, C, 1), ("rand", C + 30000, 2, 1, 10, C, 0), ("bf16", C + 4000, 2, 1, 10, C, 1), ("bf16", C + 800, 2, 1, 10, C, 0)]      # (the last one: a plane of 400 bytes, below ZN_TAIL_WG_MIN_PLANE — the merge workgroup decodes it serially)


@pytest.mark.parametrize("case", TAILS, ids=lambda c: f"{c[0]}-{c[1]}-P{c[2]}")
def test_partial_last_chunk_through_the_parallel_tail_kernel(simt_lib, case, request, monkeypatch):
    """A big partial last chunk: its Huffman planes are decoded by the tail workgroups of zn_k_decode_fused (4 ragged streams into padded
    scratch), merge workgroups at the end of the SAME launch classify the chunk's planes and interleave them (round 5: the two generic launches
    behind every ragged tensor are gone); tiny / raw tails are decoded serially by those workgroups.  Output == input."""
    kind, nb, P, rot, bm, chunk, want_tail_planes = case
    d = _gen2(kind, nb, 17)
    frame = O.compress_frame(HDR, d, P, rot, bm, chunk)
    body = torch.frombuffer(bytearray(frame[32:]), dtype=torch.uint8)
    out = torch.empty(nb, dtype=torch.uint8)
    request.addfinalizer(lambda: simt_lib.set_decode_wide(1))
    simt_lib.set_decode_wide(0)                  # (the emulated device has one CU: automatic mode gives a call this small to the 16-wave form — below)
    simt_lib.decompress_dev(body.data_ptr(), body.numel(), P, rot, bm, chunk, nb, out.data_ptr())
    assert out.numpy().tobytes() == d
    # one launch: tail workgroups at its front, the chunk's merge workgroups at its end, no generic kernels behind it
    assert simt_lib.last_kernels() == "zn_k_decode_fused^rest+tail+merge"
    assert simt_lib.last_tail_planes() == want_tail_planes
    # automatic: a sign-rotated tensor whose full chunks AND tail workgroups find a slot each (an emulated device of 32 CUs here) rides the small-input kernel —
    # tail workgroups at the front of ITS launch, the merge workgroups at its end (round 6)
    simt_lib.set_decode_wide(1)
    monkeypatch.setenv("ZN_SIMT_CUS", "48")
    out.zero_()
    simt_lib.decompress_dev(body.data_ptr(), body.numel(), P, rot, bm, chunk, nb, out.data_ptr())
    assert out.numpy().tobytes() == d
    assert simt_lib.last_tail_planes() == want_tail_planes
    slots = nb // chunk + 4 * P + 32             # full chunks + tail workgroups + merge workgroups
    form = None if not (rot == 1 and P > 1 and nb // chunk >= 1 and slots <= 96) else "zn_k_decode_wide" if slots <= 48 else "zn_k_decode_wide^2"
    assert simt_lib.last_kernels() == (form + "+tail+merge;zn_k_decode_fused^rest" if form else "zn_k_decode_fused^rest+tail+merge")


def test_corrupted_bodies_never_crash(simt_lib):
    """Random damage anywhere in the body (types, cumSizes, tree descriptions, jump tables, streams), truncation
    and over-long claims: decode either reports an error or returns (possibly different) bytes — it never crashes,
    hangs or raises anything but the documented exceptions."""
    r = np.random.default_rng(99)
    cases = [("bf16", 3 * C + 5000, 2, 1, 10, C), ("fp32", 2 * C + 4, 4, 1, 220, C), ("fp8", C + 20001, 1, 1, 10, C),
             ("skew", C + 30000, 2, 0, 10, C)]
    for kind, nb, P, rot, bm, chunk in cases:
        d = _gen2(kind, nb, 23)
        body = bytearray(O.compress_frame(HDR, d, P, rot, bm, chunk)[32:])
        K = (nb + chunk - 1) // chunk
        meta = 9 * P * K
        for trial in range(24):
            b = bytearray(body)
            mode = trial % 4
            if mode == 0:                                   # metadata byte
                b[int(r.integers(0, meta))] ^= int(r.integers(1, 256))
            elif mode == 1:                                 # start of a payload block (tree description / jump table)
                b[meta + int(r.integers(0, min(200, len(b) - meta)))] ^= int(r.integers(1, 256))
            elif mode == 2:                                 # anywhere
                for _ in range(8):
                    b[int(r.integers(0, len(b)))] ^= int(r.integers(1, 256))
            else:                                           # truncate
                b = b[: int(r.integers(1, len(b)))]
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
            out = torch.empty(nb, dtype=torch.uint8)
            try:
                simt_lib.decompress_dev(t.data_ptr(), t.numel(), P, rot, bm, chunk, nb, out.data_ptr())
            except (RuntimeError, MemoryError, ValueError):
                pass


def test_batched_compress_matches_per_tensor_frames(simt_lib):
    """zn_compress_batch_dev: tensors of all plane counts, tails, tiny / empty ones, geometries the fused encoder
    does not take — every body equals the oracle's frame body; the batch then decodes back in one call."""
    from zipnn_amd import codec
    specs = [("bf16", 3 * C + 10, 2, 1, 10, C), ("fp32", 2 * C + 4, 4, 1, 220, C), ("fp8", C + 1, 1, 1, 10, C),
             ("bf16", 7, 2, 1, 10, C), ("bf16", 0, 2, 1, 10, C), ("fp16", 5 * 16384, 2, 0, 10, 16384),
             ("rand", 4 * C, 2, 1, 10, C), ("const", 2 * C, 2, 1, 10, C), ("fp32", 1000, 4, 1, 220, C),
             ("bf16", 4096 * 5 + 2, 2, 1, 10, 4096), ("skew", C + 30000, 2, 0, 10, C)]
    datas, items = [], []
    for i, (kind, nb, P, rot, bm, chunk) in enumerate(specs):
        d = _gen2(kind, nb, 60 + i)
        datas.append(d)
        items.append((torch.frombuffer(bytearray(d), dtype=torch.uint8) if d else torch.empty(0, dtype=torch.uint8), P, rot, bm, chunk, 0.95))
    bodies = codec.compress_device_batch(simt_lib, items)
    for (kind, nb, P, rot, bm, chunk), d, b in zip(specs, datas, bodies):
        assert b.numpy().tobytes() == O.compress_frame(HDR, d, P, rot, bm, chunk)[32:], kind
    outs = codec.decompress_device_batch(simt_lib, [(b, P, rot, bm, chunk, nb) for b, (kind, nb, P, rot, bm, chunk) in zip(bodies, specs)])
    for d, o in zip(datas, outs):
        assert o.numpy().tobytes() == d


def test_batch_entry_points_reject_bad_items(simt_lib):
    """An invalid item fails the whole batch with the same error a single call gives; nothing is decoded / written."""
    d = gen_bytes("bf16", 3 * C, 1)
    src = torch.frombuffer(bytearray(d), dtype=torch.uint8)
    body = torch.empty(simt_lib.compress_bound(len(d), 2, C, 0), dtype=torch.uint8)
    good = (src.data_ptr(), len(d), 2, 1, 10, C, 0.95, body.data_ptr(), body.numel())
    with pytest.raises(RuntimeError):                      # capacity below zn_compress_bound
        simt_lib.compress_batch_dev([good, (src.data_ptr(), len(d), 2, 1, 10, C, 0.95, body.data_ptr(), 100)])
    with pytest.raises(ValueError):                        # 3 planes do not exist
        simt_lib.compress_batch_dev([good, (src.data_ptr(), len(d), 3, 1, 10, C, 0.95, body.data_ptr(), body.numel())])
    assert simt_lib.compress_batch_dev([good]) == [len(O.compress_frame(HDR, d, 2, 1, 10, C)) - 32]
    out = torch.empty(len(d), dtype=torch.uint8)
    n = simt_lib.compress_batch_dev([good])[0]
    with pytest.raises(RuntimeError):                      # body shorter than its own metadata
        simt_lib.decompress_batch_dev([(body.data_ptr(), n, 2, 1, 10, C, len(d), out.data_ptr()), (body.data_ptr(), 5, 2, 1, 10, C, len(d), out.data_ptr())])
    simt_lib.decompress_batch_dev([(body.data_ptr(), n, 2, 1, 10, C, len(d), out.data_ptr())])
    assert out.numpy().tobytes() == d


def test_randomised_geometry_sweep_bit_exact(simt_lib):
    """24 random (distribution, size, planes, rotate, chunk size, threshold) combinations under the emulator:
    frame == oracle frame, decode == input, per call and batched."""
    from test_gpu_parity import _random_cases
    from zipnn_amd import codec
    cases = _random_cases(24, 7, 70000)
    datas, frames = [], []
    for i, (kind, nb, P, rot, bm, chunk, thr) in enumerate(cases):
        d = _gen2(kind, nb, 500 + i)
        nb = len(d)                                            # 'u11' rounds down to whole bf16 values
        cases[i] = (kind, nb, P, rot, bm, chunk, thr)
        want = O.compress_frame(HDR, d, P, rot, bm, chunk, thr)
        assert bytes(simt_lib.compress(HDR, d, P, rot, bm, chunk, thr)) == want, (i, cases[i])
        if nb:
            assert bytes(simt_lib.decompress(want[32:], P, rot, bm, chunk, nb)) == d, (i, cases[i])
        datas.append(d); frames.append(want)
    flats = [torch.frombuffer(bytearray(d), dtype=torch.uint8) if d else torch.empty(0, dtype=torch.uint8) for d in datas]
    bodies = codec.compress_device_batch(simt_lib, [(f, P, rot, bm, chunk, thr) for f, (kind, nb, P, rot, bm, chunk, thr) in zip(flats, cases)])
    for i, (b, fr) in enumerate(zip(bodies, frames)):
        assert b.numpy().tobytes() == fr[32:], (i, cases[i])
    outs = codec.decompress_device_batch(simt_lib, [(b, P, rot, bm, chunk, nb) for b, (kind, nb, P, rot, bm, chunk, thr) in zip(bodies, cases)])
    for i, (o, d) in enumerate(zip(outs, datas)):
        assert o.numpy().tobytes() == d, (i, cases[i])


def _delta_pair(kind, nb, seed):
    """(tensor bytes, base bytes): `fine-tuned` = base with a sparse perturbation, so the XOR is mostly zero bytes."""
    a = np.frombuffer(_gen2(kind, nb, seed), dtype=np.uint8).copy()
    r = np.random.default_rng(seed)
    b = a.copy()
    hit = r.random(len(a)) < 0.03
    b[hit] ^= r.integers(1, 256, int(hit.sum()), dtype=np.uint8)
    return a.tobytes(), b.tobytes()


DELTA = [("bf16", 2 * C + 1234, 2, 1, 10, C), ("fp32", C + 4 * 77, 4, 1, 220, C), ("fp8", 3 * 65536 + 5, 1, 1, 10, 65536),
         ("bf16", 70001, 2, 1, 10, 4442), ("fp16", 2 * C, 2, 0, 10, C), ("rand", C, 2, 1, 10, C)]


@pytest.mark.parametrize("case", DELTA, ids=lambda c: f"{c[0]}-{c[1]}-P{c[2]}-c{c[5]}")
def test_delta_xor_fused_into_the_kernels(simt_lib, case):
    """compress(data, delta=base) == oracle frame of data ^ base (reference XORs on the host first, zipnn.py:625-640);
    decompress(frame, delta=base) == data.  Fused and generic kernels, host and device entry points, an unaligned base."""
    from zipnn_amd import codec
    kind, nb, P, rot, bm, chunk = case
    a, b = _delta_pair(kind, nb, 31)
    nb = len(a)
    x = (np.frombuffer(a, dtype=np.uint8) ^ np.frombuffer(b, dtype=np.uint8)).tobytes()
    want = O.compress_frame(HDR, x, P, rot, bm, chunk)
    assert bytes(simt_lib.compress(HDR, a, P, rot, bm, chunk, 0.95, delta=b)) == want
    if nb >= chunk and chunk % 16384 == 0:
        assert "zn_k_encode_emit^delta" in simt_lib.last_kernels()
    assert bytes(simt_lib.decompress(want[32:], P, rot, bm, chunk, nb, delta=b)) == a
    assert "zn_k_decode_fused^delta" in simt_lib.last_kernels()
    if chunk % 16384 == 0:
        assert simt_lib.last_fused_chunks() == nb // chunk     # (a sparse delta: every plane is Huffman-coded)
    # device entry points; the base at an odd address: every chunk is coded by the ragged-plane workgroups
    ta = torch.frombuffer(bytearray(a), dtype=torch.uint8)
    pad = torch.zeros(nb + 1, dtype=torch.uint8); pad[1:] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    tb = pad[1:]
    body = codec.compress_device(simt_lib, ta, P, rot, bm, chunk, 0.95, delta=tb)
    assert body.numpy().tobytes() == want[32:]
    assert "zn_k_encode_emit^delta+tail" in simt_lib.last_kernels()        # (every plane by the tail workgroups of the fused launches)
    out = codec.decompress_device(simt_lib, body, P, rot, bm, chunk, nb, delta=tb)
    assert out.numpy().tobytes() == a
    assert simt_lib.last_fused_chunks() == 0
    with pytest.raises(ValueError):
        codec.compress_device(simt_lib, ta, P, rot, bm, chunk, 0.95, delta=tb[:-1])


def test_delta_in_a_batch_is_per_tensor(simt_lib):
    """A batch mixes tensors with and without a base (and plane counts): every frame equals its own oracle frame."""
    from zipnn_amd import codec
    specs = [("bf16", 2 * C, 2, 1, 10, C, True), ("bf16", C + 10, 2, 1, 10, C, False), ("fp32", C, 4, 1, 220, C, True),
             ("fp8", 65536 * 2, 1, 1, 10, 65536, False), ("fp16", 1000, 2, 0, 10, C, True)]
    items, wants, datas = [], [], []
    for i, (kind, nb, P, rot, bm, chunk, has) in enumerate(specs):
        a, b = _delta_pair(kind, nb, 40 + i)
        x = (np.frombuffer(a, dtype=np.uint8) ^ np.frombuffer(b, dtype=np.uint8)).tobytes() if has else a
        wants.append(O.compress_frame(HDR, x, P, rot, bm, chunk)[32:])
        ta = torch.frombuffer(bytearray(a), dtype=torch.uint8)
        tb = torch.frombuffer(bytearray(b), dtype=torch.uint8) if has else None
        items.append((ta, P, rot, bm, chunk, 0.95, tb)); datas.append(a)
    bodies = codec.compress_device_batch(simt_lib, items)
    for b, w in zip(bodies, wants):
        assert b.numpy().tobytes() == w
    outs = codec.decompress_device_batch(simt_lib, [(b, P, rot, bm, chunk, len(a), tb)
                                                    for b, (ta, P, rot, bm, chunk, thr, tb), a in zip(bodies, items, datas)])
    for o, a in zip(outs, datas):
        assert o.numpy().tobytes() == a


def test_wrapped_cum_sizes_are_rejected_without_leaving_the_body(simt_lib):
    """cumSizes come out of an untrusted frame.  A plane-0 total of 2^64 - 9PK - 4096 used to wrap the
    64-bit payload base of plane 1 to 4096 bytes BEFORE the body (m.ok stayed true, the Huffman header was read out of
    bounds).  Every crafted entry below must be rejected as corrupt — under the emulator an out-of-bounds read of that
    size would fault or trip the RuntimeError for the wrong reason, so the body is also placed at the start of a page-
    aligned mapping with nothing readable before it."""
    import mmap
    d = gen_bytes("bf16", 2 * C, 5)
    f = bytearray(O.compress_frame(HDR, d, 2, 1, 10, C))
    body = bytearray(f[32:])
    P_, K = 2, 2
    PK = P_ * K
    for wrapped in ((1 << 64) - 9 * PK - 4096, (1 << 64) - 9 * PK, (1 << 64) - 1, (1 << 63)):
        bad = bytearray(body)
        bad[PK + 8 * (0 * K + K - 1): PK + 8 * (0 * K + K - 1) + 8] = int(wrapped).to_bytes(8, "little")   # last cumSize of plane 0
        m = mmap.mmap(-1, (len(bad) + 4095) // 4096 * 4096)
        m.write(bytes(bad))
        with pytest.raises(RuntimeError):
            simt_lib.decompress(memoryview(m)[:len(bad)], 2, 1, 10, C, len(d))
        del m
    # a chunk entry below its predecessor (non-monotonic) and one past the end of the body
    for p_, c_, val in ((0, 1, 3), (1, 1, len(body) * 2)):
        bad = bytearray(body)
        bad[PK + 8 * (p_ * K + c_): PK + 8 * (p_ * K + c_) + 8] = int(val).to_bytes(8, "little")
        with pytest.raises(RuntimeError):
            simt_lib.decompress(bytes(bad), 2, 1, 10, C, len(d))


def _tile_counters(reset=True):
    import ctypes, os
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt", "libzipnn_simt.so")
    raw = ctypes.CDLL(so)
    a = (ctypes.c_ulonglong * 8)()
    raw.zn_debug_tile_counters(a, 1 if reset else 0)
    return list(a)       # [tiles, tiles in the looping form, fix-up iterations, tiles written in several lane groups]


@pytest.mark.parametrize("kind,P,rot,bm,chunk", [("bf16", 2, 1, 10, 256 * 1024), ("fp32", 4, 1, 220, 256 * 1024), ("bf16", 2, 1, 10, 128 * 1024),
                                                    ("fp16", 2, 0, 10, 256 * 1024), ("fp8", 1, 0, 10, 128 * 1024)],
                         ids=["bf16-256k", "fp32-256k", "bf16-128k", "fp16-256k-dense-code", "fp8-128k-dense-code"])
def test_register_resident_form_decodes_weights_like_tensors(simt_lib, kind, P, rot, bm, chunk):
    """The fast form of the fused decoder (decode once into registers, compact): weights-like tensors at the default
    chunk size must go through it tile for tile (counters of the emulated build), including its fix-up iterations,
    and decode to the input bit for bit."""
    g = torch.Generator().manual_seed(11)
    n = 3 * chunk
    if kind == "fp8":
        x = (torch.randn(n, generator=g) * 0.02).to(torch.float8_e4m3fn)
    else:
        x = (torch.randn(n // (4 if kind == "fp32" else 2), generator=g) * 0.02).to({"fp32": torch.float32, "fp16": torch.float16}.get(kind, torch.bfloat16))
    d = x.view(torch.uint8).numpy().tobytes()
    ref = O.compress_frame(HDR, d, P, rot, bm, chunk)
    _tile_counters()
    assert bytes(simt_lib.decompress(ref[32:], P, rot, bm, chunk, len(d))) == d
    tiles, looping, fixups, groups = _tile_counters()[:4]
    if kind in ("fp16", "fp8"):     # dense codes (5-6 bits a symbol, none shorter than 4): 6-dword sub-blocks in the packed-record instance (two steps
        assert tiles > 20 and looping * 20 <= tiles and groups == 0      # to a register); a few tiles may overflow the step slots and take the looping form
    else:
        assert tiles > 20 and looping == 0 and groups == 0    # every tile took the register-resident form
    if kind in ("fp16", "fp8"):
        assert fixups > 0                                     # … and some of them needed a fix-up iteration: fp16's top-byte code re-synchronises
                                                              # slowly (0.2-0.3 fix-ups per tile at the 44-bit run-in; bf16 / fp32: a few in 1000 tiles)


@pytest.mark.parametrize("kind", ["skew", "burst", "dense3", "sparse", "onebit"])
def test_looping_form_still_decodes_what_the_fast_form_leaves(simt_lib, kind):
    """Distributions whose sub-block size is not a compile-time one (short codes, dense tiles, long sub-blocks with a short code) and
    tiles denser than one staging buffer go through the looping form (and its lane groups): same bytes."""
    chunk = 256 * 1024
    r = np.random.default_rng(3)
    if kind in ("skew", "burst"):
        d = _gen2(kind, 2 * chunk, 3); P, rot = 2, 0      # (one plane of 256 KiB would exceed huff0's 128 KiB block: stored raw)
    elif kind == "dense3":   # ≈ 5 bits a symbol but with one 3-bit code: wants long sub-blocks, is not "dense" (three symbols fit a window): run-time D, looping form
        probs = np.array([0.14] + [0.86 / 60] * 60); probs /= probs.sum()
        d = r.choice(np.arange(61, dtype=np.uint8), 2 * chunk, p=probs).tobytes(); P, rot = 1, 0; chunk = 128 * 1024
    elif kind == "sparse":
        b = np.zeros(2 * chunk, dtype=np.uint8); m = r.random(2 * chunk) < 0.08; b[m] = r.integers(0, 255, int(m.sum())); d = b.tobytes(); P, rot = 2, 1
    else:
        d = (r.random(2 * chunk) < 0.5).astype(np.uint8).tobytes(); P, rot = 2, 0
    ref = O.compress_frame(HDR, d, P, rot, 10, chunk)
    _tile_counters()
    assert bytes(simt_lib.decompress(ref[32:], P, rot, 10, chunk, len(d))) == d
    tiles, looping, _, groups = _tile_counters()[:4]
    assert tiles > 0 and looping > 0
    if kind == "burst":
        assert groups > 0                                       # tiles denser than the staging buffer: several lane groups


def test_devices_do_not_serialise_each_other(simt_lib):
    """ADVICE r1 / VERDICT r1 (multi-GPU readiness): the library used to hold ONE process-wide mutex across every call, so
    eight threads driving eight GPUs of a node ran one at a time.  The workspace lock is per device now.  The emulated
    build has two "devices" (an ordinal that is current per host thread) and a hook that holds one device's lock: a call
    on the OTHER device must finish while it is held, a call on the SAME device must wait for it."""
    import ctypes, os, threading, time
    raw = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt", "libzipnn_simt.so"))
    raw.zn_debug_hold_device_lock.argtypes = [ctypes.c_int, ctypes.c_int]
    d = gen_bytes("bf16", 2 * C, 21)
    ref = O.compress_frame(HDR, d, 2, 1, 10, C)
    assert bytes(simt_lib.decompress(ref[32:], 2, 1, 10, C, len(d), device=1)) == d       # (warm both devices' workspaces)
    assert bytes(simt_lib.decompress(ref[32:], 2, 1, 10, C, len(d), device=0)) == d
    HOLD = 1500
    done_at = {}

    def holder():
        raw.zn_debug_hold_device_lock(0, HOLD)
        done_at["hold"] = time.perf_counter()

    def worker(dev):
        assert bytes(simt_lib.decompress(ref[32:], 2, 1, 10, C, len(d), device=dev)) == d
        assert bytes(simt_lib.compress(HDR, d, 2, 1, 10, C, 0.95, device=dev)) == ref
        done_at[dev] = time.perf_counter()

    t0 = time.perf_counter()
    th = threading.Thread(target=holder); th.start()
    time.sleep(0.1)                                            # the holder has the lock of device 0 by now
    w1 = threading.Thread(target=worker, args=(1,)); w0 = threading.Thread(target=worker, args=(0,))
    w1.start(); w0.start()
    for t in (th, w1, w0):
        t.join(timeout=120)
    assert done_at[1] - t0 < HOLD / 1000 * 0.8                 # device 1 did not wait for device 0's lock
    assert done_at[0] >= done_at["hold"] - 0.01                # device 0 did
    with pytest.raises(RuntimeError):                          # and an ordinal that does not exist is an error, not device 0
        simt_lib.decompress(ref[32:], 2, 1, 10, C, len(d), device=5)


@pytest.mark.parametrize("kind,P,rot,bm,chunk,n", [("bf16", 2, 1, 10, C, 7 * C + 1234), ("fp32", 4, 1, 220, C, 5 * C + 4 * 77), ("fp8w", 1, 0, 10, C // 2, 9 * (C // 2) + 5),
                                                    ("bf16", 2, 1, 10, C, C // 2 + 3), ("bf16", 2, 1, 10, C, 2 * C)],
                         ids=["bf16-7.x-chunks", "fp32-5.x-chunks", "fp8-9.x-chunks", "bf16-one-partial-chunk", "bf16-two-chunks"])
@pytest.mark.parametrize("devices", [[0, 1], [1, 0, 1], [0, 0], [0, 1, 0, 1, 0]], ids=["2-devices", "3-ranges", "same-device-twice", "more-ranges-than-chunks-sometimes"])
def test_multi_device_entry_points_give_the_single_device_frame(simt_lib, kind, P, rot, bm, chunk, n, devices):
    """zn_compress_multi / zn_decompress_multi (SURVEY §8b's `devices, ndev` boundary): the chunk ranges are coded on the
    listed devices by one host thread each (two emulated devices here) and the host does the plane-major bookkeeping —
    the frame must be the oracle's byte for byte, and any frame must decode to the input, for ranges that are empty,
    partial-chunk-only, or more numerous than the chunks."""
    if kind == "fp8w":
        g = torch.Generator().manual_seed(9)
        d = (torch.randn(n, generator=g) * 0.02).to(torch.float8_e4m3fn).view(torch.uint8).numpy().tobytes()
    else:
        d = gen_bytes(kind, n, 31)
    ref = O.compress_frame(HDR, d, P, rot, bm, chunk)
    got = simt_lib.compress_multi(HDR, d, P, rot, bm, chunk, 0.95, devices)
    assert bytes(got) == ref
    assert bytes(simt_lib.decompress_multi(ref[32:], P, rot, bm, chunk, len(d), devices)) == d


def test_multi_device_decompress_rejects_malformed_size_tables(simt_lib):
    """The host splits the body by its cumSizes before any device sees it: entries that decrease or point past the
    payload must come back as ZN_E_CORRUPT (RuntimeError), not as an out-of-bounds memcpy on the host."""
    d = gen_bytes("bf16", 4 * C, 33)
    ref = bytearray(O.compress_frame(HDR, d, 2, 1, 10, C)[32:])
    K = 4
    bad = bytearray(ref); bad[2 * K + 8 * 1: 2 * K + 8 * 2] = (1 << 40).to_bytes(8, "little")          # plane 0, chunk 1: far past the payload
    with pytest.raises(RuntimeError):
        simt_lib.decompress_multi(bytes(bad), 2, 1, 10, C, len(d), [0, 1])
    bad = bytearray(ref); bad[2 * K + 8 * 2: 2 * K + 8 * 3] = (0).to_bytes(8, "little")                 # decreasing
    with pytest.raises(RuntimeError):
        simt_lib.decompress_multi(bytes(bad), 2, 1, 10, C, len(d), [0, 1])
    with pytest.raises(RuntimeError):                                                                     # too short for its own tables
        simt_lib.decompress_multi(bytes(ref[:30]), 2, 1, 10, C, len(d), [0, 1])


@pytest.mark.parametrize("kind,P,rot,bm,chunk,n", [("bf16", 2, 1, 10, C, 7 * C + 1234), ("fp32", 4, 1, 220, C, 5 * C + 4 * 77), ("bf16", 2, 1, 10, C, C // 2 + 3)],
                         ids=["bf16-7.x-chunks", "fp32-5.x-chunks", "bf16-one-partial-chunk"])
@pytest.mark.parametrize("devices", [[0, 1], [1, 0, 1], [0, 0], [0, 1, 0, 1, 0]], ids=["2-devices", "3-ranges", "same-device-twice", "more-ranges-than-chunks-sometimes"])
def test_multi_device_entries_with_the_tensor_resident_per_device(simt_lib, kind, P, rot, bm, chunk, n, devices):
    """zn_decompress_multi_dev / zn_compress_multi_dev: device i holds the bytes of its chunk range (zn_multi_range) in its own
    memory — what a sharded loader / saver has.  Host body in, ranges decoded in place on the devices (nothing decoded crosses
    the host); per-device ranges in, the oracle's frame out.  (Emulated devices: their 'HBM' is host memory.)"""
    d = gen_bytes(kind, n, 41)
    ref = O.compress_frame(HDR, d, P, rot, bm, chunk)
    G = len(devices)
    rng = [simt_lib.multi_range(n, chunk, G, i) for i in range(G)]
    assert sum(l for _, l in rng) == n and all(rng[i][0] + rng[i][1] == rng[i + 1][0] for i in range(G - 1) if rng[i + 1][1])
    outs = [torch.zeros(max(l, 1), dtype=torch.uint8) for _, l in rng]
    simt_lib.decompress_multi_dev(ref[32:], P, rot, bm, chunk, n, devices, [o.data_ptr() if l else 0 for o, (_, l) in zip(outs, rng)])
    assert b"".join(o.numpy().tobytes()[:l] for o, (_, l) in zip(outs, rng)) == d
    parts = [torch.frombuffer(bytearray(d[o:o + l]) or bytearray(1), dtype=torch.uint8) for o, l in rng]
    got = simt_lib.compress_multi_dev(HDR, [p.data_ptr() if l else 0 for p, (_, l) in zip(parts, rng)], n, P, rot, bm, chunk, 0.95, devices)
    assert bytes(got) == ref
    with pytest.raises(ValueError):                          # a non-empty range without a destination
        simt_lib.decompress_multi_dev(ref[32:], P, rot, bm, chunk, n, devices, [0] * G)


@pytest.fixture()
def host_slices():
    """Force the host path's pipeline slices (zn_set_host_slices) for one test; automatic again afterwards."""
    used = []

    def set_(lib, s):
        lib.set_host_slices(s)
        used.append(lib)
    yield set_
    for lib in used:
        lib.set_host_slices(0)


@pytest.mark.parametrize("kind,P,rot,bm,chunk,n", [("bf16", 2, 1, 10, C, 7 * C + 1234), ("fp32", 4, 1, 220, C, 5 * C + 4 * 77), ("fp8", 1, 1, 10, C // 2, 9 * (C // 2) + 5),
                                                    ("bf16", 2, 1, 10, C, 2 * C), ("rand", 2, 1, 10, C, 6 * C + 1), ("const", 2, 1, 10, C, 4 * C)],
                         ids=["bf16-7.x", "fp32-5.x", "fp8-9.x", "bf16-2-chunks", "incompressible", "rle"])
@pytest.mark.parametrize("slices", [2, 3, 5, 64])
def test_host_entry_points_pipelined_over_slices(simt_lib, host_slices, kind, P, rot, bm, chunk, n, slices):
    """zn_compress / zn_decompress with the three-stage pipeline (upload | code | download over slices of the chunks) forced on
    small inputs: the frame is the oracle's byte for byte — the slices' plane-0 payload placed as the slices finish, the later
    planes at the end, cumSizes re-based — and any frame decodes back, slice by slice, into the caller's buffer."""
    host_slices(simt_lib, slices)
    d = gen_bytes(kind, n, 51)
    ref = O.compress_frame(HDR, d, P, rot, bm, chunk)
    got = simt_lib.compress(HDR, d, P, rot, bm, chunk, 0.95)
    assert bytes(got) == ref
    assert bytes(simt_lib.decompress(ref[32:], P, rot, bm, chunk, len(d))) == d
    with pytest.raises(RuntimeError):                              # a truncated body is refused before any slice is uploaded
        simt_lib.decompress(ref[32:len(ref) - 50], P, rot, bm, chunk, len(d))


def test_range_entry_points_reject_bad_arguments_and_damaged_tables(simt_lib):
    """zn_decompress_range_dev / zn_decompress_multi_dev / zn_merge_range_bodies validate before they copy: chunk ranges outside the
    tensor, size tables that decrease or point past the payload, parts shorter than their own tables or whose cumSizes disagree with
    their length — errors, never an out-of-bounds copy on the host."""
    d = gen_bytes("bf16", 6 * C + 100, 61)
    body = O.compress_frame(b"", d, 2, 1, 10, C)
    K = 7
    out = torch.zeros(len(d), dtype=torch.uint8)
    for lo, hi in ((3, 2), (0, K + 1), (K + 1, K + 2)):
        with pytest.raises(ValueError):
            simt_lib.decompress_range_dev(body, 2, 1, 10, C, len(d), lo, hi, 0, out.data_ptr())
    simt_lib.decompress_range_dev(body, 2, 1, 10, C, len(d), 2, 2, 0, 0)                 # an empty range needs no destination
    bad = bytearray(body); bad[2 * K + 8 * 3: 2 * K + 8 * 4] = (1 << 50).to_bytes(8, "little")
    with pytest.raises(RuntimeError):
        simt_lib.decompress_range_dev(bytes(bad), 2, 1, 10, C, len(d), 0, 3, 0, out.data_ptr())
    with pytest.raises(RuntimeError):
        simt_lib.decompress_multi_dev(bytes(bad), 2, 1, 10, C, len(d), [0, 1], [out.data_ptr(), out.data_ptr()])
    with pytest.raises(RuntimeError):
        simt_lib.decompress_range_dev(bytes(body[:40]), 2, 1, 10, C, len(d), 0, 3, 0, out.data_ptr())
    from zipnn_amd import sharding
    parts = sharding.split_body(body, 2, C, len(d), 2)
    ks = [hi - lo for lo, hi in sharding.chunk_ranges(K, 2)]
    good = [(parts[0][0], ks[0]), (parts[1][0], ks[1])]
    assert bytes(simt_lib.merge_range_bodies(good, 2)) == body
    with pytest.raises(ValueError):                                                       # a part shorter than its own size tables
        simt_lib.merge_range_bodies([(parts[0][0][:10], ks[0]), good[1]], 2)
    with pytest.raises(RuntimeError):                                                     # … or whose payload is not what its cumSizes say
        simt_lib.merge_range_bodies([(parts[0][0][:-5], ks[0]), good[1]], 2)
    with pytest.raises(ValueError):
        simt_lib.merge_range_bodies(good, 3)


def _varied_planes(nchunks, chunk, seed):
    """One byte plane per chunk, every chunk with its own symbol statistics (alphabet size, shape, which byte values):
    what drives the tree description through its cases — FSE-coded weights with few / many weight classes, the
    secondary normalisation, raw 4-bit weights, descriptions that are not kept."""
    rng = np.random.default_rng(seed)
    out = np.empty(nchunks * chunk, dtype=np.uint8)
    for c in range(nchunks):
        shape = c % 6
        A = int(rng.integers(2, 257))
        syms = rng.permutation(256)[:A] if (c % 3) else np.arange(A) + int(rng.integers(0, 257 - A))
        if shape == 0:                                   # geometric
            p = float(rng.uniform(0.5, 0.98)) ** np.arange(A)
        elif shape == 1:                                 # power law
            p = 1.0 / (np.arange(A) + 1.0) ** float(rng.uniform(0.6, 2.5))
        elif shape == 2:                                 # near uniform
            p = np.ones(A) + rng.uniform(0, 0.3, A)
        elif shape == 3:                                 # a few heavy symbols + a long tail of rare ones
            p = np.full(A, 1e-4); p[: max(1, A // 16)] = 1.0
        elif shape == 4:                                 # two plateaus
            p = np.where(np.arange(A) < A // 2, 8.0, 1.0)
        else:                                            # gaussian bump (an exponent plane)
            p = np.exp(-0.5 * ((np.arange(A) - A / 2) / max(1.0, A / float(rng.uniform(4, 12)))) ** 2) + 1e-6
        p = p / p.sum()
        out[c * chunk:(c + 1) * chunk] = syms[rng.choice(A, size=chunk, p=p)].astype(np.uint8)
    return out.tobytes()


def _dyadic_plane(lengths, absent, chunk, seed):
    """A plane whose Huffman code lengths are exactly `lengths` (a complete code: Σ 2^-l = 1; symbol i occurs
    chunk · 2^-l_i times), over ascending byte values with `absent` unused values in between — the weight histogram
    of its tree description is then known in advance."""
    rng = np.random.default_rng(seed)
    A = len(lengths)
    assert sum(chunk >> l for l in lengths) == chunk
    gaps = set(rng.choice(np.arange(1, A + absent - 1), size=absent, replace=False).tolist()) if absent else set()
    vals = [v for v in range(A + absent) if v not in gaps]
    order = rng.permutation(A)                                      # which byte value gets which length ...
    order = np.concatenate([order[order != np.argmax(lengths)], [np.argmax(lengths)]])   # ... the highest value one of the longest
    plane = np.concatenate([np.full(chunk >> lengths[i], vals[j], dtype=np.uint8) for j, i in enumerate(order)])
    return rng.permutation(plane).tobytes()


def _tree_description_planes(chunk=16384):
    """Planes (one per chunk, P = 1) that take the tree description through its forms: see the test below."""
    rng = np.random.default_rng(9)
    parts = [_varied_planes(180, chunk, 5)]
    for k in (7, 8, 9, 10, 11):                                      # lengths 1, 2, …, k, k with a few unused values: reaches M2
        for absent in (2, 3):
            parts.append(_dyadic_plane(list(range(1, k + 1)) + [k], absent, chunk, 100 + 10 * k + absent))
    parts.append(_dyadic_plane([1, 3, 3, 3, 4, 5, 6, 7, 9, 10, 10, 10, 11, 11, 11, 11, 11, 11], 4, chunk, 7))
    parts.append(_dyadic_plane([1, 2, 3, 4, 5, 6, 8, 9, 9, 9, 9, 10, 11, 11, 11, 11, 11, 11], 4, chunk, 8))
    for i in range(40):                                              # small alphabets: raw 4-bit weights
        A = int(rng.integers(3, 40)); lo = int(rng.integers(0, 90))
        p = rng.uniform(0.2, 1.0, A) ** float(rng.uniform(1, 4)); p /= p.sum()
        parts.append((lo + rng.choice(A, size=chunk, p=p)).astype(np.uint8).tobytes())
    # every weight the same (HUF_compressWeights returns 1): 16 and 128 values equally often take the raw 4-bit form; all 256 (more than
    # 128 weights, no FSE form) make HUF_writeCTable fail — the plane is stored; and a code with one symbol per weight class (returns 0)
    for A in (16, 128, 256):
        parts.append(rng.permutation(np.repeat(np.arange(A, dtype=np.uint8), chunk // A)).tobytes())
    parts.append(_dyadic_plane([1, 2, 3, 4, 5, 6, 7, 8, 8], 0, chunk, 3))
    return b"".join(parts)


def test_tree_descriptions_written_by_the_wave_match_the_oracle(simt_lib):
    """The fused table kernel writes the tree description with the whole wave (zn_wave_write_ctable); the oracle is the
    serial statement of HUF_writeCTable.  Planes with different statistics, frames compared byte for byte; the inputs
    are shown to reach the FSE form, the raw 4-bit form and the secondary normalisation (FSE_normalizeM2: codes
    whose weight classes hold one symbol each, found by search over weight histograms with the oracle)."""
    from zipnn_amd import sharding
    chunk = 16384
    d = _tree_description_planes(chunk)
    nchunks = len(d) // chunk
    m2_before = O.lib().zo_debug_m2_calls()
    want = O.compress_frame(HDR, d, 1, 0, 10, chunk, threads=1)
    m2 = O.lib().zo_debug_m2_calls() - m2_before
    simt_lib.set_encode_onepass(True)
    try:
        got = simt_lib.compress(HDR, d, 1, 0, 10, chunk, 0.95)             # the table job inside the one-pass encoder (wave 0 of a four-wave workgroup)
        assert bytes(got) == want
        assert "zn_k_encode_onepass" in simt_lib.last_kernels()
        simt_lib.set_encode_onepass(False)
        got = simt_lib.compress(HDR, d, 1, 0, 10, chunk, 0.95)             # … and as the table kernel's one-wave workgroup
    finally:
        simt_lib.set_encode_onepass(1)
    assert bytes(got) == want
    assert "zn_k_encode_tables" in simt_lib.last_kernels()
    types, cum, payload = sharding._parse(np.frombuffer(want[32:], dtype=np.uint8), 1, nchunks)
    kinds = {"fse": 0, "raw4": 0, "stored": 0}
    for c in range(nchunks):
        start = int(cum[0, c - 1]) if c else 0
        if types[0, c] == 0:
            kinds["stored"] += 1
        else:
            kinds["fse" if payload[start] < 128 else "raw4"] += 1
    assert kinds["fse"] >= 100 and kinds["raw4"] >= 20 and m2 >= 3, (kinds, m2)
    assert bytes(simt_lib.decompress(bytes(got)[32:], 1, 0, 10, chunk, len(d))) == d


def test_ragged_planes_are_coded_inside_the_fused_launches(simt_lib):
    """a tensor with a partial last chunk (and a geometry the fused kernels do not take at all) is compressed by the
    same three launches as its full chunks — stats, tables, emit, with tail workgroups — and the scan; no split / encode / gather
    kernels any more.  Bytes identical to the oracle's."""
    for kind, nb, P, rot, bm, chunk in (("bf16", 3 * C + 250_001, 2, 1, 10, C), ("fp32", C + 4 * 999, 4, 1, 220, C), ("fp8", 2 * C + 777, 1, 1, 10, C),
                                         ("bf16", 10 * 1000 + 6, 2, 1, 10, 1000), ("bf16", 37, 2, 1, 10, C)):
        d = gen_bytes(kind, nb, 5)
        assert bytes(simt_lib.compress(HDR, d, P, rot, bm, chunk, 0.95)) == O.compress_frame(HDR, d, P, rot, bm, chunk)
        k = simt_lib.last_kernels()
        assert [x for x in k.split(";") if x != "zn_k_encode_onepass"] == ["zn_k_encode_stats+tail", "zn_k_encode_tables", "zn_k_scan_sizes", "zn_k_encode_emit+tail"], k      # (the full chunks: the one-pass kernel, or the same launches)


# ---------------------------------------------------------------------------------------------------------------------
# the small-input decoder (zn_decode_wide.hpp): one 16-wave workgroup per chunk, four waves per huff0 stream
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture()
def wide_mode(simt_lib):
    """zn_set_decode_wide for one test (0 never / 1 automatic / 2 always, 16-wave form / 3 always, 8-wave form); back to automatic afterwards."""
    yield simt_lib.set_decode_wide
    simt_lib.set_decode_wide(1)


def _slow_sync_bf16(nbytes, seed):
    """A bf16-like tensor whose top-byte code re-synchronises badly (three equally likely 2-bit codes hold 95 % of the mass) at the
    density of a weights tensor: some tiles' top sub-block guesses its start wrong, the wave decodes the tile again."""
    r = np.random.default_rng(seed)
    p = np.array([500, 500, 480] + [40] * 6 + [2] * 20 + [0.05] * 16); p = p / p.sum()
    n = nbytes // 2
    hi = (r.choice(len(p), size=n, p=p) * 3 + 40).astype(np.uint8)
    lo = r.integers(0, 256, size=n, dtype=np.uint8)
    return np.stack([lo, hi], axis=1).reshape(-1).tobytes()


@pytest.mark.parametrize("kind,nb,P,rot,bm,wide_chunks,pending", [
    ("bf16", 3 * C, 2, 1, 10, 3, 0), ("fp32", 2 * C, 4, 1, 220, 2, 0), ("bf16", 2 * C + C // 2 + 10, 2, 1, 10, 2, 0), ("fp32", C + C // 4 + 4, 4, 1, 220, 1, 0),
    ("fp8", 2 * C, 1, 0, 10, 0, 2), ("fp16", 2 * C, 2, 0, 10, 0, 2), ("rand", 2 * C, 2, 1, 10, 0, 2), ("const", 2 * C, 2, 1, 10, 0, 2), ("slowsync", 3 * C, 2, 0, 10, 3, 0)],
    ids=["bf16", "fp32", "bf16-tail", "fp32-tail", "fp8-dense-code", "fp16-dense-code", "raw-planes", "rle-planes", "slow-sync"])
@pytest.mark.parametrize("mode", [2, 3], ids=["16-waves", "8-waves"])
def test_wide_decoder_takes_weights_like_chunks_and_leaves_the_rest_pending(simt_lib, wide_mode, mode, kind, nb, P, rot, bm, wide_chunks, pending):
    """Forced on (mode 2): the full chunks of weights-like tensors are decoded by zn_k_decode_wide (counter 4 of the emulated build); dense codes,
    chunks without exactly one Huffman plane are left pending (counter 5) and taken by the fused kernel behind it; a partial last chunk is finished
    by the tail and merge workgroups of the wide launch itself (round 6) — same bytes in every case.  The slow-sync tensor makes some tile tops guess wrong: those tiles are decoded again (counter 6)."""
    d = _slow_sync_bf16(nb, 4) if kind == "slowsync" else _gen2(kind, nb, 23)
    frame = O.compress_frame(HDR, d, P, rot, bm, C)
    wide_mode(mode)                        # 2: four waves per stream (16-wave workgroups), 3: two (8-wave workgroups, two per CU)
    _tile_counters()
    assert bytes(simt_lib.decompress(frame[32:], P, rot, bm, C, len(d))) == d
    cnt = _tile_counters()
    assert cnt[4] == wide_chunks and cnt[5] == pending
    ks = simt_lib.last_kernels().split(";")
    name = "zn_k_decode_wide" if mode == 2 else "zn_k_decode_wide^2"
    # the fused kernel's `rest` instance is the only other launch (it decodes what neither takes with the generic path's own code); with a partial chunk
    # the tail workgroups ride at the front of the wide launch and the merge workgroups at its end (round 6: it was the pending pass + two generic kernels)
    if nb % C:
        assert ks == [name + "+tail+merge", "zn_k_decode_fused^rest"]
    else:
        assert ks == [name, "zn_k_decode_fused^rest"]
    if kind == "slowsync":
        assert cnt[6] > 0
    if pending == 0:
        assert cnt[0] == 0                       # no tile went through the fused kernel
    wide_mode(0)
    assert bytes(simt_lib.decompress(frame[32:], P, rot, bm, C, len(d))) == d
    assert simt_lib.last_kernels().split(";")[0].startswith("zn_k_decode_fused") and "wide" not in simt_lib.last_kernels()


def test_wide_decoder_automatic_mode_is_for_calls_of_at_most_two_chunks_per_cu(simt_lib, wide_mode, monkeypatch):
    """Mode 1 (the default): sign-rotated layouts without a delta base, at most one full chunk per CU of the device (the emulated device has one CU)
    in the 16-wave form, at most two in the 8-wave form."""
    d1 = _gen2("bf16", C, 5); d2 = _gen2("bf16", 2 * C, 7); d3 = _gen2("bf16", 3 * C, 6)
    f1 = O.compress_frame(HDR, d1, 2, 1, 10, C); f2 = O.compress_frame(HDR, d2, 2, 1, 10, C); f3 = O.compress_frame(HDR, d3, 2, 1, 10, C)
    wide_mode(1)
    assert bytes(simt_lib.decompress(f1[32:], 2, 1, 10, C, len(d1))) == d1
    assert simt_lib.last_kernels().startswith("zn_k_decode_wide;")                  # one chunk per CU: four waves per stream
    assert bytes(simt_lib.decompress(f2[32:], 2, 1, 10, C, len(d2))) == d2
    assert simt_lib.last_kernels().startswith("zn_k_decode_wide^2")                 # two: the 8-wave form
    assert bytes(simt_lib.decompress(f3[32:], 2, 1, 10, C, len(d3))) == d3
    assert simt_lib.last_kernels() == "zn_k_decode_fused^rest"                      # three: the fused kernel (its rest instance: no generic launches behind a call of whole chunks)
    dt = d1 + d1[:1000]                                                             # a partial last chunk: its eight tail workgroups want slots beside the full chunks' — one CU has two
    ft = O.compress_frame(HDR, dt, 2, 1, 10, C)
    assert bytes(simt_lib.decompress(ft[32:], 2, 1, 10, C, len(dt))) == dt
    assert simt_lib.last_kernels() == "zn_k_decode_fused^rest+tail+merge"            # (… inside ONE launch since round 5)
    dt = d1 + d1[:40000]                                                            # … whatever its length
    ft = O.compress_frame(HDR, dt, 2, 1, 10, C)
    assert bytes(simt_lib.decompress(ft[32:], 2, 1, 10, C, len(dt))) == dt
    assert simt_lib.last_kernels() == "zn_k_decode_fused^rest+tail+merge"
    monkeypatch.setenv("ZN_SIMT_CUS", "41")                                         # … on a device of 41 CUs (1 chunk + 8 tail + 32 merge workgroups): the 16-wave form, tail workgroups in its launch, merge workgroups in the one behind (round 6)
    assert bytes(simt_lib.decompress(ft[32:], 2, 1, 10, C, len(dt))) == dt
    assert simt_lib.last_kernels() == "zn_k_decode_wide+tail+merge;zn_k_decode_fused^rest"
    dt = d2 + d1[:1000]                                                             # … 42 workgroups on 41 CUs: the 8-wave form (two per CU), whatever the length of the tail
    ft = O.compress_frame(HDR, dt, 2, 1, 10, C)
    assert bytes(simt_lib.decompress(ft[32:], 2, 1, 10, C, len(dt))) == dt
    assert simt_lib.last_kernels() == "zn_k_decode_wide^2+tail+merge;zn_k_decode_fused^rest"
    monkeypatch.delenv("ZN_SIMT_CUS")
    f16 = O.compress_frame(HDR, d1, 2, 0, 10, C)                                    # no sign rotate (an fp16 layout): not in automatic mode
    assert bytes(simt_lib.decompress(f16[32:], 2, 0, 10, C, len(d1))) == d1
    assert simt_lib.last_kernels() == "zn_k_decode_fused^rest"
    with pytest.raises(ValueError):
        simt_lib.set_decode_wide(4)


def test_wide_decoder_batches_and_corrupt_streams(simt_lib, wide_mode):
    """A batch (segment table) through the wide kernel; a corrupted stream is handed to the fused kernel, which reports it."""
    from zipnn_amd import codec
    wide_mode(2)
    specs = [("bf16", 2 * C + 10, 2, 1, 10, C), ("fp32", 2 * C, 4, 1, 220, C), ("fp8", C + 1, 1, 1, 10, C), ("bf16", 7, 2, 1, 10, C), ("bf16", 0, 2, 1, 10, C)]
    datas, items = [], []
    for i, (kind, nb, P, rot, bm, chunk) in enumerate(specs):
        d = gen_bytes(kind, nb, 40 + i)
        body = O.compress_frame(HDR, d, P, rot, bm, chunk)[32:]
        datas.append(d)
        items.append((torch.frombuffer(bytearray(body), dtype=torch.uint8) if body else torch.empty(0, dtype=torch.uint8), P, rot, bm, chunk, nb))
    _tile_counters()
    outs = codec.decompress_device_batch(simt_lib, items)
    for d, o in zip(datas, outs):
        assert o.numpy().tobytes() == d
    assert _tile_counters()[4] == 2 + 2 and "zn_k_decode_wide" in simt_lib.last_kernels()
    # a stream whose last byte is zero (no end mark): the wide kernel leaves the chunk pending, the fused kernel reports it
    d = _gen2("bf16", 2 * C, 9)
    body = bytearray(O.compress_frame(HDR, d, 2, 1, 10, C)[32:])
    K = 2
    PK = 2 * K                                                            # body = types[P][K], cumulative sizes u64[P][K] (per plane), payload plane by plane
    cum = lambda p, c: int.from_bytes(body[PK + 8 * (p * K + c): PK + 8 * (p * K + c) + 8], "little")
    assert body[1 * K + 0] == 1 and cum(1, 0) < C // 2                    # plane 1 (sign-rotated exponent byte) of chunk 0 is a huff0 block
    body[9 * PK + cum(0, K - 1) + cum(1, 0) - 1] = 0                      # its last byte = the last byte of its stream 4
    _tile_counters()
    with pytest.raises(RuntimeError):
        simt_lib.decompress(bytes(body), 2, 1, 10, C, len(d))
    cnt = _tile_counters()
    assert cnt[5] >= 1


def test_rest_instance_decodes_what_neither_kernel_takes_with_the_generic_code(simt_lib, wide_mode):
    """Behind the wide kernel, in a call without partial chunks, zn_k_decode_fused^rest is the only other launch: a geometry neither kernel takes
    (a chunk size that is not a multiple of 4 P rows) is decoded chunk by chunk by the generic path's device functions inside it — same bytes —, and a
    malformed jump table is reported by that code exactly as the generic kernels would."""
    ch = 3072
    d = _gen2("bf16", 5 * ch, 12)
    frame = O.compress_frame(HDR, d, 2, 1, 10, ch)
    for mode in (2, 3):
        wide_mode(mode)
        _tile_counters()
        assert bytes(simt_lib.decompress(frame[32:], 2, 1, 10, ch, len(d))) == d
        cnt = _tile_counters()
        assert cnt[4] == 0 and cnt[5] == 5                       # the wide kernel declined all five
        assert simt_lib.last_kernels().split(";")[1:] == ["zn_k_decode_fused^rest"] and simt_lib.last_fused_chunks() == 0
    wide_mode(0)
    assert bytes(simt_lib.decompress(frame[32:], 2, 1, 10, ch, len(d))) == d
    assert simt_lib.last_kernels() == "zn_k_decode_fused^rest" and simt_lib.last_fused_chunks() == 0      # (without the wide kernel: the same instance, alone)
    # a weights-like chunk whose jump table claims more than the block holds
    d = _gen2("bf16", 2 * C, 9)
    body = bytearray(O.compress_frame(HDR, d, 2, 1, 10, C)[32:])
    K = 2; PK = 2 * K
    cum = lambda p, c: int.from_bytes(body[PK + 8 * (p * K + c): PK + 8 * (p * K + c) + 8], "little")
    blk = 9 * PK + cum(0, K - 1)                                  # plane 1 of chunk 0
    hs = 1 + body[blk]
    assert body[blk] < 128
    body[blk + hs: blk + hs + 2] = (0xFFFF).to_bytes(2, "little")
    for mode in (2, 0):
        wide_mode(mode)
        with pytest.raises(RuntimeError):
            simt_lib.decompress(bytes(body), 2, 1, 10, C, len(d))


def test_decode_group_rule_counts_rounds(simt_lib):
    """zn_decode_fused_group: the group size with the fewest rounds x (13 + 85 x size); the emulated device holds 1 CU x 4 workgroups.
    (On the device, with 1 024 slots, the same arithmetic picks the measured best column of profiles/r05_decode_group_rule.txt.)"""
    slots = 4

    def rule(K):
        best, pick = None, 1
        for m in (1, 2, 3, 4):
            wgs = max(1, -(-K // m)); rounds = -(-wgs // slots); cost = rounds * (13 + 85 * m)
            if best is None or cost <= best:
                best, pick = cost, m
        return pick
    for K in (0, 1, 3, 4, 5, 8, 9, 12, 13, 14, 16, 17, 20, 21, 64, 65, 68, 1 << 20):
        assert simt_lib.decode_group_for(K) == rule(K), K
    assert [simt_lib.decode_group_for(K) for K in (4, 5, 8, 9, 12, 16, 20)] == [1, 2, 2, 3, 3, 4, 1]
    simt_lib.set_decode_group(3)
    try:
        assert simt_lib.decode_group_for(1000) == 3
    finally:
        simt_lib.set_decode_group(0)


def test_decode_status_belongs_to_the_calling_threads_call(simt_lib):
    """zn_decode_status: a check = 0 call's verdict sits in a slot of its own — a healthy decode by another thread in between does not erase it,
    and once the slot has been handed to a later call the answer is an error, not a guess."""
    import threading
    d = gen_bytes("bf16", 2 * C, 4)
    good = O.compress_frame(HDR, d, 2, 1, 10, C)
    bad = bytearray(good)
    bad[32] = 7                                         # a chunk-type byte that is neither 0 nor 1
    gb = torch.frombuffer(bytearray(good[32:]), dtype=torch.uint8)
    bb = torch.frombuffer(bytearray(bad[32:]), dtype=torch.uint8)
    out = torch.empty(2 * C, dtype=torch.uint8)
    out2 = torch.empty(2 * C, dtype=torch.uint8)

    def other_thread(n, check):
        for _ in range(n):
            simt_lib.decompress_dev(gb.data_ptr(), gb.numel(), 2, 1, 10, C, 2 * C, out2.data_ptr(), check=check)

    simt_lib.decompress_dev(bb.data_ptr(), bb.numel(), 2, 1, 10, C, 2 * C, out.data_ptr(), check=False)
    for check in (False, True):
        t = threading.Thread(target=other_thread, args=(3, check)); t.start(); t.join()
    with pytest.raises(MemoryError):
        simt_lib.decode_status()                        # ZN_E_TYPE, as the checked call reports it
    # a healthy call of this thread: ok, whatever other threads' calls found
    simt_lib.decompress_dev(gb.data_ptr(), gb.numel(), 2, 1, 10, C, 2 * C, out.data_ptr(), check=False)
    simt_lib.decode_status()
    assert out.numpy().tobytes() == d
    # sixteen decode calls later the slot is another call's
    simt_lib.decompress_dev(gb.data_ptr(), gb.numel(), 2, 1, 10, C, 2 * C, out.data_ptr(), check=False)
    t = threading.Thread(target=other_thread, args=(16, False)); t.start(); t.join()
    with pytest.raises(RuntimeError):
        simt_lib.decode_status()
    # a thread that made no call of its own is told about the device's last decode
    res = []
    t = threading.Thread(target=lambda: res.append(simt_lib.decode_status())); t.start(); t.join()
    assert res == [None]


@pytest.mark.parametrize("mode", [4, 7])
def test_host_entry_points_staged_and_direct_bookkeeping(simt_lib, mode):
    """zn_set_host_direct on the emulated library (hipHostRegister is a no-op there: "device" memory is host memory): the direct path's bookkeeping — piece cuts on
    2 MiB boundaries, one DMA per registration, the maps of the slice pipeline, the residency rule — against the staged path, 72 MiB of mostly incompressible bytes
    (cheap for the emulated kernels) with a compressible stretch and a ragged tail; same frame as the oracle, same bytes back, fresh and recycled buffers."""
    import ctypes
    L = simt_lib._L
    n = 72 * 1024 * 1024 + 777                   # (the direct path takes calls from 64 MiB: a 16 MiB probe piece + one more registration; three slices)
    rng = np.random.default_rng(9)
    x = rng.integers(0, 256, n, dtype=np.uint8)
    x[5 * C: 9 * C] = np.frombuffer(gen_bytes("bf16", 4 * C, 3), dtype=np.uint8)
    want = O.compress_frame(HDR, x, 2, 1, 10, C, threads=8)
    hdr = np.frombuffer(HDR, dtype=np.uint8)
    cap = L.zn_compress_bound(n, 2, C, 32); sz = ctypes.c_size_t(0)
    warm_frame, warm_back = np.zeros(cap, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
    try:
        simt_lib.set_host_direct(mode)
        for slices, fresh in ((0, False), (3, True), (1, False)):
            simt_lib.set_host_slices(slices)
            fr = np.empty(cap, dtype=np.uint8) if fresh else warm_frame
            assert L.zn_compress(hdr.ctypes.data, 32, x.ctypes.data, n, 2, 1, 10, C, ctypes.c_float(0.95), 0, fr.ctypes.data, cap, ctypes.byref(sz)) == 0
            assert fr[:sz.value].tobytes() == want
            bk = np.empty(n, dtype=np.uint8) if fresh else warm_back
            assert L.zn_decompress(fr.ctypes.data + 32, sz.value - 32, 2, 1, 10, C, n, 0, bk.ctypes.data) == 0
            assert np.array_equal(bk, x)
    finally:
        simt_lib.set_host_direct(4); simt_lib.set_host_slices(0)


def test_results_come_out_of_the_pinned_arena_and_go_back(simt_lib):
    """zn_host_alloc / zn_host_free (round 6) and what stands on them: ZnLib.compress / decompress hand out results of 8 MiB and more as views of a block of the
    library's pinned arena (the reference's extension also returns memoryviews over memory it allocated: csrc/zipnn_core.c:596, 1126); the block goes back when the
    LAST view of it dies and is the block the next request of that size gets."""
    import ctypes
    import gc
    L = simt_lib._L
    p1 = L.zn_host_alloc(10 << 20); assert p1
    assert L.zn_host_free(ctypes.c_void_p(p1)) == 0
    assert L.zn_host_free(ctypes.c_void_p(p1)) == -1                   # not a live block any more
    assert L.zn_host_free(ctypes.c_void_p(0)) == 0
    p2 = L.zn_host_alloc(9 << 20); assert p2 == p1                       # recycled (within a factor of two of its size)
    p3 = L.zn_host_alloc(1 << 20); assert p3 and p3 != p2                # … but not for a request a tenth of it
    assert L.zn_host_free(ctypes.c_void_p(p2)) == 0 and L.zn_host_free(ctypes.c_void_p(p3)) == 0
    n = 9 * 1024 * 1024 + 5
    d = np.random.default_rng(3).integers(0, 256, n, dtype=np.uint8).tobytes()
    want = O.compress_frame(HDR, d, 2, 1, 10, C, threads=8)
    fr = simt_lib.compress(HDR, d, 2, 1, 10, C, 0.95)
    assert bytes(fr) == want
    base = fr.obj
    assert type(base).__name__ == "_ArenaArray"
    addr = base.ctypes.data
    back = simt_lib.decompress(fr[32:], 2, 1, 10, C, n)
    assert bytes(back) == d and type(back.obj).__name__ == "_ArenaArray"
    tail = fr[100:200]                                                   # a view of a view keeps the block alive
    del fr, base; gc.collect()
    assert bytes(tail) == want[100:200]
    assert L.zn_host_alloc.restype is not None and addr                 # (the block is still out: `tail` holds it)
    del tail, back; gc.collect()
    small = simt_lib.compress(HDR, d[:C], 2, 1, 10, C, 0.95)             # below 8 MiB: ordinary memory
    assert type(small.obj).__name__ != "_ArenaArray"


def test_decode_status_after_the_workspace_is_released(simt_lib):
    """the token of an unverified check = 0 decode outlives zn_release_workspace; the answer is then an error, once, not "ok"."""
    d = gen_bytes("bf16", 2 * C, 4)
    gb = torch.frombuffer(bytearray(O.compress_frame(HDR, d, 2, 1, 10, C)[32:]), dtype=torch.uint8)
    out = torch.empty(2 * C, dtype=torch.uint8)
    simt_lib.decompress_dev(gb.data_ptr(), gb.numel(), 2, 1, 10, C, 2 * C, out.data_ptr(), check=False)
    simt_lib.release_workspace()
    with pytest.raises(RuntimeError):
        simt_lib.decode_status()
    simt_lib.decode_status()                            # the token is spent: nothing has run since
    simt_lib.decompress_dev(gb.data_ptr(), gb.numel(), 2, 1, 10, C, 2 * C, out.data_ptr(), check=False)
    simt_lib.decode_status()
    assert out.numpy().tobytes() == d


def test_mixed_batch_of_two_full_kinds_takes_the_two_stream_path(simt_lib):
    """zn_decompress_batch_dev: a batch whose one-plane tensors AND whose multi-plane tensors each have at least 512 chunks forks the two kinds' launches onto two
    streams of the library's own and joins them again — per-kind bases in the launch-wide arrays, ragged tensors, a tensor of each kind.  Small
    chunks keep the emulated run short; the same bytes as the sources, and the kernel log says that the path was taken."""
    from zipnn_amd import codec
    c1, c2, c4 = 4096, 4096, 8192
    specs = [("fp8", 300 * c1 + 77, 1, 0, 10, c1), ("bf16", 400 * c2, 2, 1, 10, c2), ("fp8", 230 * c1, 1, 0, 10, c1),
             ("fp32", 40 * c4 + 12, 4, 1, 220, c4), ("bf16", 120 * c2 + 1000, 2, 1, 10, c2), ("fp16", 3 * c2, 2, 0, 10, c2)]
    datas, items = [], []
    for i, (kind, nb, P, rot, bm, chunk) in enumerate(specs):
        d = gen_bytes(kind, nb, 60 + i)
        frame = O.compress_frame(HDR, d, P, rot, bm, chunk)
        datas.append(d)
        items.append((torch.frombuffer(bytearray(frame[32:]), dtype=torch.uint8), P, rot, bm, chunk, nb))
    outs = codec.decompress_device_batch(simt_lib, items)
    for d, o in zip(datas, outs):
        assert o.numpy().tobytes() == d
    assert "(two streams)" in simt_lib.last_kernels()
    # … and a batch below the bar stays on the caller's stream
    outs = codec.decompress_device_batch(simt_lib, items[1:])
    for d, o in zip(datas[1:], outs):
        assert o.numpy().tobytes() == d
    assert "(two streams)" not in simt_lib.last_kernels()
