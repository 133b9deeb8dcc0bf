"""CPU tests (-m "not gpu"): pin the oracle (oracle/zn_oracle.c) against
 (1) the golden frames the reference itself produced (tests/golden/),
 (2) libzstd 1.4.8's exported huff0 (the library the reference build links), and
 (3) oracle/_ref — the reference csrc/ compiled from /root/reference — when present.
The oracle is only the checker; nothing here is product code."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


@pytest.mark.parametrize("name", G.names())
def test_oracle_decodes_golden_frames(name):
    meta, blob = G.get(name)
    out = b""
    for fr in G.split_frames(blob):
        p = G.parse_frame(fr)
        out += O.decompress_body(p["body"], p["num_buf"], p["bits_mode"], p["bytes_mode"], p["chunk"],
                                 p["orig_len"], threads=2)
    assert len(out) == meta["in_len"]
    assert G.sha(out) == meta["in_sha256"]


@pytest.mark.parametrize("name", G.names())
def test_oracle_reproduces_golden_frames_bit_exact(name):
    """compress(decode(frame)) must give back the reference's frame byte for byte."""
    meta, blob = G.get(name)
    rebuilt = b""
    for fr in G.split_frames(blob):
        p = G.parse_frame(fr)
        data = O.decompress_body(p["body"], p["num_buf"], p["bits_mode"], p["bytes_mode"], p["chunk"], p["orig_len"])
        rebuilt += O.compress_frame(p["header"], data, p["num_buf"], p["bits_mode"], p["bytes_mode"], p["chunk"],
                                    threshold=0.95, threads=2)
    assert G.sha(rebuilt) == meta["frame_sha256"]
    assert rebuilt == blob


@pytest.mark.parametrize("name", G.delta_names())
def test_oracle_on_reference_written_delta_frames(name):
    """Delta frames written by the reference's own package (tests/golden/make_golden_delta.py; reference zipnn/zipnn.py:625-640,
    983-1004: XOR with the second buffer on the host around the core call): the oracle decodes them to data ^ base and re-encodes
    data ^ base to the same bytes."""
    meta, blob, base = G.delta_get(name)
    out, rebuilt = b"", b""
    for fr in G.split_frames(blob):
        p = G.parse_frame(fr)
        x = O.decompress_body(p["body"], p["num_buf"], p["bits_mode"], p["bytes_mode"], p["chunk"], p["orig_len"], threads=2)
        rebuilt += O.compress_frame(p["header"], x, p["num_buf"], p["bits_mode"], p["bytes_mode"], p["chunk"], threshold=0.95, threads=2)
        out += x
    data = bytes(np.frombuffer(out, np.uint8) ^ np.frombuffer(base, np.uint8))
    assert len(data) == meta["in_len"] and G.sha(data) == meta["in_sha256"]
    assert rebuilt == blob and G.sha(rebuilt) == meta["frame_sha256"]
    assert meta["frame_len"] < meta["plain_frame_len"]          # the delta is what made it small


def _planes(rng):
    """byte planes with the shapes the path meets: skewed exponents, flat mantissas, tiny, RLE."""
    out = []
    for n, a, conc in ((131072, 20, 0.05), (65536, 200, 0.3), (32768, 256, 1.0), (12345, 7, 0.05),
                       (131072, 256, 5.0), (300, 5, 0.3), (13, 2, 1.0), (12, 2, 1.0), (11, 3, 1.0)):
        p = rng.dirichlet(np.ones(a) * conc)
        out.append(rng.choice(a, n, p=p).astype(np.uint8).tobytes())
    geo = 0.62 ** np.arange(60); geo /= geo.sum()
    out.append(rng.permutation(256)[:60][rng.choice(60, 100000, p=geo)].astype(np.uint8).tobytes())  # height-limited
    out.append(bytes([9]) * 4096)                                                                     # RLE
    out.append(rng.integers(0, 256, 50000, dtype=np.uint8).tobytes())                                 # incompressible
    return out


def test_huf_roundtrip_and_conventions():
    rng = np.random.default_rng(11)
    for src in _planes(rng):
        r, blob = O.huf_compress(src, cap=256 * 1024)
        assert not O.lib().zo_huf_is_error(r)
        if r == 1:
            assert len(set(src)) == 1 and blob[0] == src[0]
        elif r > 1:
            rr, back = O.huf_decompress(blob, len(src))
            assert rr == len(src) and back == src
    r, _ = O.huf_compress(bytes(128 * 1024 + 1), cap=512 * 1024)
    assert r == 2 ** 64 - 72          # "Src size is incorrect": the caller then stores the plane raw


def test_huf_matches_libzstd_148():
    z = O.libzstd()
    if z is None:
        pytest.skip("libzstd 1.4.x with exported huff0 not on this host")
    rng = np.random.default_rng(12)
    for src in _planes(rng):
        s = np.frombuffer(src, np.uint8)
        dst = np.zeros(256 * 1024, np.uint8)
        lr = z.HUF_compress(dst.ctypes.data, dst.size, s.ctypes.data, s.size)
        r, blob = O.huf_compress(src, cap=256 * 1024)
        assert r == lr
        if 1 < lr < dst.size:
            assert blob == dst[:lr].tobytes()
            back = np.zeros(s.size, np.uint8)                      # libzstd decodes the oracle's bytes
            assert z.HUF_decompress(back.ctypes.data, s.size, np.frombuffer(blob, np.uint8).ctypes.data, len(blob)) == s.size
            assert back.tobytes() == src


def _frame_cases():
    C = 256 * 1024
    cs = []
    for nb in (1, 2, 3, 6, 7, 1001, 1002, C - 2, C - 1, C, C + 1, C + 2, C + 6, 2 * C + 31338):
        cs += [("bf16", nb, 2, 1, 10, C), ("fp16", nb, 2, 0, 10, C), ("rand", nb, 2, 1, 10, C), ("const", nb, 2, 1, 10, C)]
    for nb in (4, 8, 1000, C - 4, C, C + 4, 2 * C + 4096):
        cs += [("fp32", nb, 4, 1, 220, C), ("rand", nb, 4, 1, 220, C)]
    for nb in (1, 5, 1000, 128 * 1024 - 1, 128 * 1024, 128 * 1024 + 1, 300001):
        cs += [("fp8", nb, 1, 1, 10, 128 * 1024), ("rand", nb, 1, 1, 10, 128 * 1024)]
    cs += [("bf16", 5 * 65536 + 10, 2, 1, 10, 65536), ("bf16", 9 * 16384 + 2, 2, 1, 10, 16384)]
    return cs


def gen_bytes(kind, nbytes, seed=0):
    import torch
    g = torch.Generator().manual_seed(1000 + seed)
    if kind == "bf16":
        return (torch.randn((nbytes + 1) // 2, generator=g) * 0.02).to(torch.bfloat16).view(torch.uint8).numpy().tobytes()[:nbytes]
    if kind == "fp16":
        return (torch.randn((nbytes + 1) // 2, generator=g) * 0.02).half().view(torch.uint8).numpy().tobytes()[:nbytes]
    if kind == "fp32":
        return (torch.randn((nbytes + 3) // 4, generator=g) * 0.02).view(torch.uint8).numpy().tobytes()[:nbytes]
    if kind == "fp8":
        return (torch.randn(nbytes, generator=g) * 0.5).to(torch.float8_e4m3fn).view(torch.uint8).numpy().tobytes()
    if kind == "const":
        return bytes([0x3C, 0x80]) * (nbytes // 2) + b"\x3c" * (nbytes % 2)
    if kind == "rand":
        return np.random.default_rng(seed).integers(0, 256, nbytes, dtype=np.uint8).tobytes()
    raise ValueError(kind)


def test_oracle_frames_match_reference_build():
    """Byte-identical frames vs oracle/_ref (the reference's csrc + libzstd 1.4.8)."""
    if O.ref_core() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this host)")
    hdr = bytes(range(32))
    for i, (kind, nb, P, rot, bm, chunk) in enumerate(_frame_cases()):
        d = gen_bytes(kind, nb, i)
        mine = O.compress_frame(hdr, d, P, rot, bm, chunk, threads=3)
        ref = O.ref_compress_frame(hdr, d, P, rot, bm, chunk, threads=3)
        assert mine == ref, (kind, nb)
        assert O.ref_decompress_body(mine[32:], P, rot, bm, chunk, nb, 2) == d
        assert O.decompress_body(ref[32:], P, rot, bm, chunk, nb, 2) == d


def test_oracle_frame_roundtrip_edges():
    """Empty, ragged and chunk-boundary inputs round-trip through the oracle alone."""
    hdr = bytes(32)
    for i, (kind, nb, P, rot, bm, chunk) in enumerate(_frame_cases() + [("bf16", 0, 2, 1, 10, 256 * 1024)]):
        d = gen_bytes(kind, nb, i)
        f = O.compress_frame(hdr, d, P, rot, bm, chunk)
        assert int.from_bytes(f[24:32], "little") == len(f)
        assert O.decompress_body(f[32:], P, rot, bm, chunk, nb) == d


def test_oracle_rejects_bad_type_byte():
    d = gen_bytes("bf16", 5000)
    f = bytearray(O.compress_frame(bytes(32), d, 2, 1, 10, 256 * 1024))
    f[32] = 7
    with pytest.raises(RuntimeError):
        O.decompress_body(bytes(f[32:]), 2, 1, 10, 256 * 1024, 5000)
