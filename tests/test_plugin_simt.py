"""CPU tests (-m "not gpu") of the safetensors surface, driven through the SIMT-emulated kernels:
file round trip like the reference's tests/simple_stress_tests.py:205-264 (fp16 / bf16 / fp8 matrices,
half constant, half random) and load through zipnn_safetensors() + safe_open."""
import os

import pytest
import torch


def _model(seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.ones(100, 100)
    base[50:] = torch.rand(50, 100, generator=g) * 2 - 1
    return {
        "w_bf16": base.to(torch.bfloat16),
        "w_fp16": base.to(torch.float16),
        "w_fp8": base.to(torch.float8_e4m3fn),
        "w_fp32": (torch.randn(300, 40, generator=g) * 0.02),
        "ids": torch.arange(1000, dtype=torch.int64),
        "tiny": torch.randn(3, generator=g).to(torch.bfloat16),
    }


def test_safetensors_file_roundtrip_and_plugin(use_simt, tmp_path):
    import safetensors
    import safetensors.torch
    from safetensors.torch import save_file
    from zipnn_amd import safetensors_io, zipnn_safetensors
    from zipnn_amd.zipnn import METADATA_KEY
    tensors = _model(3)
    src = os.path.join(tmp_path, "m.safetensors")
    save_file(tensors, src, {"format": "pt"})
    znn_path = safetensors_io.compress_safetensors_file(src)
    assert znn_path.endswith(".znn.safetensors") and os.path.getsize(znn_path) < os.path.getsize(src)
    with safetensors.safe_open(znn_path, "pt", "cpu") as f:       # raw view: frames are uint8 tensors
        meta = f.metadata()
        assert METADATA_KEY in meta and "w_bf16" in meta[METADATA_KEY] and "ids" not in meta[METADATA_KEY]
        assert f.get_tensor("w_bf16").dtype == torch.uint8
    # file -> file
    back = safetensors_io.decompress_safetensors_file(znn_path, out_path=os.path.join(tmp_path, "back.safetensors"))
    with safetensors.safe_open(back, "pt", "cpu") as f:
        for k, v in tensors.items():
            got = f.get_tensor(k)
            assert got.dtype == v.dtype and got.shape == v.shape
            assert got.view(torch.uint8).numpy().tobytes() == v.contiguous().view(torch.uint8).numpy().tobytes()
    # the producer's batched path (one compress call for the whole file) writes the same file
    znn_b = safetensors_io.compress_safetensors_file(src, out_path=os.path.join(tmp_path, "b.znn.safetensors"), batched=True)
    with safetensors.safe_open(znn_b, "pt", "cpu") as fb, safetensors.safe_open(znn_path, "pt", "cpu") as fa:
        assert fa.metadata() == fb.metadata() and set(fa.keys()) == set(fb.keys())      # (safetensors orders metadata keys randomly)
        for k in fa.keys():
            assert fa.get_tensor(k).dtype == fb.get_tensor(k).dtype and torch.equal(fa.get_tensor(k).view(torch.uint8), fb.get_tensor(k).view(torch.uint8)), k
    # whole file in one batched decode (device = cpu memory under the emulator)
    loaded = safetensors_io.load_file(znn_path, device="cpu")
    assert set(loaded) == set(tensors)
    for k, v in tensors.items():
        assert loaded[k].dtype == v.dtype and loaded[k].shape == v.shape
        assert loaded[k].contiguous().view(torch.uint8).numpy().tobytes() == v.contiguous().view(torch.uint8).numpy().tobytes(), k
    # plugin: patched safe_open decompresses on access
    orig_a, orig_b = safetensors.torch.safe_open, safetensors.safe_open
    try:
        zipnn_safetensors()
        with safetensors.torch.safe_open(znn_path, "pt", "cpu") as f:
            for k, v in tensors.items():
                got = f.get_tensor(k)
                assert got.dtype == v.dtype and torch.equal(got.view(torch.uint8), v.contiguous().view(torch.uint8)), k
            assert f.get_slice("ids")[:10].tolist() == list(range(10))
            assert torch.equal(f.get_slice("w_bf16")[10:60].view(torch.uint8), tensors["w_bf16"][10:60].contiguous().view(torch.uint8))   # (the reference returns NotImplementedError here, zipnn.py:1617; chunk-range decode: test_get_slice_* below)
        with safetensors.safe_open(znn_path, framework="pt", device="cpu") as f:   # keyword form used by newer callers
            assert torch.equal(f.get_tensor("w_fp32"), tensors["w_fp32"])
    finally:
        safetensors.torch.safe_open, safetensors.safe_open = orig_a, orig_b
        from zipnn_amd import zipnn as _Z
        _Z._patches_applied.pop(_Z._zipnn_safetensors, None)      # (the patcher applies a patch once per process, like the reference's: let the next test apply it again)


def test_zipnn_api_errors_and_types(use_simt):
    from zipnn_amd import ZipNN
    with pytest.raises(ValueError):
        ZipNN(compression_chunk=3000)
    with pytest.raises(ValueError):
        ZipNN(input_format="torch", is_streaming=True)
    with pytest.raises(ValueError):
        ZipNN(method="nope")
    with pytest.raises(ValueError):
        ZipNN(input_format="torch").compress(torch.arange(10))
    with pytest.raises(ValueError):
        ZipNN().decompress(b"XX" + bytes(40))
    z = ZipNN(bytearray_dtype="float32")
    raw = (torch.rand(8192) * 2 - 1).numpy().tobytes()
    frame = z.compress(raw)
    assert isinstance(frame, memoryview) and bytes(frame[:2]) == b"ZN" and frame[5] == 220 and frame[15] == 1
    assert int.from_bytes(frame[24:32], "little") == len(frame)
    assert bytes(ZipNN(bytearray_dtype="float32").decompress(frame)) == raw
    import numpy as np
    a = (np.random.default_rng(0).standard_normal((33, 77)) * 0.02).astype(np.float16)
    back = ZipNN(input_format="numpy").decompress(ZipNN(input_format="numpy").compress(a))
    assert isinstance(back, np.ndarray) and back.dtype == a.dtype and back.shape == a.shape and (back == a).all()
    # delta (XOR) and streaming compose at the Python level exactly like the reference
    b2 = bytes(x ^ 0x5A for x in raw[:4096]) + raw[4096:]
    zd = ZipNN(bytearray_dtype="float32", delta_compressed_type="byte")
    assert bytes(ZipNN(bytearray_dtype="float32", delta_compressed_type="byte").decompress(zd.compress(raw, delta_second_data=b2), delta_second_data=b2)) == raw
    zs = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 12)
    blob = zs.compress(raw)
    assert isinstance(blob, bytearray) and bytes(ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 12).decompress(blob)) == raw


def test_streaming_compress_batched_equals_per_frame_loop(use_simt):
    """The batched streaming compress produces exactly the frames of the per-piece loop (== the reference's
    streaming output), also with a delta buffer; and decompresses back."""
    import numpy as np
    from zipnn_amd import ZipNN
    g = torch.Generator().manual_seed(5)
    raw = (torch.randn(150_000, generator=g) * 0.02).to(torch.bfloat16).view(torch.uint8).numpy().tobytes() + b"\x01\x02\x03"
    z = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 16)
    blob = z.compress(raw)
    want = bytearray()
    for off in range(0, len(raw), 1 << 16):
        want += bytes(ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 16).compress_torch_numpy_byte(raw[off:off + (1 << 16)]))
    assert bytes(blob) == bytes(want)
    assert bytes(ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 16).decompress(blob)) == raw
    other = np.random.default_rng(1).integers(0, 256, len(raw), dtype=np.uint8).tobytes()
    zd = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 16, delta_compressed_type="byte")
    blob2 = zd.compress(raw, delta_second_data=other)
    back2 = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 16, delta_compressed_type="byte").decompress(blob2, delta_second_data=other)
    assert bytes(back2) == raw


def test_host_buffer_entry_points_through_the_pinned_pipe(simt_lib, monkeypatch):
    """zn_compress / zn_decompress with host buffers above the pipe's threshold: the sliced, multi-threaded
    transfer (zn_host_pipe.hpp — plain memcpy under the emulator, same slicing / barriers / stripes) must hand the
    kernels exactly the caller's bytes and return exactly theirs.  Sizes straddle slice and stripe boundaries."""
    import numpy as np
    import oracle_lib as O
    rng = np.random.default_rng(7)
    for n, threads in ((2 * 1024 * 1024 + 2, "3"), (5 * 1024 * 1024 + 4098, "8"), (17 * 1024 * 1024 + 6, "5")):
        monkeypatch.setenv("ZN_HOST_THREADS", threads)
        x = (rng.standard_normal(n // 2) * 0.02).astype(np.float32)
        raw = (x.view(np.uint32) >> 16).astype(np.uint16).tobytes()          # bf16 bit patterns
        assert len(raw) == n
        hdr = bytes(32)
        frame = simt_lib.compress(hdr, raw, 2, 1, 10, 256 * 1024, 0.95)
        want = O.compress_frame(hdr, raw, 2, 1, 10, 256 * 1024)
        assert bytes(frame[32:]) == want[32:]
        back = simt_lib.decompress(memoryview(frame)[32:], 2, 1, 10, 256 * 1024, n)
        assert bytes(back) == raw


def test_tensor_frames_with_delta_streaming_and_many_dims(use_simt):
    """a frame handed to decompress() as a torch.Tensor used to work only for the plain one-frame case —
    with a delta buffer it raised an opaque TypeError, a streaming BYTE blob decoded its first frame only, and a shape
    of more than ~9 dimensions did not fit the fixed 80-byte header window."""
    import numpy as np
    import torch
    from zipnn_amd import ZipNN
    r = np.random.default_rng(2)
    a = r.integers(0, 256, 300_000, dtype=np.uint8).tobytes()
    b = bytes(x ^ (1 if i % 50 == 0 else 0) for i, x in enumerate(a))
    z = ZipNN(input_format="byte", bytearray_dtype="bfloat16", delta_compressed_type="byte")
    f = bytes(z.compress(a, delta_second_data=b))
    ft = torch.frombuffer(bytearray(f), dtype=torch.uint8)
    assert bytes(ZipNN(input_format="byte", bytearray_dtype="bfloat16", delta_compressed_type="byte").decompress(ft, delta_second_data=b)) == a
    # a streaming blob of several frames, as a tensor
    zs = ZipNN(input_format="byte", bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=65536)
    blob = bytes(zs.compress(a))
    bt = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    assert bytes(ZipNN(input_format="byte", bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=65536).decompress(bt)) == a
    # 12 dimensions: the shape extension is 1 + 12 * 2 bytes here, up to 1 + 9 * ndim in general
    t = torch.randn(2, 1, 3, 1, 2, 1, 2, 1, 5, 1, 2, 3).to(torch.bfloat16)
    fr = ZipNN(input_format="torch").compress(t)
    back = ZipNN(input_format="torch").decompress(torch.frombuffer(bytearray(bytes(fr)), dtype=torch.uint8))
    assert back.shape == t.shape and torch.equal(back, t)
    big = torch.zeros((1,) * 20 + (4,), dtype=torch.float32)
    fb = ZipNN(input_format="torch").compress(big)
    assert ZipNN(input_format="torch").decompress(fb).shape == big.shape


def test_safe_open_without_with_closes_its_host_handle(use_simt, tmp_path):
    """SafeOpen opens a second, host-side handle for device reads; used without `with` it was never closed."""
    import gc
    import torch
    from safetensors.torch import save_file
    from zipnn_amd import zipnn as Z
    from zipnn_amd.safetensors_io import compress_safetensors_file
    src = tmp_path / "m.safetensors"
    save_file({"w": (torch.randn(70_000) * 0.02).to(torch.bfloat16)}, str(src))
    out = compress_safetensors_file(str(src))
    f = Z.SafeOpen(out, framework="pt", device="cpu")
    f._device = "meta-not-cpu"          # force the second handle the way device="cuda:0" would (no GPU here)
    h = f._host_reader()
    assert f._host is h and h is not f._f
    closed = []
    real_exit = h.__exit__
    class Spy:
        def __init__(self, inner): self.inner = inner
        def __exit__(self, *a): closed.append(1); return real_exit(*a)
        def __getattr__(self, n): return getattr(self.inner, n)
    f._host = Spy(h)
    del f
    gc.collect()
    assert closed == [1]


def test_reference_produced_checkpoint_through_emulated_kernels(use_simt):
    """The reference-produced GPT-2-shaped checkpoint (tests/golden/make_golden_safetensors.py) through the plugin and
    load_file on the emulated kernels — the CPU half of tests/test_gpu_parity.py's hardware test of the same file."""
    import hashlib
    import json
    import safetensors
    import safetensors.torch
    import torch
    from zipnn_amd import safetensors_io, zipnn_safetensors
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpt2_small_ref.znn.safetensors")
    info = json.load(open(path + ".json"))
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == info["file_sha256"]
    sha = lambda t: hashlib.sha256(t.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()   # noqa: E731
    loaded = safetensors_io.load_file(path, device="cpu")
    assert {k: sha(v) for k, v in loaded.items()} == {k: m["sha256"] for k, m in info["tensors"].items()}
    orig_a, orig_b = safetensors.torch.safe_open, safetensors.safe_open
    try:
        zipnn_safetensors()
        with safetensors.safe_open(path, framework="pt", device="cpu") as f:
            assert len(f.compressed_tensors_metadata) >= 10
            for k, m in info["tensors"].items():
                t = f.get_tensor(k)
                assert str(t.dtype) == m["dtype"] and list(t.shape) == m["shape"] and sha(t) == m["sha256"], k
    finally:
        safetensors.torch.safe_open, safetensors.safe_open = orig_a, orig_b
        from zipnn_amd import zipnn as _Z
        _Z._patches_applied.pop(_Z._zipnn_safetensors, None)      # (the patcher applies a patch once per process, like the reference's: let the next test apply it again)


def test_zipnn_devices_keyword_spreads_host_input_over_gpus(use_simt):
    """ZipNN(devices=[…]) — the keyword-only extension that takes the place of the reference's `threads`: byte / numpy /
    host-torch input goes through zn_compress_multi / zn_decompress_multi (two emulated devices) and the frames are the ones
    the plain constructor writes."""
    from zipnn_amd import ZipNN
    import numpy as np
    raw = (torch.randn(5 * 131072 + 77, generator=torch.Generator().manual_seed(4)) * 0.02).to(torch.bfloat16).view(torch.uint8).numpy().tobytes()
    one = ZipNN().compress(raw)
    two = ZipNN(devices=[0, 1]).compress(raw)
    assert bytes(two) == bytes(one)
    assert bytes(ZipNN(devices=[1, 0, 1]).decompress(one)) == raw
    t = (torch.randn(3, 70001, generator=torch.Generator().manual_seed(5)) * 0.02).to(torch.float16)
    f1 = ZipNN(input_format="torch").compress(t)
    f2 = ZipNN(input_format="torch", devices=[0, 1]).compress(t)
    assert bytes(f2) == bytes(f1)
    back = ZipNN(input_format="torch", devices=[0, 1]).decompress(f1)
    assert back.dtype == t.dtype and back.shape == t.shape and torch.equal(back, t)
    a = (np.random.default_rng(1).standard_normal(300001) * 0.02).astype(np.float32)
    assert (ZipNN(input_format="numpy", devices=[0, 1]).decompress(ZipNN(input_format="numpy", devices=[1, 0]).compress(a)) == a).all()


def test_plugin_read_ahead_decodes_the_whole_file_with_one_batched_launch(use_simt, tmp_path):
    """SafeOpen's read-ahead (what get_tensor uses for a device target): the first compressed name ships the data section once and
    decodes EVERY compressed tensor of the file with one batched launch; get_tensor then serves from the cache and drops what it hands
    out.  Under the emulator the 'device' is CPU memory, so the read-ahead is started by hand."""
    from safetensors.torch import save_file
    from zipnn_amd import zipnn as Z
    from zipnn_amd.safetensors_io import compress_safetensors_file
    tensors = _model(11)
    src = os.path.join(tmp_path, "m.safetensors")
    save_file(tensors, src, {"format": "pt"})
    out = compress_safetensors_file(src)
    with Z.SafeOpen(out, framework="pt", device="cpu") as f:
        names = set(f.compressed_tensors_metadata)
        assert len(names) >= 3
        f._read_ahead()
        assert isinstance(f._ahead, dict) and set(f._ahead) == names
        kernels = use_simt.last_kernels()
        assert kernels.count("zn_k_decode_fused") + kernels.count("zn_k_decode_planes") >= 1      # one launch set for all of them
        for k in sorted(names):
            got = f._ahead.pop(k)
            v = tensors[k]
            assert got.dtype == v.dtype and got.shape == v.shape
            assert got.contiguous().view(torch.uint8).numpy().tobytes() == v.contiguous().view(torch.uint8).numpy().tobytes(), k
        assert f._ahead == {}
        # a name asked for twice falls back to the per-tensor path
        k = sorted(names)[0]
        assert torch.equal(f.get_tensor(k).view(torch.uint8), tensors[k].contiguous().view(torch.uint8))
    assert f._ahead is False


def test_load_file_copies_plain_tensors_out_of_the_uploaded_section(use_simt, tmp_path):
    """an uncompressed tensor must not keep the whole uploaded data section (all compressed frames) alive."""
    from safetensors.torch import save_file
    from zipnn_amd.safetensors_io import compress_safetensors_file, load_file
    tensors = _model(5)
    src = os.path.join(tmp_path, "m.safetensors")
    save_file(tensors, src, {"format": "pt"})
    out = compress_safetensors_file(src)
    loaded = load_file(out, device="cpu")
    ids = loaded["ids"]
    assert torch.equal(ids, tensors["ids"])
    assert ids.untyped_storage().nbytes() <= ids.numel() * ids.element_size() + 64


def test_corrupt_frame_header_in_a_file_reports_the_real_error(use_simt, tmp_path):
    """a frame with a damaged header must surface as the header error, not as BufferError from closing the mapping."""
    from safetensors.torch import save_file
    from zipnn_amd.safetensors_io import compress_safetensors_file, load_file, _read_layout
    tensors = _model(6)
    src = os.path.join(tmp_path, "m.safetensors")
    save_file(tensors, src, {"format": "pt"})
    out = compress_safetensors_file(src)
    meta, layout, data_start = _read_layout(out)
    lo = layout["w_bf16"][2]
    blob = bytearray(open(out, "rb").read())
    blob[data_start + lo] = ord("X")                      # "ZN" -> "XN"
    open(out, "wb").write(blob)
    with pytest.raises(ValueError, match="Header should start with ZN"):
        load_file(out, device="cpu")


def test_load_file_unmaps_the_file_on_a_helper_thread_and_reports_its_phases(use_simt, tmp_path):
    """decode_file_on_device closes its mapping on a helper thread (munmap behind the multi-threaded upload costs 1.2 ms on a 256-CPU host); the thread is
    joined by the next call at the latest, the file can be replaced right away, and `timings` splits decode_s into launch / views / wait / plain copies."""
    from safetensors.torch import save_file
    from zipnn_amd import safetensors_io
    sd = {"w": (torch.randn(70000) * 0.02).to(torch.bfloat16), "ids": torch.arange(16, dtype=torch.int64)}
    src = str(tmp_path / "m.safetensors"); save_file(sd, src, {"format": "pt"})
    znn = safetensors_io.compress_safetensors_file(src)
    tm = {}
    out = safetensors_io.load_file(znn, device="cpu", timings=tm)
    assert all(torch.equal(out[k], sd[k]) for k in sd)
    for k in ("decode_launch_s", "decode_views_s", "decode_wait_s", "decode_plain_s", "read_s", "h2d_s", "decode_s"):
        assert tm[k] >= 0.0
    assert abs(tm["decode_launch_s"] + tm["decode_views_s"] + tm["decode_wait_s"] + tm["decode_plain_s"] - tm["decode_s"]) < 1e-6
    pend = list(safetensors_io._PENDING_CLOSERS)
    assert len(pend) == 1
    os.replace(znn, znn + ".moved")                      # (nothing holds the path)
    pend[0].join(10.0)
    assert not pend[0].is_alive()
    out2 = safetensors_io.load_file(znn + ".moved", device="cpu")       # joins and drops the previous call's helper
    assert pend[0] not in safetensors_io._PENDING_CLOSERS and torch.equal(out2["w"], sd["w"])


def _same_safetensors_container(a, b):
    """Two safetensors files with the same header (as JSON objects: safetensors writes the metadata map in hash order, which differs from run to run) and
    the same data section."""
    import json
    da, db = open(a, "rb").read(), open(b, "rb").read()
    na, nb = int.from_bytes(da[:8], "little"), int.from_bytes(db[:8], "little")
    return json.loads(da[8:8 + na]) == json.loads(db[8:8 + nb]) and da[8 + na:] == db[8 + nb:]


def test_file_compress_through_one_upload_one_batch_one_download_writes_the_same_file(use_simt, tmp_path):
    """_compress_file_on_device (what compress_safetensors_file does for a cuda device: the data section up in one transfer, one batched compress into an
    arena with a gap in front of every body, the arena down in one transfer, headers written into the gaps) produces byte for byte the file of the
    per-tensor path — compressed tensors, a tensor that does not shrink, integers, an empty and a tiny one, the metadata list in the same order."""
    from safetensors.torch import save_file
    from zipnn_amd import safetensors_io
    g = torch.Generator().manual_seed(3)
    tensors = {"w_bf16": (torch.randn(70, 1000, generator=g) * 0.02).to(torch.bfloat16), "w_fp32": torch.randn(51, 400, generator=g) * 0.02,
               "w_fp16": (torch.randn(30, 1001, generator=g) * 0.02).half(), "ids": torch.arange(100), "tiny": torch.randn(3, generator=g).to(torch.bfloat16),
               "noise": torch.randint(0, 256, (4000,), generator=g, dtype=torch.uint8).view(torch.float16), "empty": torch.empty(0, 7, dtype=torch.bfloat16)}
    src = str(tmp_path / "m.safetensors"); save_file(tensors, src, {"format": "pt", "who": "me"})
    a = safetensors_io._compress_file_on_device(src, str(tmp_path / "a.znn.safetensors"), torch.device("cpu"), None)
    b = safetensors_io.compress_safetensors_file(src, out_path=str(tmp_path / "b.znn.safetensors"), device="cpu")
    assert _same_safetensors_container(a, b)
    out = safetensors_io.load_file(a, device="cpu")
    assert all(torch.equal(out[k].view(torch.uint8) if out[k].numel() else out[k], tensors[k].view(torch.uint8) if tensors[k].numel() else tensors[k]) for k in tensors)


def _st_file(tmp_path, tensors, name="m"):
    from safetensors.torch import save_file
    from zipnn_amd.safetensors_io import compress_safetensors_file
    src = os.path.join(tmp_path, name + ".safetensors")
    save_file(tensors, src, {"format": "pt"})
    return compress_safetensors_file(src)


def test_get_slice_of_a_compressed_tensor_decodes_only_the_covering_chunks(use_simt, tmp_path):
    """SafeOpen.get_slice on a compressed tensor (the reference returns NotImplementedError, zipnn.py:1615-1617):
    slices equal get_tensor(name)[index] bit for bit, on every dtype, for the index forms a tensor-parallel loader uses (row ranges,
    a column range behind a full first dimension, an int, a step), and a row range decodes only the chunks that cover it."""
    from zipnn_amd import zipnn as Z
    g = torch.Generator().manual_seed(21)
    tensors = {"w_bf16": (torch.randn(1000, 640, generator=g) * 0.02).to(torch.bfloat16),        # 1.28 MB: 5 chunks, the last one partial
               "w_fp32": torch.randn(500, 300, generator=g) * 0.02,                              # 600 KB: 3 chunks
               "w_fp16": (torch.randn(700, 256, generator=g) * 0.02).half(),
               "w_fp8": (torch.randn(600, 512, generator=g) * 0.02).to(torch.float8_e4m3fn),     # 128 KiB chunks
               "vec": (torch.randn(5000, generator=g) * 0.02).to(torch.bfloat16), "ids": torch.arange(64)}
    out = _st_file(tmp_path, tensors)
    same = lambda a, b: a.dtype == b.dtype and a.shape == b.shape and a.contiguous().reshape(-1).view(torch.uint8).numpy().tobytes() == b.contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()  # noqa: E731
    with Z.SafeOpen(out, framework="pt", device="cpu") as f:
        assert f.get_slice("ids")[3:5].tolist() == [3, 4]                 # plain tensors: safetensors' own slice object
        for k in ("w_bf16", "w_fp32", "w_fp16", "w_fp8", "vec"):
            assert k in f.compressed_tensors_metadata
            v = tensors[k]
            s = f.get_slice(k)
            assert s.get_shape() == list(v.shape) and s.get_dtype() == {"w_bf16": "BF16", "w_fp32": "F32", "w_fp16": "F16", "w_fp8": "F8_E4M3", "vec": "BF16"}[k]
            rows = v.shape[0]
            for idx in [slice(0, rows), slice(rows // 2, rows), slice(1, 2), slice(rows - 3, None), slice(7, 7), slice(None, None, 3), 5, -1,
                        slice(rows // 4, rows // 2)]:
                assert same(s[idx], v[idx]), (k, idx)
            if v.dim() == 2:
                cols = v.shape[1]
                for idx in [(slice(None), slice(0, cols // 2)), (slice(10, 20), slice(cols // 2, cols)), (3, slice(1, 9)), (Ellipsis, slice(0, 4)), (slice(5, 50, 5), 2)]:
                    assert same(s[idx], v[idx]), (k, idx)
        # only the covering chunks are decoded: rows 400 .. 449 of the bf16 matrix are bytes [512000, 576000) = chunks 1 .. 2 of 256 KiB
        s = f.get_slice("w_bf16")
        part = s[400:450]
        assert s.last_chunk_range == (1, 3) and part.untyped_storage().nbytes() <= 2 * 256 * 1024
        assert same(part, tensors["w_bf16"][400:450])
        _ = s[999:1000]
        assert s.last_chunk_range == (4, 5)                                 # the partial last chunk alone
        s8 = f.get_slice("w_fp8"); _ = s8[0:256]
        assert s8.last_chunk_range == (0, 1)                                # 256 rows x 512 bytes = the first 128 KiB chunk of an fp8 tensor
        with pytest.raises(IndexError):
            s[1000]
    os.environ["ZIPNN_AMD_REFERENCE_GET_SLICE"] = "1"
    try:
        with Z.SafeOpen(out, framework="pt", device="cpu") as f:
            assert f.get_slice("w_bf16") is NotImplementedError            # the reference's answer, on request
    finally:
        del os.environ["ZIPNN_AMD_REFERENCE_GET_SLICE"]


def test_get_slice_of_a_zero_dim_compressed_tensor(use_simt, tmp_path):
    """a 0-dim tensor LEFT COMPRESSED in a file (neither writer does that — the frame is larger than the value — but a third party's file
    may): CompressedSlice used to take its 'empty' branch and hand back uninitialised memory.  A hand-built file: the frame of a scalar, listed in
    znn_compressed_vectors with shape []."""
    import json
    from safetensors.torch import save_file
    from zipnn_amd import ZipNN
    from zipnn_amd import zipnn as Z
    for val, dt in ((torch.tensor(3.25, dtype=torch.bfloat16), "BF16"), (torch.tensor(-7.5, dtype=torch.float32), "F32")):
        frame = bytes(ZipNN(input_format="torch").compress(val.clone()))
        stored = torch.frombuffer(bytearray(frame), dtype=torch.uint8)
        path = os.path.join(tmp_path, f"scalar_{dt}.znn.safetensors")
        save_file({"s": stored, "plain": torch.arange(4)}, path,
                  {"format": "pt", Z.METADATA_KEY: json.dumps({"s": Z.build_compressed_tensor_info(val)})})
        with Z.SafeOpen(path, framework="pt", device="cpu") as f:
            assert torch.equal(f.get_tensor("s"), val)
            sl = f.get_slice("s")
            assert sl.get_shape() == [] and sl.get_dtype() == dt
            for idx in ((), Ellipsis):
                got = sl[idx]
                assert got.shape == () and got.dtype == val.dtype and got.item() == val.item(), (dt, idx)
            assert sl.last_chunk_range == (0, 1)


def test_read_ahead_failures_fall_back_to_the_per_tensor_path(use_simt, tmp_path, monkeypatch):
    """whatever goes wrong inside the plugin's read-ahead (one corrupt frame in the file, an allocation that does not fit)
    only switches the read-ahead off — every healthy tensor still loads, and the error surfaces for the tensor that has it, as with the
    reference's per-tensor semantics (zipnn.py:1592-1626).  The read-ahead's tensors own their allocations (no shared arena)."""
    from zipnn_amd import zipnn as Z
    from zipnn_amd import safetensors_io
    from zipnn_amd._capi import ZnError
    tensors = _model(8)
    out = _st_file(tmp_path, tensors)
    with Z.SafeOpen(out, framework="pt", device="cpu") as f:
        f._read_ahead()
        assert isinstance(f._ahead, dict)
        t = f._ahead["w_fp8"]
        assert t.untyped_storage().nbytes() <= t.numel() + 64               # (10 000 bytes of fp8: its own allocation, not a view of an arena of all of them)
    meta, layout, data_start = safetensors_io._read_layout(out)
    lo, hi = layout["w_fp32"][2], layout["w_fp32"][3]
    blob = bytearray(open(out, "rb").read())
    for i in range(data_start + hi - 2000, data_start + hi - 1000):      # damage the payload of ONE frame (its header stays valid)
        blob[i] ^= 0xA5
    bad = os.path.join(tmp_path, "bad.znn.safetensors")
    open(bad, "wb").write(blob)
    with Z.SafeOpen(bad, framework="pt", device="cpu") as f:
        f._read_ahead()
        assert f._ahead is False                                             # the batched decode reported the damage: read-ahead off
        for k in ("w_bf16", "w_fp16", "w_fp8", "tiny", "ids"):
            assert torch.equal(f.get_tensor(k).view(torch.uint8), tensors[k].contiguous().view(torch.uint8)), k
        try:
            got = f.get_tensor("w_fp32")                                     # damaged streams either fail to decode or decode to other bytes
            assert not torch.equal(got, tensors["w_fp32"])
        except (ZnError, RuntimeError, MemoryError):
            pass
    # an allocation failure inside the read-ahead
    def boom(*a, **k):
        raise MemoryError("no room")
    monkeypatch.setattr(safetensors_io, "decode_file_on_device", boom)
    with Z.SafeOpen(out, framework="pt", device="cpu") as f:
        f._read_ahead()
        assert f._ahead is False
        assert torch.equal(f.get_tensor("w_bf16").view(torch.uint8), tensors["w_bf16"].contiguous().view(torch.uint8))


def test_frame_heads_from_a_file_are_validated(use_simt):
    """fast_frame_params reads untrusted bytes — a short head, an impossible chunk exponent, a shape that does not
    match the original length all raise ValueError (the loaders fall back to / report through the per-tensor path)."""
    from zipnn_amd import ZipNN
    from zipnn_amd.zipnn import fast_frame_params
    t = (torch.randn(40, 50) * 0.02).to(torch.bfloat16)
    frame = bytes(ZipNN(input_format="torch").compress(t))
    fp = fast_frame_params(memoryview(frame))
    assert fp[5] == 4000 and tuple(fp[7]) == (40, 50) and fp[1] == 2
    with pytest.raises(ValueError):
        fast_frame_params(memoryview(frame[:20]))
    with pytest.raises(ValueError):
        fast_frame_params(memoryview(frame[:33]))                            # the shape extension is cut off
    b = bytearray(frame); b[14] = 255
    with pytest.raises(ValueError):
        fast_frame_params(memoryview(bytes(b)))
    b = bytearray(frame); b[34] = 41                                         # shape (41, 50): 4100 bytes, the header says 4000
    with pytest.raises(ValueError):
        fast_frame_params(memoryview(bytes(b)))
