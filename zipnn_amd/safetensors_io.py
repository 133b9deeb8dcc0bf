"""Tensor-by-tensor safetensors compression (the file producer/consumer either side of the hot path;
SURVEY.md §8f-1).  Same file layout as the reference's scripts/zipnn_compress_safetensors.py:37-148 and
scripts/zipnn_decompress_safetensors.py:34-136: every floating-point tensor that shrinks is stored as a
1-D uint8 tensor holding one TORCH-format ZN frame, and listed (dtype, shape) in the file metadata under
`znn_compressed_vectors`; everything else is stored untouched.  Suffix: `.znn.safetensors`.
"""
import os
import struct

import torch

from .zipnn import (COMPRESSED_DTYPE, COMPRESSION_METHOD, METADATA_KEY, ZipNN, build_compressed_tensor_info,
                    get_compressed_tensors_metadata, set_compressed_tensors_metadata)

SUFFIX = ".znn.safetensors"


def compress_safetensors_file(filename, out_path=None, device="cpu", method=None, batched=None):
    """-> path of the compressed file.  `device` = where tensors are staged for compression
    ("cuda:N" compresses in HBM; the compressed frames come back to the host for writing).  Tensors staged in HBM
    are compressed by ONE batched call for the whole file (`batched=None`: automatic; True forces it)."""
    from safetensors import safe_open
    from safetensors.torch import save_file
    assert filename.endswith(".safetensors")
    out_path = out_path or filename[: -len(".safetensors")] + SUFFIX
    if torch.device(device).type == "cuda" and batched is not False:
        done = _compress_file_on_device(filename, out_path, torch.device(device), method)
        if done is not None:
            return done
    tensors, infos = {}, {}
    batch = []                                        # (name, tensor, header, planes, bits, bytes, chunk)
    with safe_open(filename, "pt", device) as f:
        for name in f.keys():
            t = f.get_tensor(name)
            if not torch.is_floating_point(t) or t.dtype == torch.float64 or t.numel() == 0:
                tensors[name] = t.cpu()
                continue
            znn = ZipNN(input_format="torch", bytearray_dtype=t.dtype, method=method or COMPRESSION_METHOD)
            if t.is_cuda if batched is None else batched:
                batch.append((name, t) + znn.torch_frame_plan(t) + (znn.compression_threshold,))
                continue
            frame = znn.compress(t)                       # our compress never modifies `t`
            if len(frame) >= t.element_size() * t.nelement():
                tensors[name] = t.cpu()
                continue
            tensors[name] = torch.frombuffer(bytearray(frame), dtype=COMPRESSED_DTYPE)
            infos[name] = build_compressed_tensor_info(t)
        metadata = dict(f.metadata() or {})
    if batch:
        # tensors staged in HBM: one batched compress for the whole file (zn_compress_batch_dev)
        from . import _capi, codec
        bodies = codec.compress_device_batch(_capi.lib(), [(codec.flat_bytes(t), P, bits, byts, chunk, th) for (_, t, _, P, bits, byts, chunk, th) in batch])
        for (name, t, hdr, *_), body in zip(batch, bodies):
            total = len(hdr) + body.numel()
            if total >= t.element_size() * t.nelement():
                tensors[name] = t.cpu()
                continue
            frame = codec.new_bytearray(total)
            frame[:len(hdr)] = hdr
            frame[24:32] = total.to_bytes(8, "little")     # what the core writes at zipnn_core.c:121
            codec.to_host(_capi.lib(), body, memoryview(frame)[len(hdr):])
            tensors[name] = torch.frombuffer(frame, dtype=COMPRESSED_DTYPE)
            infos[name] = build_compressed_tensor_info(t)
    if not metadata:
        metadata = {"format": "pt"}                       # the reference silently drops the list when a file has no metadata
    set_compressed_tensors_metadata(infos, metadata)
    save_file(tensors, out_path, metadata)
    return out_path


def _compress_file_on_device(filename, out_path, dev, method):
    """compress_safetensors_file with the tensors staged in HBM, the way decode_file_on_device loads: the file's data section crosses PCIe ONCE
    (pinned multi-threaded upload of the mapping), ONE batched compress writes every body into one arena — each behind a 256-byte gap —, the
    arena comes back in ONE transfer and the frame headers are written into the gaps on the host: the frames handed to safetensors are views of that one
    buffer.  (Per tensor — safe_open + get_tensor in, one transfer per body out — the same file took 35 + 64 ms for these two legs; now ≈ 10 + 10.)
    -> out_path, or None when the container names a dtype this parser does not know (the caller falls back to the per-tensor path)."""
    import contextlib
    import mmap
    import threading
    from safetensors.torch import save_file
    from . import _capi, codec
    from .zipnn import dtype_from_user
    lay = _read_layout(filename)
    if lay is None:
        return None
    metadata, layout, data_start = lay
    metadata = dict(metadata)
    lib = _capi.lib()
    from safetensors import safe_open
    with safe_open(filename, "pt", "cpu") as f0:                  # (header only: the order the per-tensor path walks the names in — the metadata list keeps it)
        order = [n for n in f0.keys() if n in layout]
    if len(order) != len(layout):
        return None
    with open(filename, "rb") as f:
        size = os.fstat(f.fileno()).st_size
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if size else None
    view = memoryview(mm) if mm is not None else None
    tensors, infos, batch = {}, {}, []
    try:
        for name in order:
            dt, shape, lo, hi = layout[name]
            numel = 1
            for d in shape:
                numel *= int(d)
            if not dt.is_floating_point or dt == torch.float64 or numel == 0 or dtype_from_user(dt) is None:
                tensors[name] = torch.frombuffer(bytearray(view[data_start + lo: data_start + hi]), dtype=dt).reshape(shape) if hi > lo else torch.empty(shape, dtype=dt)
                continue
            znn = ZipNN(input_format="torch", bytearray_dtype=dt, method=method or COMPRESSION_METHOD)
            like = torch.empty(shape, dtype=dt, device="meta")                   # (dtype and shape are all the frame plan looks at)
            batch.append((name, lo, hi, like) + znn.torch_frame_plan(like) + (znn.compression_threshold,))
        blob = codec.to_device(lib, view[data_start:], dev) if (view is not None and size > data_start) else torch.empty(0, dtype=torch.uint8, device=dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    finally:
        if view is not None:
            view.release()
        if mm is not None:
            def _close_mapping():
                with contextlib.suppress(BufferError):
                    mm.close()
            closer = threading.Thread(target=_close_mapping, daemon=True)      # (munmap behind the multi-threaded upload: 1.2 ms, see decode_file_on_device)
            closer.start()
            _PENDING_CLOSERS.append(closer)
    if batch:
        GAP = 256 * ((max(len(b[4]) for b in batch) + 255) // 256)              # room for the longest frame header in front of every body
        arena, offs, lens = codec.compress_device_batch(lib, [(blob[lo:hi], P, bits, byts, chunk, th) for (_, lo, hi, _, _, P, bits, byts, chunk, th) in batch],
                                                       gap=GAP, return_arena=True)
        end = max(o + n for o, n in zip(offs, lens))
        # one transfer: every body (and the unused tail of every slot in between), into a block of the library's pinned arena (one DMA, no page faults, nothing to munmap)
        host = codec.to_host(lib, arena[:end], out=lib.host_buffer(end))
        hv = memoryview(host).cast("B")
        for (name, lo, hi, like, hdr, *_), o, n in zip(batch, offs, lens):
            total = len(hdr) + n
            if total >= hi - lo:                                                   # did not shrink: stored as it was
                tensors[name] = blob[lo:hi].cpu().view(like.dtype).reshape(like.shape)
                continue
            s0 = o - len(hdr)
            hv[s0:o] = bytes(hdr)
            hv[s0 + 24: s0 + 32] = total.to_bytes(8, "little")                     # what the core writes at zipnn_core.c:121
            tensors[name] = torch.frombuffer(host, dtype=COMPRESSED_DTYPE, offset=s0, count=total)
            infos[name] = build_compressed_tensor_info(like)
    if not metadata:
        metadata = {"format": "pt"}                                                # the reference silently drops the list when a file has no metadata
    set_compressed_tensors_metadata(infos, metadata)
    save_file({name: tensors[name] for name in order}, out_path, metadata)
    return out_path


def decompress_safetensors_file(filename, out_path=None, device="cpu"):
    """Inverse of compress_safetensors_file -> path of the plain .safetensors file."""
    from safetensors import safe_open
    from safetensors.torch import save_file
    assert filename.endswith(SUFFIX)
    out_path = out_path or filename[: -len(SUFFIX)] + ".safetensors"
    tensors = {}
    with safe_open(filename, "pt", "cpu") as f:
        metadata = dict(f.metadata() or {})
        infos = get_compressed_tensors_metadata(metadata)
        for name in f.keys():
            t = f.get_tensor(name)
            if name in infos:
                znn = ZipNN(input_format="torch", bytearray_dtype=COMPRESSED_DTYPE, method=COMPRESSION_METHOD)
                t = znn.decompress(t, decompress_cpu_gpu=device)
            tensors[name] = t.cpu() if isinstance(t, torch.Tensor) else t
    metadata.pop(METADATA_KEY, None)
    save_file(tensors, out_path, metadata or None)
    return out_path


_ST_DTYPES = {"F64": torch.float64, "F32": torch.float32, "F16": torch.float16, "BF16": torch.bfloat16, "I64": torch.int64, "I32": torch.int32,
              "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8, "BOOL": torch.bool, "F8_E4M3": getattr(torch, "float8_e4m3fn", None),
              "F8_E5M2": getattr(torch, "float8_e5m2", None), "U16": getattr(torch, "uint16", None), "U32": getattr(torch, "uint32", None),
              "U64": getattr(torch, "uint64", None)}


def _read_layout(filename):
    """The safetensors container itself: 8-byte little-endian header length, a JSON header {name: {dtype, shape, data_offsets}}
    (+ "__metadata__"), then the tensors' bytes back to back.  -> (metadata, {name: (dtype, shape, lo, hi)}, data_start), or None
    when the header names a dtype this loader does not know (the per-tensor path then reads the file through safetensors)."""
    import json
    with open(filename, "rb") as f:
        n = int.from_bytes(f.read(8), "little")
        hdr = json.loads(f.read(n))
    meta = hdr.pop("__metadata__", None) or {}
    layout = {}
    for name, e in hdr.items():
        dt = _ST_DTYPES.get(e["dtype"])
        if dt is None:
            return None
        layout[name] = (dt, tuple(e["shape"]), int(e["data_offsets"][0]), int(e["data_offsets"][1]))
    return meta, layout, 8 + n


def _contiguous_strides(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= max(int(d), 1)
    return tuple(reversed(st))


_PENDING_CLOSERS = []      # helper threads still unmapping the file of an earlier decode_file_on_device call


def decode_file_on_device(filename, device, compressed_only=False, timings=None, use_arena=None):
    """The batched loader behind load_file and behind the plugin's read-ahead (SafeOpen.get_tensor): the file's data section
    crosses PCIe ONCE, as it lies on disk, through the library's pinned multi-threaded transfer; every compressed tensor is then
    decoded by ONE batched launch (zn_decompress_batch_dev) from where its frame landed in HBM into one output arena (every
    tensor at a 256-byte boundary of it).  All frame headers are parsed from the host mapping before anything moves — one pass over
    the mmap with zipnn.fast_frame_params, no ZipNN object and no device read-back per tensor — and the bodies and destinations
    go to the library as plain addresses.  -> {name: tensor} (with compressed_only: the compressed tensors alone), or None when the
    container names a dtype this parser does not know.
    The uncompressed tensors of the file are COPIED out of the uploaded section, so that nothing keeps the compressed bytes
    resident once the decode has run; the decoded tensors share the arena (they live and die together in a model load; set
    ZIPNN_AMD_LOAD_ARENA=0, or pass use_arena=False, for one allocation per tensor — what the plugin's read-ahead does, whose
    consumers may keep any subset of a shard)."""
    import contextlib
    import mmap
    import threading
    import time
    from . import _capi, codec
    from .zipnn import fast_frame_params
    dev = torch.device(device)
    while _PENDING_CLOSERS:                            # mappings of earlier calls, unmapped on helper threads (below)
        _PENDING_CLOSERS.pop().join()
    lay = _read_layout(filename)
    if lay is None:
        return None
    lib = _capi.lib()
    t0 = time.perf_counter()
    metadata, layout, data_start = lay
    infos = get_compressed_tensors_metadata(dict(metadata))
    with open(filename, "rb") as f:
        size = os.fstat(f.fileno()).st_size
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if size else None
    view = memoryview(mm) if mm is not None else None
    head_len = 32 + 1 + 9 * 255                        # header + the largest shape extension (zipnn._frame_head)
    if use_arena is None:                              # (load_file: everything is returned together; SafeOpen's read-ahead passes False)
        use_arena = os.environ.get("ZIPNN_AMD_LOAD_ARENA", "1") != "0"
    try:
        # ---- host side: one pass over the mapping ----
        plan, total = [], 0                            # (name, lo + body_off, hi, fp, arena offset)
        for name, (dt, shape, lo, hi) in layout.items():
            if name in infos:
                fp = fast_frame_params(view[data_start + lo: data_start + min(hi, lo + head_len)])
                plan.append((name, lo + fp[0], hi, fp, total))
                total += (fp[5] + 255) & ~255
        t1 = time.perf_counter()
        blob = codec.to_device(lib, view[data_start:], dev) if (view is not None and size > data_start) else torch.empty(0, dtype=torch.uint8, device=dev)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
    finally:
        if view is not None:
            view.release()
        closer = None
        if mm is not None:
            # Unmapping a file that the upload's worker threads have just read costs ≈ 1.2 ms on a 256-CPU host (TLB shoot-downs) — as long as
            # the whole decode of a GPT-2 checkpoint.  mmap.close() drops the GIL around munmap, so it runs on a helper thread under the launches below
            # (measured: decode_s 1.65 -> 0.59 ms; process exit unmaps whatever a daemon thread has not).
            def _close_mapping():
                with contextlib.suppress(BufferError):     # (a traceback may still hold slices of the mapping: the real error must not be replaced by this one)
                    mm.close()
            closer = threading.Thread(target=_close_mapping, daemon=True)
            closer.start()
            _PENDING_CLOSERS.append(closer)                # (joined by the next call — by then long finished — not by this one: nothing below needs the mapping gone)
    return _decode_uploaded(lib, codec, dev, layout, infos, plan, total, blob, use_arena, compressed_only, timings, (t0, t1, t2))


def _decode_uploaded(lib, codec, dev, layout, infos, plan, total, blob, use_arena, compressed_only, timings, marks):
    """Second half of decode_file_on_device: the data section is in HBM (`blob`), `plan` holds every compressed tensor's frame parameters."""
    import time
    from . import _capi
    t0, t1, t2 = marks
    out = {}
    ta_ = tb_ = tc_ = t2
    if plan:
        arena = torch.empty(max(total, 16), dtype=torch.uint8, device=dev) if use_arena else None
        outs = [None] * len(plan) if use_arena else [torch.empty(fp[5], dtype=torch.uint8, device=dev) for (_, _, _, fp, _) in plan]
        base_in, base_out = blob.data_ptr(), (arena.data_ptr() if use_arena else 0)
        pack = struct.Struct(_capi.ZN_BATCH_ITEM_FMT).pack
        packed = b"".join([pack(base_in + b0, hi - b0, (((base_out + off) if use_arena else outs[i].data_ptr()) if fp[5] else 0), fp[5],
                                fp[1], fp[2], fp[3], fp[4], 0) for i, (_, b0, hi, fp, off) in enumerate(plan)])
        stream = codec._stream_handle(blob)
        with torch.cuda.device(dev) if dev.type == "cuda" else codec._nullctx():
            lib.decompress_batch_dev_packed(packed, len(plan), stream, check=False)      # asynchronous: the verdict is asked for below
        ta_ = time.perf_counter()
        # (views while the kernels run: one typed view of the arena per dtype, one as_strided per tensor)
        typed = {}
        for i, (name, _, _, fp, off) in enumerate(plan):
            n, dt, shape = fp[5], fp[6], tuple(fp[7]) if fp[7] is not None else None
            if n == 0:
                out[name] = torch.empty(shape if shape is not None else (0,), dtype=dt, device=dev)
                continue
            if use_arena:
                ta = typed.get(dt)
                if ta is None:
                    ta = typed[dt] = arena.view(dt)
                es = ta.element_size()
                shp = shape if shape is not None else (n // es,)
                out[name] = torch.as_strided(ta, shp, _contiguous_strides(shp), off // es)
            else:
                out[name] = outs[i].view(dt).reshape(shape) if shape is not None else outs[i].view(dt)
    tb_ = time.perf_counter()
    if plan:
        with torch.cuda.device(dev) if dev.type == "cuda" else codec._nullctx():
            lib.decode_status(stream)                  # waits for the decode; raises for a corrupt frame exactly as a checked call would
    tc_ = time.perf_counter()
    if not compressed_only:
        for name, (dt, shape, lo, hi) in layout.items():
            if name not in infos:
                # (a copy, not a view: one surviving int tensor would otherwise pin the whole uploaded section — all compressed frames — in HBM)
                out[name] = blob[lo:hi].clone().view(dt).reshape(shape) if hi > lo else torch.empty(shape, dtype=dt, device=dev)
    if timings is not None:
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        # (decode_s in parts: arena + item table + launch | views built while the kernels run | wait for the kernels' verdict | copies of the plain tensors + sync)
        timings.update(decode_launch_s=ta_ - t2, decode_views_s=tb_ - ta_, decode_wait_s=tc_ - tb_, decode_plain_s=t3 - tc_)
        timings.update(read_s=t1 - t0, h2d_s=t2 - t1, decode_s=t3 - t2, compressed_tensors=len(plan), h2d_bytes=int(blob.numel()),
                       compressed_bytes=int(sum(hi - b0 for (_, b0, hi, _, _) in plan)), decoded_bytes=int(sum(p[3][5] for p in plan)))
    return out


def load_file(filename, device="cuda:0", timings=None):
    """Load a (possibly ZipNN-compressed) safetensors file straight onto `device` -> {name: tensor}: one transfer of the file's data
    section, one batched decode (decode_file_on_device).  The batched counterpart of looping SafeOpen.get_tensor (reference
    zipnn.py:1592-1626, scripts/zipnn_decompress_safetensors.py:75-120) — and what SafeOpen itself uses behind get_tensor for a
    device target.  timings: an optional dict that receives the seconds spent mapping the file and parsing every header (`read_s`),
    moving the data section to the device (`h2d_s`) and decoding (`decode_s`), each ended by a device sync."""
    dev = torch.device(device)
    out = decode_file_on_device(filename, dev, timings=timings)
    if out is None:
        return _load_file_per_tensor(filename, dev, timings)
    # (file order, as safetensors' own load_file returns it)
    return out


def _load_file_per_tensor(filename, dev, timings=None):
    """load_file through safetensors' own reader, tensor by tensor (CPU targets, dtypes the container parser above does not know)."""
    import time
    from safetensors import safe_open
    from . import _capi, codec

    def _sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out, host, meta = {}, [], []
    with safe_open(filename, "pt", "cpu") as f:
        infos = get_compressed_tensors_metadata(dict(f.metadata() or {}))
        for name in f.keys():
            host.append((name, f.get_tensor(name)))
    t1 = time.perf_counter()
    items = []
    for name, t in host:
        if name not in infos:
            out[name] = t.to(dev, non_blocking=True)
            continue
        znn = ZipNN(input_format="torch", bytearray_dtype=COMPRESSED_DTYPE, method=COMPRESSION_METHOD)
        fp = znn.frame_params(t)
        host_body = t.reshape(-1).view(torch.uint8)[fp["body_off"]:]
        # (large bodies through the library's pinned multi-threaded transfer; small ones are not worth its threads)
        body = codec.to_device(_capi.lib(), host_body.numpy(), dev) if host_body.numel() >= (2 << 20) and dev.type == "cuda" \
            else host_body.to(dev, non_blocking=True)
        items.append((body, fp["num_buf"], fp["bits_mode"], fp["bytes_mode"], fp["chunk"], fp["orig_size"]))
        meta.append((name, fp["torch_dtype"], fp["shape"]))
    if timings is not None:
        _sync()
    t2 = time.perf_counter()
    flats = codec.decompress_device_batch(_capi.lib(), items)
    for (name, dtype, shape), flat in zip(meta, flats):
        out[name] = flat.view(dtype).reshape(shape) if flat.numel() else torch.empty(shape, dtype=dtype, device=dev)
    if timings is not None:
        _sync()
        t3 = time.perf_counter()
        timings.update(read_s=t1 - t0, h2d_s=t2 - t1, decode_s=t3 - t2, compressed_tensors=len(items),
                       compressed_bytes=int(sum(it[0].numel() for it in items)), decoded_bytes=int(sum(it[5] for it in items)))
    return out
