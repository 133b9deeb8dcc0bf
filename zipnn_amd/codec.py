"""torch <-> C-ABI glue for buffers that live in HBM.

torch is plumbing here (device memory, streams); the work is in libzipnn_hip.so.  The
functions take a `lib` (zipnn_amd._capi.ZnLib) so that the CPU test-suite can drive the
same code through the SIMT-emulated build of the kernels with CPU tensors.
"""
import torch


def current_device():
    return torch.cuda.current_device() if torch.cuda.is_available() else 0


def flat_bytes(t):
    """Contiguous 1-D uint8 view of a tensor's storage bytes (no copy when contiguous)."""
    t = t.contiguous()
    if t.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        return t.view(torch.uint8).reshape(-1)
    return t.reshape(-1).view(torch.uint8)


def to_device(lib, buf, dev):
    """bytes-like (pageable host memory) -> uint8 tensor on `dev`, through the library's pinned, multi-threaded
    transfer (zn_copy_to_device) instead of a pageable torch copy.  The copy runs on the library's own stream and
    is complete on return; torch's stream is drained first because the allocator may hand out a block that kernels
    queued earlier are still using."""
    mv = memoryview(buf).cast("B")
    t = torch.empty(mv.nbytes, dtype=torch.uint8, device=dev)
    if mv.nbytes:
        if t.is_cuda:
            with torch.cuda.device(t.device):
                torch.cuda.current_stream().synchronize()
                lib.copy_to_device(t.data_ptr(), mv)
        else:
            lib.copy_to_device(t.data_ptr(), mv)
    return t


def new_bytearray(n):
    """bytearray of n bytes WITHOUT the zero fill of bytearray(n) (128 ms for 256 MiB: every page touched by one
    thread before the data arrives) — the transfer that fills it touches the pages from several threads."""
    import ctypes
    try:
        f = ctypes.pythonapi.PyByteArray_FromStringAndSize
        f.restype = ctypes.py_object
        f.argtypes = [ctypes.c_char_p, ctypes.c_ssize_t]
        ba = f(None, n)
        if isinstance(ba, bytearray) and len(ba) == n:
            return ba
    except Exception:
        pass
    return bytearray(n)


def to_host(lib, t, out=None):
    """uint8 tensor (device) -> writable host buffer (a new bytearray unless `out` is given), the same way back."""
    t = t.contiguous()
    n = t.numel()
    if out is None:
        out = new_bytearray(n)
    if n:
        if t.is_cuda:
            with torch.cuda.device(t.device):
                torch.cuda.current_stream().synchronize()      # the kernels that produced `t` ran on torch's stream
                lib.copy_to_host(memoryview(out).cast("B")[:n], t.data_ptr())
        else:
            lib.copy_to_host(memoryview(out).cast("B")[:n], t.data_ptr())
    return out


def _stream_handle(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _delta_ptr(delta, like):
    """delta base: None, or a uint8 tensor of the same length on the same device (contiguous)."""
    if delta is None:
        return None, None
    delta = delta.contiguous()
    if delta.dtype != torch.uint8 or delta.numel() != like.numel() or delta.device != like.device:
        raise ValueError("delta base must be a uint8 tensor of the same length on the same device")
    return delta, (delta.data_ptr() if delta.numel() else None)


def compress_device(lib, flat, num_buf, bits_mode, bytes_mode, chunk, threshold, delta=None, body=None):
    """flat: uint8 tensor in HBM -> uint8 tensor (same device) holding the frame BODY
    (types ‖ cumSizes ‖ payload).  One 8-byte read-back for the length.
    delta: optional uint8 tensor of the same length — the body then encodes flat ^ delta, the XOR being fused
    into the kernels that read the tensor (the reference's delta step, zipnn/zipnn.py:625-640).
    body: optional preallocated uint8 buffer on the same device (>= zn_compress_bound bytes) the body is written into —
    a caller that compresses repeatedly (a checkpoint writer, bench.py) allocates it once; the result is a slice of it."""
    n = flat.numel()
    delta, dptr = _delta_ptr(delta, flat)
    cap = lib.compress_bound(n, num_buf, chunk, 0)
    if body is None:
        body = torch.empty(max(cap, 16), dtype=torch.uint8, device=flat.device)
    elif body.dtype != torch.uint8 or body.device != flat.device or body.numel() < max(cap, 16) or not body.is_contiguous():
        raise ValueError("body must be a contiguous uint8 tensor on the same device with at least zn_compress_bound bytes")
    with torch.cuda.device(flat.device) if flat.is_cuda else _nullctx():
        used = lib.compress_dev(flat.data_ptr() if n else 0, n, num_buf, bits_mode, bytes_mode, chunk, threshold,
                                body.data_ptr(), body.numel(), _stream_handle(flat), delta_ptr=dptr)
    return body[:used]


def compress_device_to_frame(lib, header, flat, num_buf, bits_mode, bytes_mode, chunk, threshold):
    """Device tensor -> host frame bytes (header ‖ body); only compressed bytes cross PCIe."""
    body = compress_device(lib, flat, num_buf, bits_mode, bytes_mode, chunk, threshold)
    hl = len(header)
    total = hl + body.numel()
    # (from 8 MiB up a block of the library's pinned arena — ZnLib.host_buffer —: the body then crosses PCIe as one DMA into memory that needs no page faults and
    #  goes back to the arena, not to munmap, when the caller drops the frame; the caller sees a memoryview either way, as from the reference's extension)
    frame = lib.host_buffer(total) if hasattr(lib, "host_buffer") and total >= getattr(lib, "HOST_ARENA_MIN", 1 << 62) else new_bytearray(total)
    fv = memoryview(frame).cast("B")
    fv[:hl] = bytes(header)
    if hl >= 32:
        fv[24:32] = total.to_bytes(8, "little")           # what the reference core writes at zipnn_core.c:121
    to_host(lib, body, fv[hl:])
    return frame


def decompress_device(lib, body, num_buf, bits_mode, bytes_mode, chunk, orig_size, out=None, check=True, delta=None):
    """body: uint8 tensor in HBM (frame minus header) -> uint8 tensor of orig_size bytes on
    the same device.  delta: optional uint8 tensor of orig_size bytes XORed into the output inside the
    kernels that write it (reference zipnn/zipnn.py:983-1004)."""
    if out is None:
        out = torch.empty(orig_size, dtype=torch.uint8, device=body.device)
    if orig_size == 0:
        return out
    body = body.contiguous()
    delta, dptr = _delta_ptr(delta, out)
    with torch.cuda.device(body.device) if body.is_cuda else _nullctx():
        lib.decompress_dev(body.data_ptr(), body.numel(), num_buf, bits_mode, bytes_mode, chunk, orig_size,
                           out.data_ptr(), _stream_handle(body), check, delta_ptr=dptr)
    return out


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def decompress_device_batch(lib, items, check=True, into=None):
    """Many tensors, one set of kernel launches (zn_decompress_batch_dev): small tensors fill the device together.
    items: iterable of (body, num_buf, bits_mode, bytes_mode, chunk, orig_size[, delta]) with `body` a uint8 tensor on
    the device (frame minus header) and `delta` an optional uint8 base of orig_size bytes.  Returns the list of
    decoded uint8 tensors (same device)."""
    items = list(items)
    deltas = [(it[6].contiguous() if len(it) > 6 and it[6] is not None else None) for it in items]
    items = [(it[0].contiguous(),) + tuple(it[1:6]) for it in items]
    if not items:
        return []
    for d, it in zip(deltas, items):
        if d is not None and (d.dtype != torch.uint8 or d.numel() != it[5] or d.device != it[0].device):
            raise ValueError("delta base must be a uint8 tensor of orig_size bytes on the same device")
    dev = items[0][0].device
    if into is not None:                       # one preallocated buffer: tensor i lands at the running offset
        assert into.numel() >= sum(n for (_, _, _, _, _, n) in items)
        offs, o = [], 0
        for (_, _, _, _, _, n) in items:
            offs.append(o); o += n
        outs = [into[o:o + n] for o, (_, _, _, _, _, n) in zip(offs, items)]
    else:
        outs = [torch.empty(n, dtype=torch.uint8, device=dev) for (_, _, _, _, _, n) in items]
    with torch.cuda.device(dev) if dev.type == "cuda" else _nullctx():
        lib.decompress_batch_dev(((b.data_ptr(), b.numel(), nb, bi, by, ch, n, o.data_ptr() if n else 0,
                                   d.data_ptr() if (d is not None and n) else None)
                                  for (b, nb, bi, by, ch, n), o, d in zip(items, outs, deltas)), _stream_handle(items[0][0]), check)
    return outs


def compress_device_batch(lib, items, gap=0, return_arena=False):
    """Many tensors, one launch per stage (zn_compress_batch_dev), one read-back of all lengths.
    items: iterable of (flat_uint8_device_tensor, num_buf, bits_mode, bytes_mode, chunk, threshold[, delta]).
    Returns the list of body tensors (uint8, same device; slices of one arena).
    gap: bytes left free in FRONT of every body (a multiple of 256: the bodies stay 256-byte aligned) — room for the frame header when the arena goes
    to the host in one piece; return_arena: -> (arena, [offset of each body in it], [body lengths]) instead of the slices."""
    items = list(items)
    deltas = [(it[6].contiguous() if len(it) > 6 and it[6] is not None else None) for it in items]
    items = [(it[0].contiguous(),) + tuple(it[1:6]) for it in items]
    if not items:
        return []
    for d, it in zip(deltas, items):
        if d is not None and (d.dtype != torch.uint8 or d.numel() != it[0].numel() or d.device != it[0].device):
            raise ValueError("delta base must be a uint8 tensor of the same length on the same device")
    dev = items[0][0].device
    caps = [max(lib.compress_bound(f.numel(), nb, ch, 0), 16) for (f, nb, _, _, ch, _) in items]
    if gap % 256:
        raise ValueError("gap must be a multiple of 256")
    offs, o = [], 0
    for c in caps:
        o += gap
        offs.append(o); o += (c + 255) // 256 * 256
    arena = torch.empty(max(o, 16), dtype=torch.uint8, device=dev)
    base = arena.data_ptr()
    with torch.cuda.device(dev) if dev.type == "cuda" else _nullctx():
        lens = lib.compress_batch_dev(((f.data_ptr() if f.numel() else 0, f.numel(), nb, bi, by, ch, th, base + b0, c,
                                        d.data_ptr() if (d is not None and f.numel()) else None)
                                       for (f, nb, bi, by, ch, th), c, b0, d in zip(items, caps, offs, deltas)), _stream_handle(arena))
    if return_arena:
        return arena, offs, lens
    return [arena[b0:b0 + n] for b0, n in zip(offs, lens)]
