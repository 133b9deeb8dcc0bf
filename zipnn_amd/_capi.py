"""ctypes binding of libzipnn_hip.so (include/zipnn_hip.h).

This is the only route from Python to the codec: there is no CPU implementation behind
it.  If the shared library has not been built (``python -c "import __graft_entry__ as g;
g.build()"``) importing the codec fails loudly.

The two host-buffer calls mirror the reference's C extension one for one:
``zipnn_core.zipnn_core`` (reference csrc/zipnn_core.c:401, called at zipnn/zipnn.py:714)
-> :meth:`ZnLib.compress`, and ``zipnn_core.combine_dtype`` (csrc/zipnn_core.c:881, called
at zipnn/zipnn.py:1143) -> :meth:`ZnLib.decompress`.
"""
import ctypes
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libzipnn_hip.so"

ZN_OK, ZN_E_ARG, ZN_E_HIP, ZN_E_CAP, ZN_E_CORRUPT, ZN_E_TYPE, ZN_E_NODEV, ZN_E_ALLOC, ZN_E_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7, -8


class ZnError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(text)
        self.status = status


class ZnBatchItem(ctypes.Structure):
    """struct zn_batch_item of include/zipnn_hip.h"""
    _fields_ = [("d_body", ctypes.c_void_p), ("body_len", ctypes.c_size_t), ("d_dst", ctypes.c_void_p),
                ("orig_size", ctypes.c_size_t), ("num_buf", ctypes.c_int), ("bits_mode", ctypes.c_int),
                ("bytes_mode", ctypes.c_int), ("chunk", ctypes.c_size_t), ("d_delta", ctypes.c_void_p)]


ZN_BATCH_ITEM_FMT = "<QQQQiii4xQQ"          # struct zn_batch_item as struct.pack sees it (64 bytes)
assert ctypes.sizeof(ZnBatchItem) == 64


class ZnCBatchItem(ctypes.Structure):
    """struct zn_cbatch_item of include/zipnn_hip.h"""
    _fields_ = [("d_src", ctypes.c_void_p), ("n", ctypes.c_size_t), ("num_buf", ctypes.c_int), ("bits_mode", ctypes.c_int),
                ("bytes_mode", ctypes.c_int), ("chunk", ctypes.c_size_t), ("threshold", ctypes.c_float),
                ("d_body", ctypes.c_void_p), ("body_cap", ctypes.c_size_t), ("body_len", ctypes.c_size_t),
                ("d_delta", ctypes.c_void_p)]


class ZnLib:
    """A loaded libzipnn_hip.so."""

    def __init__(self, path):
        self.path = path
        L = ctypes.CDLL(path)
        sz, vp, ci, cf = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.zn_abi_version.restype = ci
        L.zn_strerror.restype = ctypes.c_char_p
        L.zn_strerror.argtypes = [ci]
        L.zn_last_hip_error.restype = ctypes.c_char_p
        L.zn_last_kernels.restype = ctypes.c_char_p
        L.zn_device_count.restype = ci
        L.zn_num_chunks.restype = sz
        L.zn_num_chunks.argtypes = [sz, sz]
        L.zn_compress_bound.restype = sz
        L.zn_compress_bound.argtypes = [sz, ci, sz, sz]
        L.zn_compress.restype = ci
        L.zn_compress.argtypes = [vp, sz, vp, sz, ci, ci, ci, sz, cf, ci, vp, sz, ctypes.POINTER(sz)]
        L.zn_decompress.restype = ci
        L.zn_decompress.argtypes = [vp, sz, ci, ci, ci, sz, sz, ci, vp]
        L.zn_compress_multi.restype = ci
        L.zn_compress_multi.argtypes = [vp, sz, vp, sz, ci, ci, ci, sz, cf, ctypes.POINTER(ci), ci, vp, sz, ctypes.POINTER(sz)]
        L.zn_decompress_multi.restype = ci
        L.zn_decompress_multi.argtypes = [vp, sz, ci, ci, ci, sz, sz, ctypes.POINTER(ci), ci, vp]
        vpp = ctypes.POINTER(ctypes.c_void_p)
        L.zn_multi_range.restype = ci
        L.zn_multi_range.argtypes = [sz, sz, ci, ci, ctypes.POINTER(sz), ctypes.POINTER(sz)]
        L.zn_decompress_multi_dev.restype = ci
        L.zn_decompress_multi_dev.argtypes = [vp, sz, ci, ci, ci, sz, sz, ctypes.POINTER(ci), ci, vpp]
        L.zn_compress_multi_dev.restype = ci
        L.zn_compress_multi_dev.argtypes = [vp, sz, vpp, sz, ci, ci, ci, sz, cf, ctypes.POINTER(ci), ci, vp, sz, ctypes.POINTER(sz)]
        L.zn_decompress_range_dev.restype = ci
        L.zn_decompress_range_dev.argtypes = [vp, sz, ci, ci, ci, sz, sz, sz, sz, ci, vp]
        L.zn_merge_range_bodies.restype = ci
        L.zn_merge_range_bodies.argtypes = [vpp, ctypes.POINTER(sz), ctypes.POINTER(sz), ci, ci, vp, sz, ctypes.POINTER(sz)]
        L.zn_set_host_slices.restype = ci
        L.zn_set_host_slices.argtypes = [ci]
        L.zn_set_host_direct.restype = ci
        L.zn_set_host_direct.argtypes = [ci]
        L.zn_host_alloc.restype = vp
        L.zn_host_alloc.argtypes = [sz]
        L.zn_host_free.restype = ci
        L.zn_host_free.argtypes = [vp]
        L.zn_set_decode_group.restype = ci
        L.zn_set_decode_group.argtypes = [ci]
        L.zn_decode_group_for.restype = ci
        L.zn_decode_group_for.argtypes = [ctypes.c_ulonglong]
        L.zn_set_encode_onepass.restype = ci
        L.zn_set_encode_onepass.argtypes = [ci]
        L.zn_set_decode_wide.restype = ci
        L.zn_set_decode_wide.argtypes = [ci]
        L.zn_compress_dev.restype = ci
        L.zn_compress_dev.argtypes = [vp, sz, ci, ci, ci, sz, cf, vp, sz, ctypes.POINTER(sz), vp]
        L.zn_decompress_dev.restype = ci
        L.zn_decompress_dev.argtypes = [vp, sz, ci, ci, ci, sz, sz, vp, vp, ci]
        L.zn_compress_delta.restype = ci
        L.zn_compress_delta.argtypes = [vp, sz, vp, vp, sz, ci, ci, ci, sz, cf, ci, vp, sz, ctypes.POINTER(sz)]
        L.zn_decompress_delta.restype = ci
        L.zn_decompress_delta.argtypes = [vp, sz, vp, ci, ci, ci, sz, sz, ci, vp]
        L.zn_compress_delta_dev.restype = ci
        L.zn_compress_delta_dev.argtypes = [vp, vp, sz, ci, ci, ci, sz, cf, vp, sz, ctypes.POINTER(sz), vp]
        L.zn_decompress_delta_dev.restype = ci
        L.zn_decompress_delta_dev.argtypes = [vp, sz, vp, ci, ci, ci, sz, sz, vp, vp, ci]
        L.zn_compress_batch_dev.restype = ci
        L.zn_compress_batch_dev.argtypes = [ctypes.POINTER(ZnCBatchItem), sz, vp]
        L.zn_decompress_batch_dev.restype = ci
        L.zn_decompress_batch_dev.argtypes = [ctypes.POINTER(ZnBatchItem), sz, vp, ci]
        L.zn_copy_to_device.restype = ci; L.zn_copy_to_device.argtypes = [vp, vp, sz]
        L.zn_copy_to_host.restype = ci; L.zn_copy_to_host.argtypes = [vp, vp, sz]
        L.zn_release_workspace.restype = ci
        L.zn_set_legacy_tree_descriptions.restype = ci
        L.zn_set_legacy_tree_descriptions.argtypes = [ci]
        L.zn_decode_status.restype = ci
        L.zn_decode_status.argtypes = [vp]
        L.zn_last_fused_chunks.restype = ctypes.c_longlong
        L.zn_last_tail_planes.restype = ctypes.c_longlong
        self._L = L
        if L.zn_abi_version() != 3:
            raise ImportError(f"{path}: unexpected ABI version {L.zn_abi_version()}")

    # -- error mapping: the Python-visible exceptions of the reference ------------------
    def _check(self, rc):
        if rc == ZN_OK:
            return
        text = self._L.zn_strerror(rc).decode()
        if rc == ZN_E_HIP:
            text += ": " + self._L.zn_last_hip_error().decode()
        if rc in (ZN_E_TYPE, ZN_E_ALLOC):
            # reference raises MemoryError for bad type bytes and failed allocations
            # (csrc/zipnn_core.c:993-996 and the malloc checks around it)
            raise MemoryError(text)
        if rc == ZN_E_ARG:
            raise ValueError(text)
        # worker failure in the reference surfaces as RuntimeError("Thread processing failed")
        # (csrc/zipnn_core.c:520-523,1088-1091)
        raise ZnError(rc, "Thread processing failed: " + text)

    def device_count(self):
        return self._L.zn_device_count()

    def compress_bound(self, n, num_buf, chunk, hdr_len):
        return self._L.zn_compress_bound(n, num_buf, chunk, hdr_len)

    def last_kernels(self):
        return self._L.zn_last_kernels().decode()

    # -- host buffers --------------------------------------------------------------------
    HOST_ARENA_MIN = 8 << 20      # results from this size up come out of the library's pinned arena (zn_host_alloc)

    def host_buffer(self, n):
        """An uninitialised uint8 numpy array of n bytes for a RESULT that crosses PCIe: from 8 MiB up a block of the library's pinned arena (zn_host_alloc — one
        DMA instead of a staged copy, no first-touch faults, recycled when the last view of it dies), below that (or when the driver has no pinned memory left, or
        ZIPNN_AMD_HOST_ARENA_MB=0) plain np.empty.  What the reference's extension does too: its results are memoryviews over memory it allocated (csrc/zipnn_core.c:596, 1126)."""
        n = max(int(n), 1)
        if n >= self.HOST_ARENA_MIN and os.environ.get("ZIPNN_AMD_HOST_ARENA_MB") != "0":
            p = self._L.zn_host_alloc(n)
            if p:
                try:
                    arr = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p)).view(_ArenaArray)
                    weakref.finalize(arr, self._L.zn_host_free, ctypes.c_void_p(p))     # every view / memoryview of it keeps `arr` alive
                    return arr
                except Exception:
                    self._L.zn_host_free(ctypes.c_void_p(p))
                    raise
        return np.empty(n, dtype=np.uint8)

    def compress(self, header, data, num_buf, bits_mode, bytes_mode, chunk, threshold, device=0, delta=None):
        """header/data: bytes-like (not modified).  Returns the frame as a writable memoryview over an
        uninitialised numpy buffer (a zero-filled 1 GiB bytearray alone costs 180 ms).  delta: bytes-like of
        the same length — the frame then holds data ^ delta (XOR fused into the device kernels)."""
        hv = memoryview(header).cast("B")
        dv = memoryview(data).cast("B")
        n = dv.nbytes
        cap = self._L.zn_compress_bound(n, num_buf, chunk, hv.nbytes)
        out = self.host_buffer(cap)
        out_len = ctypes.c_size_t(0)
        hb = (ctypes.c_char * max(hv.nbytes, 1)).from_buffer_copy(hv.tobytes() or b"\0")
        src = _as_c_buffer(dv)
        dl = _as_c_buffer(memoryview(delta).cast("B")) if delta is not None else None
        if dl is not None and memoryview(delta).nbytes != n:
            raise ValueError("delta buffer and data differ in length")
        if dl is None:      # the entry point the reference-side binding binds
            rc = self._L.zn_compress(ctypes.addressof(hb), hv.nbytes, src.addr, n, num_buf, bits_mode, bytes_mode, chunk,
                                     threshold, device, out.ctypes.data, cap, ctypes.byref(out_len))
        else:
            rc = self._L.zn_compress_delta(ctypes.addressof(hb), hv.nbytes, src.addr, dl.addr, n, num_buf, bits_mode,
                                           bytes_mode, chunk, threshold, device, out.ctypes.data, cap, ctypes.byref(out_len))
        self._check(rc)
        return memoryview(out)[:out_len.value]

    def decompress(self, body, num_buf, bits_mode, bytes_mode, chunk, orig_size, device=0, delta=None):
        """body: bytes-like after the header.  Returns orig_size bytes as a writable memoryview (numpy-backed).
        delta: bytes-like of orig_size bytes XORed into the output on the device."""
        bv = memoryview(body).cast("B")
        out = self.host_buffer(orig_size)
        src = _as_c_buffer(bv)
        dl = _as_c_buffer(memoryview(delta).cast("B")) if delta is not None else None
        if dl is not None and memoryview(delta).nbytes != orig_size:
            raise ValueError("delta buffer and original size differ")
        if dl is None:
            rc = self._L.zn_decompress(src.addr, bv.nbytes, num_buf, bits_mode, bytes_mode, chunk, orig_size, device, out.ctypes.data)
        else:
            rc = self._L.zn_decompress_delta(src.addr, bv.nbytes, dl.addr, num_buf, bits_mode, bytes_mode, chunk,
                                             orig_size, device, out.ctypes.data)
        self._check(rc)
        return memoryview(out)[:orig_size]

    def compress_multi(self, header, data, num_buf, bits_mode, bytes_mode, chunk, threshold, devices):
        """compress() with the chunks spread over several GPUs of the node (zn_compress_multi): device i codes the
        contiguous chunk range [i K / G, (i + 1) K / G) on its own host thread and stream; same frame bytes."""
        hv = memoryview(header).cast("B")
        dv = memoryview(data).cast("B")
        n = dv.nbytes
        cap = self._L.zn_compress_bound(n, num_buf, chunk, hv.nbytes)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        out_len = ctypes.c_size_t(0)
        hb = (ctypes.c_char * max(hv.nbytes, 1)).from_buffer_copy(hv.tobytes() or b"\0")
        src = _as_c_buffer(dv)
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._L.zn_compress_multi(ctypes.addressof(hb), hv.nbytes, src.addr, n, num_buf, bits_mode, bytes_mode, chunk,
                                       threshold, devs, len(devices), out.ctypes.data, cap, ctypes.byref(out_len))
        self._check(rc)
        return memoryview(out)[:out_len.value]

    def decompress_multi(self, body, num_buf, bits_mode, bytes_mode, chunk, orig_size, devices):
        """decompress() with the chunk ranges decoded on several GPUs (zn_decompress_multi); same bytes."""
        bv = memoryview(body).cast("B")
        out = np.empty(max(orig_size, 1), dtype=np.uint8)
        src = _as_c_buffer(bv)
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._L.zn_decompress_multi(src.addr, bv.nbytes, num_buf, bits_mode, bytes_mode, chunk, orig_size, devs, len(devices),
                                         out.ctypes.data)
        self._check(rc)
        return memoryview(out)[:orig_size]

    def multi_range(self, n, chunk, ndev, i):
        """(byte offset, byte length) of the chunk range device i of ndev codes (zn_multi_range)."""
        off, ln = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self._check(self._L.zn_multi_range(n, chunk, ndev, i, ctypes.byref(off), ctypes.byref(ln)))
        return off.value, ln.value

    def decompress_multi_dev(self, body, num_buf, bits_mode, bytes_mode, chunk, orig_size, devices, dst_ptrs):
        """zn_decompress_multi_dev: host body -> range i decoded into device memory dst_ptrs[i] of devices[i]
        (the decoded bytes never cross PCIe)."""
        bv = memoryview(body).cast("B")
        src = _as_c_buffer(bv)
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        ptrs = (ctypes.c_void_p * len(devices))(*[ctypes.c_void_p(p or None) for p in dst_ptrs])
        self._check(self._L.zn_decompress_multi_dev(src.addr, bv.nbytes, num_buf, bits_mode, bytes_mode, chunk, orig_size, devs,
                                                    len(devices), ptrs))

    def compress_multi_dev(self, header, src_ptrs, n, num_buf, bits_mode, bytes_mode, chunk, threshold, devices):
        """zn_compress_multi_dev: range i of the tensor lies at device pointer src_ptrs[i] on devices[i]; -> frame (host)."""
        hv = memoryview(header).cast("B")
        cap = self._L.zn_compress_bound(n, num_buf, chunk, hv.nbytes)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        out_len = ctypes.c_size_t(0)
        hb = (ctypes.c_char * max(hv.nbytes, 1)).from_buffer_copy(hv.tobytes() or b"\0")
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        ptrs = (ctypes.c_void_p * len(devices))(*[ctypes.c_void_p(p or None) for p in src_ptrs])
        self._check(self._L.zn_compress_multi_dev(ctypes.addressof(hb), hv.nbytes, ptrs, n, num_buf, bits_mode, bytes_mode, chunk,
                                                  threshold, devs, len(devices), out.ctypes.data, cap, ctypes.byref(out_len)))
        return memoryview(out)[:out_len.value]

    def decompress_range_dev(self, body, num_buf, bits_mode, bytes_mode, chunk, orig_size, chunk_lo, chunk_hi, device, dst_ptr):
        """zn_decompress_range_dev: chunks [chunk_lo, chunk_hi) of a host-resident body decoded into device memory at dst_ptr."""
        bv = memoryview(body).cast("B")
        src = _as_c_buffer(bv)
        self._check(self._L.zn_decompress_range_dev(src.addr, bv.nbytes, num_buf, bits_mode, bytes_mode, chunk, orig_size, chunk_lo,
                                                    chunk_hi, int(device), dst_ptr or None))

    def merge_range_bodies(self, parts, num_buf):
        """zn_merge_range_bodies: [(body bytes-like, num_chunks)] of consecutive chunk ranges -> one body (memoryview, numpy-backed)."""
        parts = [(memoryview(b).cast("B"), int(k)) for b, k in parts]
        bufs = [_as_c_buffer(b) for b, _ in parts]
        n = len(parts)
        ptrs = (ctypes.c_void_p * max(n, 1))(*[ctypes.c_void_p(c.addr) for c in bufs])
        lens = (ctypes.c_size_t * max(n, 1))(*[b.nbytes for b, _ in parts])
        ks = (ctypes.c_size_t * max(n, 1))(*[k for _, k in parts])
        cap = sum(b.nbytes for b, _ in parts)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        out_len = ctypes.c_size_t(0)
        self._check(self._L.zn_merge_range_bodies(ptrs, lens, ks, n, num_buf, out.ctypes.data, cap, ctypes.byref(out_len)))
        return memoryview(out)[:out_len.value]

    def set_host_slices(self, slices):
        """Tuning knob (zn_set_host_slices): slices of the pipelined host path; 0 = automatic, 1 = one shot."""
        self._check(self._L.zn_set_host_slices(int(slices)))

    def set_host_direct(self, mode):
        """zn_set_host_direct: 4 (default) = huge-page hint only, 7 = pinned, direct DMA for recycled buffers as well, 0 = neither."""
        self._check(self._L.zn_set_host_direct(int(mode)))

    def set_legacy_tree_descriptions(self, on):
        """zn_set_legacy_tree_descriptions: True = write tree descriptions the way the reference's PyPI wheels do (-1 markers)."""
        self._check(self._L.zn_set_legacy_tree_descriptions(1 if on else 0))

    def set_encode_onepass(self, mode):
        """Developer / test knob (zn_set_encode_onepass): 0 / False = the four-kernel encoder only, 1 = automatic (default: large bf16 calls), 2 / True =
        every call with full chunks through the one-pass encoder."""
        self._check(self._L.zn_set_encode_onepass(2 if mode is True else 0 if mode is False else int(mode)))

    def set_decode_wide(self, mode):
        """Tuning knob (zn_set_decode_wide): the small-input decoder — 0 never, 1 automatic (default), 2 / 3 every call without a delta base in its 16- / 8-wave form."""
        self._check(self._L.zn_set_decode_wide(int(mode)))

    def set_decode_group(self, chunks_per_workgroup):
        """Tuning knob (zn_set_decode_group): chunks per workgroup of the fused decoder, 1..4; 0 = automatic."""
        self._check(self._L.zn_set_decode_group(int(chunks_per_workgroup)))

    def decode_group_for(self, chunks):
        """zn_decode_group_for: chunks per workgroup the fused decoder gives a launch of `chunks` chunks on the current device."""
        return int(self._L.zn_decode_group_for(int(chunks)))

    # -- device pointers (ints), used by zipnn_amd.codec with torch tensors -----------------
    def compress_dev(self, src_ptr, n, num_buf, bits_mode, bytes_mode, chunk, threshold, body_ptr, body_cap, stream=0,
                     delta_ptr=None):
        out_len = ctypes.c_size_t(0)
        rc = self._L.zn_compress_delta_dev(src_ptr, delta_ptr, n, num_buf, bits_mode, bytes_mode, chunk, threshold, body_ptr,
                                           body_cap, ctypes.byref(out_len), stream)
        self._check(rc)
        return out_len.value

    def decompress_dev(self, body_ptr, body_len, num_buf, bits_mode, bytes_mode, chunk, orig_size, dst_ptr, stream=0,
                       check=True, delta_ptr=None):
        rc = self._L.zn_decompress_delta_dev(body_ptr, body_len, delta_ptr, num_buf, bits_mode, bytes_mode, chunk, orig_size,
                                             dst_ptr, stream, 1 if check else 0)
        self._check(rc)

    def compress_batch_dev(self, items, stream=0):
        """items: iterable of (src_ptr, n, num_buf, bits_mode, bytes_mode, chunk, threshold, body_ptr, body_cap[, delta_ptr])
        -> list of body lengths."""
        items = list(items)
        arr = (ZnCBatchItem * max(len(items), 1))()
        for i, it in enumerate(items):
            sp, n, nb, bi, by, ch, th, bp, cap = it[:9]
            arr[i].d_delta = it[9] if len(it) > 9 else None
            arr[i].d_src = sp; arr[i].n = n; arr[i].num_buf = nb; arr[i].bits_mode = bi; arr[i].bytes_mode = by
            arr[i].chunk = ch; arr[i].threshold = th; arr[i].d_body = bp; arr[i].body_cap = cap
        self._check(self._L.zn_compress_batch_dev(arr, len(items), stream))
        return [int(arr[i].body_len) for i in range(len(items))]

    def decompress_batch_dev(self, items, stream=0, check=True):
        """items: iterable of (body_ptr, body_len, num_buf, bits_mode, bytes_mode, chunk, orig_size, dst_ptr[, delta_ptr])."""
        items = list(items)
        arr = (ZnBatchItem * max(len(items), 1))()
        for i, it in enumerate(items):
            bp, bl, nb, bi, by, ch, n, dp = it[:8]
            arr[i].d_delta = it[8] if len(it) > 8 else None
            arr[i].d_body = bp; arr[i].body_len = bl; arr[i].d_dst = dp; arr[i].orig_size = n
            arr[i].num_buf = nb; arr[i].bits_mode = bi; arr[i].bytes_mode = by; arr[i].chunk = ch
        self._check(self._L.zn_decompress_batch_dev(arr, len(items), stream, 1 if check else 0))

    def decompress_batch_dev_packed(self, packed, count, stream=0, check=True):
        """zn_decompress_batch_dev over items already laid out as `struct zn_batch_item` (64 bytes each:
        struct.pack(ZN_BATCH_ITEM_FMT, d_body, body_len, d_dst, orig_size, num_buf, bits_mode, bytes_mode, chunk, d_delta or 0)) —
        for loaders that hand over hundreds of tensors per call."""
        buf = (ctypes.c_char * len(packed)).from_buffer_copy(packed)
        self._check(self._L.zn_decompress_batch_dev(ctypes.cast(buf, ctypes.POINTER(ZnBatchItem)), count, stream, 1 if check else 0))

    def decode_status(self, stream=0):
        """zn_decode_status: wait for `stream`, raise what the last check=False decode call on this device would have raised."""
        self._check(self._L.zn_decode_status(ctypes.c_void_p(stream or None)))

    def last_fused_chunks(self):
        """Chunks of the last decompress_dev call that took the fused single-pass kernel."""
        n = self._L.zn_last_fused_chunks()
        if n < 0:
            self._check(int(n))
        return int(n)

    def last_tail_planes(self):
        """Huffman planes of partial last chunks that the parallel tail kernel decoded in the last decompress call."""
        n = self._L.zn_last_tail_planes()
        if n < 0:
            self._check(int(n))
        return int(n)

    # -- pageable host buffer <-> device pointer, through the library's pinned multi-threaded pipe --------------
    def copy_to_device(self, dst_ptr, buf):
        cb = _as_c_buffer(memoryview(buf).cast("B"))
        self._check(self._L.zn_copy_to_device(dst_ptr, cb.addr, memoryview(buf).nbytes))

    def copy_to_host(self, buf, src_ptr, n=None):
        mv = memoryview(buf).cast("B")
        cb = _as_c_buffer(mv)
        self._check(self._L.zn_copy_to_host(cb.addr, src_ptr, mv.nbytes if n is None else n))

    def release_workspace(self):
        self._check(self._L.zn_release_workspace())


class _ArenaArray(np.ndarray):
    """A numpy view of a block of the library's pinned arena (ZnLib.host_buffer): a subclass only so that it can carry a finalizer."""


class _CBuf:
    """Address of a (possibly read-only) Python buffer, zero-copy, kept alive for a call."""

    def __init__(self, mv):
        self.arr = np.frombuffer(mv, dtype=np.uint8) if mv.nbytes else None
        self.addr = self.arr.ctypes.data if mv.nbytes else None


def _as_c_buffer(mv):
    return _CBuf(mv)


def _addr_of_bytearray(ba):
    return ctypes.addressof((ctypes.c_char * len(ba)).from_buffer(ba))


_LIB = None


def lib():
    """The process-wide library handle; raises ImportError when the extension is not built."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, LIB_NAME)
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). zipnn_amd has no CPU fallback.")
        _LIB = ZnLib(path)
    return _LIB
