"""zipnn_core.py — the reference-side binding of INTEGRATION.md §1, as a file that is actually executed.

Put this directory on sys.path IN PLACE OF the reference's compiled `zipnn_core` extension and the stock
`/root/reference/zipnn` Python package runs on libzipnn_hip.so: the same two functions the extension exports
(csrc/zipnn_core_module.c:9-23), called at zipnn/zipnn.py:714-725 and :1143-1151.

The library is taken from $ZIPNN_HIP_LIB (tests point it at the SIMT-emulated build, tests/simt/libzipnn_simt.so, where
no GPU exists) and defaults to the in-tree zipnn_amd/libzipnn_hip.so.  TEST INFRASTRUCTURE of this repository only in the
sense that it lives under tests/: it is the file a maintainer of zipnn/zipnn would add.
"""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_L = ctypes.CDLL(os.environ.get("ZIPNN_HIP_LIB") or os.path.join(_HERE, "..", "..", "zipnn_amd", "libzipnn_hip.so"))
_sz, _vp, _i, _f = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_L.zn_compress_bound.restype = _sz
_L.zn_compress_bound.argtypes = [_sz, _i, _sz, _sz]
_L.zn_compress.restype = _i
_L.zn_compress.argtypes = [_vp, _sz, _vp, _sz, _i, _i, _i, _sz, _f, _i, _vp, _sz, ctypes.POINTER(_sz)]
_L.zn_decompress.restype = _i
_L.zn_decompress.argtypes = [_vp, _sz, _i, _i, _i, _sz, _sz, _i, _vp]
_L.zn_strerror.restype = ctypes.c_char_p
_DEVICE = int(os.environ.get("ZIPNN_HIP_DEVICE", "0"))


_L.zn_host_alloc.restype = _vp
_L.zn_host_alloc.argtypes = [_sz]
_L.zn_host_free.argtypes = [_vp]
_ARENA_MIN = 8 << 20


class _Pinned(np.ndarray):
    """numpy view of a block of the library's pinned arena; a subclass only so that it can carry a finalizer"""


def _result(n):
    """Result memory: from 8 MiB up a block of the library's pinned arena (zn_host_alloc: one DMA instead of a staged copy, recycled when the last view of it
    dies) — the reference's extension also returns memoryviews over memory it allocated itself (csrc/zipnn_core.c:596, 1126) —, a bytearray below that or when
    the driver has no pinned memory left.  -> (object to keep / slice, its address as a ctypes argument)"""
    n = max(int(n), 1)
    if n >= _ARENA_MIN and os.environ.get("ZIPNN_AMD_HOST_ARENA_MB") != "0":
        p = _L.zn_host_alloc(n)
        if p:
            arr = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p)).view(_Pinned)
            weakref.finalize(arr, _L.zn_host_free, ctypes.c_void_p(p))
            return arr, ctypes.c_void_p(p)
    out = bytearray(n)
    return out, (ctypes.c_char * n).from_buffer(out)


def _addr(buf):
    """zero-copy view + address of any bytes-like object"""
    a = np.frombuffer(buf, dtype=np.uint8)
    return a, (a.ctypes.data if a.size else None)


def zipnn_core(header, data, numBuf, bits_mode, bytes_mode, is_redata, origChunkSize, compThreshold, checkThAfterPercent, threads):
    """csrc/zipnn_core.c:401 ("y*y*iiiinfii"): header ‖ types ‖ cumSizes ‖ payload as a memoryview.
    is_redata, checkThAfterPercent and threads are accepted and ignored (dead in the reference core too)."""
    h, hp = _addr(header)
    d, dp = _addr(data)
    cap = _L.zn_compress_bound(d.size, numBuf, origChunkSize, h.size)
    out, outp = _result(cap)
    n = _sz(0)
    rc = _L.zn_compress(hp, h.size, dp, d.size, numBuf, bits_mode, bytes_mode, origChunkSize, compThreshold, _DEVICE,
                        outp, cap, ctypes.byref(n))
    if rc:
        raise RuntimeError("Thread processing failed: " + _L.zn_strerror(rc).decode() + f" [{_L._name}]")
    try:
        header[24:32] = bytes(memoryview(out)[24:32])       # the reference writes the total length into the caller's header (zipnn_core.c:121)
    except TypeError:
        pass                             # (an immutable header object: the frame carries the length anyway)
    return memoryview(out)[: n.value]


def combine_dtype(data, numBuf, bits_mode, bytes_mode, origChunkSize, origSize, threads):
    """csrc/zipnn_core.c:881 ("y*iiinni"): the body after the header -> origSize bytes."""
    d, dp = _addr(data)
    out, outp = _result(origSize)
    rc = _L.zn_decompress(dp, d.size, numBuf, bits_mode, bytes_mode, origChunkSize, origSize, _DEVICE, outp)
    if rc == -5:
        raise MemoryError("Compress Type is not correct in Decompression function")   # zipnn_core.c:993-996
    if rc:
        raise RuntimeError("Thread processing failed: " + _L.zn_strerror(rc).decode())
    return memoryview(out)[:origSize]
