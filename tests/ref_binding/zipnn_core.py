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
    out = bytearray(max(cap, 1))
    n = _sz(0)
    rc = _L.zn_compress(hp, h.size, dp, d.size, numBuf, bits_mode, bytes_mode, origChunkSize, compThreshold, _DEVICE,
                        (ctypes.c_char * len(out)).from_buffer(out), cap, ctypes.byref(n))
    if rc:
        raise RuntimeError("Thread processing failed: " + _L.zn_strerror(rc).decode() + f" [{_L._name}]")
    try:
        header[24:32] = out[24:32]       # the reference writes the total length into the caller's header (zipnn_core.c:121)
    except TypeError:
        pass                             # (an immutable header object: the frame carries the length anyway)
    return memoryview(out)[: n.value]


def combine_dtype(data, numBuf, bits_mode, bytes_mode, origChunkSize, origSize, threads):
    """csrc/zipnn_core.c:881 ("y*iiinni"): the body after the header -> origSize bytes."""
    d, dp = _addr(data)
    out = bytearray(max(origSize, 1))
    rc = _L.zn_decompress(dp, d.size, numBuf, bits_mode, bytes_mode, origChunkSize, origSize, _DEVICE,
                          (ctypes.c_char * len(out)).from_buffer(out))
    if rc == -5:
        raise MemoryError("Compress Type is not correct in Decompression function")   # zipnn_core.c:993-996
    if rc:
        raise RuntimeError("Thread processing failed: " + _L.zn_strerror(rc).decode())
    return memoryview(out)[:origSize]
