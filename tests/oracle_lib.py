"""ctypes access to the CPU oracle (oracle/libzn_oracle.so) and, when present, to
oracle/_ref/zipnn_core.so (the reference csrc/ compiled from /root/reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (zipnn_amd/) never imports this module.
"""
import contextlib
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None
_REF = None


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, capture_output=True)


def lib():
    """The plain-C restatement."""
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libzn_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = ctypes.CDLL(path)
        sz, vp, u8p = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p
        L.zo_huf_compress.restype = sz
        L.zo_huf_compress.argtypes = [vp, sz, vp, sz]
        L.zo_huf_decompress.restype = sz
        L.zo_huf_decompress.argtypes = [vp, sz, vp, sz]
        L.zo_huf_is_error.restype = ctypes.c_uint
        L.zo_huf_is_error.argtypes = [sz]
        L.zo_optimal_table_log.restype = ctypes.c_uint
        L.zo_optimal_table_log.argtypes = [ctypes.c_uint, sz, ctypes.c_uint, ctypes.c_uint]
        L.zo_huf_build_ctable.restype = sz
        L.zo_huf_build_ctable.argtypes = [vp, ctypes.c_uint, ctypes.c_uint, vp, vp]
        L.zo_huf_write_ctable.restype = sz
        L.zo_huf_write_ctable.argtypes = [vp, sz, vp, ctypes.c_uint, ctypes.c_uint]
        L.zo_huf_read_stats.restype = sz
        L.zo_huf_read_stats.argtypes = [vp, vp, vp, vp, sz]
        L.zo_fse_normalize_count.restype = sz
        L.zo_fse_normalize_count.argtypes = [vp, ctypes.c_uint, vp, sz, ctypes.c_uint, ctypes.c_int]
        L.zo_compress_bound.restype = sz
        L.zo_compress_bound.argtypes = [sz, ctypes.c_int, sz, sz]
        L.zo_compress_frame.restype = ctypes.c_int
        L.zo_compress_frame.argtypes = [u8p, sz, u8p, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        sz, ctypes.c_float, ctypes.c_int, u8p, sz, vp]
        L.zo_decompress_body.restype = ctypes.c_int
        L.zo_decompress_body.argtypes = [u8p, sz, ctypes.c_int, ctypes.c_int, ctypes.c_int, sz, sz,
                                         ctypes.c_int, u8p]
        L.zo_set_weight_low_prob.restype = None
        L.zo_set_weight_low_prob.argtypes = [ctypes.c_int]
        L.zo_get_weight_low_prob.restype = ctypes.c_int
        L.zo_debug_m2_calls.restype = ctypes.c_ulong
        _LIB = L
    return _LIB


@contextlib.contextmanager
def legacy_weights():
    """Inside this block the oracle's ENCODER writes tree descriptions the way the legacy FiniteStateEntropy huff0
    does — the one the reference's PyPI wheels bundle (/root/reference/setup.py:23-28): weight counts that round below
    one FSE cell are written as -1 ("less than one"), where zstd >= 1.4.7 writes +1.  Both decode everywhere; the
    frames differ in the tree-description bytes only.  Used to feed the decoders what real wheels produce."""
    L = lib()
    old = L.zo_get_weight_low_prob()
    L.zo_set_weight_low_prob(-1)
    try:
        yield
    finally:
        L.zo_set_weight_low_prob(old)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def huf_compress(src, cap=None):
    """-> (ret, bytes). ret follows HUF_compress: 0, 1, size, or error (> 2**64-120)."""
    s = _u8(src)
    cap = cap if cap is not None else max(2 * s.size, 64)
    dst = np.zeros(cap, dtype=np.uint8)
    r = lib().zo_huf_compress(_ptr(dst), cap, _ptr(s), s.size)
    return r, (dst[:r].tobytes() if r <= cap else b"")


def huf_decompress(csrc, dst_size):
    s = _u8(csrc)
    dst = np.zeros(max(dst_size, 1), dtype=np.uint8)
    r = lib().zo_huf_decompress(_ptr(dst), dst_size, _ptr(s), s.size)
    return r, dst[:dst_size].tobytes()


def compress_frame(header, data, num_buf, bits_mode, bytes_mode, chunk, threshold=0.95, threads=1):
    """Full ZN frame (header ‖ types ‖ cumSizes ‖ payload) as bytes."""
    h, s = _u8(header), _u8(data)
    cap = lib().zo_compress_bound(s.size, num_buf, chunk, h.size)
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    out_len = ctypes.c_size_t(0)
    rc = lib().zo_compress_frame(_ptr(h), h.size, _ptr(s), s.size, num_buf, bits_mode, bytes_mode,
                                 chunk, threshold, threads, _ptr(dst), cap, ctypes.byref(out_len))
    if rc != 0:
        raise RuntimeError(f"oracle compress failed rc={rc}")
    return dst[: out_len.value].tobytes()


def decompress_body(body, num_buf, bits_mode, bytes_mode, chunk, orig_size, threads=1):
    b = _u8(body)
    dst = np.zeros(max(orig_size, 1), dtype=np.uint8)
    rc = lib().zo_decompress_body(_ptr(b), b.size, num_buf, bits_mode, bytes_mode, chunk, orig_size,
                                  threads, _ptr(dst))
    if rc != 0:
        raise RuntimeError(f"oracle decompress failed rc={rc}")
    return dst[:orig_size].tobytes()


# ---------------------------------------------------------------------------
# oracle/_ref : the reference's own C extension (zipnn_core), if it was built
# ---------------------------------------------------------------------------
def ref_core():
    """The reference CPython extension `zipnn_core` built into oracle/_ref, or None."""
    global _REF
    if _REF is None:
        path = os.path.join(ORACLE_DIR, "_ref", "zipnn_core.so")
        if not os.path.exists(path):
            _REF = False
        else:
            try:
                spec = importlib.util.spec_from_file_location("zipnn_core", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _REF = mod
            except Exception:  # e.g. libzstd.so.1 missing on this host
                _REF = False
    return _REF or None


def ref_compress_frame(header, data, num_buf, bits_mode, bytes_mode, chunk, threshold=0.95, threads=1):
    """zipnn_core.zipnn_core(...) of the reference; `data` is copied first because the
    reference rotates its input in place (csrc/data_manipulation_dtype16.c:68)."""
    core = ref_core()
    d = bytearray(bytes(data))
    return bytes(core.zipnn_core(bytearray(bytes(header)), d, num_buf, bits_mode, bytes_mode, 0,
                                 chunk, threshold, 10, threads))


def ref_decompress_body(body, num_buf, bits_mode, bytes_mode, chunk, orig_size, threads=1):
    core = ref_core()
    return bytes(core.combine_dtype(bytes(body), num_buf, bits_mode, bytes_mode, chunk, orig_size, threads))


def libzstd():
    """System libzstd.so.1 (zstd 1.4.8) exporting the huff0 stage functions, or None."""
    try:
        z = ctypes.CDLL("libzstd.so.1")
        z.ZSTD_versionNumber.restype = ctypes.c_uint
        if z.ZSTD_versionNumber() // 100 != 104:  # 1.4.x only: 1.5 changed tie-breaking
            return None
        z.HUF_compress.restype = ctypes.c_size_t
        z.HUF_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        z.HUF_decompress.restype = ctypes.c_size_t
        z.HUF_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        return z
    except OSError:
        return None


if __name__ == "__main__":
    build_oracle()
    print("oracle:", lib(), "ref:", ref_core(), "libzstd:", libzstd())
    sys.exit(0)
