"""ZN frame header, dtype codes and the shape ext-header (wire format: SURVEY.md Appendix A).

Semantics follow the reference's zipnn/util_header.py:5-45 (enums), zipnn/util_torch.py:89-159
(shape packing), :176-234 (dtype codes) and zipnn/zipnn.py:287-438 (the 32-byte header).
Only what the compress/decompress hot path needs is kept.
"""
import struct
from enum import Enum

import numpy as np
import torch

HEADER_LEN = 32
VERSION = (0, 5, 3)   # frames are interchangeable with the reference release this was built against


class _CaseInsensitive(Enum):
    @classmethod
    def _missing_(cls, value):
        if isinstance(value, str):
            return cls.__members__.get(value.upper())
        return None


class EnumMethod(_CaseInsensitive):
    AUTO = 0
    HUFFMAN = 1
    ZSTD = 2
    LZ4 = 3
    SNAPPY = 4


class EnumFormat(_CaseInsensitive):
    BYTE = 1
    TORCH = 2
    NUMPY = 3
    FILE = 4


class EnumLossy(_CaseInsensitive):
    NONE = 0
    INTEGER = 1
    UNSIGN = 2


# dtype code (header byte 15) -> what the codec needs to know about it.
#   planes      numBuf passed to the core          (reference zipnn/zipnn.py:786-815,1103-1141)
#   rotate      bit_reorder (sign-bit rotate)
#   byte_mode   byte_reorder
class DType:
    __slots__ = ("code", "name", "torch", "numpy", "planes", "rotate", "byte_mode")

    def __init__(self, code, name, torch_dtype, numpy_dtype, planes, rotate, byte_mode):
        self.code, self.name, self.torch, self.numpy = code, name, torch_dtype, numpy_dtype
        self.planes, self.rotate, self.byte_mode = planes, rotate, byte_mode


_DTYPES = [
    DType(1, "float32", torch.float32, np.float32, 4, 1, 220),
    DType(2, "float", torch.float32, np.float32, 4, 1, 220),     # alias code; never produced (torch.float is float32)
    DType(4, "float16", torch.float16, np.float16, 2, 0, 10),
    DType(5, "half", torch.float16, np.float16, 2, 0, 10),
    DType(6, "bfloat16", torch.bfloat16, None, 2, 1, 10),
    DType(29, "float8_e4m3fn", torch.float8_e4m3fn, None, 1, 1, 10),
    DType(30, "float8_e5m2", torch.float8_e5m2, None, 1, 1, 10),
]
_BY_CODE = {d.code: d for d in _DTYPES}
UINT32_CODE = 15
FLOAT64_NAMES = ("float64",)


def dtype_from_code(code):
    """Header byte 15 -> DType; raises like reference zipnn/zipnn.py:1119-1123,1136."""
    if code == UINT32_CODE:
        raise ValueError("Unsupported uinit32 in this version yet! please try version 0.1.1")
    d = _BY_CODE.get(code)
    if d is None:
        raise ValueError(f"Unsupported Dtype {code}")
    return d


def dtype_from_user(spec):
    """A torch dtype, numpy dtype or dtype string -> DType or None when it is not a float
    type this path codes (first match wins, like ZipNNDtypeEnum.from_dtype, util_torch.py:219-225)."""
    if isinstance(spec, str):
        spec = spec.lower()
        for d in _DTYPES:
            if d.name == spec:
                return d
        return None
    for d in _DTYPES:
        if spec is d.torch:
            return d
    try:
        nd = np.dtype(spec)
    except TypeError:
        return None
    for d in _DTYPES:
        if d.numpy is not None and nd == np.dtype(d.numpy):
            return d
    return None


def pack_shape(shape):
    """ndim:u8, then per dim a width byte (1/2/4/8) and the dim in that many LE bytes."""
    out = bytearray([len(shape)])
    for dim in shape:
        dim = int(dim)
        for width, fmt in ((1, "<B"), (2, "<H"), (4, "<I"), (8, "<Q")):
            if dim < (1 << (8 * width)):
                out.append(width)
                out += struct.pack(fmt, dim)
                break
    return bytes(out)


def unpack_shape(buf):
    """-> (shape tuple, bytes consumed)."""
    mv = memoryview(buf)
    ndim, pos, dims = mv[0], 1, []
    fmts = {1: "<B", 2: "<H", 4: "<I", 8: "<Q"}
    for _ in range(ndim):
        width = mv[pos]
        pos += 1
        dims.append(struct.unpack_from(fmts.get(width, "<Q"), mv, pos)[0])
        pos += width if width in fmts else 8
    return tuple(dims), pos


def is_pow2(x):
    return x > 0 and (x & (x - 1)) == 0
