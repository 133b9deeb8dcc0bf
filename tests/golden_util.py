"""Loader for tests/golden/golden_v1.npz (frames produced by the reference itself;
see tests/golden/make_golden.py).  Test infrastructure."""
import hashlib
import json
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")
_CACHE = None

# header byte 15 -> (num_buf, fp8?)  (reference zipnn/zipnn.py:1103-1141)
_NUM_BUF = {1: 4, 2: 4, 4: 2, 5: 2, 6: 2, 29: 1, 30: 1}


def load():
    global _CACHE
    if _CACHE is None:
        z = np.load(_PATH)
        meta = json.loads(bytes(z["meta.json"]).decode())
        _CACHE = [(m, bytes(z[m["name"] + ".frame"])) for m in meta]
    return _CACHE


def names():
    return [m["name"] for m, _ in load()]


def get(name):
    for m, f in load():
        if m["name"] == name:
            return m, f
    raise KeyError(name)


def split_frames(blob):
    """A streaming container is back-to-back frames, each self-sized by header[24:32]
    (reference zipnn/zipnn.py:977-992); a plain frame is the one-element case."""
    out, off = [], 0
    while off < len(blob):
        total = int.from_bytes(blob[off + 24:off + 32], "little")
        out.append(blob[off:off + total])
        off += total
    assert off == len(blob)
    return out


def parse_frame(frame):
    """-> dict(header, ext_len, num_buf, bits_mode, bytes_mode, chunk, orig_len, body)."""
    h = frame[:32]
    assert h[:2] == b"ZN"
    fmt, dtype = h[8], h[15]
    ext_len = 0
    if fmt in (2, 3):  # TORCH / NUMPY carry a packed shape (reference util_torch.py:89-159)
        nd = frame[32]
        ext_len, i = 1, 33
        for _ in range(nd):
            w = frame[i]
            ext_len += 1 + w
            i += 1 + w
    num_buf = _NUM_BUF[dtype]
    chunk = 1 << h[14]
    if num_buf == 1:
        chunk = min(chunk, 128 * 1024)  # fp8: reference zipnn/zipnn.py:721,1148
    return dict(header=frame[:32 + ext_len], ext_len=ext_len, num_buf=num_buf, bits_mode=h[6],
                bytes_mode=h[5], chunk=chunk, orig_len=int.from_bytes(h[16:24], "little"),
                body=frame[32 + ext_len:])


# ---- delta (XOR) frames written by the reference: tests/golden/make_golden_delta.py ----
_DELTA_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_delta_v1.npz")
_DELTA = None


def delta_load():
    """[(meta, frame bytes, base bytes)] — `base` is the delta_second_data the reference was given."""
    global _DELTA
    if _DELTA is None:
        z = np.load(_DELTA_PATH)
        meta = json.loads(bytes(z["meta.json"]).decode())
        _DELTA = [(m, bytes(z[m["name"] + ".frame"]), bytes(z[m["name"] + ".base"])) for m in meta]
    return _DELTA


def delta_names():
    return [m["name"] for m, _, _ in delta_load()]


def delta_get(name):
    for m, f, b in delta_load():
        if m["name"] == name:
            return m, f, b
    raise KeyError(name)


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()
