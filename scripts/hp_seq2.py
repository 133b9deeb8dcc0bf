#!/usr/bin/env python
"""Developer probe: host-buffer decompress times call by call under different allocation patterns of the result buffer."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import _capi
lib = _capi.lib(); L = lib._L
n = 1 << 30
x = (torch.randn(n // 2, device="cuda") * 0.02).to(torch.bfloat16).cpu().view(torch.uint8).numpy()
hdr = np.zeros(32, dtype=np.uint8); cap = L.zn_compress_bound(n, 2, 262144, 32); sz = ctypes.c_size_t(0)
frame = np.empty(cap, dtype=np.uint8)
lib.set_host_slices(int(os.environ.get("SLICES", "1")))
L.zn_compress(hdr.ctypes.data, 32, x.ctypes.data, n, 2, 1, 10, 262144, ctypes.c_float(0.95), 0, frame.ctypes.data, cap, ctypes.byref(sz))
def dec(o):
    t0 = time.perf_counter(); rc = L.zn_decompress(frame.ctypes.data + 32, sz.value - 32, 2, 1, 10, 262144, n, 0, o.ctypes.data); assert rc == 0; return (time.perf_counter() - t0) * 1e3
warm = np.empty(n, dtype=np.uint8); dec(warm); dec(warm)
def fresh(): return np.empty(n, dtype=np.uint8)
out = []
keep = []
for _ in range(5): o = fresh(); out.append(f"{dec(o):.1f}"); keep.append(o)
print("A fresh, kept           :", " ".join(out), flush=True); out = []
t0 = time.perf_counter(); del keep, o; print(f"   (free of 5: {(time.perf_counter() - t0) * 1e3:.0f} ms)")
for _ in range(5): o = fresh(); out.append(f"{dec(o):.1f}"); del o
print("B fresh, freed          :", " ".join(out), flush=True); out = []
for _ in range(4): out.append(f"w{dec(warm):.1f}"); o = fresh(); out.append(f"f{dec(o):.1f}"); del o
print("C warm, fresh, freed    :", " ".join(out), flush=True); out = []
for _ in range(4): o = fresh(); out.append(f"{dec(o):.1f}"); del o; time.sleep(0.05)
print("D fresh, freed, sleep 50:", " ".join(out), flush=True); out = []
for _ in range(4): out.append(f"w{dec(warm):.1f}"); time.sleep(0.05); o = fresh(); out.append(f"f{dec(o):.1f}"); del o
print("E warm, sleep, fresh, freed:", " ".join(out), flush=True); out = []
for _ in range(4): out.append(f"w{dec(warm):.1f}"); o = fresh(); o[::4096] = 1; out.append(f"f{dec(o):.1f}"); del o
print("F warm, fresh PRE-TOUCHED by the caller, freed:", " ".join(out), flush=True); out = []
for _ in range(6): out.append(f"{dec(warm):.1f}")
print("G warm                  :", " ".join(out), flush=True)
