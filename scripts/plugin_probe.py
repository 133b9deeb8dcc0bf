#!/usr/bin/env python
"""Developer tool (GPU): where load_file's decode_s goes for the GPT-2-shaped checkpoint of the bench (148 fp32 tensors)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as B
from safetensors.torch import save_file
from zipnn_amd import _capi, safetensors_io
dev = torch.device("cuda", 0); lib = _capi.lib()
tmp = tempfile.mkdtemp(prefix="zn_probe_")
sd = B.gpt2_state(dev)
src = os.path.join(tmp, "gpt2.safetensors"); save_file({k: v.cpu() for k, v in sd.items()}, src, {"format": "pt"})
znn = safetensors_io.compress_safetensors_file(src, device=str(dev))
for rep in range(5):
    tm = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = safetensors_io.load_file(znn, device=str(dev), timings=tm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print({k: round(v * 1e3, 3) for k, v in tm.items() if k.endswith("_s")}, "total_ms", round(dt * 1e3, 3), lib.last_kernels(), flush=True)
    del out

# ---- the launch part, step by step (the same calls decode_file_on_device makes) ----
import struct, mmap
from zipnn_amd import codec
from zipnn_amd.zipnn import fast_frame_params
from zipnn_amd.safetensors_io import _read_layout, get_compressed_tensors_metadata
metadata, layout, data_start = _read_layout(znn)
infos = get_compressed_tensors_metadata(dict(metadata))
f = open(znn, "rb"); mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ); view = memoryview(mm)
plan, total = [], 0
for name, (dt, shape, lo, hi) in layout.items():
    if name in infos:
        fp = fast_frame_params(view[data_start + lo: data_start + min(hi, lo + 32 + 1 + 9 * 255)])
        plan.append((name, lo + fp[0], hi, fp, total)); total += (fp[5] + 255) & ~255
blob = codec.to_device(lib, view[data_start:], dev); torch.cuda.synchronize()
for rep in range(4):
    t = [time.perf_counter()]
    arena = torch.empty(max(total, 16), dtype=torch.uint8, device=dev); t.append(time.perf_counter())
    base_in, base_out = blob.data_ptr(), arena.data_ptr()
    pack = struct.Struct(_capi.ZN_BATCH_ITEM_FMT).pack
    packed = b"".join([pack(base_in + b0, hi - b0, (base_out + off) if fp[5] else 0, fp[5], fp[1], fp[2], fp[3], fp[4], 0) for i, (_, b0, hi, fp, off) in enumerate(plan)]); t.append(time.perf_counter())
    stream = codec._stream_handle(blob); t.append(time.perf_counter())
    lib.decompress_batch_dev_packed(packed, len(plan), stream, check=False); t.append(time.perf_counter())
    lib.decode_status(stream); t.append(time.perf_counter())
    print("arena %.3f  pack %.3f  stream %.3f  library call %.3f  wait %.3f ms" % tuple((t[i + 1] - t[i]) * 1e3 for i in range(5)), flush=True)
    del arena

# ---- what closing the mapping costs after the multi-threaded upload has read it ----
for rep in range(3):
    f2 = open(znn, "rb"); mm2 = mmap.mmap(f2.fileno(), 0, access=mmap.ACCESS_READ); v2 = memoryview(mm2)
    t0 = time.perf_counter(); b2 = codec.to_device(lib, v2[data_start:], dev); torch.cuda.synchronize(); t1 = time.perf_counter()
    v2.release(); t2 = time.perf_counter(); mm2.close(); t3 = time.perf_counter(); f2.close()
    with torch.cuda.device(dev):
        t4 = time.perf_counter()
    print("upload %.3f  release %.3f  munmap %.3f  device ctx %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
    del b2
