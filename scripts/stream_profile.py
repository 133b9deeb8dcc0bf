#!/usr/bin/env python
"""Developer tool (GPU box): cProfile of the streaming `.znn` decompress (256 MiB in 1 MiB frames), second call."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zipnn_amd import ZipNN
x = (torch.randn(128 << 20, device="cuda") * 0.02).to(torch.bfloat16).cpu().view(torch.uint8).numpy()
raw = x.tobytes()
blob = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 20).compress(raw)
for rep in range(3):
    t0 = time.perf_counter()
    back = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 20).decompress(blob)
    print("decompress call", (time.perf_counter() - t0) * 1e3, "ms")
    del back
pr = cProfile.Profile(); pr.enable()
back = ZipNN(bytearray_dtype="bfloat16", is_streaming=True, streaming_chunk=1 << 20).decompress(blob)
pr.disable()
assert bytes(back) == raw
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
