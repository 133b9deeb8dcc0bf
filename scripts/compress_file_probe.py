#!/usr/bin/env python
"""Developer tool (GPU): where compress_safetensors_file's time goes for the GPT-2-shaped checkpoint of the bench."""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bench as B
from safetensors.torch import save_file
from zipnn_amd import _capi, safetensors_io
dev = torch.device("cuda", 0); lib = _capi.lib()
tmp = tempfile.mkdtemp(prefix="zn_probe_")
sd = B.gpt2_state(dev)
src = os.path.join(tmp, "gpt2.safetensors"); save_file({k: v.cpu() for k, v in sd.items()}, src, {"format": "pt"})
safetensors_io.compress_safetensors_file(src, device=str(dev)); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); safetensors_io.compress_safetensors_file(src, device=str(dev)); torch.cuda.synchronize(); print("total %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); safetensors_io.compress_safetensors_file(src, device=str(dev)); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
import hashlib, shutil
a = safetensors_io.compress_safetensors_file(src, device=str(dev)); ha = hashlib.sha256(open(a, "rb").read()).hexdigest()
b = safetensors_io.compress_safetensors_file(src, out_path=os.path.join(tmp, "per_tensor.znn.safetensors"), device=str(dev), batched=False); hb = hashlib.sha256(open(b, "rb").read()).hexdigest()
c = safetensors_io.compress_safetensors_file(src, out_path=os.path.join(tmp, "cpu_staged.znn.safetensors"), device="cpu"); hc = hashlib.sha256(open(c, "rb").read()).hexdigest()
from safetensors import safe_open
fa, fc = safe_open(a, "pt", "cpu"), safe_open(c, "pt", "cpu")
same_c = fa.metadata() == fc.metadata() and list(fa.keys()) == list(fc.keys()) and all(torch.equal(fa.get_tensor(k), fc.get_tensor(k)) for k in fa.keys())
print("same file as the per-tensor path (byte for byte; the metadata map is written in hash order, so this may differ from run to run):", ha == hb, " same tensors and metadata as the host-staged path (its container orders the header differently):", same_c)
out = safetensors_io.load_file(a, device=str(dev)); print("loads back exact:", all(torch.equal(out[k], sd[k]) for k in sd))
shutil.rmtree(tmp, ignore_errors=True)
