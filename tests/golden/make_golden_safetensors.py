"""Generate tests/golden/gpt2_small_ref.znn.safetensors with the REFERENCE's own producer script.

Runs only in the build container (needs /root/reference and oracle/_ref/zipnn_core.so = the reference csrc/ compiled
from where it lies + libzstd 1.4.8 huff0, oracle/Makefile): a GPT-2-shaped checkpoint — the real tensor names and
layout of `transformers.GPT2LMHeadModel`, scaled down (2 layers, width 96, vocabulary 640, 48 positions: ~0.8 MB
instead of 498 MB) and random-initialised the way GPT-2 is (N(0, 0.02) weights, zero biases, unit LayerNorm gains;
there is no network for the real checkpoint) — is written with `save_file`, then compressed by the reference's
`scripts/zipnn_compress_safetensors.py:compress_safetensors_file` (zipnn/zipnn.py + zipnn_core).  BASELINE.json
configs[3] / SURVEY.md §8d-4: the consumer under test is this repository's `zipnn_safetensors()` plugin and
`safetensors_io.load_file`; the producer is the reference.

    python tests/golden/make_golden_safetensors.py      # rewrites the fixture and its .json (sha256 of every tensor)
"""
import hashlib
import importlib.util
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))   # zipnn_core (reference C ext)
sys.dont_write_bytecode = True                               # (nothing is written under /root/reference, not even __pycache__)
sys.path.insert(1, "/root/reference")                        # zipnn (reference Python)

LAYERS, WIDTH, VOCAB, NPOS = 2, 96, 640, 48


def gpt2_state(seed=20):
    g = torch.Generator().manual_seed(seed)
    n = lambda *s: torch.randn(*s, generator=g) * 0.02       # noqa: E731
    sd = {"transformer.wte.weight": n(VOCAB, WIDTH), "transformer.wpe.weight": n(NPOS, WIDTH)}
    for i in range(LAYERS):
        p = f"transformer.h.{i}."
        sd.update({
            p + "ln_1.weight": torch.ones(WIDTH), p + "ln_1.bias": torch.zeros(WIDTH),
            p + "attn.c_attn.weight": n(WIDTH, 3 * WIDTH), p + "attn.c_attn.bias": torch.zeros(3 * WIDTH),
            p + "attn.c_proj.weight": n(WIDTH, WIDTH), p + "attn.c_proj.bias": torch.zeros(WIDTH),
            p + "ln_2.weight": torch.ones(WIDTH), p + "ln_2.bias": torch.zeros(WIDTH),
            p + "mlp.c_fc.weight": n(WIDTH, 4 * WIDTH), p + "mlp.c_fc.bias": torch.zeros(4 * WIDTH),
            p + "mlp.c_proj.weight": n(4 * WIDTH, WIDTH), p + "mlp.c_proj.bias": torch.zeros(WIDTH),
        })
    sd.update({"transformer.ln_f.weight": torch.ones(WIDTH), "transformer.ln_f.bias": torch.zeros(WIDTH)})
    # an integer buffer (left alone by the producer), a bf16 and an fp16 tensor so that every decode geometry is in the file
    sd["transformer.h.0.attn.bias_mask"] = torch.tril(torch.ones(NPOS, NPOS)).to(torch.int64)
    sd["extra.bf16"] = n(300, 257).to(torch.bfloat16)
    sd["extra.fp16"] = n(129, 64).to(torch.float16)
    return sd


def main():
    from safetensors.torch import save_file
    spec = importlib.util.spec_from_file_location("ref_compress_safetensors", "/root/reference/scripts/zipnn_compress_safetensors.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sd = gpt2_state()
    src = os.path.join(HERE, "gpt2_small_ref.safetensors")
    save_file(sd, src, metadata={"format": "pt"})
    mod.compress_safetensors_file(src, delete=True, force=True)        # -> gpt2_small_ref.znn.safetensors
    out = src[:-len(".safetensors")] + ".znn.safetensors"
    info = {"producer": "/root/reference/scripts/zipnn_compress_safetensors.py over oracle/_ref (reference csrc + libzstd 1.4.8 huff0)",
            "file_sha256": hashlib.sha256(open(out, "rb").read()).hexdigest(),
            "tensors": {k: {"dtype": str(v.dtype), "shape": list(v.shape),
                            "sha256": hashlib.sha256(v.contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()} for k, v in sd.items()}}
    with open(out + ".json", "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    print(out, os.path.getsize(out), "bytes,", len(sd), "tensors")


if __name__ == "__main__":
    main()
