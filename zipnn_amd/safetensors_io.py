"""Tensor-by-tensor safetensors compression (the file producer/consumer either side of the hot path;
SURVEY.md §8f-1).  Same file layout as the reference's scripts/zipnn_compress_safetensors.py:37-148 and
scripts/zipnn_decompress_safetensors.py:34-136: every floating-point tensor that shrinks is stored as a
1-D uint8 tensor holding one TORCH-format ZN frame, and listed (dtype, shape) in the file metadata under
`znn_compressed_vectors`; everything else is stored untouched.  Suffix: `.znn.safetensors`.
"""
import os

import torch

from .zipnn import (COMPRESSED_DTYPE, COMPRESSION_METHOD, METADATA_KEY, ZipNN, build_compressed_tensor_info,
                    get_compressed_tensors_metadata, set_compressed_tensors_metadata)

SUFFIX = ".znn.safetensors"


def compress_safetensors_file(filename, out_path=None, device="cpu", method=None, batched=None):
    """-> path of the compressed file.  `device` = where tensors are staged for compression
    ("cuda:N" compresses in HBM; the compressed frames come back to the host for writing).  Tensors staged in HBM
    are compressed by ONE batched call for the whole file (`batched=None`: automatic; True forces it)."""
    from safetensors import safe_open
    from safetensors.torch import save_file
    assert filename.endswith(".safetensors")
    out_path = out_path or filename[: -len(".safetensors")] + SUFFIX
    tensors, infos = {}, {}
    batch = []                                        # (name, tensor, header, planes, bits, bytes, chunk)
    with safe_open(filename, "pt", device) as f:
        for name in f.keys():
            t = f.get_tensor(name)
            if not torch.is_floating_point(t) or t.dtype == torch.float64 or t.numel() == 0:
                tensors[name] = t.cpu()
                continue
            znn = ZipNN(input_format="torch", bytearray_dtype=t.dtype, method=method or COMPRESSION_METHOD)
            if t.is_cuda if batched is None else batched:
                batch.append((name, t) + znn.torch_frame_plan(t) + (znn.compression_threshold,))
                continue
            frame = znn.compress(t)                       # our compress never modifies `t`
            if len(frame) >= t.element_size() * t.nelement():
                tensors[name] = t.cpu()
                continue
            tensors[name] = torch.frombuffer(bytearray(frame), dtype=COMPRESSED_DTYPE)
            infos[name] = build_compressed_tensor_info(t)
        metadata = dict(f.metadata() or {})
    if batch:
        # tensors staged in HBM: one batched compress for the whole file (zn_compress_batch_dev)
        from . import _capi, codec
        bodies = codec.compress_device_batch(_capi.lib(), [(codec.flat_bytes(t), P, bits, byts, chunk, th) for (_, t, _, P, bits, byts, chunk, th) in batch])
        for (name, t, hdr, *_), body in zip(batch, bodies):
            total = len(hdr) + body.numel()
            if total >= t.element_size() * t.nelement():
                tensors[name] = t.cpu()
                continue
            frame = codec.new_bytearray(total)
            frame[:len(hdr)] = hdr
            frame[24:32] = total.to_bytes(8, "little")     # what the core writes at zipnn_core.c:121
            codec.to_host(_capi.lib(), body, memoryview(frame)[len(hdr):])
            tensors[name] = torch.frombuffer(frame, dtype=COMPRESSED_DTYPE)
            infos[name] = build_compressed_tensor_info(t)
    if not metadata:
        metadata = {"format": "pt"}                       # the reference silently drops the list when a file has no metadata
    set_compressed_tensors_metadata(infos, metadata)
    save_file(tensors, out_path, metadata)
    return out_path


def decompress_safetensors_file(filename, out_path=None, device="cpu"):
    """Inverse of compress_safetensors_file -> path of the plain .safetensors file."""
    from safetensors import safe_open
    from safetensors.torch import save_file
    assert filename.endswith(SUFFIX)
    out_path = out_path or filename[: -len(SUFFIX)] + ".safetensors"
    tensors = {}
    with safe_open(filename, "pt", "cpu") as f:
        metadata = dict(f.metadata() or {})
        infos = get_compressed_tensors_metadata(metadata)
        for name in f.keys():
            t = f.get_tensor(name)
            if name in infos:
                znn = ZipNN(input_format="torch", bytearray_dtype=COMPRESSED_DTYPE, method=COMPRESSION_METHOD)
                t = znn.decompress(t, decompress_cpu_gpu=device)
            tensors[name] = t.cpu() if isinstance(t, torch.Tensor) else t
    metadata.pop(METADATA_KEY, None)
    save_file(tensors, out_path, metadata or None)
    return out_path


def load_file(filename, device="cuda:0"):
    """Load a (possibly ZipNN-compressed) safetensors file straight onto `device`: compressed tensors cross
    PCIe compressed and are decoded by ONE batched launch (zn_decompress_batch_dev), so a file of many small
    tensors decodes at the rate of one large tensor.  -> {name: tensor}.  The batched counterpart of looping
    SafeOpen.get_tensor (reference zipnn.py:1592-1626, scripts/zipnn_decompress_safetensors.py:75-120)."""
    from safetensors import safe_open
    from . import _capi, codec
    dev = torch.device(device)
    out, items, meta = {}, [], []
    with safe_open(filename, "pt", "cpu") as f:
        infos = get_compressed_tensors_metadata(dict(f.metadata() or {}))
        for name in f.keys():
            t = f.get_tensor(name)
            if name not in infos:
                out[name] = t.to(dev, non_blocking=True)
                continue
            znn = ZipNN(input_format="torch", bytearray_dtype=COMPRESSED_DTYPE, method=COMPRESSION_METHOD)
            fp = znn.frame_params(t)
            host_body = t.reshape(-1).view(torch.uint8)[fp["body_off"]:]
            # (large bodies through the library's pinned multi-threaded transfer; small ones are not worth its threads)
            body = codec.to_device(_capi.lib(), host_body.numpy(), dev) if host_body.numel() >= (2 << 20) and dev.type == "cuda" \
                else host_body.to(dev, non_blocking=True)
            items.append((body, fp["num_buf"], fp["bits_mode"], fp["bytes_mode"], fp["chunk"], fp["orig_size"]))
            meta.append((name, fp["torch_dtype"], fp["shape"]))
    flats = codec.decompress_device_batch(_capi.lib(), items)
    for (name, dtype, shape), flat in zip(meta, flats):
        out[name] = flat.view(dtype).reshape(shape) if flat.numel() else torch.empty(shape, dtype=dtype, device=dev)
    return out
