"""zipnn_amd — MI355X (gfx950) implementation of ZipNN's compress/decompress hot path.

Same public names as the reference package (`from zipnn import ZipNN, zipnn_hf,
zipnn_safetensors`, reference zipnn/__init__.py:1); `zipnn_hf` is outside this
repository's scope and raises NotImplementedError.
"""
from .zipnn import ZipNN, SafeOpen, zipnn_safetensors, zipnn_hf, decompress_safetensors_tensor  # noqa: F401

__all__ = ["ZipNN", "SafeOpen", "zipnn_safetensors", "zipnn_hf", "decompress_safetensors_tensor"]
