"""`import zipnn` → the MI355X library (zipnn_amd), for consumers that spell the reference's import name
(reference zipnn/__init__.py:1: `from zipnn import ZipNN, zipnn_hf, zipnn_safetensors` — vLLM, Hugging Face loaders,
the reference's own scripts).  Put this directory on the path INSTEAD of the reference package:

    PYTHONPATH=<repo>/compat:<repo> python -c "from zipnn import ZipNN, zipnn_safetensors"

or, inside a process that already imported zipnn_amd, call `zipnn_amd.install_as_zipnn()`.
Not at the repository root on purpose: there it would shadow the stock reference package that the parity tests and
the golden-vector generators import under the same name."""
from zipnn_amd import ZipNN, zipnn_hf, zipnn_safetensors  # noqa: F401
from zipnn_amd.zipnn import *  # noqa: F401,F403

__all__ = ["ZipNN", "zipnn_hf", "zipnn_safetensors"]
