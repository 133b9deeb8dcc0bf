"""zipnn_amd — MI355X (gfx950) implementation of ZipNN's compress/decompress hot path.

Same public names as the reference package (`from zipnn import ZipNN, zipnn_hf,
zipnn_safetensors`, reference zipnn/__init__.py:1); `zipnn_hf` is outside this
repository's scope and raises NotImplementedError.
"""
from .zipnn import ZipNN, SafeOpen, zipnn_safetensors, zipnn_hf, decompress_safetensors_tensor  # noqa: F401

__all__ = ["ZipNN", "SafeOpen", "zipnn_safetensors", "zipnn_hf", "decompress_safetensors_tensor"]


def install_as_zipnn():
    """Make `import zipnn` / `from zipnn import ZipNN, zipnn_safetensors` / `import zipnn.zipnn` resolve to this package inside the
    running process (a sys.modules hook; the path-based form is the `compat/` directory of the repository).  Refuses to replace a
    different `zipnn` that is already imported — the reference package and this one cannot both own the name in one process."""
    import sys
    from . import zipnn as _impl
    this = sys.modules[__name__]
    other = sys.modules.get("zipnn")
    if other is not None and other is not this and getattr(other, "ZipNN", None) is not ZipNN:
        raise RuntimeError("another `zipnn` package is already imported in this process")
    sys.modules["zipnn"] = this
    sys.modules["zipnn.zipnn"] = _impl
    return this


def set_legacy_tree_descriptions(on=True):
    """Make compress() write the tree descriptions of its Huffman planes the way the reference's PyPI wheels do (the huff0 of the
    bundled FiniteStateEntropy library: rare code-weight values as "-1" markers) instead of zstd >= 1.4.7's way (the default; what the
    reference writes when built against a current libzstd).  Every decoder reads both; only with this on are the frames byte-identical
    to a wheel's.  Process-wide (zn_set_legacy_tree_descriptions)."""
    from . import _capi
    _capi.lib().set_legacy_tree_descriptions(bool(on))
