import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) where no device is visible, e.g. the build container."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---------------------------------------------------------------------------------------
# SIMT-emulated build of the kernels (tests/simt): lets the CPU suite execute the product's
# kernel sources and host logic where no GPU exists.  Test infrastructure only — the
# product never loads it (zipnn_amd._capi.lib() only opens libzipnn_hip.so).
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="session")
def simt_lib():
    import glob
    import subprocess
    from zipnn_amd._capi import ZnLib
    here = os.path.join(ROOT, "tests", "simt")
    so = os.path.join(here, "libzipnn_simt.so")
    srcs = glob.glob(os.path.join(ROOT, "zipnn_amd", "csrc", "*")) + [os.path.join(here, "hip", "hip_runtime.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["sh", os.path.join(here, "build.sh")], check=True, capture_output=True)
    return ZnLib(so)


@pytest.fixture()
def use_simt(simt_lib, monkeypatch):
    """Route zipnn_amd through the emulated kernels for one test (CPU tensors as 'device' memory)."""
    import zipnn_amd._capi as capi
    monkeypatch.setattr(capi, "_LIB", simt_lib)
    return simt_lib


@pytest.fixture()
def decode_group():
    """Force the fused decoder's chunks-per-workgroup (zn_set_decode_group) for one test; back to automatic afterwards."""
    used = []

    def set_(lib, group):
        lib.set_decode_group(group)
        used.append(lib)
    yield set_
    for lib in used:
        lib.set_decode_group(0)
