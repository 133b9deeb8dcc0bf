"""CPU tests (-m "not gpu"): the C-ABI library builds, loads and exports every symbol that
include/zipnn_hip.h declares; without a GPU the product fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    from zipnn_amd.build import build_extension
    build_extension()
    from zipnn_amd import _capi
    return _capi.lib()


def test_every_declared_symbol_is_exported(hip_lib):
    text = open(os.path.join(ROOT, "include", "zipnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(zn_[a-z_0-9]+)\s*\(", text)))
    assert len(names) >= 12
    raw = ctypes.CDLL(hip_lib.path)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/zipnn_hip.h but not exported"


def test_no_torch_types_in_abi():
    text = open(os.path.join(ROOT, "include", "zipnn_hip.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S).lower()
    assert 'extern "C"' in text


def test_size_helpers_and_errors(hip_lib):
    L = hip_lib._L
    assert L.zn_num_chunks(0, 262144) == 0 and L.zn_num_chunks(262145, 262144) == 2
    assert L.zn_compress_bound(1 << 20, 2, 1 << 18, 38) == 38 + 9 * 2 * 4 + (1 << 20)
    assert L.zn_strerror(0) == b"ok" and b"Compress Type" in L.zn_strerror(-5)


def test_product_fails_loudly_without_gpu(hip_lib):
    """No device → an exception, never a silent CPU result."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from zipnn_amd import ZipNN
    with pytest.raises(RuntimeError):
        ZipNN(bytearray_dtype="bfloat16").compress(bytes(4096))
    with pytest.raises(RuntimeError):
        hip_lib.decompress(bytes(100), 2, 1, 10, 262144, 10)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under zipnn_amd/ may reference it."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "zipnn_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                s = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_lib" not in s and "zn_oracle" not in s and "libzn_oracle" not in s, f


def test_cited_profiles_exist():
    """Every `profiles/<file>` a kernel source, the header, bench.py or a document cites is a file of this repository (measured evidence is quoted by name:
    a dangling name is a claim without its record)."""
    import glob
    import re
    cited = {}
    files = glob.glob(os.path.join(ROOT, "zipnn_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "zipnn_amd", "*.py")) + \
        [os.path.join(ROOT, f) for f in ("bench.py", "DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md"))]
    for f in files:
        for m in re.finditer(r"profiles/([A-Za-z0-9_.\-]+\.(?:txt|json|md))", open(f, encoding="utf-8", errors="replace").read()):
            cited.setdefault(m.group(1), f)
    missing = {name: src for name, src in cited.items() if not os.path.exists(os.path.join(ROOT, "profiles", name))}
    assert not missing, missing
    assert len(cited) > 20


def test_design_quotes_the_binary_s_register_and_spill_counts(hip_lib):
    """DESIGN.md's register / spill / LDS figures are a generated table (scripts/codeobj_stats.py --write), and this test reads the same
    fields (.vgpr_count, .vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size, .group_segment_fixed_size) out of the code objects the library was
    just linked from: a kernel change that moves them fails here until the document is regenerated."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no llvm-readelf on this machine")
    spec = importlib.util.spec_from_file_location("codeobj_stats", os.path.join(ROOT, "scripts", "codeobj_stats.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    if not os.path.isdir(m.OBJDIR) or not any(f.endswith(".hip.o") for f in os.listdir(m.OBJDIR)):
        pytest.skip("the library was not built on this machine (prebuilt .so only)")
    stated = m.design_block()
    assert stated is not None, "DESIGN.md lost its codeobj block"
    assert stated == m.table().strip(), "DESIGN.md's register/spill table differs from the built code objects: python scripts/codeobj_stats.py --write"
    assert "(not in this build)" not in stated


def test_bench_runs_its_multi_device_leg_in_a_child_process_with_a_time_limit(monkeypatch):
    """bench.py's in-process fan-out over several devices has never met more than one real GPU: it runs in a child process, last, with a time limit, so that an
    exception, a crash or a hang there cannot take the bench line down.  Here (no GPU): the child finds fewer than two devices and says `null`; a child that
    does not finish in time is killed and the leg reports it."""
    import sys
    import torch
    sys.path.insert(0, ROOT)
    import bench
    assert bench.multi_dev_isolated() is None                      # fewer than two devices: nothing is started
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert bench.multi_dev_isolated(timeout_s=600) is None         # the child (a real second process: it sees this machine's devices) prints null
    r = bench.multi_dev_isolated(timeout_s=0.05)
    assert r["devices"] == 2 and "timed out" in r["error"]
