"""CPU test (-m "not gpu"; skipped where /root/reference does not exist, i.e. on the GPU box): INTEGRATION.md §1 EXECUTED.

The stock reference Python package (`/root/reference/zipnn`, unmodified, read where it lies) is run twice over the
cases of its own tests/simple_stress_tests.py — once on its own compiled extension (oracle/_ref/zipnn_core.so = the
reference csrc/ + libzstd 1.4.8 huff0) and once with tests/ref_binding/zipnn_core.py on sys.path instead, i.e. on this
repository's library through the C ABI (the SIMT-emulated build of the kernels, since there is no GPU here).  Every
frame must be byte-identical between the two and decode back to its input: the drop-in claim, checked end to end."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
REF_CORE = os.path.join(ROOT, "oracle", "_ref", "zipnn_core.so")
RUNNER = os.path.join(ROOT, "tests", "run_reference_cases.py")      # (NOT inside ref_binding/: the script directory is sys.path[0] and would shadow the module under test)


def _run(first_on_path, env_extra):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([first_on_path, REF])
    env["PYTHONDONTWRITEBYTECODE"] = "1"          # /root/reference is read-only to this repository: no __pycache__ next to its modules
    env.update(env_extra)
    r = subprocess.run([sys.executable, RUNNER], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "zipnn")), reason="/root/reference is not on this machine")
def test_stock_reference_package_runs_on_this_library_with_identical_frames(simt_lib):
    if not os.path.exists(REF_CORE):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
    if not os.path.exists(REF_CORE):
        pytest.skip("oracle/_ref could not be built here")
    ours = _run(os.path.join(ROOT, "tests", "ref_binding"), {"ZIPNN_HIP_LIB": simt_lib.path})
    theirs = _run(os.path.join(ROOT, "oracle", "_ref"), {})
    assert ours.pop("zipnn_core").endswith(os.path.join("tests", "ref_binding", "zipnn_core.py"))      # the binding really was the module in use
    assert theirs.pop("zipnn_core").endswith("zipnn_core.so")
    assert set(ours) == set(theirs) and len(ours) >= 15
    for name in sorted(ours):
        assert ours[name]["roundtrip"] and theirs[name]["roundtrip"], name
        assert ours[name] == theirs[name], name            # same frame bytes (sha256 + length) from both cores


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "tests", "simple_stress_tests.py")), reason="/root/reference is not on this machine")
def test_the_reference_s_own_stress_test_passes_on_this_library(simt_lib, tmp_path):
    """`/root/reference/tests/simple_stress_tests.py`, unmodified and at its own sizes (bf16 tensors and random bytes around the
    256 KiB / 1 MiB boundaries, streaming with five chunk sizes, delta, float32, and the safetensors scripts for fp16 / bf16 /
    fp8), run by pytest in a subprocess with tests/ref_binding/ ahead of the reference on sys.path: the stock Python package, its
    own assertions, this library underneath (emulated kernels here).  Runs in a scratch directory — the test writes its
    temporary safetensors files into the cwd — and writes nothing under /root/reference (no bytecode, no pytest cache)."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "ref_binding"), REF])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    env["ZIPNN_HIP_LIB"] = simt_lib.path
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REF, "tests", "simple_stress_tests.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "--rootdir", str(tmp_path)], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
    assert not os.path.isdir(os.path.join(REF, "zipnn", "__pycache__"))
