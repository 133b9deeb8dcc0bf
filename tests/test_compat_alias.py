"""CPU tests: the reference's import name (`from zipnn import ZipNN, zipnn_hf, zipnn_safetensors`, reference zipnn/__init__.py:1)
resolves to this library — through the `compat/` directory on the path, or through zipnn_amd.install_as_zipnn()."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE_PATH = """
import zipnn, zipnn.zipnn, zipnn_amd
from zipnn import ZipNN, zipnn_hf, zipnn_safetensors
from zipnn.zipnn import SafeOpen
assert ZipNN is zipnn_amd.ZipNN and zipnn_safetensors is zipnn_amd.zipnn_safetensors and SafeOpen is zipnn_amd.SafeOpen
assert zipnn.__file__.replace('\\\\', '/').endswith('compat/zipnn/__init__.py')
z = ZipNN(input_format='torch', bytearray_dtype='bfloat16')          # the reference's constructor keywords
print('OK')
"""

CODE_HOOK = """
import zipnn_amd
zipnn_amd.install_as_zipnn()
from zipnn import ZipNN, zipnn_safetensors
import zipnn.zipnn as zz
assert ZipNN is zipnn_amd.ZipNN and zz.SafeOpen is zipnn_amd.SafeOpen
print('OK')
"""


def _run(code, path):
    env = dict(os.environ); env["PYTHONPATH"] = os.pathsep.join(path); env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd="/")
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_compat_directory_makes_import_zipnn_resolve_to_this_library():
    _run(CODE_PATH, [os.path.join(ROOT, "compat"), ROOT])


def test_sys_modules_hook():
    _run(CODE_HOOK, [ROOT])
