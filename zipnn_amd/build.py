"""Builds zipnn_amd/libzipnn_hip.so from zipnn_amd/csrc/*.hip with hipcc for gfx950.

In-tree on purpose: the built library travels with the source snapshot to the GPU box and
is the file the driver sees loaded.  hipcc cross-compiles without a GPU.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzipnn_hip.so")
ARCH = "gfx950"


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = glob.glob(os.path.join(CSRC, "*")) + [os.path.join(os.path.dirname(HERE), "include", "zipnn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force=False, verbose=False):
    """Compile every kernel translation unit for gfx950 and link the C-ABI library."""
    if not force and not is_stale():
        return OUT
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-o", OUT] + sources()
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libzipnn_hip.so")
    return OUT


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
