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


def build_extension(force=False, verbose=False, defines=(), out=None):
    """Compile every kernel translation unit for gfx950 (one hipcc process per unit, side by side) and link the C-ABI library.
    defines / out: a variant build for A/B runs on the GPU box (scripts/): extra -D flags, another output name."""
    out = out or OUT
    if not force and out == OUT and not is_stale():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    hipcc = hipcc_path()
    objdir = os.path.join(HERE, "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + [f"-D{d}" for d in defines]

    headers = [f for f in glob.glob(os.path.join(CSRC, "*")) if not f.endswith(".hip")] + [os.path.join(os.path.dirname(HERE), "include", "zipnn_hip.h")]
    stamp = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    newest_header = max(os.path.getmtime(h) for h in headers)

    def one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        # (an object is kept when it is newer than its unit and than every header, and was built with the same flags)
        if same_flags and not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj, subprocess.CompletedProcess([], 0, "", "")
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        return obj, r
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(one, sources()))
    for _, r in res:
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("hipcc failed building " + os.path.basename(out))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + [o for o, _ in res]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking " + os.path.basename(out))
    return out


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose=True))
