#!/usr/bin/env python
"""Register / spill / LDS figures of the hot kernels, read from the BUILT code objects (zipnn_amd/build/libzipnn_hip/*.hip.o: the
gfx950 code object inside each unit's .hip_fatbin, its AMDGPU metadata notes) — the numbers DESIGN.md quotes, so that the document
cannot drift from the binary: tests/test_abi.py compares the table between the `codeobj` marks of DESIGN.md with
what this prints.
    python scripts/codeobj_stats.py            # the table
    python scripts/codeobj_stats.py --write    # … and rewrite the block in DESIGN.md
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJDIR = os.path.join(ROOT, "zipnn_amd", "build", "libzipnn_hip")
BEGIN, END = "<!-- codeobj:begin (scripts/codeobj_stats.py --write) -->", "<!-- codeobj:end -->"
# the kernels DESIGN.md makes statements about (substring of the mangled name -> what the table calls it)
HOT = [("zn_k_decode_fusedILi1ELb0ELb0E", "decode_fused<1> plain"), ("zn_k_decode_fusedILi2ELb0ELb0E", "decode_fused<2> plain (headline)"),
       ("zn_k_decode_fusedILi4ELb0ELb0E", "decode_fused<4> plain"),
       ("zn_k_decode_fusedILi1ELb0ELb1E", "decode_fused<1> rest"), ("zn_k_decode_fusedILi2ELb0ELb1E", "decode_fused<2> rest"), ("zn_k_decode_fusedILi4ELb0ELb1E", "decode_fused<4> rest"),
       ("zn_k_decode_fusedILi2ELb1ELb0E", "decode_fused<2> delta"),
       ("zn_k_decode_wideILi2ELi4E", "decode_wide<2> 16 waves"), ("zn_k_decode_wideILi2ELi2E", "decode_wide<2> 8 waves"),
       ("zn_k_encode_statsILi2ELb0E", "encode_stats<2>"), ("zn_k_encode_emitILi2ELb0E", "encode_emit<2>"), ("zn_k_encode_onepassILi2ELb0E", "encode_onepass<2>"),
       ("zn_k_encode_tables", "encode_tables")]


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "f.fat"), os.path.join(td, "f.co")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True)
        if r.returncode != 0:
            return {}                            # (a unit without device code: zn_api.hip)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                       check=True, capture_output=True)
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in txt.split("- .agpr_count")[1:]:
        f = {k: v for k, v in re.findall(r"\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size|sgpr_count):\s+(\S+)", blk)}
        if "name" in f:
            out[f["name"]] = f
    return out


def table():
    ks = {}
    for fn in sorted(os.listdir(OBJDIR)):
        if fn.endswith(".hip.o"):
            ks.update(kernels_of(os.path.join(OBJDIR, fn)))
    rows = ["| kernel instance | VGPRs | spilled VGPRs | spilled SGPRs | scratch B/lane | LDS bytes |", "|---|---|---|---|---|---|"]
    for sub, label in HOT:
        m = [v for k, v in ks.items() if sub in k]
        if not m:
            rows.append(f"| {label} | (not in this build) | | | | |")
            continue
        f = m[0]
        rows.append(f"| {label} | {f.get('vgpr_count')} | {f.get('vgpr_spill_count')} | {f.get('sgpr_spill_count')} | {f.get('private_segment_fixed_size')} | {f.get('group_segment_fixed_size')} |")
    return "\n".join(rows)


def design_block():
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    if BEGIN not in s or END not in s:
        return None
    return s.split(BEGIN, 1)[1].split(END, 1)[0].strip()


if __name__ == "__main__":
    t = table()
    print(t)
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        assert BEGIN in s and END in s, "DESIGN.md has no codeobj marks"
        s = s.split(BEGIN, 1)[0] + BEGIN + "\n" + t + "\n" + END + s.split(END, 1)[1]
        open(p, "w").write(s)
        print("DESIGN.md updated")
