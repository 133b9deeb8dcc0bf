#!/usr/bin/env python
"""Developer tool: compile zn_decode_fused.hip to ISA and report, per kernel, the scratch (spill) instructions that sit
inside the hot tile loop of the compile-time-D instances (between the ZN_HOT_TILE_BEGIN / ZN_HOT_TILE_END marks), i.e. on
the path every tile takes.  A spill reload there waits on the same counter as the HBM loads in flight (s_waitcnt vmcnt)
and costs thousands of cycles per tile; spills in the rare paths (looping form, fix-ups, tails) do not matter.
    python scripts/hot_spills.py [-Dflag ...]
The marks only bracket the loop body textually; blocks the compiler moved out of line (rare paths) are listed separately
by their "looping form" neighbourhood: a scratch op counts as hot when no v_cmp/branch-to-cold label separates it — so
read the listing it prints, not just the count."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "zipnn_amd", "csrc", "zn_decode_fused.hip")
out = os.path.join(tempfile.gettempdir(), "zn_hot.s")
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", out, src,
                    "-Rpass-analysis=kernel-resource-usage"] + flags, capture_output=True, text=True)
res = {}
cur = None
for l in r.stderr.split("\n"):
    m = re.search(r"Function Name: (\S+)", l)
    if m: cur = m.group(1); res[cur] = {}
    for key in ("VGPRs:", "ScratchSize [bytes/lane]:", "VGPRs Spill:", "SGPRs Spill:"):
        if key in l and cur: res[cur][key] = l.split(key)[1].split("[")[0].strip()
L = open(out).read().split("\n")
kern = None; depth = 0; hot = {}
for i, l in enumerate(L):
    m = re.match(r"^(_Z\w+):", l)
    if m: kern = m.group(1); depth = 0
    if "ZN_HOT_TILE_BEGIN" in l: depth += 1; hot.setdefault(kern, []).append([i, None, []])
    elif "ZN_HOT_TILE_END" in l and hot.get(kern) and hot[kern][-1][1] is None: hot[kern][-1][1] = i; depth = 0
    elif depth and "scratch_" in l and hot.get(kern): hot[kern][-1][2].append((i + 1, l.strip()))
for k, v in res.items():
    if "decode_fused" not in k: continue
    print(k[:40], v)
    for b, e, sc in hot.get(k, []):
        print(f"   tile loop lines {b + 1}-{(e or 0) + 1}: {len(sc)} scratch ops textually inside")
        for n, t in sc[:12]: print(f"      {n}: {t}")
