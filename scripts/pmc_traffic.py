#!/usr/bin/env python
"""Developer tool: turn the per-dtype PMC summaries of scripts/gpu_pmc_dtypes.sh (gpurun_out/<tag>/<dtype>/summary.txt) into
profiles/traffic_pmc.json — HBM bytes per 1 GiB launch of the decode and encode kernels, per dtype — and copy the summaries to
profiles/<prefix>_pmc_<dtype>.txt.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of streaming
reads (MI355X_MICROARCH.md §HBM; calibrated on our own access widths in profiles/r02_counter_calibration.txt), WRITE_SIZE is exact.
    python scripts/pmc_traffic.py gpurun_out/r03a_pmc r03a"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(src, prefix):
    out = {"unit": "bytes per 1 GiB tensor per launch", "fetch_correction": 2.0, "decode": {}, "encode": {},
           "csrc_sha256_16": __import__("importlib").import_module("bench").csrc_digest(),
           "kernels_commit": subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "zipnn_amd/csrc"], capture_output=True, text=True).stdout.strip()}
    for kind in sorted(os.listdir(src)):
        f = os.path.join(src, kind, "summary.txt")
        if not os.path.exists(f):
            continue
        vals = {}
        for l in open(f):
            m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+per-dispatch\s+([\d.]+)", l)
            if m:
                k = re.sub(r"void |<.*", "", m.group(1).strip())
                vals.setdefault(k, {})[m.group(2)] = float(m.group(3)) * 1024.0
        dst = os.path.join(ROOT, "profiles", f"{prefix}_pmc_{kind}.txt")
        with open(dst, "w") as o:
            o.write(f"# rocprofv3 --kernel-trace --pmc <one set per run>  -- python scripts/dtype_probe.py {kind} 1.0 2   (scripts/gpu_pmc_dtypes.sh; MI355X; per-dispatch averages, 1 GiB tensor)\n")
            o.write("# SQ_* wave-level counts / quad-cycles; FETCH_SIZE, WRITE_SIZE in KiB (FETCH_SIZE reports half the streamed bytes on gfx950)\n")
            o.write(open(f).read())
        d = vals.get("zn_k_decode_fused")
        if d and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            out["decode"][kind] = {"hbm_bytes_per_gib_launch": int(2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]), "read": int(2 * d["FETCH_SIZE"]), "written": int(d["WRITE_SIZE"]),
                                   "from": f"profiles/{prefix}_pmc_{kind}.txt"}
        enc = {k: v for k, v in vals.items() if k.startswith("zn_k_encode") or k == "zn_k_scan_sizes"}
        if enc:
            out["encode"][kind] = {k: {"read": int(2 * v.get("FETCH_SIZE", 0)), "written": int(v.get("WRITE_SIZE", 0))} for k, v in enc.items()}
            out["encode"][kind]["hbm_bytes_per_gib_call"] = int(sum(2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0) for v in enc.values()))
            out["encode"][kind]["from"] = f"profiles/{prefix}_pmc_{kind}.txt"
    with open(os.path.join(ROOT, "profiles", "traffic_pmc.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out["decode"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
