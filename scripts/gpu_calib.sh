#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (MI355X_MICROARCH.md §HBM: the counters are only calibrated for
# 16 B/lane streaming reads; this kernel reads with 4-, 8- and 16-byte loads): the hand-written copy kernels of
# scripts/ubench/ubench.hip (4 GiB buffers, 4 / 8 / 16 B per lane) under rocprofv3, one --pmc pass per counter.
set -u
TAG="${1:-calib}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/$C" -o pmc -- "$R/scripts/ubench/ubench" copy > "$OUT/$C.log" 2>&1
done
python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
root = sys.argv[1]
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob(os.path.join(root, C, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
        namecol = "kernel_name" if "kernel_name" in cols else "name"
        q = f"select {namecol}, count(*), sum(value) from counters_collection where counter_name=? group by {namecol}"
        for k, n, tot in con.execute(q, (C,)):
            if "k_copy" in k:
                print(f"{C:11s} {k[:70]:70s} dispatches {n:3d}  avg {tot / n:14.1f}")
PY
find "$OUT" -name '*.db' -delete
