#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel statistics of small-tensor decodes (64 MiB / 16 MiB bf16) with (mode 1) and without
# (mode 0) the wide kernel (zn_decode_wide.hpp).  Summaries: gpurun_out/<tag>/kernel_stats_m<mode>_<GiB>.txt
set -u
TAG="${1:-wideprof}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  for gib in 0.0625 0.015625; do
    d="$OUT/prof_m${mode}_$gib"
    (cd "$R" && ZN_WIDE_MODE=$mode PYTHONPATH="$R" timeout 120 rocprofv3 --kernel-trace --stats -d "$d" -o ks -- python "$R/scripts/dtype_probe.py" bf16 $gib 30 > "$OUT/m${mode}_$gib.log" 2>&1)
    DB=$(find "$d" -name '*results.db' | head -1)
    [ -n "$DB" ] && python "$R/scripts/prof_summary.py" "$DB" > "$OUT/kernel_stats_m${mode}_$gib.txt"
    rm -rf "$d"
    echo "== mode $mode GiB $gib"; head -12 "$OUT/kernel_stats_m${mode}_$gib.txt"
  done
done
