#!/bin/bash
# rocprofv3 --kernel-trace --stats of scripts/dtype_probe.py per dtype (1 GiB tensors): per-kernel times of every dtype's instances.
# Usage: scripts/gpu_kernel_stats_dtypes.sh <tag> "bf16 fp16 fp32 fp8"     (summaries: gpurun_out/<tag>/kernel_stats_<dtype>.txt)
set -u
TAG="${1:-ks}"; KINDS="${2:-bf16 fp16 fp32 fp8}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for K in $KINDS; do
  (cd "$R" && PYTHONPATH="$R" timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$K" -o ks -- python "$R/scripts/dtype_probe.py" "$K" 1.0 10 > "$OUT/$K.log" 2>&1)
  DB=$(find "$OUT/prof_$K" -name '*results.db' | head -1)
  [ -n "$DB" ] && python "$R/scripts/prof_summary.py" "$DB" > "$OUT/kernel_stats_$K.txt"
  rm -rf "$OUT/prof_$K"
  echo "== $K"; head -7 "$OUT/kernel_stats_$K.txt"
done
