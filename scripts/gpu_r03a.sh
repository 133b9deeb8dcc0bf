#!/bin/bash
# round 3, GPU call A2: kernel stats + PMC per dtype (raw rocprof output removed on the box: gpurun_out is capped at 64 MiB)
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/r03a"; mkdir -p "$OUT"
cd "$R"
(rocminfo | grep -E 'Marketing Name|gfx' | head -4; nproc; lscpu | grep 'Model name') > "$OUT/env.log" 2>&1
echo "== kernel stats per dtype =="
cd /tmp && export TMPDIR=/tmp
for K in bf16 fp8 fp16 fp32; do
  (cd "$R" && PYTHONPATH="$R" timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/ks_$K" -o ks -- python scripts/dtype_probe.py $K 1.0 8 > "$OUT/ks_$K.log" 2>&1)
  DB=$(find "$OUT/ks_$K" -name '*results.db' | head -1)
  [ -n "$DB" ] && python "$R/scripts/prof_summary.py" "$DB" | grep -E "^#|zn_k" > "$OUT/kernel_stats_$K.txt"
  cat "$OUT/kernel_stats_$K.txt"
  rm -rf "$OUT/ks_$K"
done
cd "$R"
echo "== PMC per dtype =="; bash scripts/gpu_pmc_dtypes.sh r03a_pmc "fp8 fp16 fp32 bf16" > "$OUT/pmc.log" 2>&1; tail -3 "$OUT/pmc.log"
du -sh "$R/gpurun_out"
echo "== done =="
