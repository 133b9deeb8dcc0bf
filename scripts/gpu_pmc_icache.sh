#!/bin/bash
# Instruction-cache counters of the decode / encode kernels (one dtype), each --pmc set in its own run, kernel-trace only.
# Usage: scripts/gpu_pmc_icache.sh <tag> <dtype>      (summary under gpurun_out/<tag>/<dtype>/summary.txt)
set -u
TAG="${1:-icache}"; K="${2:-bf16}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
OUT="$R/gpurun_out/$TAG/$K"; mkdir -p "$OUT"
i=0
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd "$R" && PYTHONPATH="$R" timeout 200 rocprofv3 --kernel-trace --pmc $SET -d "$OUT/p$i" -o pmc -- python "$R/scripts/dtype_probe.py" "$K" 1.0 2 > "$OUT/p$i.log" 2>&1)
  tail -1 "$OUT/p$i.log" | cut -c1-200
done
python "$R/scripts/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
for d in "$OUT"/p*/; do rm -rf "$d"; done
