#!/bin/bash
# PMC passes per dtype (decode and encode kernels of that dtype's instances); each --pmc set in its own run, kernel-trace only.
# Usage: scripts/gpu_pmc_dtypes.sh <tag> "fp8 fp16 fp32" [GiB=1.0]      (outputs under gpurun_out/<tag>/<dtype>/; GiB 0.0625 = the small-input kernels)
set -u
TAG="${1:-pmcd}"; KINDS="${2:-fp8 fp16 fp32}"; GIB="${3:-1.0}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
for K in $KINDS; do
  OUT="$R/gpurun_out/$TAG/$K"; mkdir -p "$OUT"
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    (cd "$R" && PYTHONPATH="$R" timeout 200 rocprofv3 --kernel-trace --pmc $SET -d "$OUT/p$i" -o pmc -- python "$R/scripts/dtype_probe.py" "$K" "$GIB" 2 > "$OUT/p$i.log" 2>&1)
    tail -1 "$OUT/p$i.log" | cut -c1-200
  done
  python "$R/scripts/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
  for d in "$OUT"/p*/; do rm -rf "$d"; done      # (raw rocprof output: the merged gpurun_out is capped at 64 MiB)
done
