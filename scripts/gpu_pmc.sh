#!/bin/bash
# PMC passes for the decode kernel (each --pmc set in its own run, kernel-trace only).
set -u
TAG="${1:-pmc}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_all.txt" 2>&1
grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+" "$OUT/counters_all.txt" | sort -u > "$OUT/counters.txt"
wc -l "$OUT/counters.txt"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d "$OUT/p$i" -o pmc -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-dtypes --gib ${GIB:-1} > "$OUT/p$i.log" 2>&1
  tail -2 "$OUT/p$i.log" | cut -c1-300
done
python "$R/scripts/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
find "$OUT" -name '*.db' -size +30M -delete
