#!/bin/bash
# Runs on the GPU box via gpurun: resident waves of the decode kernel (SQ_WAVE_CYCLES / kernel time) for the product library.
set -u
TAG="${1:-occ}"; KIND="${2:-bf16}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$OUT/p1" -o pmc -- python "$R/scripts/dtype_probe.py" $KIND 1.0 2 > "$OUT/p1.log" 2>&1
tail -2 "$OUT/p1.log" | cut -c1-300
python "$R/scripts/pmc_summary.py" "$OUT" | grep decode_fused | tee "$OUT/summary.txt"
find "$OUT" -name '*.db' -size +30M -delete
