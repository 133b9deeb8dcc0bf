#!/bin/bash
# Runs on the GPU box via gpurun: parity tests, smoke, bench, rocprofv3 kernel stats.
# Usage: scripts/gpu_run.sh <tag> [steps]     (outputs under gpurun_out/<tag>/)
set -u
TAG="${1:-run}"; STEPS="${2:-50}"; WARM="${3:-20}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd "$R"
echo "== rocminfo ==" > "$OUT/env.log"; (rocminfo | grep -E 'Marketing Name|gfx|Compute Unit' | head -8; nproc; lscpu | grep 'Model name') >> "$OUT/env.log" 2>&1
echo "== pytest -m gpu =="; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log"
echo "== smoke =="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$OUT/smoke.log"
echo "== bench =="; timeout 900 python bench.py --steps "$STEPS" --warmup "$WARM" 2>&1 | tail -5 | tee "$OUT/bench.log"
echo "== other dtypes (1 GiB) =="; timeout 300 python scripts/bench_dtypes.py 1.0 2>&1 | grep GiB | tee "$OUT/dtypes.log"
echo "== rocprofv3 kernel stats =="
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$R/bench.py" --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline --no-other-dtypes > "$OUT/rocprof.log" 2>&1
tail -3 "$OUT/rocprof.log"
find "$OUT/prof" -name '*kernel_stats*' | head -3 | while read f; do echo "--- $f"; head -12 "$f"; done
# keep only the small summaries (the merged gpurun_out is capped at 64 MiB)
find "$OUT/prof" -name '*kernel_trace*' -size +20M -delete 2>/dev/null
echo "== done =="
