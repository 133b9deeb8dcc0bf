#!/bin/bash
# Runs on the GPU box via gpurun: the round's records — GPU tests, smoke, bench (default + under rocprofv3), kernel stats, PMC per dtype.
# Usage: scripts/gpu_final.sh <tag>     (summaries under gpurun_out/<tag>/; raw rocprof output is removed on the box: gpurun_out is capped at 64 MiB)
set -u
TAG="${1:-final}"
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"
cd "$R"
(rocminfo | grep -E 'Marketing Name|gfx' | head -4; nproc; lscpu | grep 'Model name') > "$OUT/env.log" 2>&1
echo "== pytest -m gpu =="; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$OUT/pytest_gpu.log"
echo "== smoke =="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
echo "== bench =="; (time timeout 900 python bench.py) > "$OUT/bench.log" 2>&1; grep "^{" "$OUT/bench.log" > "$OUT/bench.json"; cut -c1-400 "$OUT/bench.json"
echo "== bench under rocprofv3 --kernel-trace --stats =="
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$R/bench.py" --no-cpu-baseline --no-other-dtypes --no-plugin --no-llama8b > "$OUT/rocprof.log" 2>&1
grep "^{" "$OUT/rocprof.log" > "$OUT/bench_under_rocprof.json"
DB=$(find "$OUT/prof" -name '*results.db' | head -1)
[ -n "$DB" ] && python "$R/scripts/prof_summary.py" "$DB" > "$OUT/kernel_stats.txt"; head -8 "$OUT/kernel_stats.txt"
rm -rf "$OUT/prof"
cd "$R"
echo "== bench under torch.distributed.run, one rank (the RCCL path) =="
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-dtypes --no-plugin 2>/dev/null | grep "^{" > "$OUT/bench_rccl_1rank.json"; python -c "
import json,sys; j=json.load(open('$OUT/bench_rccl_1rank.json')); print('rccl_ranks', j['rccl_ranks'], 'value', j['value'], 'llama8b', j['llama8b']['value'], j['llama8b']['n_gpus'])"
echo "== PMC per dtype =="; bash scripts/gpu_pmc_dtypes.sh "${TAG}_pmc" "bf16 fp16 fp32 fp8" > "$OUT/pmc.log" 2>&1; tail -2 "$OUT/pmc.log"
echo "== host path =="; timeout 300 python scripts/host_path_check.py 2>&1 | grep -v amdgpu | grep "host-buffer" | tee "$OUT/host_path.txt"
du -sh "$R/gpurun_out"
echo "== done =="
