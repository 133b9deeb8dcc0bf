#!/bin/bash
# round 3, call E: full GPU suite + smoke + bench (new bench line) 
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/r03e"; mkdir -p "$OUT"
cd "$R"
echo "== pytest -m gpu =="; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/pytest_gpu.log"
echo "== smoke =="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log"
echo "== bench =="; (time timeout 900 python bench.py) > "$OUT/bench.log" 2>&1; tail -4 "$OUT/bench.log" | cut -c1-6000
echo "== done =="
