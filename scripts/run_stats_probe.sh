#!/bin/bash
# Developer tool (GPU box): per-kernel times of the encoder for each variant library (scripts/stats_probe.py under rocprofv3)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/stats_probe"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/sp_$v
  timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/sp_$v -o r -- python $R/scripts/stats_probe.py $v > "$OUT/$v.log" 2>&1
  f=$(find /tmp/sp_$v -name "*_results.db" | head -1)
  echo "== $v"; tail -1 "$OUT/$v.log" | cut -c1-200
  [ -n "$f" ] && python $R/scripts/prof_summary.py "$f" | grep -E "zn_k_encode|zn_k_scan" | cut -c1-130
done
