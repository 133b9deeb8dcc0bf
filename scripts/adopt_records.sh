#!/bin/bash
# Developer tool: adopt the output of scripts/gpu_final.sh <tag> (merged back under gpurun_out/) as the round's records:
# profiles/<tag>_*, profiles/traffic_pmc.json, and the tag cited in DESIGN.md / README.md / profiles/README.md.
#   scripts/adopt_records.sh <new tag> <tag the documents cite now>
set -eu
NEW="$1"; OLD="$2"; R="$(cd "$(dirname "$0")/.." && pwd)"; G="$R/gpurun_out/$NEW"
for f in bench.json bench_under_rocprof.json bench_rccl_1rank.json kernel_stats.txt host_path.txt; do cp "$G/$f" "$R/profiles/${NEW}_$f"; done
cp "$G/pytest_gpu.log" "$R/profiles/${NEW}_pytest_gpu.txt"
python "$R/scripts/pmc_traffic.py" "$R/gpurun_out/${NEW}_pmc" "$NEW"
git -C "$R" rm -q --cached "profiles/${OLD}_"* 2>/dev/null || true; rm -f "$R/profiles/${OLD}_"*
sed -i "s/${OLD}_/${NEW}_/g; s/(${OLD})/(${NEW})/g; s/\`${OLD}\`/\`${NEW}\`/g" "$R/DESIGN.md" "$R/README.md" "$R/profiles/README.md" "$R/INTEGRATION.md"
grep -n "$OLD" "$R/DESIGN.md" "$R/README.md" "$R/profiles/README.md" "$R/INTEGRATION.md" || echo "no mention of $OLD left"
