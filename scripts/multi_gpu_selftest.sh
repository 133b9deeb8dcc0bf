#!/bin/bash
# The first thing to run on a box with >= 2 GPUs (nothing in this repository has ever executed on two devices: the builder has one):
# the two auto-skipping multi-device GPU tests, then bench.py on two ranks over RCCL with a small tensor.  < 60 s.
#   bash scripts/multi_gpu_selftest.sh [N=2]
set -u
N="${1:-2}"
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python - <<'PY'
import torch
print("visible GPUs:", torch.cuda.device_count())
PY
echo "== the multi-device tests (skip themselves below 2 GPUs) =="
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "multi_device_entries_with_the_tensor_resident_in_hbm or replicated_decode_over_rccl_between_two_gpus" 2>&1 | tail -3
if [ "$(python -c 'import torch; print(torch.cuda.device_count())')" -lt "$N" ]; then
  echo "== fewer GPUs than ranks: DRY RUN of bench.py, $N ranks sharing the visible GPU(s) over gloo (logic only, numbers mean nothing) =="
  export ZN_BENCH_SHARE_GPU=1 ZN_BENCH_BACKEND=gloo
fi
echo "== bench.py, $N ranks, 0.25 GiB per rank =="
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus "$N" --gib 0.25 --steps 10 --warmup 5 \
  --no-cpu-baseline --no-other-dtypes --no-plugin --layers 2 2>/dev/null | grep "^{" | python -c "
import json, sys
j = json.loads(sys.stdin.readline())
print('n_gpus', j['n_gpus'], 'rccl_ranks', j['rccl_ranks'], 'value', j['value'], 'GB/s', 'rank_ms', j.get('rank_ms_per_step'), 'exact', j['bit_exact_roundtrip'])
print('llama8b', j['llama8b']['value'], 'GB/s', 'rank_ms', j['llama8b'].get('rank_ms_per_step'), 'exact', j['llama8b']['bit_exact_roundtrip'])"
echo "== bench.py, one process, every visible GPU through zn_decompress_multi_dev =="
timeout 300 python bench.py --gib 0.25 --steps 5 --warmup 2 --no-cpu-baseline --no-other-dtypes --no-plugin --no-llama8b 2>/dev/null | grep "^{" | python -c "
import json, sys
print('multi_dev_inprocess', json.loads(sys.stdin.readline()).get('multi_dev_inprocess'))"
