#!/bin/bash
# Developer tool (GPU box): shader / memory clocks and power WHILE the decode kernel runs (boxes of the pool differ: the same library
# decodes the 4 GiB tensor in 1.46-1.69 ms), then the bench headline on the same box.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
python - <<'PY' &
import torch, time
from zipnn_amd import _capi, codec
import bench as B
lib = _capi.lib(); dev = torch.device("cuda", 0)
x = B.make_tensor(1 << 30, dev, 1); flat = codec.flat_bytes(x)
body = codec.compress_device(lib, flat, B.P, B.ROT, B.BMODE, B.CHUNK, B.THR).clone()
dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(200):
        codec.decompress_device(lib, body, B.P, B.ROT, B.BMODE, B.CHUNK, 1 << 30, out=dst, check=False)
    torch.cuda.synchronize()
PY
BG=$!
sleep 6
rocm-smi --showclocks --showpower --showperflevel --showtemp 2>/dev/null | grep -v "^$" | head -40
sleep 2
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk"
wait $BG
timeout 200 python bench.py --no-cpu-baseline --no-plugin --no-llama8b --no-other-dtypes 2>/dev/null | cut -c1-260
