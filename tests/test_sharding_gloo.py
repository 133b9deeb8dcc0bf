"""CPU tests (-m "not gpu") of the N>1 path: chunk ranges shard across ranks with no data-path
collective.  world_size-2 gloo processes each code their own chunk range (the oracle stands in for the
GPU codec here — this test is about the sharding arithmetic and the process plumbing); the merged
frame must be byte-identical to the single-rank frame, and split bodies must decode independently."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from test_oracle import gen_bytes
from zipnn_amd import sharding

C = 64 * 1024
CASES = [("bf16", 7 * C + 1234, 2, 1, 10), ("fp32", 5 * C, 4, 1, 220), ("fp8", 3 * C + 5, 1, 1, 10), ("bf16", C - 7, 2, 1, 10)]


def _worker(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kind, nb, P, rot, bm = case
    data = gen_bytes(kind, nb, 9)
    K = (nb + C - 1) // C
    lo, hi = sharding.chunk_ranges(K, world)[rank]
    mine = data[lo * C: min(hi * C, nb)]
    body = O.compress_frame(b"", mine, P, rot, bm, C) if hi > lo else b""
    # metadata-only exchange: every rank learns every body length (no payload collective on the data path)
    lens = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(lens, torch.tensor([len(body)], dtype=torch.int64))
    q.put((rank, body, hi - lo, [int(x) for x in lens]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}")
def test_two_ranks_merge_to_single_rank_frame(case):
    kind, nb, P, rot, bm = case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + nb) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    [p.start() for p in procs]
    got = sorted([q.get(timeout=120) for _ in range(2)])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got[0][3] == [len(got[0][1]), len(got[1][1])]
    merged = sharding.merge_bodies([(b, k) for _, b, k, _ in got], P)
    data = gen_bytes(kind, nb, 9)
    assert merged == O.compress_frame(b"", data, P, rot, bm, C)          # byte-identical to the 1-rank frame
    assert O.decompress_body(merged, P, rot, bm, C, nb) == data


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_split_bodies_decode_independently(world):
    kind, nb, P, rot, bm = CASES[0]
    data = gen_bytes(kind, nb, 9)
    body = O.compress_frame(b"", data, P, rot, bm, C)
    parts = sharding.split_body(body, P, C, nb, world)
    out = b"".join(O.decompress_body(sub, P, rot, bm, C, length) if length else b"" for sub, off, length in parts)
    assert out == data
    K = (nb + C - 1) // C
    again = sharding.merge_bodies([(sub, hi - lo) for (sub, _, _), (lo, hi) in zip(parts, sharding.chunk_ranges(K, world))], P)
    assert again == body
