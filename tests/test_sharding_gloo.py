"""CPU tests (-m "not gpu") of the N>1 path: chunk ranges shard across ranks with no data-path
collective.  world_size-2 gloo processes each code their own chunk range with the PRODUCT's kernels and C ABI
(the SIMT-emulated build, tests/simt/libzipnn_simt.so — there is no GPU here); the oracle is only the checker: the
merged frame must be byte-identical to the single-rank oracle frame, and split bodies must decode independently."""
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
    from zipnn_amd._capi import ZnLib
    lib = ZnLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt", "libzipnn_simt.so"))   # the product's kernels, emulated
    body = bytes(lib.compress(b"", mine, P, rot, bm, C, 0.95)) if hi > lo else b""
    if hi > lo:                                   # and the rank's own body decodes back with the same library
        assert bytes(lib.decompress(body, P, rot, bm, C, len(mine))) == mine
    # metadata-only exchange: every rank learns every body length (no payload collective on the data path)
    lens = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(lens, torch.tensor([len(body)], dtype=torch.int64))
    q.put((rank, body, hi - lo, [int(x) for x in lens]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-{c[1]}")
def test_two_ranks_merge_to_single_rank_frame(case, simt_lib):
    kind, nb, P, rot, bm = case          # (simt_lib: builds the emulated library before the ranks start)
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
    assert sharding.merge_bodies([(b, k) for _, b, k, _ in got], P, lib=simt_lib) == merged      # the library's placement (zn_merge_range_bodies)
    assert O.decompress_body(merged, P, rot, bm, C, nb) == data


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_split_bodies_decode_independently(world, simt_lib):
    kind, nb, P, rot, bm = CASES[0]
    data = gen_bytes(kind, nb, 9)
    body = O.compress_frame(b"", data, P, rot, bm, C)
    parts = sharding.split_body(body, P, C, nb, world)
    out = b"".join(O.decompress_body(sub, P, rot, bm, C, length) if length else b"" for sub, off, length in parts)
    assert out == data
    K = (nb + C - 1) // C
    again = sharding.merge_bodies([(sub, hi - lo) for (sub, _, _), (lo, hi) in zip(parts, sharding.chunk_ranges(K, world))], P)
    assert again == body
    assert sharding.merge_bodies([(sub, hi - lo) for (sub, _, _), (lo, hi) in zip(parts, sharding.chunk_ranges(K, world))], P, lib=simt_lib) == body
    # one range alone, decoded by the library straight from the whole body (zn_decompress_range_dev; emulated device memory = host)
    for lo, hi in sharding.chunk_ranges(K, world):
        if hi > lo:
            n_r = min(hi * C, nb) - lo * C
            out = torch.zeros(n_r, dtype=torch.uint8)
            simt_lib.decompress_range_dev(body, P, rot, bm, C, nb, lo, hi, 0, out.data_ptr())
            assert out.numpy().tobytes() == data[lo * C: lo * C + n_r]


def _replicated_worker(rank, world, port, case, q):
    """Each rank decodes its chunk range with the PRODUCT kernels (SIMT-emulated build, CPU tensors), gloo all-gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zipnn_amd._capi import ZnLib
    lib = ZnLib(os.path.join(os.path.dirname(os.path.abspath(__file__)), "simt", "libzipnn_simt.so"))
    kind, nb, P, rot, bm = case
    data = gen_bytes(kind, nb, 9)
    body = O.compress_frame(b"", data, P, rot, bm, C)
    out = sharding.decompress_replicated(lib, body, P, rot, bm, C, nb, torch.device("cpu"))
    q.put((rank, out.numpy().tobytes() == data, out.numel()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [CASES[0], CASES[3], ("fp32", 3 * C, 4, 1, 220)], ids=lambda c: f"{c[0]}-{c[1]}")
def test_replicated_decode_all_gathers_the_shards(case, simt_lib):
    """every rank decodes 1/G of the chunks, one all-gather leaves the whole tensor on every rank
    (3 chunks on 2 ranks, a single partial chunk on 2 ranks — rank 1 idle —, and 8 chunks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + case[1]) % 2000
    procs = [ctx.Process(target=_replicated_worker, args=(r, 2, port, case, q)) for r in range(2)]
    [p.start() for p in procs]
    got = sorted([q.get(timeout=180) for _ in range(2)])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert [g[1] for g in got] == [True, True] and all(g[2] == case[1] for g in got)


def test_uniform_ranges_cover_every_chunk_once():
    for K in (0, 1, 2, 7, 8, 9, 1000):
        for G in (1, 2, 3, 8):
            r = sharding.uniform_ranges(K, G)
            assert r[0][0] == 0 and r[-1][1] == K and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert len({hi - lo for lo, hi in r if hi - lo} | {0}) <= 3
