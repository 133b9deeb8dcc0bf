"""Chunk-range sharding of one ZN frame body across GPUs (host-side bookkeeping only).

Chunks are independent, so G ranks can code disjoint chunk ranges with no data-path collective
(SURVEY.md §8e).  The wire format is plane-major, so a rank's output lands in `num_buf` disjoint
regions of the final payload; these helpers do that arithmetic:

  chunk_ranges(K, G)                 contiguous ranges [g*K/G, (g+1)*K/G)
  sub_body(body, P, chunk, n, lo, hi)  the standalone body of ONE chunk range (types / re-based cumSizes / payload slices)
  split_body(body, P, chunk, n, G)   … of every rank's range (tests, tools; a rank that needs only its own calls sub_body)
  merge_bodies(parts, P, lib=None)   the inverse: bodies of consecutive chunk ranges -> one body; with `lib` the copying is the
                                     library's (zn_merge_range_bodies: one thread per part), without it plain numpy
  decompress_replicated(...)         every rank decodes its chunk range in HBM, one all-gather (RCCL over xGMI)
                                     leaves the whole tensor on every GPU (replicated-weight loading)

A merged body is byte-identical to the body a single device produces for the whole buffer.
"""
import numpy as np


def chunk_ranges(num_chunks, world):
    return [(g * num_chunks // world, (g + 1) * num_chunks // world) for g in range(world)]


def _parse(body, P, K):
    b = np.frombuffer(body, dtype=np.uint8)
    types = b[: P * K].reshape(P, K)
    cum = np.frombuffer(b[P * K: 9 * P * K].tobytes(), dtype=np.uint64).reshape(P, K).astype(np.int64)
    payload = b[9 * P * K:]
    return types, cum, payload


def uniform_ranges(num_chunks, world):
    """ceil(K / G) chunks per rank (the last ranks may get fewer or none): equal shard sizes, as an all-gather wants."""
    per = (num_chunks + world - 1) // world if world else 0
    return [(min(g * per, num_chunks), min((g + 1) * per, num_chunks)) for g in range(world)]


def sub_body(body, num_buf, chunk, orig_size, lo, hi):
    """-> (sub_body: bytes, byte_offset, byte_length) of the chunk range [lo, hi) alone: only that range's bytes are touched."""
    P = num_buf
    K = (orig_size + chunk - 1) // chunk
    k = hi - lo
    if k <= 0:
        return b"", lo * chunk, 0
    types, cum, payload = _parse(body, P, K)
    plane_base = np.concatenate([[0], np.cumsum(cum[:, -1])[:-1]])
    pieces, cums = [], []
    for p in range(P):
        start = int(cum[p, lo - 1]) if lo else 0
        end = int(cum[p, hi - 1])
        pieces.append(payload[int(plane_base[p]) + start: int(plane_base[p]) + end])
        cums.append((cum[p, lo:hi] - start).astype(np.uint64))
    sub = b"".join([types[:, lo:hi].tobytes()] + [c.tobytes() for c in cums] + [x.tobytes() for x in pieces])
    off = lo * chunk
    return sub, off, min(hi * chunk, orig_size) - off


def split_body(body, num_buf, chunk, orig_size, world, ranges=None):
    """-> list of (sub_body: bytes, byte_offset, byte_length) for each rank's chunk range
    (ranges: explicit [(lo, hi)] per rank; default chunk_ranges(K, world))."""
    K = (orig_size + chunk - 1) // chunk
    return [sub_body(body, num_buf, chunk, orig_size, lo, hi) for lo, hi in (ranges if ranges is not None else chunk_ranges(K, world))]


def merge_bodies(parts, num_buf, lib=None):
    """parts: list of (body: bytes-like, num_chunks) for consecutive chunk ranges -> one body (bytes).
    lib: a zipnn_amd._capi.ZnLib — the plane-major placement is then done by the library (zn_merge_range_bodies)."""
    if lib is not None:
        return bytes(lib.merge_range_bodies([(b, k) for b, k in parts if k], num_buf))
    P = num_buf
    types_all, cum_all, pay_all = [[] for _ in range(P)], [[] for _ in range(P)], [[] for _ in range(P)]
    run = np.zeros(P, dtype=np.int64)
    for body, k in parts:
        if k == 0:
            continue
        types, cum, payload = _parse(body, P, k)
        base = 0
        for p in range(P):
            tot = int(cum[p, -1])
            types_all[p].append(types[p])
            cum_all[p].append((cum[p] + run[p]).astype(np.uint64))
            pay_all[p].append(payload[base: base + tot])
            run[p] += tot
            base += tot
    cat = lambda xs, dt: (np.concatenate(xs).astype(dt).tobytes() if xs else b"")  # noqa: E731
    return (b"".join(cat(types_all[p], np.uint8) for p in range(P)) + b"".join(cat(cum_all[p], np.uint64) for p in range(P))
            + b"".join(cat(pay_all[p], np.uint8) for p in range(P)))


def decompress_replicated(lib, body, num_buf, bits_mode, bytes_mode, chunk, orig_size, device, group=None):
    """Replicated-weight loading: `body` (host bytes-like, the same on every rank) -> the decoded tensor bytes on
    EVERY rank's device.  Rank g uploads and decodes only its chunk range (1/G of the compressed bytes cross its PCIe
    link, 1/G of the decode work), then one all-gather of equal shards — RCCL over xGMI on GPUs — fills in the rest.
    This is the one place the path has a real exchange step; plain sharded decode (each rank keeps its slice) has none.
    Returns a uint8 tensor of orig_size bytes on `device`."""
    import torch
    import torch.distributed as dist
    device = torch.device(device)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    K = (orig_size + chunk - 1) // chunk
    ranges = uniform_ranges(K, world)
    shard = ((K + world - 1) // world) * chunk if world else 0          # bytes per rank in the gathered buffer
    full = torch.empty(max(shard * world, 1), dtype=torch.uint8, device=device)
    lo, hi = ranges[rank]
    if hi > lo:
        # only this rank's range is touched: its re-based size tables are built by the library, its payload slices go from `body`
        # through the pinned pipe straight into HBM (zn_decompress_range_dev) and are decoded into the rank's shard of `full`
        off = lo * chunk
        dev_index = device.index if (device.type == "cuda" and device.index is not None) else (torch.cuda.current_device() if device.type == "cuda" else 0)
        if device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()             # (`full` may be a block kernels queued earlier still use)
        lib.decompress_range_dev(body, num_buf, bits_mode, bytes_mode, chunk, orig_size, lo, hi, dev_index, full.data_ptr() + off)
    if dist.is_initialized():                                           # (a group of one still goes through the collective: RCCL on GPUs)
        mine = full[rank * shard:(rank + 1) * shard].clone()            # (gloo does not take an aliasing input)
        dist.all_gather_into_tensor(full[:shard * world], mine, group=group)
    return full[:orig_size]
