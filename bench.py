#!/usr/bin/env python
"""bench.py — the hot path's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W                      (default workload: BASELINE.json configs[1])
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 \
           bench.py --gpus N --steps K --warmup W                      (N > 1: one rank per GPU)
    python bench.py --workload llama8b [--gpus N ...]                  (BASELINE.json configs[4]: strong scaling)

Default workload.  One "step" = one pass of the decompress hot path over one batch: a 4 GiB synthetic bf16 tensor
(N(0, 0.02) like model weights, generated on the device, 2 147 483 648 elements, 16 384 chunks of 256 KiB), compressed
body already resident in HBM when the timed region starts, decoded into an HBM buffer.  GB/s = uncompressed bytes /
seconds (the reference README's convention).  At N > 1 each rank codes its own tensor (chunks shard with no data-path
collective: weak scaling) and `value` is the sum of bytes over ranks / the slowest rank's time.

Before the W warm-up steps the decode is launched `--settle-launches` times (default 16, untimed, reported on the
JSON line): after idle or other kernels the clocks need ~12 launches (25 ms) of this kernel to settle
(profiles/r01z2_launch_time_ramp.txt); every timed step is a full step either way.

Also on the JSON line:
  roofline           decompress: (N + C_payload) bytes / avg launch time over the K timed steps, measured with HIP events on
                     the launch stream, vs the 8 TB/s HBM3E peak
  compress_*         the same tensor compressed K/2 times into a body buffer allocated ONCE outside the loop, every step
                     timed with HIP events (avg / min / median); compress_roofline prices (N + C_payload) on the avg
  other_dtypes       fp16 / fp32 / fp8-e4m3 tensors of 1 GiB each (BASELINE.json configs[2]): event-timed decode and
                     compress, GB/s, ratio, (N + C)/t as a fraction of 8 TB/s, exact round trip (N = 1 only)
  cpu_baseline       the reference's own C core (oracle/_ref: reference csrc/ + libzstd 1.4.8 huff0; "port" = our C
                     restatement if that build is absent) timed on the host cores, and the WHOLE GPU frame compared
                     with the CPU frame of the same tensor (gpu_frame_equals_cpu_frame)

--workload llama8b.  The 291 tensors of a Llama-3-8B checkpoint (bf16, N(0, 0.02), ~16 GB) plus an fp8-e4m3 copy of
its linear weights (~7 GB), synthesised on the devices.  Every tensor's chunks are split into WORLD_SIZE contiguous
ranges (zipnn_amd.sharding.chunk_ranges); rank g codes range g of every tensor with one batched call per direction.
The total work is fixed, so the line says "scaling": "strong"; at N = 1 the whole model runs on one GPU.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
CHUNK = 256 * 1024
P, ROT, BMODE, THR = 2, 1, 10, 0.95


def make_tensor(n_bytes, device, seed, dtype=torch.bfloat16):
    """N(0, 0.02) in 256 MiB slabs (seeds seed+k) in `dtype`, as a tensor in HBM."""
    es = torch.empty(0, dtype=dtype).element_size()
    out = torch.empty(n_bytes // es, dtype=dtype, device=device)
    slab = 128 * 1024 * 1024
    g = torch.Generator(device=device)
    for k, off in enumerate(range(0, out.numel(), slab)):
        g.manual_seed(seed + k)
        m = min(slab, out.numel() - off)
        out[off:off + m] = (torch.randn(m, generator=g, device=device) * 0.02).to(dtype)
    return out


def stats(ms):
    s = sorted(ms)
    return {"avg": sum(ms) / len(ms), "min": s[0], "median": s[len(s) // 2]}


def time_events(fn, steps):
    """K calls of fn(), each bracketed by events on the stream the library launches on (torch's current stream)."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def cpu_baseline(raw, want_body):
    """Time the CPU reference on the host cores (rank 0, N = 1 only) and compare its frame with the GPU's.
    Test infrastructure is used here strictly as the thing being compared AGAINST."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    threads = min(os.cpu_count() or 1, 16)            # the reference default (zipnn/zipnn.py:176-177)
    hdr = bytes(32)
    kind = "reference" if O.ref_core() is not None else "port"
    best_c = best_d = 1e9
    frame = None
    reps = 2 if raw.size > (1 << 30) else 3
    for _ in range(reps):
        if kind == "reference":
            buf = bytearray(raw.tobytes())             # the reference rotates its input in place
            t0 = time.perf_counter()
            frame = O.ref_core().zipnn_core(bytearray(hdr), buf, P, ROT, BMODE, 0, CHUNK, THR, 10, threads)
            best_c = min(best_c, time.perf_counter() - t0)
            del buf
            t0 = time.perf_counter()
            back = O.ref_core().combine_dtype(memoryview(frame)[32:], P, ROT, BMODE, CHUNK, raw.size, threads)
            best_d = min(best_d, time.perf_counter() - t0)
        else:
            t0 = time.perf_counter()
            frame = O.compress_frame(hdr, raw, P, ROT, BMODE, CHUNK, THR, threads)
            best_c = min(best_c, time.perf_counter() - t0)
            t0 = time.perf_counter()
            back = O.decompress_body(frame[32:], P, ROT, BMODE, CHUNK, raw.size, threads)
            best_d = min(best_d, time.perf_counter() - t0)
    import numpy as np
    assert np.array_equal(np.frombuffer(back, dtype=np.uint8), raw)
    fb = np.frombuffer(memoryview(frame)[32:], dtype=np.uint8)
    parity = bool(fb.size == want_body.size and np.array_equal(fb, want_body)) if want_body is not None else None
    gb = raw.size / 1e9
    return {"value": round(gb / best_d, 3), "unit": "GB/s", "cores": threads, "kind": kind,
            "compress_GBps": round(gb / best_c, 3),
            "sample": f"the first {raw.size >> 20} MiB of the same tensor (all of it when that is its size), decompress best of {reps}, {threads} threads",
            "gpu_frame_equals_cpu_frame": parity, "frame_compare_bytes": int(raw.size)}


def other_dtypes(lib, codec, device, steps):
    """BASELINE.json configs[2] (+ fp8): 1 GiB each, event-timed; the bytes of record for these dtypes on the driver's run."""
    out = {}
    cases = [("fp16", torch.float16, 2, 0, 10, CHUNK), ("fp32", torch.float32, 4, 1, 220, CHUNK)]
    f8 = getattr(torch, "float8_e4m3fn", None)
    if f8 is not None:
        cases.append(("fp8_e4m3", f8, 1, 0, 10, CHUNK // 2))     # the reference caps fp8 chunks at 128 KiB (zipnn.py:721)
    n = 1 << 30
    for name, dt, nb, rot, bm, chunk in cases:
        x = make_tensor(n, device, 4321, dt)
        flat = codec.flat_bytes(x)
        cap = lib.compress_bound(n, nb, chunk, 0)
        body = torch.empty(cap, dtype=torch.uint8, device=device)
        used = codec.compress_device(lib, flat, nb, rot, bm, chunk, THR, body=body).numel()
        dst = torch.empty(n, dtype=torch.uint8, device=device)
        codec.decompress_device(lib, body[:used], nb, rot, bm, chunk, n, out=dst)
        torch.cuda.synchronize()
        exact = bool(torch.equal(dst, flat))
        for _ in range(6):
            codec.decompress_device(lib, body[:used], nb, rot, bm, chunk, n, out=dst, check=False)
        d = stats(time_events(lambda: codec.decompress_device(lib, body[:used], nb, rot, bm, chunk, n, out=dst, check=False), steps))
        codec.compress_device(lib, flat, nb, rot, bm, chunk, THR, body=body)
        c = stats(time_events(lambda: codec.compress_device(lib, flat, nb, rot, bm, chunk, THR, body=body), max(2, steps // 2)))
        exact = exact and bool(torch.equal(dst, flat))
        cpl = used - 9 * nb * ((n + chunk - 1) // chunk)
        out[name] = {"GiB": 1.0, "ratio": round((used + 32) / n, 5), "exact": exact,
                     "decompress_GBps": round(n / d["avg"] / 1e6, 1), "decompress_ms": round(d["avg"], 4), "decompress_ms_min": round(d["min"], 4),
                     "decompress_roofline_frac": round((n + cpl) / (d["avg"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "compress_GBps": round(n / c["avg"] / 1e6, 1), "compress_ms": round(c["avg"], 4), "compress_ms_min": round(c["min"], 4),
                     "compress_roofline_frac": round((n + cpl) / (c["avg"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        del x, flat, body, dst
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------
# --workload llama8b (BASELINE.json configs[4])
# ---------------------------------------------------------------------------------------------------------------
def llama8b_shapes(layers=32, hidden=4096, inter=14336, vocab=128256, kv_heads=8, heads=32):
    """(name, shape, is_linear) of the 291 tensors of a Llama-3-8B checkpoint."""
    kv = hidden // heads * kv_heads
    t = [("model.embed_tokens.weight", (vocab, hidden), False)]
    for i in range(layers):
        p = f"model.layers.{i}."
        t += [(p + "self_attn.q_proj.weight", (hidden, hidden), True), (p + "self_attn.k_proj.weight", (kv, hidden), True),
              (p + "self_attn.v_proj.weight", (kv, hidden), True), (p + "self_attn.o_proj.weight", (hidden, hidden), True),
              (p + "mlp.gate_proj.weight", (inter, hidden), True), (p + "mlp.up_proj.weight", (inter, hidden), True),
              (p + "mlp.down_proj.weight", (hidden, inter), True),
              (p + "input_layernorm.weight", (hidden,), False), (p + "post_attention_layernorm.weight", (hidden,), False)]
    t += [("model.norm.weight", (hidden,), False), ("lm_head.weight", (vocab, hidden), False)]
    return t


def run_llama8b(args, lib, codec, device, world, rank, dist, td):
    from zipnn_amd import sharding
    shapes = llama8b_shapes(layers=args.layers)
    f8 = getattr(torch, "float8_e4m3fn", None)
    g = torch.Generator(device=device)
    items, total_bytes, my_bytes = [], 0, 0
    for i, (name, shape, linear) in enumerate(shapes):
        numel = 1
        for s in shape:
            numel *= s
        variants = [(torch.bfloat16, 2, 1, 10, CHUNK)] + ([(f8, 1, 0, 10, CHUNK // 2)] if (linear and f8 is not None) else [])
        for dt, nb, rot, bm, chunk in variants:
            nbytes = numel * torch.empty(0, dtype=dt).element_size()
            total_bytes += nbytes
            K = (nbytes + chunk - 1) // chunk
            lo, hi = sharding.chunk_ranges(K, world)[rank]
            if hi <= lo:
                continue
            a, b = lo * chunk, min(hi * chunk, nbytes)
            es = torch.empty(0, dtype=dt).element_size()
            # (only the rank's own chunk range is materialised, from a per-range seed: the same distribution, and the
            #  set-up cost stays 1/G of the model per rank)
            g.manual_seed(7000 + 1000 * i + lo)
            x = (torch.randn((b - a) // es, generator=g, device=device) * 0.02).to(dt)
            items.append((codec.flat_bytes(x), nb, rot, bm, chunk, THR))
            my_bytes += b - a

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    bodies = codec.compress_device_batch(lib, items)
    c_bytes = sum(b.numel() for b in bodies)
    ditems = [(b, nb, rot, bm, chunk, f.numel()) for b, (f, nb, rot, bm, chunk, _) in zip(bodies, items)]
    into = torch.empty(my_bytes, dtype=torch.uint8, device=device)
    outs = codec.decompress_device_batch(lib, ditems, into=into)
    torch.cuda.synchronize()
    decode_kernels = lib.last_kernels()
    exact = all(bool(torch.equal(o, it[0])) for o, it in zip(outs, items))
    for _ in range(args.warmup):
        codec.decompress_device_batch(lib, ditems, check=False, into=into)
    barrier()
    t0 = time.perf_counter()
    dms = time_events(lambda: codec.decompress_device_batch(lib, ditems, check=False, into=into), args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    csteps = max(1, args.steps // 2)
    codec.compress_device_batch(lib, items)
    encode_kernels = lib.last_kernels()
    barrier()
    t1 = time.perf_counter()
    for _ in range(csteps):
        codec.compress_device_batch(lib, items)
    barrier()
    c_elapsed = time.perf_counter() - t1
    exact = exact and all(bool(torch.equal(o, it[0])) for o, it in zip(outs, items))
    if dist:
        tt = torch.tensor([elapsed, c_elapsed, float(not exact)], device=device, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed, c_elapsed, bad = tt.tolist()
        exact = bad == 0.0
        cb = torch.tensor([float(c_bytes)], device=device, dtype=torch.float64)
        td.all_reduce(cb)
        c_bytes = int(cb.item())
    if rank == 0:
        d = stats(dms)
        line = {"metric": "Llama-3-8B-shaped checkpoint (bf16 + fp8 copy of the linears): decompress GB/s, chunk ranges sharded over the GPUs",
                "value": round(total_bytes * args.steps / elapsed / 1e9, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": f"llama8b: {len(shapes)} tensor shapes of Llama-3-8B ({args.layers} layers) as bf16 N(0,0.02) + fp8-e4m3 copy of the linear weights, "
                                       f"{total_bytes / 1e9:.2f} GB in all; every tensor's chunks split into {world} contiguous ranges, one batched call per rank "
                                       "(BASELINE.json configs[4])",
                           "tensors_per_rank": len(items), "parallelism": f"chunk-range-sharded x{world}, no collectives"},
                "compress_GBps": round(total_bytes * csteps / c_elapsed / 1e9, 2), "compress_ms_per_step": round(c_elapsed / csteps * 1e3, 3),
                "ratio": round(c_bytes / total_bytes, 5), "bit_exact_roundtrip": exact,
                "rank0_decode_ms": {k: round(v, 4) for k, v in d.items()},
                "kernels": {"decompress": decode_kernels, "compress": encode_kernels}}
        print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the clocks need ~12 launches (≈25 ms) of this kernel to settle after anything else has run
    # (profiles/r01z2_launch_time_ramp.txt: 2.2 → 1.76 ms per launch), so the default warm-up covers that
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-launches", type=int, default=16,
                    help="untimed decode launches before the W warm-up steps: the GPU's clocks take ~12 launches (25 ms) of this "
                         "kernel to settle after idle or other kernels; reported on the JSON line (0 = off)")
    ap.add_argument("--gib", type=float, default=4.0, help="uncompressed tensor size per GPU (GiB)")
    ap.add_argument("--workload", choices=["bf16", "llama8b"], default="bf16")
    ap.add_argument("--layers", type=int, default=32, help="llama8b: transformer layers to synthesise (32 = the real model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-dtypes", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=4096, help="bytes of the tensor the CPU reference is timed on and the GPU frame is compared on")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1 or ("LOCAL_RANK" in os.environ and "MASTER_ADDR" in os.environ)   # (under torch.distributed.run even a single rank takes the RCCL path)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    td = None
    if dist:
        import torch.distributed as td
        td.init_process_group(backend="nccl", device_id=device)

    from zipnn_amd import _capi, codec
    lib = _capi.lib()

    if args.workload == "llama8b":
        run_llama8b(args, lib, codec, device, world, rank, dist, td)
        if dist:
            td.barrier()
            td.destroy_process_group()
        return

    n_bytes = int(args.gib * (1 << 30)) // CHUNK * CHUNK
    x = make_tensor(n_bytes, device, 1234 + 1000 * rank)
    flat = codec.flat_bytes(x)
    cap = lib.compress_bound(n_bytes, P, CHUNK, 0)
    body_buf = torch.empty(cap, dtype=torch.uint8, device=device)     # allocated ONCE: the compress loop below reuses it
    body = codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR, body=body_buf).clone()
    c_payload = body.numel() - 9 * P * (n_bytes // CHUNK)
    out = torch.empty(n_bytes, dtype=torch.uint8, device=device)
    codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n_bytes, out=out)
    torch.cuda.synchronize()
    exact = bool(torch.equal(out, flat))
    assert exact, "decompressed bytes differ from the input"
    decode_kernels = lib.last_kernels()

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    # ---- decompress: clock settle + W warm-up + exactly K timed steps ------------------
    for _ in range(max(0, args.settle_launches)):
        codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n_bytes, out=out, check=False)
    for _ in range(args.warmup):
        codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n_bytes, out=out, check=False)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = time_events(lambda: codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n_bytes, out=out, check=False), args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    d = stats(kernel_ms)
    # the output of the LAST timed step, and one more call with the device-side status read back
    exact = exact and bool(torch.equal(out, flat))
    out.zero_()
    codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n_bytes, out=out, check=True)
    exact = exact and bool(torch.equal(out, flat))

    # ---- compress: same tensor, K/2 timed steps into the preallocated body buffer -------
    for _ in range(max(2, min(args.warmup, 4))):
        codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR, body=body_buf)
    barrier()
    t1 = time.perf_counter()
    csteps = max(1, args.steps // 2)
    used = [0]

    def one_compress():
        used[0] = codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR, body=body_buf).numel()
    comp_ms = time_events(one_compress, csteps)
    barrier()
    c_elapsed = time.perf_counter() - t1
    c = stats(comp_ms)
    same_body = used[0] == body.numel() and bool(torch.equal(body_buf[:used[0]], body))
    encode_kernels = lib.last_kernels()

    if dist:
        tt = torch.tensor([elapsed, c_elapsed, float(not (exact and same_body))], device=device, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed, c_elapsed, bad = tt.tolist()
        exact = exact and bad == 0.0

    if rank == 0:
        total_bytes = n_bytes * world
        value = total_bytes * args.steps / elapsed / 1e9
        cvalue = total_bytes * csteps / c_elapsed / 1e9
        alg = n_bytes + c_payload
        achieved = alg / (d["avg"] * 1e-3) / 1e9
        c_achieved = alg / (c["avg"] * 1e-3) / 1e9
        traffic = None     # HBM bytes per launch from the committed PMC pass, scaled to this size
        try:
            with open(os.path.join(ROOT, "profiles", "decode_traffic_pmc.json")) as f:
                t = json.load(f)
            traffic = int(t["hbm_bytes_per_gib_launch"] * (n_bytes / (1 << 30)) / t["gib"])
        except Exception:
            pass
        line = {
            "metric": "bf16 decompress GB/s (uncompressed bytes / s; compress GB/s beside it)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle_launches": max(0, args.settle_launches),
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{n_bytes / (1 << 30):g} GiB synthetic bf16 N(0,0.02) tensor per GPU, 256 KiB chunks, "
                                   "byte-split + huff0 (BASELINE.json configs[1])",
                       "chunks_per_gpu": n_bytes // CHUNK, "parallelism": f"chunk-sharded x{world}, no collectives"},
            "compress_GBps": round(cvalue, 2), "compress_ms_per_step": round(c_elapsed / csteps * 1e3, 3),
            "compress_steps": csteps, "compress_frame_identical_every_step": same_body,
            "ratio": round((body.numel() + 32) / n_bytes, 5), "bit_exact_roundtrip": exact,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": decode_kernels, "avg_launch_ms": round(d["avg"], 4),
                         "min_launch_ms": round(d["min"], 4), "median_launch_ms": round(d["median"], 4),
                         "algorithmic_bytes": alg},
            "compress_roofline": {"bound": "hbm", "achieved": round(c_achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(c_achieved / HBM_PEAK_GBPS, 4), "kernel": encode_kernels,
                                  "avg_call_ms": round(c["avg"], 4), "min_call_ms": round(c["min"], 4), "median_call_ms": round(c["median"], 4),
                                  "algorithmic_bytes": alg,
                                  "note": "events bracket the whole zn_compress_dev call: four kernels + one 8-byte length read-back"},
            "kernels": {"decompress": decode_kernels, "compress": encode_kernels},
        }
        if world == 1 and not args.no_other_dtypes:
            del out
            torch.cuda.empty_cache()
            line["other_dtypes"] = other_dtypes(lib, codec, device, max(4, min(args.steps, 20)))
        if world == 1 and not args.no_cpu_baseline:
            sample = min(args.cpu_sample_mib << 20, n_bytes) // CHUNK * CHUNK
            if sample:
                if sample == n_bytes:
                    sbody = body
                else:
                    sbody = codec.compress_device(lib, flat[:sample], P, ROT, BMODE, CHUNK, THR)
                line["cpu_baseline"] = cpu_baseline(flat[:sample].cpu().numpy(), sbody.cpu().numpy())
        print(json.dumps(line), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
