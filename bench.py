#!/usr/bin/env python
"""bench.py — the hot path's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W

Before the W warm-up steps the decode is launched `--settle-launches` times (default 16, untimed, reported on the
JSON line): after idle or other kernels the clocks need ~12 launches (25 ms) of this kernel to settle
(profiles/r01z2_launch_time_ramp.txt); every timed step is a full step either way.

One "step" = one pass of the decompress hot path over one batch: a 4 GiB synthetic bf16
tensor (BASELINE.json configs[1]; N(0, 0.02) like model weights, generated on the device,
2 147 483 648 elements, 16 384 chunks of 256 KiB), compressed body already resident in HBM
when the timed region starts, decoded into an HBM buffer.  GB/s = uncompressed bytes /
seconds (the reference README's convention).  Compress GB/s on the same tensor is timed in
a second loop and reported beside it (`compress_GBps`).  At N > 1 each rank codes its own
tensor (chunks shard with no data-path collective: weak scaling) and `value` is the sum of
bytes over ranks / the slowest rank's time.

Also on the JSON line:
  roofline      achieved (N + C_payload) bytes / avg decode time over the timed launches,
                measured with HIP events on the launch stream, vs the 8 TB/s HBM3E peak
  cpu_baseline  the reference's own C core (oracle/_ref: reference csrc/ + libzstd 1.4.8
                huff0; "port" = our C restatement if that build is absent) timed on the
                host cores on a bounded sample of the same tensor
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


def make_tensor(n_bytes, device, seed):
    """bf16 N(0, 0.02) in 256 MiB slabs (seeds seed+k), as raw bytes in HBM."""
    out = torch.empty(n_bytes // 2, dtype=torch.bfloat16, device=device)
    slab = 128 * 1024 * 1024
    g = torch.Generator(device=device)
    for k, off in enumerate(range(0, out.numel(), slab)):
        g.manual_seed(seed + k)
        m = min(slab, out.numel() - off)
        out[off:off + m] = (torch.randn(m, generator=g, device=device) * 0.02).to(torch.bfloat16)
    return out


def cpu_baseline(sample_u8, want_body):
    """Time the CPU reference on the host cores over a bounded sample (rank 0, N = 1 only).
    Test infrastructure is used here strictly as the thing being compared AGAINST."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    threads = min(os.cpu_count() or 1, 16)            # the reference default (zipnn/zipnn.py:176-177)
    raw = sample_u8.numpy()
    hdr = bytes(32)
    kind = "reference" if O.ref_core() is not None else "port"
    best_c = best_d = 1e9
    frame = None
    for _ in range(3):
        if kind == "reference":
            buf = bytearray(raw.tobytes())             # the reference rotates its input in place
            t0 = time.perf_counter()
            frame = O.ref_core().zipnn_core(bytearray(hdr), buf, P, ROT, BMODE, 0, CHUNK, THR, 10, threads)
            best_c = min(best_c, time.perf_counter() - t0)
            frame = bytes(frame)
            t0 = time.perf_counter()
            back = O.ref_core().combine_dtype(frame[32:], P, ROT, BMODE, CHUNK, raw.size, threads)
            best_d = min(best_d, time.perf_counter() - t0)
        else:
            t0 = time.perf_counter()
            frame = O.compress_frame(hdr, raw, P, ROT, BMODE, CHUNK, THR, threads)
            best_c = min(best_c, time.perf_counter() - t0)
            t0 = time.perf_counter()
            back = O.decompress_body(frame[32:], P, ROT, BMODE, CHUNK, raw.size, threads)
            best_d = min(best_d, time.perf_counter() - t0)
    assert bytes(back) == raw.tobytes()
    parity = (frame[32:] == want_body) if want_body is not None else None
    gb = raw.size / 1e9
    return {"value": round(gb / best_d, 3), "unit": "GB/s", "cores": threads, "kind": kind,
            "compress_GBps": round(gb / best_c, 3),
            "sample": f"first {raw.size >> 20} MiB of the same tensor, decompress best of 3, {threads} threads",
            "gpu_frame_equals_cpu_frame": parity}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=512)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if dist:
        import torch.distributed as td
        td.init_process_group(backend="nccl", device_id=device)

    from zipnn_amd import _capi, codec
    lib = _capi.lib()

    n_bytes = int(args.gib * (1 << 30)) // CHUNK * CHUNK
    x = make_tensor(n_bytes, device, 1234 + 1000 * rank)
    flat = codec.flat_bytes(x)
    body = codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR).clone()
    c_payload = body.numel() - 9 * P * (n_bytes // CHUNK)
    out = torch.empty(n_bytes, dtype=torch.uint8, device=device)
    codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n_bytes, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, flat), "decompressed bytes differ from the input"
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
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n_bytes, out=out, check=False)
        ev[i + 1].record()     # same stream the library launches on (torch's current stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
    step_ms_min, step_ms_median = min(kernel_ms), sorted(kernel_ms)[len(kernel_ms) // 2]

    # ---- compress: same tensor, K timed steps ------------------------------------------
    for _ in range(min(args.warmup, 2)):
        codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR)
    barrier()
    t1 = time.perf_counter()
    csteps = max(1, args.steps // 2)
    for _ in range(csteps):
        cb = codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR)
    barrier()
    c_elapsed = time.perf_counter() - t1
    assert cb.numel() == body.numel()
    encode_kernels = lib.last_kernels()

    if dist:
        tt = torch.tensor([elapsed, c_elapsed], device=device, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed, c_elapsed = tt.tolist()

    if rank == 0:
        total_bytes = n_bytes * world
        value = total_bytes * args.steps / elapsed / 1e9
        cvalue = total_bytes * csteps / c_elapsed / 1e9
        achieved = (n_bytes + c_payload) / (avg_kernel_ms * 1e-3) / 1e9
        traffic = None     # HBM bytes per launch from the committed PMC pass (FETCH_SIZE x2 + WRITE_SIZE), scaled to this size
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
            "ratio": round((body.numel() + 32) / n_bytes, 5), "bit_exact_roundtrip": True,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": decode_kernels, "avg_launch_ms": round(avg_kernel_ms, 4),
                         "min_launch_ms": round(step_ms_min, 4), "median_launch_ms": round(step_ms_median, 4),
                         "algorithmic_bytes": n_bytes + c_payload},
            "kernels": {"decompress": decode_kernels, "compress": encode_kernels},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = min(args.cpu_sample_mib << 20, n_bytes) // CHUNK * CHUNK
            sbody = None
            if sample:
                sb = codec.compress_device(lib, flat[:sample], P, ROT, BMODE, CHUNK, THR)
                sbody = sb.cpu().numpy().tobytes()
                line["cpu_baseline"] = cpu_baseline(flat[:sample].cpu(), sbody)
        print(json.dumps(line), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
