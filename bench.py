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
                     restatement if that build is absent) timed on the host cores at T = min(nproc, 16) (the reference's
                     default) and at T = nproc, with nproc / CPU model / NUMA nodes / malloc environment on the line, and
                     the WHOLE GPU frame compared with the CPU frame of the same tensor (gpu_frame_equals_cpu_frame)
  host_path          zn_compress / zn_decompress with HOST buffers (what the INTEGRATION stub binds), 1 GiB, PCIe inside the time: GB/s with fresh and with
                     recycled result buffers, in the library's default mode and with zn_set_host_direct(7)
  plugin_gpt2        BASELINE.json configs[3] (N = 1): a real-size GPT-2 checkpoint (148 fp32 tensors, 498 MB, synthesised),
                     compressed once, then loaded onto cuda:0 through zipnn_safetensors() + safe_open and through
                     safetensors_io.load_file — seconds split into file read / H2D / decode — beside the reference's
                     per-tensor CPU decode of the same frames
  llama8b            BASELINE.json configs[4] on the same line at every N: the --workload llama8b run (strong scaling: the
                     chunk ranges of every tensor split over the ranks), value / ms / (N + C)/t fraction, and at N = 1 a sample
                     of its batched bodies compared with the CPU oracle's
  rccl_ranks         ranks that took part in an all_reduce over RCCL (N > 1 or under torch.distributed.run)

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
    return {"avg": sum(ms) / len(ms), "min": s[0], "median": s[len(s) // 2], "max": s[-1]}


def time_events(fn, steps):
    """K calls of fn(), each bracketed by events on the stream the library launches on (torch's current stream)."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def device_state_under_load(burst):
    """Shader clock / power of the busy GPU while `burst()` (asynchronous launches worth ~0.2 s) is in flight — outside every timed region.
    The boxes of a pool differ (the same library: 1.46-1.69 ms per step); this says whether a slow line is a slow box.  Read from the amdgpu
    hwmon files of every card of the node (the busy one is the one with the highest shader clock); None where they cannot be read."""
    import glob
    try:
        burst()
        time.sleep(0.06)
        best = _busiest_gpu_state(glob)
        if best is not None:
            time.sleep(0.10)                 # a second look 160 ms in: a box that throttles under sustained load shows it here
            late = _busiest_gpu_state(glob)
            if late is not None:
                best["sclk_MHz_late"], best["power_W_late"] = late["sclk_MHz"], late["power_W"]
        torch.cuda.synchronize()
        return best
    except Exception:       # noqa: BLE001 — a diagnostic, never a reason to fail the bench
        torch.cuda.synchronize()
        return None


def _busiest_gpu_state(glob):
    if True:
        best = None
        for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            def rd(name):
                try:
                    with open(os.path.join(hw, name)) as f:
                        return int(f.read().strip())
                except (OSError, ValueError):
                    return None
            sclk = rd("freq1_input")
            if sclk is not None and (best is None or sclk > best["sclk_MHz"] * 1e6):
                p, cap, mclk, t = rd("power1_input"), rd("power1_cap"), rd("freq2_input"), rd("temp2_input")
                best = {"sclk_MHz": round(sclk / 1e6), "mclk_MHz": None if mclk is None else round(mclk / 1e6), "power_W": None if p is None else round(p / 1e6),
                        "power_cap_W": None if cap is None else round(cap / 1e6), "temp_C": None if t is None else round(t / 1e3)}
        return best


def rank_spread(td, device, my_ms, world):
    """Every rank's wall milliseconds per step, gathered: {min, max, all}."""
    mine = torch.tensor([my_ms], device=device, dtype=torch.float64)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    td.all_gather(allv, mine)
    v = [round(float(t.item()), 4) for t in allv]
    return {"min": min(v), "max": max(v), "all": v}


def size_sweep(lib, codec, device, steps):
    """Decode GB/s by tensor size (bf16, device-resident, event-timed): small tensors are bound by the ~100 us a workgroup needs per chunk
    group and by the launch, not by HBM."""
    out = {}
    # (64: the 16-wave small-input kernel, 128: its 8-wave form, from 256 on the fused kernel; 576 MiB = 2 304 chunks lies BETWEEN two rounds of workgroups: the
    #  size class the round-counting group rule of zn_decode_fused_group is for; "100MiB+250000" is a RAGGED tensor — 400 chunks and a partial one,
    #  what every real tensor whose size is not a multiple of 256 KiB looks like — next to the same tensor without its tail)
    for mib, extra in ((64, 0), (100, 0), (100, 250000), (128, 0), (256, 0), (576, 0), (1024, 0)):
        n = (mib << 20) + extra
        x = make_tensor(n, device, 99 + mib)
        flat = codec.flat_bytes(x)
        body = codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR).clone()
        dst = torch.empty(n, dtype=torch.uint8, device=device)
        for _ in range(8):
            codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n, out=dst, check=False)
        d = stats(time_events(lambda: codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n, out=dst, check=False), steps))
        c = stats(time_events(lambda: codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR), max(2, steps // 2)))
        torch.cuda.synchronize()
        out[f"{mib}MiB" + (f"+{extra}" if extra else "")] = {"decompress_GBps": round(n / d["avg"] / 1e6, 1), "decompress_ms": round(d["avg"], 4), "decompress_ms_median": round(d["median"], 4),
                            "compress_GBps": round(n / c["avg"] / 1e6, 1), "compress_ms": round(c["avg"], 4), "compress_ms_median": round(c["median"], 4),
                            "compress_ms_max": round(c["max"], 4), "exact": bool(torch.equal(dst, flat))}
        del x, flat, body, dst
    torch.cuda.empty_cache()
    return out


def host_path(lib, device, n_bytes=1 << 30):
    """The host-buffer entry points — zn_compress / zn_decompress through raw ctypes: what the INTEGRATION stub (tests/ref_binding/zipnn_core.py) and
    ZipNN().compress(bytes) call; reference zipnn/zipnn.py:714-725, 1143-1151 — on 1 GiB of bf16, PCIe both ways inside the time: GB/s = uncompressed
    bytes / wall seconds, best of 3.  `fresh` = the result buffer is allocated (np.empty) inside the timed region, what a caller that does not recycle
    buffers pays; `warm` = result buffers recycled.  Two modes of the library: the default (staged through its own pinned buffers, result buffers hinted to
    huge pages) and zn_set_host_direct(7) (the caller's buffers pinned, DMA straight between them and HBM — for callers that recycle buffers:
    profiles/r06_host_path.txt says why it is opt-in).  Never part of `value`."""
    import ctypes
    import numpy as np
    L = lib._L
    try:
        n = n_bytes // CHUNK * CHUNK
        x = make_tensor(n, device, 31337).view(torch.uint8).cpu().numpy()
        hdr = np.zeros(32, dtype=np.uint8)
        cap = L.zn_compress_bound(n, P, CHUNK, 32)
        sz = ctypes.c_size_t(0)
        dev = device.index or 0

        def comp(o):
            rc = L.zn_compress(hdr.ctypes.data, 32, x.ctypes.data, n, P, ROT, BMODE, CHUNK, ctypes.c_float(THR), dev, o.ctypes.data, cap, ctypes.byref(sz))
            assert rc == 0, rc
            return sz.value

        def dec(fr, flen, o):
            rc = L.zn_decompress(fr.ctypes.data + 32, flen - 32, P, ROT, BMODE, CHUNK, n, dev, o.ctypes.data)
            assert rc == 0, rc
        frame = np.empty(cap, dtype=np.uint8); flen = comp(frame)
        back = np.empty(n, dtype=np.uint8); dec(frame, flen, back)
        out = {"GiB": n / (1 << 30), "exact": bool(np.array_equal(back, x)), "unit": "GB/s, uncompressed bytes / wall seconds, PCIe inside; best of 3"}
        for mode, tag in ((4, "default"), (7, "direct")):
            lib.set_host_direct(mode)
            r = {}
            for name in ("compress", "decompress"):
                warm = fresh = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    comp(frame) if name == "compress" else dec(frame, flen, back)
                    warm = min(warm, time.perf_counter() - t0)
                    t0 = time.perf_counter()
                    o = np.empty(cap if name == "compress" else n, dtype=np.uint8)
                    comp(o) if name == "compress" else dec(frame, flen, o)
                    fresh = min(fresh, time.perf_counter() - t0)
                    del o
                r[name] = {"warm_GBps": round(n / warm / 1e9, 1), "warm_ms": round(warm * 1e3, 1), "fresh_GBps": round(n / fresh / 1e9, 1), "fresh_ms": round(fresh * 1e3, 1)}
            out[tag] = r
        lib.set_host_direct(4)
        # results from the library's pinned arena (zn_host_alloc: what _capi.ZnLib.compress / decompress and the INTEGRATION stub hand out from 8 MiB up):
        # a FRESH block per call — allocated, filled and released inside the timed region — next to the same with a fresh pageable np.empty
        r = {}
        for name in ("compress", "decompress"):
            ta = tp = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                o = lib.host_buffer(cap if name == "compress" else n)
                comp(o) if name == "compress" else dec(frame, flen, o)
                del o
                ta = min(ta, time.perf_counter() - t0)
                t0 = time.perf_counter()
                o = np.empty(cap if name == "compress" else n, dtype=np.uint8)
                comp(o) if name == "compress" else dec(frame, flen, o)
                del o
                tp = min(tp, time.perf_counter() - t0)
            r[name] = {"arena_GBps": round(n / ta / 1e9, 1), "arena_ms": round(ta * 1e3, 1), "pageable_GBps": round(n / tp / 1e9, 1), "pageable_ms": round(tp * 1e3, 1)}
        out["fresh_result_allocated_and_released_inside"] = r
        o = lib.host_buffer(n); dec(frame, flen, o)
        out["exact"] = out["exact"] and bool(np.array_equal(back, x)) and bool(np.array_equal(o, x))
        del o
        return out
    except Exception as e:                                 # (never let the optional leg take the bench line down)
        try:
            lib.set_host_direct(4)
        except Exception:
            pass
        return {"error": repr(e)[:300]}


def multi_dev_inprocess(lib, codec, n_bytes=1 << 30):
    """device_count() > 1 only: ONE process driving every visible GPU through the library's own fan-out (zn_decompress_multi_dev: one host
    thread, stream and pinned pipe per device; the frame is a host buffer, the tensor ends up resident, chunk range i on device i).
    PCIe-bound by construction (the compressed body crosses it), reported beside the per-rank numbers; never part of `value`."""
    nd = torch.cuda.device_count()
    if nd < 2:
        return None
    try:
        dev0 = torch.device("cuda", 0)
        x = make_tensor(n_bytes, dev0, 777)
        flat = codec.flat_bytes(x)
        body = codec.compress_device(lib, flat, P, ROT, BMODE, CHUNK, THR).cpu().numpy().tobytes()
        devices = list(range(nd))
        parts = []
        for i in devices:
            off, ln = lib.multi_range(n_bytes, CHUNK, nd, i)
            parts.append(torch.empty(max(ln, 16), dtype=torch.uint8, device=torch.device("cuda", i)))
        for i in devices:
            torch.cuda.synchronize(i)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            lib.decompress_multi_dev(body, P, ROT, BMODE, CHUNK, n_bytes, devices, [t.data_ptr() for t in parts])
            for i in devices:
                torch.cuda.synchronize(i)
            best = min(best, time.perf_counter() - t0)
        ok = True
        for i in devices:
            off, ln = lib.multi_range(n_bytes, CHUNK, nd, i)
            ok = ok and bool(torch.equal(parts[i][:ln].cpu(), flat[off:off + ln].cpu()))
        return {"devices": nd, "GiB": n_bytes / (1 << 30), "seconds": round(best, 4), "GBps": round(n_bytes / best / 1e9, 2), "exact": ok,
                "what": "zn_decompress_multi_dev from a host frame, chunk range i resident on device i (PCIe-inclusive)"}
    except Exception as e:                                 # (never let the optional leg take the bench line down)
        return {"devices": nd, "error": repr(e)[:300]}


def multi_dev_isolated(timeout_s=240):
    """multi_dev_inprocess in a CHILD process with a time limit.  The fan-out over several devices in one process has never run on more than one real GPU (the builder
    box has one): on the first box that has several, whatever it does — an exception, a crash, a hang — must not take the bench line with it."""
    import subprocess
    nd = torch.cuda.device_count()
    if nd < 2:
        return None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--multi-dev-leg"], capture_output=True, text=True, timeout=timeout_s)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{") or ln == "null":
                return json.loads(ln)
        return {"devices": nd, "error": f"child rc={r.returncode}: " + (r.stderr or "")[-300:]}
    except subprocess.TimeoutExpired:
        return {"devices": nd, "error": f"child process timed out after {timeout_s} s and was killed (the rest of this line was measured before it)"}
    except Exception as e:       # noqa: BLE001
        return {"devices": nd, "error": repr(e)[:300]}


def host_info():
    """What BASELINE.md §3 wants next to every CPU number."""
    info = {"nproc": os.cpu_count() or 1, "cpu_model": None, "numa_nodes": None,
            "malloc_env": {k: v for k, v in os.environ.items() if k.startswith("MALLOC_") or k in ("LD_PRELOAD",)}}
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                info["cpu_model"] = ln.split(":", 1)[1].strip(); break
    except OSError:
        pass
    try:
        info["numa_nodes"] = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except OSError:
        pass
    return info


def cpu_baseline(raw, want_body):
    """Time the CPU reference on the host cores (rank 0, N = 1 only) and compare its frame with the GPU's.
    Test infrastructure is used here strictly as the thing being compared AGAINST."""
    import hashlib
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    hi = host_info()
    nproc = hi["nproc"]
    t_ref = min(nproc, 16)                            # the reference default (zipnn/zipnn.py:176-177)
    hdr = bytes(32)
    kind = "reference" if O.ref_core() is not None else "port"

    def run(buf_np, threads, reps):
        best_c = best_d = 1e9
        frame = back = None
        for _ in range(reps):
            if kind == "reference":
                buf = bytearray(buf_np.tobytes())     # the reference rotates its input in place
                t0 = time.perf_counter()
                frame = O.ref_core().zipnn_core(bytearray(hdr), buf, P, ROT, BMODE, 0, CHUNK, THR, 10, threads)
                best_c = min(best_c, time.perf_counter() - t0)
                del buf
                t0 = time.perf_counter()
                back = O.ref_core().combine_dtype(memoryview(frame)[32:], P, ROT, BMODE, CHUNK, buf_np.size, threads)
                best_d = min(best_d, time.perf_counter() - t0)
            else:
                t0 = time.perf_counter()
                frame = O.compress_frame(hdr, buf_np, P, ROT, BMODE, CHUNK, THR, threads)
                best_c = min(best_c, time.perf_counter() - t0)
                t0 = time.perf_counter()
                back = O.decompress_body(frame[32:], P, ROT, BMODE, CHUNK, buf_np.size, threads)
                best_d = min(best_d, time.perf_counter() - t0)
        return best_c, best_d, frame, back
    reps = 2 if raw.size > (1 << 30) else 3
    best_c, best_d, frame, back = run(raw, t_ref, reps)
    assert np.array_equal(np.frombuffer(back, dtype=np.uint8), raw)
    fb = np.frombuffer(memoryview(frame)[32:], dtype=np.uint8)
    parity = bool(fb.size == want_body.size and np.array_equal(fb, want_body)) if want_body is not None else None
    gb = raw.size / 1e9
    out = {"value": round(gb / best_d, 3), "unit": "GB/s", "cores": t_ref, "kind": kind,
           "huff0": "libzstd 1.4.8 substitute: the reference's FiniteStateEntropy submodule is un-vendored, its csrc/ is compiled against zstd 1.4.8's huff0 "
                    "(same published algorithm; PyPI wheels write the tree description's low-probability counts as -1, this build as +1; both decode both)",
           "compress_GBps": round(gb / best_c, 3),
           "sample": f"the first {raw.size >> 20} MiB of the same tensor (all of it when that is its size), decompress best of {reps}, {t_ref} threads",
           "gpu_frame_equals_cpu_frame": parity, "frame_compare_bytes": int(raw.size),
           "frame_sha256": hashlib.sha256(memoryview(frame)).hexdigest(), "ratio": round(len(frame) / raw.size, 5)}
    out.update(hi)
    if nproc != t_ref:                                # T = nproc beside the reference default (1 GiB sample, best of 2)
        sub = raw[: min(raw.size, 1 << 30) // CHUNK * CHUNK]
        tn = nproc if kind == "reference" else min(nproc, 64)        # (the port caps its pthreads at 64; the reference's core does not)
        c2, d2, _, _ = run(sub, tn, 2)
        out["all_cores"] = {"cores": tn, "value": round(sub.size / 1e9 / d2, 3), "compress_GBps": round(sub.size / 1e9 / c2, 3),
                            "sample": f"the first {sub.size >> 20} MiB, best of 2, T = nproc"}
    return out


def cpu_frame_compare(flat, body, nb, rot, bm, chunk):
    """The GPU's body of a tensor against the CPU reference's frame of the same bytes (whole buffers, sha256 of each)."""
    import hashlib
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    try:
        raw = flat.cpu().numpy()
        got = body.cpu().numpy()
        threads = min(os.cpu_count() or 1, 16)
        if O.ref_core() is not None:
            kind = "reference"
            frame = O.ref_core().zipnn_core(bytearray(32), bytearray(raw.tobytes()), nb, rot, bm, 0, chunk, THR, 10, threads)     # (a copy: the core rotates its input in place)
            back = O.ref_core().combine_dtype(memoryview(got), nb, rot, bm, chunk, raw.size, threads)                             # … and the reference decodes the GPU's body
        else:
            kind = "port"
            frame = O.compress_frame(bytes(32), raw, nb, rot, bm, chunk, THR, threads)
            back = O.decompress_body(got, nb, rot, bm, chunk, raw.size, threads)
        want = np.frombuffer(memoryview(frame)[32:], dtype=np.uint8)
        return {"kind": kind, "bytes": int(raw.size), "equal": bool(want.size == got.size and np.array_equal(want, got)),
                "cpu_decodes_gpu_body": bool(np.array_equal(np.frombuffer(back, dtype=np.uint8), raw)),
                "gpu_body_sha256": hashlib.sha256(memoryview(got)).hexdigest(), "cpu_body_sha256": hashlib.sha256(memoryview(want)).hexdigest()}
    except Exception as e:                                 # noqa: BLE001 — a missing checker is reported, not fatal
        return {"error": repr(e)[:200]}


def csrc_digest():
    """sha256 over the kernel sources (zipnn_amd/csrc/*, sorted by name): what profiles/traffic_pmc.json was measured on must be what runs."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "zipnn_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def traffic_of(kind, n_bytes):
    """HBM bytes per decode launch of `kind` from the committed PMC pass (profiles/traffic_pmc.json: FETCH_SIZE x 2 on gfx950 +
    WRITE_SIZE, per GiB; scripts/pmc_traffic.py writes it from a rocprofv3 --pmc run of the kernels named there), scaled to this
    size; the source file and the digest of the kernel sources it was measured on travel with the number.  A record measured on
    OTHER kernel sources than the ones in the tree is refused: traffic = null and the reason on the line (there is no .git on the
    GPU box, so the check is a digest of zipnn_amd/csrc, not a commit)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_pmc.json")) as f:
            t = json.load(f)
        e = t["decode"][kind]
        src = {"file": "profiles/traffic_pmc.json", "from": e.get("from"), "kernels_commit": t.get("kernels_commit"), "csrc_sha256_16": t.get("csrc_sha256_16")}
        now = csrc_digest()
        if t.get("csrc_sha256_16") != now:
            src["stale"] = f"measured on kernel sources {t.get('csrc_sha256_16')}, the tree has {now}: re-run scripts/gpu_pmc_dtypes.sh + scripts/pmc_traffic.py"
            return None, src
        return int(e["hbm_bytes_per_gib_launch"] * (n_bytes / (1 << 30))), src
    except Exception:
        return None, None


def other_dtypes(lib, codec, device, steps, cpu_compare=True):
    """BASELINE.json configs[2] (+ fp8): 1 GiB each, event-timed; the bytes of record for these dtypes on the driver's run.
    cpu_compare: after the timed regions the WHOLE 1 GiB body is compared with the frame the CPU reference writes for the same tensor
    (oracle/_ref = the reference's C core, or the C restatement where that build is absent) — sha256 of both on the line."""
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
        tr, _ = traffic_of({"fp8_e4m3": "fp8"}.get(name, name), n)
        cmp_ = cpu_frame_compare(flat, body[:used], nb, rot, bm, chunk) if cpu_compare else None      # (outside every timed region; the checker, never the thing measured)
        out[name] = {"GiB": 1.0, "ratio": round((used + 32) / n, 5), "exact": exact, "body_vs_cpu_reference": cmp_, "decompress_traffic": tr, "algorithmic_bytes": int(n + cpl),
                     "decompress_GBps": round(n / d["avg"] / 1e6, 1), "decompress_ms": round(d["avg"], 4), "decompress_ms_min": round(d["min"], 4),
                     "decompress_roofline_frac": round((n + cpl) / (d["avg"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "compress_GBps": round(n / c["avg"] / 1e6, 1), "compress_ms": round(c["avg"], 4), "compress_ms_min": round(c["min"], 4),
                     "compress_roofline_frac": round((n + cpl) / (c["avg"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        del x, flat, body, dst
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------
# plugin_gpt2 (BASELINE.json configs[3]; reference zipnn/zipnn.py:1584-1643, scripts/zipnn_compress_safetensors.py:74-123)
# ---------------------------------------------------------------------------------------------------------------
def gpt2_state(device, layers=12, width=768, vocab=50257, npos=1024, seed=4242):
    """The 148 tensors of `transformers.GPT2LMHeadModel(GPT2Config())` (497.8 MB fp32), initialised the way GPT-2 is:
    N(0, 0.02) weights, zero biases, unit LayerNorm gains (no network for the real checkpoint)."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    n = lambda *sh: torch.randn(*sh, generator=g, device=device) * 0.02       # noqa: E731
    sd = {"transformer.wte.weight": n(vocab, width), "transformer.wpe.weight": n(npos, width)}
    for i in range(layers):
        p = f"transformer.h.{i}."
        sd.update({p + "ln_1.weight": torch.ones(width, device=device), p + "ln_1.bias": torch.zeros(width, device=device),
                   p + "attn.c_attn.weight": n(width, 3 * width), p + "attn.c_attn.bias": torch.zeros(3 * width, device=device),
                   p + "attn.c_proj.weight": n(width, width), p + "attn.c_proj.bias": torch.zeros(width, device=device),
                   p + "ln_2.weight": torch.ones(width, device=device), p + "ln_2.bias": torch.zeros(width, device=device),
                   p + "mlp.c_fc.weight": n(width, 4 * width), p + "mlp.c_fc.bias": torch.zeros(4 * width, device=device),
                   p + "mlp.c_proj.weight": n(4 * width, width), p + "mlp.c_proj.bias": torch.zeros(width, device=device)})
    sd.update({"transformer.ln_f.weight": torch.ones(width, device=device), "transformer.ln_f.bias": torch.zeros(width, device=device)})
    return sd


def plugin_gpt2(lib, device):
    """Load-time decompress of a GPT-2 `.znn.safetensors` on one GPU, end to end from the file: the plugin path
    (zipnn_safetensors() + safe_open(device=...) + get_tensor per tensor) and the batched path (safetensors_io.load_file),
    each best of 3 with the file in the page cache; beside them the reference's own per-tensor decode on the host cores
    (oracle/_ref = the reference C core, called once per compressed tensor as its plugin does)."""
    import shutil
    import tempfile
    import safetensors
    import safetensors.torch
    from safetensors.torch import save_file
    from zipnn_amd import safetensors_io, zipnn_safetensors
    from zipnn_amd import zipnn as _Z
    tmp = tempfile.mkdtemp(prefix="zn_gpt2_")
    res = {}
    try:
        sd = gpt2_state(device)
        raw_bytes = sum(v.numel() * v.element_size() for v in sd.values())
        src = os.path.join(tmp, "gpt2.safetensors")
        save_file({k: v.cpu() for k, v in sd.items()}, src, {"format": "pt"})
        t0 = time.perf_counter()
        znn = safetensors_io.compress_safetensors_file(src, device=str(device))        # staged in HBM, one batched compress
        torch.cuda.synchronize()
        res["compress_file_s"] = round(time.perf_counter() - t0, 4)
        res.update(tensors=len(sd), raw_bytes=int(raw_bytes), file_bytes=os.path.getsize(znn), ratio=round(os.path.getsize(znn) / raw_bytes, 5))

        def check(loaded):
            return all(bool(torch.equal(loaded[k], sd[k])) and loaded[k].is_cuda for k in sd) and set(loaded) == set(sd)
        best, split, exact = 1e9, None, True
        for _ in range(3):
            tm = {}
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loaded = safetensors_io.load_file(znn, device=str(device), timings=tm)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            exact = exact and check(loaded)
            if dt < best:
                best, split = dt, tm
            del loaded
        res["load_file"] = {"seconds": round(best, 4), "GBps": round(raw_bytes / best / 1e9, 2), "file_read_s": round(split["read_s"], 4),
                            "h2d_s": round(split["h2d_s"], 4), "decode_s": round(split["decode_s"], 4),
                            "decode_GBps": round(split["decoded_bytes"] / max(split["decode_s"], 1e-9) / 1e9, 1),
                            "compressed_tensors": split["compressed_tensors"], "kernels": lib.last_kernels(), "bit_exact": exact}
        orig_a, orig_b = safetensors.torch.safe_open, safetensors.safe_open
        try:
            zipnn_safetensors()
            best, exact = 1e9, True
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                loaded = {}
                with safetensors.safe_open(znn, framework="pt", device=str(device)) as f:
                    for k in f.keys():
                        loaded[k] = f.get_tensor(k)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
                exact = exact and check(loaded)
                del loaded
            res["plugin_safe_open"] = {"seconds": round(best, 4), "GBps": round(raw_bytes / best / 1e9, 2), "bit_exact": exact,
                                       "note": "zipnn_safetensors() + safetensors.safe_open(device=cuda) + get_tensor per tensor — served by the read-ahead: one transfer of "
                                               "the data section and ONE batched decode of every compressed tensor at the first compressed name (SafeOpen._read_ahead)"}
        finally:
            safetensors.torch.safe_open, safetensors.safe_open = orig_a, orig_b
            _Z._patches_applied.pop(_Z._zipnn_safetensors, None)
        # the reference's plugin on the host cores: its C core, once per compressed tensor (zipnn.py:1592-1626 -> :1143)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        if O.ref_core() is not None:
            from zipnn_amd.zipnn import ZipNN, COMPRESSED_DTYPE, COMPRESSION_METHOD, get_compressed_tensors_metadata
            threads = min(os.cpu_count() or 1, 16)
            with safetensors.safe_open(znn, "pt", "cpu") as f:
                infos = get_compressed_tensors_metadata(dict(f.metadata() or {}))
                frames = []
                for k in f.keys():
                    if k in infos:
                        t = f.get_tensor(k)
                        fp = ZipNN(input_format="torch", bytearray_dtype=COMPRESSED_DTYPE, method=COMPRESSION_METHOD).frame_params(t)
                        frames.append((k, t.numpy().tobytes()[fp["body_off"]:], fp))
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                ok = True
                for k, body, fp in frames:
                    back = O.ref_core().combine_dtype(body, fp["num_buf"], fp["bits_mode"], fp["bytes_mode"], fp["chunk"], fp["orig_size"], threads)
                    ok = ok and len(back) == fp["orig_size"]
                best = min(best, time.perf_counter() - t0)
            dec = sum(fp["orig_size"] for _, _, fp in frames)
            res["cpu_reference_core"] = {"seconds": round(best, 4), "GBps": round(dec / best / 1e9, 2), "threads": threads, "tensors": len(frames),
                                         "what": "oracle/_ref (the reference's C core + libzstd 1.4.8 huff0), combine_dtype once per compressed tensor as its plugin "
                                                 "calls it, frames already in host memory; the reference's Python package is not on this box, its per-tensor Python "
                                                 "overhead is not in this number (BASELINE.md §2 measured the whole plugin at 0.46 s on 8 vCPUs)"}
        else:
            res["cpu_reference_core"] = None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res


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


def llama8b_partition(world, layers=32):
    """What every rank of a `world`-rank llama8b run gets, computed on paper (no device): chunks and bytes per rank, max / min — the load balance of
    zipnn_amd.sharding.chunk_ranges over the 291 + 224 tensors (known before the first 8-GPU run, and on its line)."""
    from zipnn_amd import sharding
    f8 = getattr(torch, "float8_e4m3fn", None)
    chunks, nbytes_r = [0] * world, [0] * world
    for name, shape, linear in llama8b_shapes(layers=layers):
        numel = 1
        for d in shape:
            numel *= d
        for es, chunk in [(2, CHUNK)] + ([(1, CHUNK // 2)] if (linear and f8 is not None) else []):
            nb = numel * es
            K = (nb + chunk - 1) // chunk
            for r, (lo, hi) in enumerate(sharding.chunk_ranges(K, world)):
                if hi > lo:
                    chunks[r] += hi - lo
                    nbytes_r[r] += min(hi * chunk, nb) - lo * chunk
    return {"ranks": world, "chunks_per_rank_min": min(chunks), "chunks_per_rank_max": max(chunks), "bytes_per_rank_min": min(nbytes_r), "bytes_per_rank_max": max(nbytes_r),
            "imbalance": round(max(nbytes_r) / max(1, sum(nbytes_r) / world), 5)}


def pin_rank_to_its_gpus_numa_node(dev_index):
    """N > 1: this rank's host threads onto the CPUs of the NUMA node its GPU hangs off (/sys/bus/pci/devices/<bdf>/local_cpulist) — the pinned staging buffers,
    the launch thread and the RCCL proxy then sit next to the device instead of wherever the launcher left them.  Best effort: returns what it did, never raises."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        dom, bus, dv = getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", 0)
        if bus is None:
            return {"pinned": False, "why": "no pci_bus_id on this torch"}
        bdf = f"{dom:04x}:{bus:02x}:{dv:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        with open(base + "/local_cpulist") as f:
            txt = f.read().strip()
        cpus = set()
        for part in txt.split(","):
            if "-" in part:
                a, b = part.split("-"); cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        node = None
        try:
            node = int(open(base + "/numa_node").read().strip())
        except (OSError, ValueError):
            pass
        allowed = cpus & set(os.sched_getaffinity(0))
        if not allowed:
            return {"pinned": False, "pci": bdf, "numa_node": node, "why": "no local CPU in this process's affinity mask"}
        os.sched_setaffinity(0, allowed)
        return {"pinned": True, "pci": bdf, "numa_node": node, "cpus": len(allowed)}
    except Exception as e:       # noqa: BLE001
        return {"pinned": False, "why": repr(e)[:120]}


def run_llama8b(args, lib, codec, device, world, rank, dist, td, sample_oracle=False):
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
    rank_ms = None
    if dist:
        rank_ms = rank_spread(td, device, elapsed / args.steps * 1e3, world)
        tt = torch.tensor([elapsed, c_elapsed, float(not exact)], device=device, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed, c_elapsed, bad = tt.tolist()
        exact = bad == 0.0
        cb = torch.tensor([float(c_bytes)], device=device, dtype=torch.float64)
        td.all_reduce(cb)
        c_bytes = int(cb.item())
    line = None
    if rank == 0:
        d = stats(dms)
        c_payload = c_bytes - sum(9 * nb * ((f.numel() + chunk - 1) // chunk) for (f, nb, rot, bm, chunk, _) in items) if world == 1 else None
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
                "rank0_decode_ms": {k: round(v, 4) for k, v in d.items()}, "rank_ms_per_step": rank_ms,
                "partition": llama8b_partition(world, args.layers), "partition_at_8_ranks": llama8b_partition(8, args.layers),
                "kernels": {"decompress": decode_kernels, "compress": encode_kernels}}
        if c_payload is not None:                     # (N + C) / t on rank 0's event-timed launches, as a fraction of 8 TB/s
            line["decompress_roofline_frac"] = round((total_bytes + c_payload) / (d["avg"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        if world == 1 and sample_oracle:
            # the batched bodies against the CPU oracle's frames of the same tensors: the FIRST tensor of every (size, dtype) class of the
            # model — embedding, q/o, k/v, gate/up, down, a norm vector, as bf16 and (the linears) as fp8 — and the last tensor (lm_head)
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            seen, pick = set(), []
            for i, it in enumerate(items):
                key = (it[0].numel(), it[1])
                if key not in seen:
                    seen.add(key); pick.append(i)
            if len(items) - 1 not in pick:
                pick.append(len(items) - 1)
            same, nbytes, bad = True, 0, []
            for i in pick:
                f, nb, rot, bm, chunk, th = items[i]
                want = O.compress_frame(b"", f.cpu().numpy(), nb, rot, bm, chunk, th, threads=min(os.cpu_count() or 1, 16))
                ok_i = bodies[i].cpu().numpy().tobytes() == want
                same = same and ok_i
                if not ok_i:
                    bad.append(i)
                nbytes += f.numel()
            line["bodies_equal_oracle"] = {"tensors": len(pick), "classes": len(seen), "bytes": int(nbytes), "equal": same, "differing_items": bad}
    return line


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
    ap.add_argument("--no-plugin", action="store_true", help="skip plugin_gpt2 (BASELINE.json configs[3])")
    ap.add_argument("--no-llama8b", action="store_true", help="skip the llama8b sub-run on the default line (BASELINE.json configs[4])")
    ap.add_argument("--multi-dev-leg", action="store_true", help="(internal) run only the in-process multi-device leg and print its JSON: the child of multi_dev_isolated()")
    ap.add_argument("--cpu-sample-mib", type=int, default=4096, help="bytes of the tensor the CPU reference is timed on and the GPU frame is compared on")
    args = ap.parse_args()
    if args.multi_dev_leg:                       # the child of multi_dev_isolated(): its own process, its own contexts on every device, one JSON value on stdout
        from zipnn_amd import _capi, codec
        print(json.dumps(multi_dev_inprocess(_capi.lib(), codec)), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1 or ("LOCAL_RANK" in os.environ and "MASTER_ADDR" in os.environ)   # (under torch.distributed.run even a single rank takes the RCCL path)
    # ZN_BENCH_SHARE_GPU=1 (+ ZN_BENCH_BACKEND=gloo): a DRY RUN of the N > 1 logic — barriers, max over ranks, per-rank times, the llama8b partition,
    # rank 0's line — on a box with fewer GPUs than ranks (scripts/multi_gpu_selftest.sh on the one-GPU builder box).  The ranks share devices, so
    # the numbers of such a line mean nothing; it says "dry_run_shared_gpu": true.
    share = os.environ.get("ZN_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % max(1, torch.cuda.device_count()) if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    affinity = pin_rank_to_its_gpus_numa_node(dev_index) if (world > 1 and not share) else None      # (N = 1: the launcher's mask stays — the CPU baseline wants every core)
    td = None
    if dist:
        import torch.distributed as td
        backend = os.environ.get("ZN_BENCH_BACKEND", "nccl") if share else "nccl"
        td.init_process_group(backend=backend, **({"device_id": device} if backend == "nccl" else {}))

    from zipnn_amd import _capi, codec
    lib = _capi.lib()

    rccl_ranks = None
    if dist:                                      # how many ranks are really there, over the collective library itself
        ones = torch.ones(1, device=device, dtype=torch.float64)
        td.all_reduce(ones)
        rccl_ranks = int(ones.item())

    if args.workload == "llama8b":
        line = run_llama8b(args, lib, codec, device, world, rank, dist, td, sample_oracle=(world == 1))
        if rank == 0:
            line["rccl_ranks"] = rccl_ranks
            print(json.dumps(line), flush=True)
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

    rank_ms = None
    if dist:
        rank_ms = rank_spread(td, device, elapsed / args.steps * 1e3, world)
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
        traffic, traffic_src = traffic_of("bf16", n_bytes)
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
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": decode_kernels, "avg_launch_ms": round(d["avg"], 4),
                         "min_launch_ms": round(d["min"], 4), "median_launch_ms": round(d["median"], 4),
                         "algorithmic_bytes": alg},
            "compress_roofline": {"bound": "hbm", "achieved": round(c_achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(c_achieved / HBM_PEAK_GBPS, 4), "kernel": encode_kernels,
                                  "avg_call_ms": round(c["avg"], 4), "min_call_ms": round(c["min"], 4), "median_call_ms": round(c["median"], 4),
                                  "algorithmic_bytes": alg,
                                  "note": "events bracket the whole zn_compress_dev call — the one-pass encoder + the size scan for a bf16 call of this size (zn_k_encode_onepass: N + C over HBM, the chunk's second read aimed at the Infinity Cache), the four-kernel encoder (2 N + C) otherwise: see `kernel` — + one read-back of length and status"},
            "kernels": {"decompress": decode_kernels, "compress": encode_kernels},
        }
        if share:
            line["dry_run_shared_gpu"] = True
        line["device_under_load"] = device_state_under_load(lambda: [codec.decompress_device(lib, body, P, ROT, BMODE, CHUNK, n_bytes, out=out, check=False) for _ in range(120)])
        line["rccl_ranks"] = rccl_ranks
        line["rank0_cpu_affinity"] = affinity
        line["rank_ms_per_step"] = rank_ms          # every rank's own wall time per step (a straggler shows as max >> min); null without torch.distributed
    # ---- BASELINE.json configs[4] on the same line, at every N (every rank takes part; strong scaling) ----
    del x, flat, body_buf, out
    torch.cuda.empty_cache()
    if not args.no_llama8b:
        import copy
        a2 = copy.copy(args); a2.steps = max(2, min(args.steps, 10)); a2.warmup = min(args.warmup, 3)
        llama = run_llama8b(a2, lib, codec, device, world, rank, dist, td, sample_oracle=(world == 1))
        torch.cuda.empty_cache()
        if rank == 0:
            line["llama8b"] = llama
    if rank == 0:
        if world == 1 and not args.no_other_dtypes:
            line["other_dtypes"] = other_dtypes(lib, codec, device, max(4, min(args.steps, 20)), cpu_compare=not args.no_cpu_baseline)
        if world == 1 and not args.no_other_dtypes:
            line["sizes"] = size_sweep(lib, codec, device, max(4, min(args.steps, 20)))
        if world == 1 and not args.no_plugin:
            line["plugin_gpt2"] = plugin_gpt2(lib, device)
        if world == 1 and not args.no_plugin:
            line["host_path"] = host_path(lib, device)
        if world == 1 and not args.no_cpu_baseline:
            sample = min(args.cpu_sample_mib << 20, n_bytes) // CHUNK * CHUNK
            if sample:
                xs = make_tensor(sample, device, 1234 + 1000 * rank)       # (the same seeds: the first `sample` bytes of the timed tensor)
                fs = codec.flat_bytes(xs)
                sbody = body if sample == n_bytes else codec.compress_device(lib, fs, P, ROT, BMODE, CHUNK, THR)
                line["cpu_baseline"] = cpu_baseline(fs.cpu().numpy(), sbody.cpu().numpy())
                del xs, fs, sbody
        if world == 1:                           # (last, and in a child process with a time limit: first contact with several devices in one process)
            del body
            torch.cuda.empty_cache()
            line["multi_dev_inprocess"] = multi_dev_isolated()
        print(json.dumps(line), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
