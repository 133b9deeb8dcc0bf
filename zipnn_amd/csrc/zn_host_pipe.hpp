// zn_host_pipe.hpp — pageable host memory <-> HBM for the host-buffer entry points (zn_compress / zn_decompress).
//
// A plain hipMemcpy of pageable memory is staged by the runtime on one thread; the result buffer of a call is freshly
// allocated, so its pages are first touched during the copy as well (1 GiB: ~150 ms, 7 GB/s end to end).  Here the
// transfer is cut into slices that go through two pinned bounce buffers: worker threads move slice i+1 between the
// caller's buffer and a bounce buffer (page faults spread over the threads) while the DMA engine moves slice i.
// Host-side plumbing only — no data is transformed; the reference has no counterpart (its buffers never leave the host).
#pragma once

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>
#include <sys/mman.h>

struct ZnHostPipe {
  void* pin[2] = {nullptr, nullptr};
  size_t slice = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
};

// ---- the DIRECT path (round 6): DMA straight between the caller's buffer and HBM ------------------------------
// Measured on the MI355X boxes (scripts/ubench/hostreg.hip, profiles/r06_host_path.txt), 1 GiB: the DMA itself takes 18.8 ms either way
// (57 GB/s); what a caller's pageable buffer costs on top depends on its PAGES.  4 KiB pages: first touch 36-90 ms (8 threads), hipHostRegister
// 20-47 ms, free() 70-110 ms — the staged copy above (≈ 20 ms + the faults) is the better deal.  2 MiB pages (transparent huge pages; this kernel
// runs THP in `madvise` mode): first touch 5.7 ms, register 2.7 ms, unregister ≈ 0.  So:
//   * a DESTINATION range gets madvise(MADV_HUGEPAGE) before it is first written — a hint about page size, nothing the caller can observe but
//     speed; fresh result buffers (np.empty, bytearray, malloc ≥ 128 KiB: their own mmap) then fault in 2 MiB at a time;
//   * big transfers go piece by piece (64 MiB): worker threads touch the piece, hipHostRegister pins it (timed), the DMA engine moves it while
//     the next piece is prepared; pieces are unpinned when the stream is idle (16 at a time at most, or by the caller: ZnPinList);
//   * a piece whose registration runs at 4 KiB-page speed (> 11 ms per GiB) ends the direct path: the rest of the buffer takes the staged copy.
// OPT-IN (zn_set_host_direct(7) / ZIPNN_AMD_HOST_DIRECT=7): see direct_mode() below for why; the huge-page hint alone is the default.
namespace zn_host_pipe_detail {
// zn_set_host_direct / ZIPNN_AMD_HOST_DIRECT: bit 0 = device-to-host transfers may go direct, bit 1 = host-to-device ones, bit 2 = the huge-page hint.
// DEFAULT 4: the hint only — pinning a caller's memory is OPT-IN (7).  Why, measured (profiles/r06_host_path.txt): in a long-lived process (Python + torch)
// that has pinned user memory at some point, calls that have to FAULT their result buffer in — the common pattern: a fresh np.empty / bytearray per call —
// run 55-60 ms per GiB instead of 35, the DMAs in flight crawling while the pages are faulted; a bare C++ process does not show it (scripts/ubench/hostreg4.hip:
// faults 6 ms per GiB with and without pinned memory), so it is a property of the process, not of the path, and not one the library can see.  With recycled
// (resident) buffers the direct pipeline moves 1 GiB in 23 ms each way (46 GB/s) against 31-36 ms staged.
inline std::atomic<int>& direct_mode_ref() {
  static std::atomic<int> m{[] { const char* e = getenv("ZIPNN_AMD_HOST_DIRECT"); return (e && e[0] >= '0' && e[0] <= '7') ? e[0] - '0' : 4; }()};
  return m;
}
inline int direct_mode() { return direct_mode_ref().load(std::memory_order_relaxed); }
inline bool direct_enabled() { return (direct_mode() & 3) != 0; }
inline bool thp_available() {
  static const int ok = [] {
    char b[128] = {0};
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    if (!f) return 0;
    const bool got = fgets(b, sizeof b - 1, f) != nullptr;
    fclose(f);
    return (got && (strstr(b, "[always]") || strstr(b, "[madvise]"))) ? 1 : 0;
  }();
  return ok != 0;
}
}  // namespace zn_host_pipe_detail
// Is [h, h + n) already backed by pages (a recycled, warm buffer — or one the caller has written before)?  64 samples with mincore().
// Only RESIDENT destinations are pinned (see the rule at the top of the direct path): measured, scripts/hp_seq2.py / profiles/r06_host_path.txt — a buffer that
// is pinned while fresh, then freed, leaves its address range in a state where the NEXT fresh mapping at that address faults in slowly and stalls every DMA
// of the process meanwhile (1 GiB: 35-50 ms instead of 23); buffers that were resident when they were pinned do not do that, and neither do fresh buffers that
// are never pinned (the staged copy faults them in from its worker threads, 2 MiB at a time after the hint above).
inline bool zn_host_resident(const void* h, size_t n) {
  if (n < 4096) return true;
  const uintptr_t lo = ((uintptr_t)h + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)h + n) & ~(uintptr_t)4095;
  if (hi <= lo) return true;
  const size_t pages = (hi - lo) >> 12;
  unsigned have = 0, asked = 0;
  for (unsigned i = 0; i < 64; i++) {
    const size_t pg = (size_t)((double)i / 64.0 * (double)pages);
    unsigned char v = 0;
    if (mincore((void*)(lo + (pg << 12)), 4096, &v) != 0) return false;      // (not a mapping mincore knows: leave it alone)
    asked++; have += v & 1u;
  }
  return have * 100u >= asked * 95u;
}
// huge pages, please, for the part of [h, h + n) that whole 2 MiB pages cover (a hint: errors are ignored — not an anonymous mapping, THP off)
inline void zn_host_thp_hint(void* h, size_t n) {
  using namespace zn_host_pipe_detail;
  if (!(direct_mode() & 4) || !thp_available() || n < ((size_t)8 << 20)) return;
  const uintptr_t lo = ((uintptr_t)h + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1), hi = ((uintptr_t)h + n) & ~(uintptr_t)(((size_t)2 << 20) - 1);
  if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
}

namespace zn_host_pipe_detail {

// reusable barrier for (workers + 1) participants
struct Barrier {
  std::mutex m; std::condition_variable cv; unsigned n, waiting = 0, gen = 0;
  explicit Barrier(unsigned n_) : n(n_) {}
  void resize(unsigned n_) { std::lock_guard<std::mutex> lk(m); n = n_; }     // only while nobody can complete a round
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned g = gen;
    if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

inline unsigned worker_count(size_t n) {
  unsigned t = 8;
  if (const char* e = getenv("ZN_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) t = (unsigned)v; }
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw && t > hw) t = hw;
  const size_t by_size = n / (4u << 20);        // at least 4 MiB of work per thread
  if (by_size < t) t = by_size ? (unsigned)by_size : 1u;
  return t;
}

// (round 6, measured and not kept: the workers' copies with non-temporal AVX2 stores, 16 / 32 threads, 64 MiB slices — the staged path moves 1 GiB in 33-37 ms each
//  way in every combination, profiles/r06_host_path.txt: one staged direction already runs at the DMA engine's pace, 18.8 ms per GiB, the one-shot call is their sum)
// the stripe of [0, len) that worker w of t copies (64-byte aligned cuts)
inline void stripe(size_t len, unsigned w, unsigned t, size_t* lo, size_t* hi) {
  const size_t per = ((len + t - 1) / t + 63) & ~(size_t)63;
  *lo = (size_t)w * per < len ? (size_t)w * per : len;
  *hi = *lo + per < len ? *lo + per : len;
}

}  // namespace zn_host_pipe_detail

inline void zn_host_pipe_release(ZnHostPipe& p) {
  for (int i = 0; i < 2; i++) {
    if (p.pin[i]) { (void)hipHostFree(p.pin[i]); p.pin[i] = nullptr; }
    if (p.ev[i]) { (void)hipEventDestroy(p.ev[i]); p.ev[i] = nullptr; }
  }
  if (p.stream) { (void)hipStreamDestroy(p.stream); p.stream = nullptr; }
  p.slice = 0;
}

inline hipError_t zn_host_pipe_init(ZnHostPipe& p) {
  if (p.pin[0]) return hipSuccess;
  size_t slice = 32u << 20;
  if (const char* e_ = getenv("ZN_HOST_SLICE_MB")) { const int v = atoi(e_); if (v >= 1 && v <= 256) slice = (size_t)v << 20; }   // tuning knob
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipHostMalloc(&p.pin[i], slice, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p.ev[i], hipEventDisableTiming);
  }
  // (non-blocking: a transfer is complete when zn_host_pipe_copy returns — nothing relies on stream order against the null stream —
  //  and the pipelined host path runs two of these next to a kernel stream: none of them may wait for the others)
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
  if (e != hipSuccess) { zn_host_pipe_release(p); (void)hipGetLastError(); return e; }   // all or nothing
  p.slice = slice;
  return hipSuccess;
}

// to_device: host `h` -> device `d`; else device `d` -> host `h`.  Returns when the transfer is complete.
// ---- the library's own pinned host memory (round 6): zn_host_alloc / zn_host_free ---------------------------------------------------
// Result buffers that the LIBRARY hands out — what the reference's extension does as well: its two functions return memoryviews over memory it
// malloc'ed itself (csrc/zipnn_core.c:596, 1126) — come from an arena of hipHostMalloc'ed blocks that are recycled: no first-touch faults, no 50-130 ms
// munmap per GiB on release, and a transfer between such a block and HBM is ONE asynchronous DMA (57 GB/s; with the upload of the call running beside it
// the copy engines reach their full duplex).  Driver-allocated pinned memory has none of the side effects measured for hipHostRegister'ed user memory
// (profiles/r06_host_path.txt): the library has staged through two such blocks since round 1.
// Blocks are kept when freed, up to ZIPNN_AMD_HOST_ARENA_MB (default 8192) MiB of free blocks; a request takes the smallest free block that fits and is
// not more than twice too big, else a new one (rounded up to 2 MiB).
struct ZnArena {
  struct Block { uint8_t* p; size_t cap; bool used; };
  std::mutex m;
  std::vector<Block> blocks;
  size_t free_bytes = 0;
  static size_t limit() {
    static const size_t v = [] { const char* e = getenv("ZIPNN_AMD_HOST_ARENA_MB"); const long long mb = e ? atoll(e) : 8192; return (size_t)(mb < 0 ? 0 : mb) << 20; }();
    return v;
  }
  void* alloc(size_t n) {
    if (n == 0) n = 1;
    {
      std::lock_guard<std::mutex> lk(m);
      int best = -1;
      for (size_t i = 0; i < blocks.size(); i++)
        if (!blocks[i].used && blocks[i].cap >= n && blocks[i].cap / 2 <= n && (best < 0 || blocks[i].cap < blocks[(size_t)best].cap)) best = (int)i;
      if (best >= 0) { blocks[(size_t)best].used = true; free_bytes -= blocks[(size_t)best].cap; return blocks[(size_t)best].p; }
    }
    const size_t cap = (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      trim();                                      // (free blocks of other sizes may be what is in the way: hand them back and ask once more)
      if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    try { std::lock_guard<std::mutex> lk(m); blocks.push_back(Block{(uint8_t*)p, cap, true}); }
    catch (...) { (void)hipHostFree(p); return nullptr; }
    return p;
  }
  bool release(void* p) {                         // false: not a block of this arena
    void* drop = nullptr;
    {
      std::lock_guard<std::mutex> lk(m);
      size_t i = 0;
      for (; i < blocks.size(); i++) if (blocks[i].p == (uint8_t*)p && blocks[i].used) break;
      if (i == blocks.size()) return false;
      if (free_bytes + blocks[i].cap > limit()) { drop = blocks[i].p; blocks.erase(blocks.begin() + (long)i); }
      else { blocks[i].used = false; free_bytes += blocks[i].cap; }
    }
    if (drop) (void)hipHostFree(drop);
    return true;
  }
  // is [h, h + n) inside one (in-use) block?
  bool covers(const void* h, size_t n) {
    std::lock_guard<std::mutex> lk(m);
    for (const Block& b : blocks) if (b.used && (const uint8_t*)h >= b.p && (const uint8_t*)h + n <= b.p + b.cap) return true;
    return false;
  }
  void trim() {                                   // every free block back to the driver (zn_release_workspace)
    std::vector<void*> drop;
    { std::lock_guard<std::mutex> lk(m); for (size_t i = blocks.size(); i-- > 0;) if (!blocks[i].used) { drop.push_back(blocks[i].p); blocks.erase(blocks.begin() + (long)i); } free_bytes = 0; }
    for (void* q : drop) (void)hipHostFree(q);
  }
};
inline ZnArena& zn_arena() { static ZnArena a; return a; }

// keep (optional): pieces of the caller's buffer that the direct path pinned are left pinned and listed there — the caller unregisters them
// (zn_host_unpin) once every stream that may touch them is idle.
struct ZnPinList { std::mutex m; std::vector<void*> v; };
inline void zn_host_unpin(ZnPinList& l) { std::lock_guard<std::mutex> lk(l.m); for (void* q : l.v) (void)hipHostUnregister(q); l.v.clear(); }
inline hipError_t zn_host_pipe_copy(ZnHostPipe& p, void* d, void* h, size_t n, bool to_device, ZnPinList* keep = nullptr, bool staged_only = false) {
  using namespace zn_host_pipe_detail;
  if (n == 0) return hipSuccess;
  hipError_t e = zn_host_pipe_init(p);
  if (n < (2u << 20)) {                         // small: one plain copy, on this pipe's stream (a null-stream copy would wait for the other pipe's transfers)
    if (e != hipSuccess) return to_device ? hipMemcpy(d, h, n, hipMemcpyHostToDevice) : hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
    e = to_device ? hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, p.stream) : hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, p.stream);
    const hipError_t e2 = hipStreamSynchronize(p.stream);
    return e != hipSuccess ? e : e2;
  }
  if (e != hipSuccess)                          // no pinned memory to be had: the plain, slower way
    return to_device ? hipMemcpy(d, h, n, hipMemcpyHostToDevice) : hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
  if (zn_arena().covers(h, n)) {                // the library's own pinned memory (zn_host_alloc): one DMA, nothing to stage (cut into 4-64 MiB pieces: no difference, measured)
    e = to_device ? hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, p.stream) : hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, p.stream);
    const hipError_t e2 = hipStreamSynchronize(p.stream);
    return e != hipSuccess ? e : e2;
  }
  const size_t S = p.slice;
  const unsigned T = worker_count(n);
  Barrier go(T + 1), done(T + 1);
  unsigned nworkers = T;
  std::atomic<bool> stop{false};
  // what the workers copy in the current round
  struct Round { uint8_t* dst; const uint8_t* src; size_t len; bool touch; } cur = {nullptr, nullptr, 0, false};
  // (a worker that cannot be started — thread limit — just leaves the others more to do: the barriers are sized
  //  to the workers that exist before anyone can complete a round; no exception leaves this function)
  std::vector<std::thread> pool;
  unsigned started = 0;
  try {
    pool.reserve(T);
    for (unsigned w = 0; w < T; w++) {
      pool.emplace_back([&, w] {
        for (;;) {
          go.wait();
          if (stop.load()) return;
          size_t lo, hi; stripe(cur.len, w, nworkers, &lo, &hi);
          if (hi > lo && !cur.touch) memcpy(cur.dst + lo, cur.src + lo, hi - lo);
          else if (hi > lo) { volatile uint8_t* q = cur.dst; for (size_t o = lo; o < hi; o += 4096) q[o] = 0; }      // first touch of a destination piece (it is overwritten by the DMA that follows)
          done.wait();
        }
      });
      started++;
    }
  } catch (...) { }
  if (started == 0) return to_device ? hipMemcpy(d, h, n, hipMemcpyHostToDevice) : hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
  nworkers = started; go.resize(started + 1); done.resize(started + 1);
  const bool trace = getenv("ZN_HOST_PIPE_TRACE") != nullptr;
  double t_copy = 0, t_wait = 0, t_issue = 0;
  auto now = [] { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; };
  const double t_begin = now();
  auto round = [&](uint8_t* dst, const uint8_t* src, size_t len) { const double t0 = now(); cur.dst = dst; cur.src = src; cur.len = len; cur.touch = false; go.wait(); done.wait(); t_copy += now() - t0; };
  uint8_t* hb = (uint8_t*)h; uint8_t* db = (uint8_t*)d;
  // ---- the direct phase: a prefix of the buffer, piece by piece, for as long as registration runs at huge-page speed ----
  size_t direct = 0; double t_touch = 0, t_reg = 0; unsigned pieces = 0; bool gave_up = false;
  double t_unreg = 0;
  if (!to_device) zn_host_thp_hint(h, n);
  if (!staged_only && (direct_mode() & (to_device ? 2 : 1)) && n >= ((size_t)128 << 20) && (to_device || zn_host_resident(h, n))) {
    const size_t PIECE = (size_t)64 << 20, PROBE = (size_t)16 << 20;      // (the first piece is small: it is the one that finds out what the pages are — 4 KiB pages cost it ≈ 1 ms)
    // Pinned pieces stay pinned until the stream is idle: hipHostUnregister with DMAs in flight waits for them (measured: 1 ms per piece, the whole
    // pipelining gone).  An epoch of 16 pieces (1 GiB) ends with one synchronise and their unregistration; `keep` hands that to the caller instead
    // (the pipelined entry points, whose two directions must not wait for each other's DMAs).
    std::vector<void*> mine;
    std::vector<void*>& pinned = keep ? keep->v : mine;
    auto unpin_all = [&] { const double t0 = now(); for (void* q : mine) (void)hipHostUnregister(q); mine.clear(); t_unreg += now() - t0; };
    try {
      while (direct < n && e == hipSuccess) {
        // (cuts on 2 MiB boundaries of the host address: no huge page is split between two registrations; the last piece takes the tail)
        size_t len = (pieces ? PIECE : PROBE) - (size_t)(((uintptr_t)hb + direct) & (((size_t)2 << 20) - 1));
        if (direct + len + (PIECE >> 2) > n) len = n - direct;
        if (!keep && mine.size() >= 16) {
          const double t0 = now(); e = hipStreamSynchronize(p.stream); t_wait += now() - t0;
          unpin_all();
          if (e != hipSuccess) break;
        }
        if (!to_device) { const double t0 = now(); cur.dst = hb + direct; cur.src = nullptr; cur.len = len; cur.touch = true; go.wait(); done.wait(); t_touch += now() - t0; }
        const double t0 = now();
        if (hipHostRegister(hb + direct, len, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); gave_up = true; break; }      // (not registrable: the staged copy takes it from here)
        const double dt = now() - t0; t_reg += dt;
        if (keep) { std::lock_guard<std::mutex> lk(keep->m); pinned.push_back(hb + direct); } else pinned.push_back(hb + direct);
        const double t1 = now();
        e = to_device ? hipMemcpyAsync(db + direct, hb + direct, len, hipMemcpyHostToDevice, p.stream) : hipMemcpyAsync(hb + direct, db + direct, len, hipMemcpyDeviceToHost, p.stream);
        t_issue += now() - t1;
        direct += len; pieces++;
        if (dt > 11e-3 * ((double)len / (double)((size_t)1 << 30))) { gave_up = true; break; }      // 4 KiB pages underneath: pinning them costs more than staging them
      }
    } catch (...) { gave_up = true; }                                  // (std::bad_alloc from the list: what is pinned is in it, the rest goes the staged way)
    { const double t0 = now(); const hipError_t es = hipStreamSynchronize(p.stream); if (e == hipSuccess) e = es; t_wait += now() - t0; }
    unpin_all();
    hb += direct; db += direct; n -= direct;
  }
  const size_t slices_left = (n + S - 1) / S;
  auto len_of = [&](size_t i) { return (i + 1) * S <= n ? S : n - i * S; };
  if (n == 0 || e != hipSuccess) { }
  else if (to_device) {
    for (size_t i = 0; i < slices_left && e == hipSuccess; i++) {
      const int b = (int)(i & 1);
      { const double t0 = now(); if (i >= 2) e = hipEventSynchronize(p.ev[b]); t_wait += now() - t0; }   // the DMA that last read this bounce buffer is done
      if (e != hipSuccess) break;
      round((uint8_t*)p.pin[b], hb + i * S, len_of(i));
      const double t1 = now();
      e = hipMemcpyAsync(db + i * S, p.pin[b], len_of(i), hipMemcpyHostToDevice, p.stream);
      if (e == hipSuccess) e = hipEventRecord(p.ev[b], p.stream);
      t_issue += now() - t1;
    }
  } else {
    e = hipMemcpyAsync(p.pin[0], db, len_of(0), hipMemcpyDeviceToHost, p.stream);
    if (e == hipSuccess) e = hipEventRecord(p.ev[0], p.stream);
    for (size_t i = 0; i < slices_left && e == hipSuccess; i++) {
      const int b = (int)(i & 1);
      if (i + 1 < slices_left) {                                         // (its previous contents were copied out in round i - 1)
        e = hipMemcpyAsync(p.pin[b ^ 1], db + (i + 1) * S, len_of(i + 1), hipMemcpyDeviceToHost, p.stream);
        if (e == hipSuccess) e = hipEventRecord(p.ev[b ^ 1], p.stream);
        if (e != hipSuccess) break;
      }
      { const double t0 = now(); e = hipEventSynchronize(p.ev[b]); t_wait += now() - t0; }
      if (e != hipSuccess) break;
      round(hb + i * S, (const uint8_t*)p.pin[b], len_of(i));
    }
  }
  stop.store(true);
  go.wait();
  for (auto& t : pool) t.join();
  const hipError_t e2 = hipStreamSynchronize(p.stream);
  if (trace) fprintf(stderr, "[zn host pipe] %s %.1f MiB, %u workers: direct %.1f MiB in %u pieces (touch %.1f, register %.1f, unregister %.1f ms%s), staged %.1f MiB in %zu slices: total %.1f ms (worker copies %.1f, waiting for DMA %.1f, issuing %.1f)\n",
                     to_device ? "H2D" : "D2H", (n + direct) / 1048576.0, nworkers, direct / 1048576.0, pieces, t_touch * 1e3, t_reg * 1e3, t_unreg * 1e3, gave_up ? "; gave up: 4 KiB pages" : "", n / 1048576.0, slices_left,
                     (now() - t_begin) * 1e3, t_copy * 1e3, t_wait * 1e3, t_issue * 1e3);
  return e != hipSuccess ? e : e2;
}

// ---- ZnHostMap: a caller's buffer made DMA-able AHEAD of the transfers that will use it (the pipelined entry points) ----------------
// A helper thread walks [h, h + n) in address order, piece by piece (the first one small: it finds out what the pages are) and pins each with
// hipHostRegister; progress is published.  A destination is only taken when its pages are RESIDENT (zn_host_resident: a fresh result buffer goes the
// staged way and is never pinned).  A transfer asks wait(off, len): true = that range is pinned, a plain asynchronous DMA moves it; false =
// the map gave up (4 KiB pages: registration slower than 11 ms per GiB, or not registrable at all) and the caller takes the staged copy.
// `limit` bounds how far ahead of the caller's needs a DESTINATION is prepared (a compress result uses two thirds of its capacity).
// The destructor joins the thread and unpins everything: by then every stream that touched the buffer must be idle.
struct ZnHostMap {
  uint8_t* h = nullptr; size_t n = 0; bool write = false;
  std::mutex m; std::condition_variable cv;
  size_t ready = 0, limit = 0; bool gave_up = false, stop = false, started = false;
  std::vector<void*> pinned;
  std::vector<size_t> cuts;                        // piece i covers [cuts[i], cuts[i + 1]): a DMA must stay inside ONE registration (the runtime resolves a host pointer to the registration it starts in)
  std::thread th;
  double t_touch = 0, t_reg = 0;
  static double now_() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
  void start(void* h_, size_t n_, bool write_, size_t limit_, int dev) {
    using namespace zn_host_pipe_detail;
    h = (uint8_t*)h_; n = n_; write = write_; limit = limit_ < n_ ? limit_ : n_;
    if (write_) zn_host_thp_hint(h_, n_);
    if (!(direct_mode() & (write_ ? 1 : 2)) || n < ((size_t)64 << 20) || (write_ && !zn_host_resident(h_, limit))) { gave_up = true; return; }
    try {
      th = std::thread([this, dev] {
        if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); std::lock_guard<std::mutex> lk(m); gave_up = true; cv.notify_all(); return; }
        const size_t PIECE = (size_t)64 << 20, PROBE = (size_t)16 << 20;
        size_t at = 0; unsigned pieces = 0;
        for (;;) {
          size_t lim;
          { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return stop || at < limit; }); if (stop) return; lim = limit; }
          size_t len = (pieces ? PIECE : PROBE) - (size_t)(((uintptr_t)h + at) & (((size_t)2 << 20) - 1));
          if (at + len + (PIECE >> 2) > n) len = n - at;
          (void)lim;
          const double t0 = now_();
          const bool ok = hipHostRegister(h + at, len, hipHostRegisterDefault) == hipSuccess;
          const double dt = now_() - t0; t_reg += dt;
          if (!ok) (void)hipGetLastError();
          std::lock_guard<std::mutex> lk(m);
          if (ok) { try { pinned.push_back(h + at); if (cuts.empty()) cuts.push_back(0); cuts.push_back(at + len); } catch (...) { (void)hipHostUnregister(h + at); if (pinned.size() >= cuts.size() && !pinned.empty()) pinned.pop_back(); gave_up = true; cv.notify_all(); return; } at += len; pieces++; ready = at; }
          if (!ok || dt > 11e-3 * ((double)len / (double)((size_t)1 << 30))) { gave_up = true; cv.notify_all(); return; }      // (a slow piece stays usable; nothing further is pinned)
          cv.notify_all();
          if (at >= n) return;
        }
      });
      started = true;
    } catch (...) { gave_up = true; }
  }
  void raise_limit(size_t x) { std::lock_guard<std::mutex> lk(m); if (x > n) x = n; if (x > limit) { limit = x; cv.notify_all(); } }
  // is [off, off + len) pinned (waiting for the helper if it is still on its way there)?
  bool wait(size_t off, size_t len) {
    if (len == 0) return true;
    std::unique_lock<std::mutex> lk(m);
    if (off + len > limit && !gave_up) { limit = off + len < n ? off + len : n; cv.notify_all(); }
    cv.wait(lk, [&] { return ready >= off + len || gave_up || stop; });
    return ready >= off + len;
  }
  ~ZnHostMap() {
    { std::lock_guard<std::mutex> lk(m); stop = true; cv.notify_all(); }
    if (started && th.joinable()) th.join();
    for (void* q : pinned) (void)hipHostUnregister(q);
  }
};
// one transfer of the pipelined paths: through the map when the range is pinned (one asynchronous DMA on the pipe's stream, waited for), the staged copy otherwise
inline hipError_t zn_host_copy_mapped(ZnHostMap& map, ZnHostPipe& p, void* d, void* h, size_t len, bool to_device) {
  if (len == 0) return hipSuccess;
  if (len >= ((size_t)1 << 20) && zn_arena().covers(h, len)) return zn_host_pipe_copy(p, d, h, len, to_device, nullptr, true);      // (its arena branch: one DMA)
  const size_t off = (size_t)((uint8_t*)h - map.h);
  if (len >= ((size_t)1 << 20) && (uint8_t*)h >= map.h && off + len <= map.n && map.wait(off, len)) {
    hipError_t e = zn_host_pipe_init(p);
    if (e != hipSuccess) return e;
    std::vector<size_t> cuts;
    { std::lock_guard<std::mutex> lk(map.m); cuts = map.cuts; }
    for (size_t i = 0; i + 1 < cuts.size() && e == hipSuccess; i++) {       // one DMA per registration the range crosses
      const size_t lo = cuts[i] > off ? cuts[i] : off, hi = cuts[i + 1] < off + len ? cuts[i + 1] : off + len;
      if (hi <= lo) continue;
      uint8_t* hh = map.h + lo; uint8_t* dd = (uint8_t*)d + (lo - off);
      e = to_device ? hipMemcpyAsync(dd, hh, hi - lo, hipMemcpyHostToDevice, p.stream) : hipMemcpyAsync(hh, dd, hi - lo, hipMemcpyDeviceToHost, p.stream);
    }
    const hipError_t e2 = hipStreamSynchronize(p.stream);
    return e != hipSuccess ? e : e2;
  }
  return zn_host_pipe_copy(p, d, h, len, to_device, nullptr, true);
}
