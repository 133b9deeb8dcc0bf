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

struct ZnHostPipe {
  void* pin[2] = {nullptr, nullptr};
  size_t slice = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
};

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
inline hipError_t zn_host_pipe_copy(ZnHostPipe& p, void* d, void* h, size_t n, bool to_device) {
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
  const size_t S = p.slice, slices = (n + S - 1) / S;
  const unsigned T = worker_count(n);
  Barrier go(T + 1), done(T + 1);
  unsigned nworkers = T;
  std::atomic<bool> stop{false};
  // what the workers copy in the current round
  struct Round { uint8_t* dst; const uint8_t* src; size_t len; } cur = {nullptr, nullptr, 0};
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
          if (hi > lo) memcpy(cur.dst + lo, cur.src + lo, hi - lo);
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
  auto round = [&](uint8_t* dst, const uint8_t* src, size_t len) { const double t0 = now(); cur.dst = dst; cur.src = src; cur.len = len; go.wait(); done.wait(); t_copy += now() - t0; };
  auto len_of = [&](size_t i) { return (i + 1) * S <= n ? S : n - i * S; };
  uint8_t* hb = (uint8_t*)h; uint8_t* db = (uint8_t*)d;
  if (to_device) {
    for (size_t i = 0; i < slices && e == hipSuccess; i++) {
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
    for (size_t i = 0; i < slices && e == hipSuccess; i++) {
      const int b = (int)(i & 1);
      if (i + 1 < slices) {                                         // (its previous contents were copied out in round i - 1)
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
  if (trace) fprintf(stderr, "[zn host pipe] %s %.1f MiB, %u workers, %zu slices: total %.1f ms (worker copies %.1f, waiting for DMA %.1f, issuing %.1f)\n",
                     to_device ? "H2D" : "D2H", n / 1048576.0, nworkers, slices, (now() - t_begin) * 1e3, t_copy * 1e3, t_wait * 1e3, t_issue * 1e3);
  return e != hipSuccess ? e : e2;
}
