// hostreg2.hip — follow-up probe: what munmap / hipHostUnregister cost for huge-page and 4 KiB-page buffers, registered whole or in 64 MiB pieces.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static void touch_mt(uint8_t* p, size_t n, int T) {
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back([=] { size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1); for (size_t i = lo; i < hi; i += 4096) p[i] = 1; });
  for (auto& t : th) t.join();
}
static long anon_huge_kb() { FILE* f = fopen("/proc/self/smaps_rollup", "r"); char b[256]; long v = -1; while (f && fgets(b, 255, f)) if (!strncmp(b, "AnonHugePages:", 14)) v = atol(b + 14); if (f) fclose(f); return v; }
int main() {
  const size_t N = (size_t)1 << 30;
  uint8_t* d; CK(hipMalloc(&d, N)); CK(hipMemset(d, 7, N));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int rep = 0; rep < 2; rep++) for (int mode = 0; mode < 6; mode++) {
    // 0: THP, touch, munmap      1: 4K, touch, munmap      2: THP, touch, register whole, DMA, unregister, munmap
    // 3: THP, pieces of 64 MiB: register, DMA, (sync) unregister each; munmap     4: as 3 but all unregisters at the end   5: malloc + madvise (what the library does), pieces, free
    const bool thp = mode != 1;
    uint8_t* a; void* base = nullptr;
    if (mode == 5) { base = malloc(N + 64); a = (uint8_t*)base + 16; uintptr_t lo = ((uintptr_t)a + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1), hi = ((uintptr_t)a + N) & ~(uintptr_t)((2 << 20) - 1); madvise((void*)lo, hi - lo, MADV_HUGEPAGE); }
    else { a = (uint8_t*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); if (thp) madvise(a, N, MADV_HUGEPAGE); }
    double t0 = now(); touch_mt(a, N, 8); const double t_touch = now() - t0;
    const long huge = anon_huge_kb();
    double t_reg = 0, t_dma = 0, t_unreg = 0;
    if (mode == 2) {
      t0 = now(); CK(hipHostRegister(a, N, 0)); t_reg = now() - t0;
      t0 = now(); CK(hipMemcpyAsync(a, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t_dma = now() - t0;
      t0 = now(); CK(hipHostUnregister(a)); t_unreg = now() - t0;
    } else if (mode >= 3) {
      const size_t S = 64u << 20;
      const double tb = now();
      for (size_t o = 0; o < N; o += S) {
        t0 = now(); CK(hipHostRegister(a + o, S, 0)); t_reg += now() - t0;
        CK(hipMemcpyAsync(a + o, d + o, S, hipMemcpyDeviceToHost, s));
        if (mode == 3 && o >= 3 * S) { CK(hipStreamSynchronize(s)); t0 = now(); CK(hipHostUnregister(a + o - 3 * S)); t_unreg += now() - t0; }
      }
      CK(hipStreamSynchronize(s)); t_dma = now() - tb;
      t0 = now();
      for (size_t o = (mode == 3 ? N - 3 * S : 0); o < N; o += S) CK(hipHostUnregister(a + o));
      t_unreg += now() - t0;
    }
    t0 = now(); if (mode == 5) free(base); else munmap(a, N); const double t_free = now() - t0;
    const char* nm[] = {"THP touch munmap", "4K touch munmap", "THP register whole", "THP 64 MiB pieces, unregister as it goes", "THP 64 MiB pieces, unregister at the end", "malloc + madvise, pieces, free"};
    printf("%-42s touch %5.1f (AnonHuge %ld MB) register %5.1f  dma(total) %5.1f  unregister %5.1f  free %5.1f ms\n", nm[mode], t_touch * 1e3, huge >> 10, t_reg * 1e3, t_dma * 1e3, t_unreg * 1e3, t_free * 1e3);
  }
  return 0;
}
