// hostreg5.hip — probe: is the slow-fault / slow-DMA effect of pinned user memory a matter of process AGE (automatic NUMA balancing starts scanning a task's address
// space about a second after it starts)?  The pattern of scripts/hp_seq2.py "C" in C++: warm call (pin src + dst, DMA both ways, unpin), fresh dst (malloc, pin piecewise
// with touch, DMA, unpin, free), repeated for several seconds, with busy threads touching memory in between.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static const size_t N = (size_t)1 << 30, S = 64u << 20;
static void touch_mt(uint8_t* p, size_t n, int T) {
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back([=] { size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1); for (size_t i = lo; i < hi; i += 4096) p[i] = 1; });
  for (auto& t : th) t.join();
}
static uint8_t* fresh() { uint8_t* b = (uint8_t*)malloc(N + 64); uint8_t* a = b + 16; uintptr_t lo = ((uintptr_t)a + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1), hi = ((uintptr_t)a + N) & ~(uintptr_t)((2 << 20) - 1); madvise((void*)lo, hi - lo, MADV_HUGEPAGE); return b; }
int main(int argc, char** argv) {
  const double run_s = argc > 1 ? atof(argv[1]) : 6.0;
  uint8_t* d; CK(hipMalloc(&d, N)); CK(hipMemset(d, 7, N));
  uint8_t* d2; CK(hipMalloc(&d2, N));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint8_t* srcb = fresh(); uint8_t* src = srcb + 16; touch_mt(src, N, 8); memset(src, 5, 1 << 20);
  { FILE* f = fopen("/proc/sys/kernel/numa_balancing", "r"); char b[32] = {0}; if (f) { fgets(b, 31, f); fclose(f); } printf("numa_balancing = %s", b[0] ? b : "?\n"); }
  const double t_start = now();
  int it = 0;
  while (now() - t_start < run_s) {
    // fresh destination: pinned source H2D (678 MiB), then dst touched + pinned piecewise, D2H, unpin, free
    uint8_t* b = fresh(); uint8_t* a = b + 16;
    double t0 = now();
    std::vector<void*> pins;
    const size_t n_up = (size_t)678 << 20;
    for (size_t o = 0; o < n_up; o += S) { const size_t m = n_up - o < S ? n_up - o : S; CK(hipHostRegister(src + o, m, 0)); pins.push_back(src + o); CK(hipMemcpyAsync(d2 + o, src + o, m, hipMemcpyHostToDevice, s)); }
    CK(hipStreamSynchronize(s));
    const double t_h2d = now() - t0; t0 = now();
    for (void* q : pins) CK(hipHostUnregister(q)); pins.clear();
    for (size_t o = 0; o < N; o += S) { touch_mt(a + o, S, 8); CK(hipHostRegister(a + o, S, 0)); pins.push_back(a + o); CK(hipMemcpyAsync(a + o, d + o, S, hipMemcpyDeviceToHost, s)); }
    CK(hipStreamSynchronize(s));
    for (void* q : pins) CK(hipHostUnregister(q));
    const double t_d2h = now() - t0; t0 = now();
    free(b);
    const double t_free = now() - t0;
    if (it % 4 == 0 || t_h2d > 0.02) printf("t = %5.2f s  it %3d: H2D 678 MiB (pinned user memory) %5.1f ms   D2H 1 GiB into a fresh buffer %5.1f ms   free %5.1f ms\n", now() - t_start, it, t_h2d * 1e3, t_d2h * 1e3, t_free * 1e3);
    it++;
  }
  return 0;
}
