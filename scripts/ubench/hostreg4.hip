// hostreg4.hip — probe: do page faults get slower once the process has pinned (and unpinned) user memory?  And while memory is pinned?
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
static const size_t N = (size_t)1 << 30;
static void touch_mt(uint8_t* p, size_t n, int T) {
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back([=] { size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1); for (size_t i = lo; i < hi; i += 4096) p[i] = 1; });
  for (auto& t : th) t.join();
}
static uint8_t* fresh() { uint8_t* b = (uint8_t*)malloc(N + 64); uint8_t* a = b + 16; uintptr_t lo = ((uintptr_t)a + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1), hi = ((uintptr_t)a + N) & ~(uintptr_t)((2 << 20) - 1); madvise((void*)lo, hi - lo, MADV_HUGEPAGE); return b; }
static double fault_ms(bool with_dma, uint8_t* d, uint8_t* pin, hipStream_t s, double* dma_ms) {
  uint8_t* b = fresh();
  double t0 = now();
  if (with_dma) CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s));
  touch_mt(b + 16, N, 8);
  const double t = now() - t0;
  if (with_dma) { CK(hipStreamSynchronize(s)); *dma_ms = (now() - t0) * 1e3; }
  free(b);
  return t * 1e3;
}
int main() {
  uint8_t* d; CK(hipMalloc(&d, N)); CK(hipMemset(d, 7, N));
  uint8_t* pin; CK(hipHostMalloc((void**)&pin, N, 0)); memset(pin, 3, N);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double dm = 0;
  for (int i = 0; i < 3; i++) printf("before any registration: fault 1 GiB (8 threads) %.1f ms\n", fault_ms(false, d, pin, s, &dm));
  for (int i = 0; i < 3; i++) { double f = fault_ms(true, d, pin, s, &dm); printf("before any registration: fault %.1f ms WITH an H2D DMA from pinned memory in flight (DMA done after %.1f ms)\n", f, dm); }
  uint8_t* w = fresh(); touch_mt(w + 16, N, 8);
  CK(hipHostRegister(w + 16, N, 0));
  for (int i = 0; i < 3; i++) printf("1 GiB of user memory PINNED: fault %.1f ms\n", fault_ms(false, d, pin, s, &dm));
  for (int i = 0; i < 3; i++) { double f = fault_ms(true, d, pin, s, &dm); printf("1 GiB pinned: fault %.1f ms with an H2D DMA in flight (DMA done after %.1f ms)\n", f, dm); }
  CK(hipHostUnregister(w + 16));
  for (int i = 0; i < 3; i++) printf("after hipHostUnregister: fault %.1f ms\n", fault_ms(false, d, pin, s, &dm));
  for (int i = 0; i < 3; i++) { double f = fault_ms(true, d, pin, s, &dm); printf("after unregister: fault %.1f ms with an H2D DMA in flight (DMA done after %.1f ms)\n", f, dm); }
  free(w);
  for (int i = 0; i < 3; i++) printf("after freeing that buffer: fault %.1f ms\n", fault_ms(false, d, pin, s, &dm));
  for (int i = 0; i < 3; i++) { double f = fault_ms(true, d, pin, s, &dm); printf("after freeing: fault %.1f ms with an H2D DMA in flight (DMA done after %.1f ms)\n", f, dm); }
  return 0;
}
