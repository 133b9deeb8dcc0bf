// hostreg.hip — developer probe for the host-buffer entry points: what does it cost to make a caller's pageable buffer DMA-able
// (hipHostRegister) against staging it through pinned bounce buffers, for fresh (never touched) and warm pages, both directions?
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/ubench/hostreg scripts/ubench/hostreg.hip -lpthread
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
static void touch_mt(uint8_t* p, size_t n, int T, int val) {
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back([=] { size_t lo = n / T * t, hi = t == T - 1 ? n : n / T * (t + 1); if (val < 0) { for (size_t i = lo; i < hi; i += 4096) p[i] = 1; } else memset(p + lo, val, hi - lo); });
  for (auto& t : th) t.join();
}
int main(int argc, char** argv) {
  const size_t N = (size_t)1 << 30;
  uint8_t* d; CK(hipMalloc(&d, N)); CK(hipMemset(d, 7, N));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  { void* w; CK(hipHostMalloc(&w, 1 << 20, 0)); CK(hipMemcpy(d, w, 1 << 20, hipMemcpyHostToDevice)); CK(hipHostFree(w)); }
  for (int rep = 0; rep < 2; rep++) {
    printf("--- rep %d\n", rep);
    for (int mode = 0; mode < 5; mode++) {
      // 0: fresh malloc, register, D2H, unregister      1: fresh malloc, touched first by 8 threads (page stride), register, D2H
      // 2: warm (memset before), register, D2H          3: warm, plain hipMemcpy D2H (runtime staging)   4: fresh, plain hipMemcpy D2H
      uint8_t* h = (uint8_t*)malloc(N + 4096);
      uint8_t* a = (uint8_t*)(((uintptr_t)h + 4095) & ~(uintptr_t)4095);
      double t_touch = 0, t_reg = 0, t_copy = 0, t_unreg = 0;
      if (mode == 1) { double t0 = now(); touch_mt(a, N, 8, -1); t_touch = now() - t0; }
      if (mode == 2 || mode == 3) touch_mt(a, N, 8, 3);
      if (mode <= 2) {
        double t0 = now(); CK(hipHostRegister(a, N, hipHostRegisterDefault)); t_reg = now() - t0;
        t0 = now(); CK(hipMemcpyAsync(a, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t_copy = now() - t0;
        t0 = now(); CK(hipHostUnregister(a)); t_unreg = now() - t0;
      } else { double t0 = now(); CK(hipMemcpy(a, d, N, hipMemcpyDeviceToHost)); t_copy = now() - t0; }
      const char* nm[] = {"fresh, register + DMA", "fresh, 8-thread touch, register + DMA", "warm, register + DMA", "warm, plain hipMemcpy", "fresh, plain hipMemcpy"};
      printf("D2H 1 GiB %-40s touch %6.1f  register %6.1f  copy %6.1f  unregister %6.1f  total %6.1f ms  (%5.1f GB/s)  check %d\n", nm[mode], t_touch * 1e3, t_reg * 1e3, t_copy * 1e3, t_unreg * 1e3,
             (t_touch + t_reg + t_copy + t_unreg) * 1e3, N / (t_touch + t_reg + t_copy + t_unreg) / 1e9, a[N - 1]);
      double t0 = now(); free(h); printf("      free %.1f ms\n", (now() - t0) * 1e3);
    }
    // H2D from a warm caller buffer: register + DMA vs plain
    {
      uint8_t* h = (uint8_t*)malloc(N + 4096); uint8_t* a = (uint8_t*)(((uintptr_t)h + 4095) & ~(uintptr_t)4095);
      touch_mt(a, N, 8, 5);
      double t0 = now(); CK(hipHostRegister(a, N, hipHostRegisterDefault)); double t_reg = now() - t0;
      t0 = now(); CK(hipMemcpyAsync(d, a, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t_copy = now() - t0;
      t0 = now(); CK(hipHostUnregister(a)); double t_unreg = now() - t0;
      printf("H2D 1 GiB warm, register + DMA: register %.1f copy %.1f unregister %.1f ms\n", t_reg * 1e3, t_copy * 1e3, t_unreg * 1e3);
      t0 = now(); CK(hipMemcpy(d, a, N, hipMemcpyHostToDevice)); printf("H2D 1 GiB warm, plain hipMemcpy: %.1f ms\n", (now() - t0) * 1e3);
      // registering in slices of 64 MiB on a second thread while the DMA of the previous slice runs
      { const size_t S = 64u << 20; double t00 = now();
        std::thread reg([&] { for (size_t o = 0; o < N; o += S) CK(hipHostRegister(a + o, S, hipHostRegisterDefault)); });
        reg.join(); double t_r = now() - t00;
        t00 = now(); for (size_t o = 0; o < N; o += S) CK(hipMemcpyAsync(d + o, a + o, S, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t_c = now() - t00;
        t00 = now(); for (size_t o = 0; o < N; o += S) CK(hipHostUnregister(a + o)); double t_u = now() - t00;
        printf("H2D 1 GiB warm, 16 slices of 64 MiB: register %.1f copy %.1f unregister %.1f ms\n", t_r * 1e3, t_c * 1e3, t_u * 1e3); }
      free(h);
    }
    // mmap with MAP_POPULATE / MADV_HUGEPAGE as the fresh destination
    for (int hp = 0; hp < 2; hp++) {
      double t0 = now();
      uint8_t* a = (uint8_t*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (hp) madvise(a, N, MADV_HUGEPAGE);
      double t_map = now() - t0; t0 = now();
      touch_mt(a, N, 8, -1); double t_touch = now() - t0;
      t0 = now(); CK(hipHostRegister(a, N, hipHostRegisterDefault)); double t_reg = now() - t0;
      t0 = now(); CK(hipMemcpyAsync(a, d, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t_copy = now() - t0;
      t0 = now(); CK(hipHostUnregister(a)); double t_unreg = now() - t0;
      printf("D2H 1 GiB fresh mmap%s: map %.1f touch(8 thr) %.1f register %.1f copy %.1f unregister %.1f ms\n", hp ? " + MADV_HUGEPAGE" : "", t_map * 1e3, t_touch * 1e3, t_reg * 1e3, t_copy * 1e3, t_unreg * 1e3);
      munmap(a, N);
    }
  }
  { FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char b[128] = {0}; if (f) { fgets(b, 127, f); fclose(f); } printf("THP: %s", b); }
  return 0;
}
