// hostreg3.hip — probe: does a freshly mapped (THP-hinted) destination that is later pinned piece by piece slow down the H2D DMA that runs BEFORE it is touched?
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
int main(int argc, char** argv) {
  const int variant = argc > 1 ? atoi(argv[1]) : 0;   // 0: fresh dst each iteration, direct D2H   1: same dst kept (warm)   2: fresh dst, D2H via pinned bounce + memcpy (staged)  3: fresh, direct, but NO free at the end (leak)
  const size_t N = (size_t)1 << 30, S = 64u << 20;
  uint8_t* d; CK(hipMalloc(&d, N)); CK(hipMemset(d, 7, N));
  uint8_t* pin; CK(hipHostMalloc((void**)&pin, N, 0)); memset(pin, 3, N);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint8_t* keep = nullptr;
  for (int it = 0; it < 6; it++) {
    double t0 = now();
    uint8_t* base = (variant == 1 && keep) ? keep : (uint8_t*)malloc(N + 64);
    uint8_t* a = base + 16;
    if (!(variant == 1 && keep)) { uintptr_t lo = ((uintptr_t)a + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1), hi = ((uintptr_t)a + N) & ~(uintptr_t)((2 << 20) - 1); madvise((void*)lo, hi - lo, MADV_HUGEPAGE); }
    keep = base;
    const double t_alloc = now() - t0; t0 = now();
    CK(hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));        // "the staged upload": DMA out of pinned memory
    const double t_h2d = now() - t0; t0 = now();
    double t_touch = 0, t_reg = 0;
    if (variant != 2) {
      std::vector<void*> pinned;
      for (size_t o = 0; o < N; o += S) {
        double t1 = now(); touch_mt(a + o, S, 8); t_touch += now() - t1;
        t1 = now(); CK(hipHostRegister(a + o, S, 0)); t_reg += now() - t1; pinned.push_back(a + o);
        CK(hipMemcpyAsync(a + o, d + o, S, hipMemcpyDeviceToHost, s));
      }
      CK(hipStreamSynchronize(s));
      for (void* q : pinned) CK(hipHostUnregister(q));
    } else {
      for (size_t o = 0; o < N; o += S) { CK(hipMemcpyAsync(pin, d + o, S, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); std::vector<std::thread> th; for (int t = 0; t < 8; t++) th.emplace_back([=] { memcpy(a + o + S / 8 * t, pin + S / 8 * t, S / 8); }); for (auto& t : th) t.join(); }
    }
    const double t_d2h = now() - t0; t0 = now();
    if (variant == 0 || variant == 2) free(base);
    const double t_free = now() - t0;
    printf("variant %d it %d: alloc %.2f  H2D(pinned, 1 GiB) %.1f  D2H %.1f (touch %.1f reg %.1f)  free %.1f ms\n", variant, it, t_alloc * 1e3, t_h2d * 1e3, t_d2h * 1e3, t_touch * 1e3, t_reg * 1e3, t_free * 1e3);
  }
  return 0;
}
