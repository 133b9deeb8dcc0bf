// duplex.hip — probe: is PCIe full duplex reachable?  H2D and D2H at once: both by the copy engines; H2D by a kernel that reads pinned host memory while a copy engine moves D2H; both by kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = __builtin_nontemporal_load(src + i);
}
int main() {
  const size_t N = (size_t)1 << 30;
  uint8_t *d1, *d2, *h1, *h2;
  CK(hipMalloc(&d1, N)); CK(hipMalloc(&d2, N)); CK(hipHostMalloc((void**)&h1, N, 0)); CK(hipHostMalloc((void**)&h2, N, 0));
  memset(h1, 1, N); memset(h2, 2, N); CK(hipMemset(d1, 3, N)); CK(hipMemset(d2, 4, N));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto run = [&](const char* nm, int up, int down) {   // 0 none, 1 copy engine, 2 kernel (blocks = third arg)
    double best = 1e9;
    for (int r = 0; r < 4; r++) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      if (up == 1) CK(hipMemcpyAsync(d1, h1, N, hipMemcpyHostToDevice, s1));
      if (up >= 2) hipLaunchKernelGGL(k_copy, dim3(up), dim3(256), 0, s1, (const v4u*)h1, (v4u*)d1, N / 16);
      if (down == 1) CK(hipMemcpyAsync(h2, d2, N, hipMemcpyDeviceToHost, s2));
      if (down >= 2) hipLaunchKernelGGL(k_copy, dim3(down), dim3(256), 0, s2, (const v4u*)d2, (v4u*)h2, N / 16);
      CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
      const double t = now() - t0; if (t < best) best = t;
    }
    const double gb = ((up ? 1 : 0) + (down ? 1 : 0)) * (double)N / 1e9;
    printf("%-70s %6.1f ms  aggregate %5.1f GB/s\n", nm, best * 1e3, gb / best);
  };
  // registered (malloc + MADV_HUGEPAGE + hipHostRegister) memory instead of hipHostMalloc'ed, per direction
  uint8_t *r1, *r2;
  { void* b1 = malloc(N + (4 << 20)); void* b2 = malloc(N + (4 << 20));
    r1 = (uint8_t*)(((uintptr_t)b1 + (2 << 20)) & ~(uintptr_t)((2 << 20) - 1)); r2 = (uint8_t*)(((uintptr_t)b2 + (2 << 20)) & ~(uintptr_t)((2 << 20) - 1));
    madvise(r1, N, MADV_HUGEPAGE); madvise(r2, N, MADV_HUGEPAGE); memset(r1, 5, N); memset(r2, 6, N);
    CK(hipHostRegister(r1, N, 0)); CK(hipHostRegister(r2, N, 0)); }
  uint8_t *ph1 = h1, *ph2 = h2;
  for (int reg = 0; reg < 4; reg++) {
    h1 = (reg & 1) ? r1 : ph1; h2 = (reg & 2) ? r2 : ph2;
    printf("--- H2D source: %s, D2H destination: %s\n", (reg & 1) ? "REGISTERED" : "hipHostMalloc", (reg & 2) ? "REGISTERED" : "hipHostMalloc");
    run("H2D alone, copy engine", 1, 0);
    run("D2H alone, copy engine", 0, 1);
    run("H2D + D2H at once, copy engines", 1, 1);
  }
  h1 = ph1; h2 = ph2;
  run("H2D alone, copy engine", 1, 0);
  run("D2H alone, copy engine", 0, 1);
  run("H2D + D2H at once, copy engines", 1, 1);
  run("H2D alone, kernel reading pinned host memory (256 blocks)", 256, 0);
  run("H2D alone, kernel (1024 blocks)", 1024, 0);
  run("H2D alone, kernel (64 blocks)", 64, 0);
  run("D2H alone, kernel writing pinned host memory (256 blocks)", 0, 256);
  run("H2D kernel (256) + D2H copy engine", 256, 1);
  run("H2D kernel (64) + D2H copy engine", 64, 1);
  run("H2D copy engine + D2H kernel (256)", 1, 256);
  run("H2D kernel (256) + D2H kernel (256)", 256, 256);
  return 0;
}
