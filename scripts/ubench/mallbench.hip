// mallbench.hip — developer probe (round 5): can a one-pass encoder re-read its chunk from the Infinity Cache?
// The fused encoder would read a 256 KiB chunk (histogram), build its code table (tens of microseconds), and read the chunk AGAIN to emit —
// "2 N + C" of HBM traffic unless the second read is served on chip.  This kernel has the access pattern and nothing else: one workgroup per
// chunk reads it (16-byte loads, four in flight per thread), waits `delay` microseconds, re-reads it (MODE 0), or re-reads the same chunk of
// ANOTHER buffer nobody has touched (MODE 2: the second read certainly comes from HBM), or nothing (MODE 1), and writes 0.66 x the chunk.
// Workgroups per CU are set with dynamic LDS.  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/mallbench scripts/ubench/mallbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define CHUNK (256u * 1024u)

template <int MODE, int NT1, int NT2>
__global__ __launch_bounds__(256) void k_reread(const uint8_t* __restrict__ src, const uint8_t* __restrict__ other, uint8_t* __restrict__ dst,
                                                uint32_t* __restrict__ ticket, uint32_t delay_100ns, uint32_t out_bytes) {
  extern __shared__ uint32_t lds[];
  if (threadIdx.x == 0) lds[0] = atomicAdd(ticket, 1u);
  __syncthreads();
  const uint32_t c = lds[0];
  const v4u* p = (const v4u*)(src + (size_t)c * CHUNK);
  v4u acc = {0, 0, 0, 0};
  for (uint32_t i = threadIdx.x; i < CHUNK / 16u; i += 1024u) {
    v4u v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = NT1 ? __builtin_nontemporal_load(p + i + 256u * u) : p[i + 256u * u];
#pragma unroll
    for (int u = 0; u < 4; u++) acc ^= v[u];
  }
  __syncthreads();
  if (delay_100ns) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < delay_100ns / 1u) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
  const v4u* q = (MODE == 2) ? (const v4u*)(other + (size_t)c * CHUNK) : p;
  v4u* o = (v4u*)(dst + (size_t)c * out_bytes);
  const uint32_t nout = out_bytes / 16u;
  for (uint32_t i = threadIdx.x; i < CHUNK / 16u; i += 1024u) {
    v4u v[4];
    if (MODE != 1) {
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = NT2 ? __builtin_nontemporal_load(q + i + 256u * u) : q[i + 256u * u];
    } else {
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) { acc ^= v[u]; const uint32_t j = i + 256u * u; if (j < nout) __builtin_nontemporal_store(acc, o + j); }
  }
}

__global__ void k_spin(uint32_t ticks, uint64_t* out) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = __builtin_readcyclecounter() - c0;
}

int main(int argc, char** argv) {
  const size_t n = (argc > 1 ? (size_t)atol(argv[1]) : 4096) << 20;        // MiB
  const uint32_t nchunks = (uint32_t)(n / CHUNK), out_bytes = 173u * 1024u;
  uint8_t *src, *other, *dst; uint32_t* ticket;
  CK(hipMalloc(&src, n)); CK(hipMalloc(&other, n)); CK(hipMalloc(&dst, (size_t)nchunks * out_bytes)); CK(hipMalloc(&ticket, 4));
  CK(hipMemset(src, 1, n)); CK(hipMemset(other, 2, n));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int mode, int nt1, int nt2, uint32_t lds_bytes, uint32_t delay_us) {
    float best = 1e9f;
    for (int r = 0; r < 6; r++) {
      CK(hipMemsetAsync(ticket, 0, 4, 0));
      CK(hipEventRecord(e0, 0));
#define GO(M, A, B) hipLaunchKernelGGL((k_reread<M, A, B>), dim3(nchunks), dim3(256), lds_bytes, 0, src, other, dst, ticket, delay_us * 100u, out_bytes)   /* s_memrealtime: 100 ticks per microsecond (calibrated below) */
      if (mode == 0) { if (nt1 && nt2) GO(0, 1, 1); else if (nt1) GO(0, 1, 0); else if (nt2) GO(0, 0, 1); else GO(0, 0, 0); }
      else if (mode == 1) { if (nt1) GO(1, 1, 0); else GO(1, 0, 0); }
      else { if (nt1 && nt2) GO(2, 1, 1); else if (nt1) GO(2, 1, 0); else if (nt2) GO(2, 0, 1); else GO(2, 0, 0); }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0 && ms < best) best = ms;
    }
    return best;
  };
  const uint32_t lds_for[] = {0, 0, 80 * 1024, 53 * 1024, 40 * 1024, 32 * 1024, 26 * 1024, 0, 20 * 1024};
  {   // what is one tick of s_memrealtime?  (one thread spins for 100000 ticks; the launch is timed with events)
    uint64_t* d; CK(hipMalloc(&d, 16)); uint64_t h[2];
    for (int r = 0; r < 2; r++) {
      CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(k_spin, dim3(1), dim3(1), 0, 0, 100000u, d); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
      printf("# calibration: %llu ticks of s_memrealtime (%llu shader cycles) took %.3f ms -> %.1f ticks per us\n", (unsigned long long)h[0], (unsigned long long)h[1], ms, h[0] / (ms * 1e3));
    }
  }
  printf("# %zu MiB, %u chunks; time in ms per launch (best of 5): no re-read | re-read SAME chunk | re-read from an untouched buffer (HBM)\n", n >> 20, nchunks);
  for (int nt = 0; nt < 4; nt++)
    for (int occ : {2, 3, 4, 5, 8})
      for (uint32_t d : {0u, 10u, 20u, 40u, 80u, 120u}) {
        const int nt1 = nt & 1, nt2 = nt >> 1;
        const float a = run(1, nt1, 0, lds_for[occ], d), b = run(0, nt1, nt2, lds_for[occ], d), c = run(2, nt1, nt2, lds_for[occ], d);
        printf("first read %s, second read %s, %d wg/CU (%3u MiB in flight), delay %3u us:  %7.3f | %7.3f | %7.3f   saved %4.0f %% of the second read\n",
               nt1 ? "nt   " : "plain", nt2 ? "nt   " : "plain", occ, occ * 256u * 256u >> 10, d, a, b, c, 100.0 * (c - b) / (c - a > 1e-6 ? c - a : 1e-6));
        fflush(stdout);
      }
  return 0;
}
