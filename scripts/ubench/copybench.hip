// copybench.hip — developer probe: what hand-written copy / fill kernels get from HBM on an MI355X, by access shape.
// (the yardstick for DESIGN.md's "practical ceiling": same bytes as the decode / encode kernels, no compute)
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/copybench scripts/ubench/copybench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

// ST: 0 plain, 1 non-temporal.  LD likewise.
template <typename T, int LD, int ST>
__device__ __forceinline__ T ld(const T* p) { return LD ? __builtin_nontemporal_load(p) : *p; }
template <typename T, int ST>
__device__ __forceinline__ void st(T* p, T v) { if (ST) __builtin_nontemporal_store(v, p); else *p = v; }

// each workgroup owns a contiguous span of `span` bytes of the destination (and the proportional span of the source),
// walks it in steps of 256 lanes x sizeof(T) x U; rd_per_wr_256 = bytes read per 256 bytes written (fixed point)
template <typename T, int LD, int ST, int U>
__global__ __launch_bounds__(256) void k_span(const T* __restrict__ src, T* __restrict__ dst, size_t n_wr, size_t n_rd, size_t span_el) {
  const size_t spans = (n_wr > n_rd ? n_wr : n_rd) / span_el;
  T acc = {};
  for (size_t s = blockIdx.x; s < spans; s += gridDim.x) {
    const size_t b = s * span_el;
    for (size_t i = threadIdx.x; i < span_el; i += 256 * U) {
      T v[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const size_t j = b + i + (size_t)u * 256; v[u] = (j < n_rd) ? ld<T, LD, ST>(src + j) : acc; }
#pragma unroll
      for (int u = 0; u < U; u++) { const size_t j = b + i + (size_t)u * 256; if (j < n_wr) st<T, ST>(dst + j, v[u]); else acc ^= v[u]; }
    }
  }
  if (n_wr < n_rd) { uint32_t x = ((uint32_t*)&acc)[0]; if (x == 0x12345u) ((uint32_t*)dst)[0] = x; }
}

// grid-stride (each wave's consecutive accesses are a whole grid apart)
template <typename T, int LD, int ST, int U>
__global__ __launch_bounds__(256) void k_grid(const T* __restrict__ src, T* __restrict__ dst, size_t n_wr, size_t n_rd) {
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t n = n_wr > n_rd ? n_wr : n_rd;
  T acc = {};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += U * stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t j = i + u * stride; v[u] = (j < n_rd) ? ld<T, LD, ST>(src + j) : acc; }
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t j = i + u * stride; if (j < n_wr) st<T, ST>(dst + j, v[u]); else if (j < n) acc ^= v[u]; }
  }
  if (n_wr < n_rd) { uint32_t x = ((uint32_t*)&acc)[0]; if (x == 0x12345u) ((uint32_t*)dst)[0] = x; }
}

// decode-shaped: per 256 KiB "chunk" (one workgroup at a time): read C bytes (two streams: 43 KB "huffman" + 128 KiB "raw"),
// write 256 KiB as 4 quarters (wave w owns quarter w), 8 rows of 1 KiB per burst — the store pattern of zn_k_decode_fused
// the same with the waves' regions skewed against each other (SKEW bytes more than 64 KiB apart; a benchmark only: the regions overlap),
// BR rows per burst, and (ADJ) the four waves writing adjacent 1 KiB rows of one span instead of four quarters
template <int ST, int SKEW, int BR, int ADJ>
__global__ __launch_bounds__(256) void k_decode_shape2(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t chunks, size_t cbytes) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const uint8_t* s = src + c * cbytes + (size_t)wave * (cbytes / 4 & ~15ull);
    uint8_t* d = dst + c * 262144 + (ADJ ? (size_t)wave * 1024 : (size_t)wave * (65536 + SKEW));
    const int bursts = 64 / BR;
    const size_t rd_per_burst = ((cbytes / 4) / bursts) & ~15ull;
    for (int burst = 0; burst < bursts; burst++) {
      v4u v[BR];
      v4u a = {};
      for (size_t o = (size_t)lane * 16; o < rd_per_burst; o += 1024) a ^= __builtin_nontemporal_load((const v4u*)(s + burst * rd_per_burst + o));
#pragma unroll
      for (int r = 0; r < BR; r++) { v[r] = a; v[r].x += r; }
#pragma unroll
      for (int r = 0; r < BR; r++) {
        size_t off = ADJ ? ((size_t)(burst * BR + r) * 4096 + lane * 16) : ((size_t)burst * BR * 1024 + r * 1024 + lane * 16);
        if (!ADJ && off + 16 > 65536 - (size_t)(SKEW > 0 ? 3 * SKEW : 0)) off = lane * 16;
        v4u* p = (v4u*)(d + off); if (ST) __builtin_nontemporal_store(v[r], p); else *p = v[r];
      }
    }
  }
}

template <int ST>
__global__ __launch_bounds__(256) void k_decode_shape(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t chunks, size_t cbytes) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
    const uint8_t* s = src + c * cbytes + (size_t)wave * (cbytes / 4 & ~15ull);
    uint8_t* d = dst + c * 262144 + (size_t)wave * 65536;
    const size_t rd_per_burst = ((cbytes / 4) / 8) & ~15ull;     // bytes read per 8-row burst
    for (int burst = 0; burst < 8; burst++) {
      v4u v[8];
      // reads: rd_per_burst bytes, 16 B per lane
      v4u a = {};
      for (size_t o = (size_t)lane * 16; o < rd_per_burst; o += 1024) a ^= __builtin_nontemporal_load((const v4u*)(s + burst * rd_per_burst + o));
#pragma unroll
      for (int r = 0; r < 8; r++) { v[r] = a; v[r].x += r; }
#pragma unroll
      for (int r = 0; r < 8; r++) { v4u* p = (v4u*)(d + burst * 8192 + r * 1024 + lane * 16); if (ST) __builtin_nontemporal_store(v[r], p); else *p = v[r]; }
    }
  }
}

static double time_kernel(std::function<void()> f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
  return best;
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && (!strcmp(argv[1], "quick") || !strcmp(argv[1], "shapes"));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int CUS = pr.multiProcessorCount;
  const size_t N = (size_t)4 << 30;
  const size_t C = 2844282670ull & ~(size_t)255;
  uint8_t *a, *b; CK(hipMalloc(&a, N)); CK(hipMalloc(&b, N));
  CK(hipMemset(a, 1, N)); CK(hipMemset(b, 2, N));
  printf("# %s, %d CUs; best of 10; GB/s = (bytes read + bytes written) / time\n", pr.gcnArchName, CUS);
  printf("%-86s %9s %9s\n", "kernel", "ms", "GB/s");
  auto row = [&](const char* name, size_t bytes, std::function<void()> f) { const double ms = time_kernel(f, 10); printf("%-86s %9.3f %9.0f\n", name, ms, (double)bytes / ms / 1e6); fflush(stdout); };
  char nm[160];
  // hipMemcpyAsync D2D / hipMemsetAsync as the runtime's own yardstick
  row("hipMemcpyAsync D2D 4 GiB", 2 * N, [&] { CK(hipMemcpyAsync(b, a, N, hipMemcpyDeviceToDevice, 0)); });
  row("hipMemsetAsync 4 GiB", N, [&] { CK(hipMemsetAsync(b, 3, N, 0)); });
  for (int shape = 0; shape < 4; shape++) {          // 0 copy, 1 write only, 2 read only, 3 decode mix (C in, N out)
    const size_t nrd = shape == 1 ? 0 : (shape == 3 ? C : N), nwr = shape == 2 ? 0 : N;
    const char* sn[] = {"copy 4 GiB->4 GiB", "write 4 GiB", "read 4 GiB", "read 2.84 GB + write 4.29 GB"};
    for (int st_ = 0; st_ < 2; st_++) {
      for (size_t span : {(size_t)16384, (size_t)65536, (size_t)262144}) for (int wgpc : {4, 8, 16}) {
        if (quick && !(span == 65536 && wgpc == 8)) continue;
        snprintf(nm, sizeof nm, "%s: span %zu KiB per WG step, 16 B/lane x4, %s, %d WG/CU", sn[shape], span >> 10, st_ ? "nt" : "plain", wgpc);
        const int blocks = CUS * wgpc;
        if (st_) row(nm, nrd + nwr, [&] { hipLaunchKernelGGL((k_span<v4u, 1, 1, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, nwr / 16, nrd / 16, span / 16); });
        else row(nm, nrd + nwr, [&] { hipLaunchKernelGGL((k_span<v4u, 0, 0, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, nwr / 16, nrd / 16, span / 16); });
      }
      for (int wgpc : {8, 32}) {
        snprintf(nm, sizeof nm, "%s: grid-stride, 16 B/lane x4, %s, %d WG/CU", sn[shape], st_ ? "nt" : "plain", wgpc);
        const int blocks = CUS * wgpc;
        if (st_) row(nm, nrd + nwr, [&] { hipLaunchKernelGGL((k_grid<v4u, 1, 1, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, nwr / 16, nrd / 16); });
        else row(nm, nrd + nwr, [&] { hipLaunchKernelGGL((k_grid<v4u, 0, 0, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, nwr / 16, nrd / 16); });
      }
    }
  }
  // one-grid-per-element launches (what torch's fill_/copy_ do): grid = n / (256 * U)
  {
    const size_t n16 = N / 16;
    row("copy: one pass, no loop (grid = n/1024 WGs), 16 B x4, plain", 2 * N, [&] { hipLaunchKernelGGL((k_span<v4u, 0, 0, 4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, n16, n16, (size_t)1024); });
    row("copy: one pass, no loop (grid = n/1024 WGs), 16 B x4, nt", 2 * N, [&] { hipLaunchKernelGGL((k_span<v4u, 1, 1, 4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, n16, n16, (size_t)1024); });
    row("write: one pass, no loop (grid = n/1024 WGs), 16 B x4, plain", N, [&] { hipLaunchKernelGGL((k_span<v4u, 0, 0, 4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, n16, (size_t)0, (size_t)1024); });
    row("write: one pass, no loop (grid = n/1024 WGs), 16 B x4, nt", N, [&] { hipLaunchKernelGGL((k_span<v4u, 1, 1, 4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, n16, (size_t)0, (size_t)1024); });
  }
  for (int st_ = 0; st_ < 2; st_++) for (int wgpc : {4, 8, 16, 64}) {
    snprintf(nm, sizeof nm, "decode-shaped per-chunk pattern (16384 chunks, 173.6 KB in, 256 KiB out), %s stores, %d WG/CU", st_ ? "nt" : "plain", wgpc);
    const int blocks = wgpc == 64 ? 16384 : CUS * wgpc;
    if (st_) row(nm, 16384ull * 173600 + N, [&] { hipLaunchKernelGGL((k_decode_shape<1>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
    else row(nm, 16384ull * 173600 + N, [&] { hipLaunchKernelGGL((k_decode_shape<0>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
  }
  {
    const size_t bytes = 16384ull * 173600 + N; const int blocks = CUS * 4;
    row("decode-shaped, nt, 4 WG/CU: 8 rows per burst, quarters 64 KiB apart (as above)", bytes, [&] { hipLaunchKernelGGL((k_decode_shape2<1, 0, 8, 0>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
    row("decode-shaped, nt, 4 WG/CU: 8 rows per burst, quarters 64 KiB + 4352 B apart", bytes, [&] { hipLaunchKernelGGL((k_decode_shape2<1, 4352, 8, 0>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
    row("decode-shaped, nt, 4 WG/CU: 8 rows per burst, quarters 64 KiB + 1280 B apart", bytes, [&] { hipLaunchKernelGGL((k_decode_shape2<1, 1280, 8, 0>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
    row("decode-shaped, nt, 4 WG/CU: 4 rows per burst, quarters 64 KiB apart", bytes, [&] { hipLaunchKernelGGL((k_decode_shape2<1, 0, 4, 0>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
    row("decode-shaped, nt, 4 WG/CU: 2 rows per burst, quarters 64 KiB apart", bytes, [&] { hipLaunchKernelGGL((k_decode_shape2<1, 0, 2, 0>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
    row("decode-shaped, nt, 4 WG/CU: 16 rows per burst, quarters 64 KiB apart", bytes, [&] { hipLaunchKernelGGL((k_decode_shape2<1, 0, 16, 0>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
    row("decode-shaped, nt, 4 WG/CU: 8 rows per burst, the four waves' rows ADJACENT (one 4 KiB span per step)", bytes, [&] { hipLaunchKernelGGL((k_decode_shape2<1, 0, 8, 1>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
    row("decode-shaped, nt, 4 WG/CU: 4 rows per burst, the four waves' rows ADJACENT", bytes, [&] { hipLaunchKernelGGL((k_decode_shape2<1, 0, 4, 1>), dim3(blocks), dim3(256), 0, 0, a, b, (size_t)16384, (size_t)173600); });
  }
  {
    // per-lane access width: the decode reads its raw rows 8 bytes per lane and its stream tiles 4 bytes per lane
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    const int blocks = CUS * 4;
    row("read 4 GiB: span 64 KiB per WG step, 16 B/lane x4, nt, 4 WG/CU", N, [&] { hipLaunchKernelGGL((k_span<v4u, 1, 1, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, (size_t)0, N / 16, (size_t)65536 / 16); });
    row("read 4 GiB: span 64 KiB per WG step,  8 B/lane x4, nt, 4 WG/CU", N, [&] { hipLaunchKernelGGL((k_span<v2u, 1, 1, 4>), dim3(blocks), dim3(256), 0, 0, (const v2u*)a, (v2u*)b, (size_t)0, N / 8, (size_t)65536 / 8); });
    row("read 4 GiB: span 64 KiB per WG step,  8 B/lane x8, nt, 4 WG/CU", N, [&] { hipLaunchKernelGGL((k_span<v2u, 1, 1, 8>), dim3(blocks), dim3(256), 0, 0, (const v2u*)a, (v2u*)b, (size_t)0, N / 8, (size_t)65536 / 8); });
    row("read 4 GiB: span 64 KiB per WG step,  4 B/lane x4, nt, 4 WG/CU", N, [&] { hipLaunchKernelGGL((k_span<uint32_t, 1, 1, 4>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, (size_t)0, N / 4, (size_t)65536 / 4); });
    row("read 4 GiB: span 64 KiB per WG step,  4 B/lane x8, nt, 4 WG/CU", N, [&] { hipLaunchKernelGGL((k_span<uint32_t, 1, 1, 8>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, (size_t)0, N / 4, (size_t)65536 / 4); });
    row("read 4 GiB: span 64 KiB per WG step,  4 B/lane x16, nt, 4 WG/CU", N, [&] { hipLaunchKernelGGL((k_span<uint32_t, 1, 1, 16>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, (size_t)0, N / 4, (size_t)65536 / 4); });
  }
  return 0;
}
