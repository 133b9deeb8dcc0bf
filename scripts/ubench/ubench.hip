// ubench.hip — developer microbenchmarks for gfx950 (not part of the product): what one wave-instruction costs on the
// VALU / LDS pipes of an MI355X CU, whether byte-unaligned DS accesses are cheap, and what a hand-written copy gets
// from HBM.  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/ubench scripts/ubench/ubench.hip
// Run on the GPU box: scripts/ubench/ubench [valu|lds|chain|copy|all]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------
// VALU issue rate: 8 independent accumulators x 8 unrolled = 64 instructions per loop trip
// ------------------------------------------------------------------------------------------------
#define R8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
#define R64(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S)

template <int OP>
__global__ __launch_bounds__(256) void k_valu(uint32_t* out, uint64_t* cyc, int trips, uint32_t kk) {
  uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
  uint32_t k = kk;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < trips; i++) {
    if (OP == 0) {
#define S(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 1) {
#define S(x) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(b##x) : "v"(k));
#define ba0 b0
#define ba1 b1
#define ba2 b2
#define ba3 b3
#define ba4 b4
#define ba5 b5
#define ba6 b6
#define ba7 b7
      R64(S)
#undef S
    } else if (OP == 2) {
#define S(x) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 3) {
#define S(x) asm volatile("v_bfe_u32 %0, %0, %1, 7" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 4) {
#define S(x) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 5) {
#define S(x) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 6) {
#define S(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : );
      R64(S)
#undef S
    } else if (OP == 7) {
#define S(x) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 8) {
#define S(x) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
      R64(S)
#undef S
    } else if (OP == 9) {
#define S(x) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc");
      R64(S)
#undef S
    } else if (OP == 10) {
#define S(x) asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 11) {
#define S(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 12) {
#define S(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 13) {
#define S(x) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(b##x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 14) {
#define S(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    } else if (OP == 15) {
#define S(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
      R64(S)
#undef S
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7);
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------
// LDS: 8 accesses per trip at per-lane addresses held in registers (random or conflict-free), one wait per trip
// ------------------------------------------------------------------------------------------------
// MODE: 0 ds_read_b32, 1 ds_read_b64, 2 ds_write_b32, 3 ds_or_b32 (no return), 4 ds_write_b64, 5 ds_read_b128, 6 ds_read_u8, 7 ds_write_b8
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, uint64_t* cyc, int trips, int pattern, uint32_t span) {
  extern __shared__ uint8_t lds[];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  for (uint32_t i = tid; i < span / 4; i += blockDim.x) ((uint32_t*)lds)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t ad[8];
  uint32_t h = (blockIdx.x * 256 + tid) * 2654435761u + 12345u;
  const uint32_t width = (MODE == 1 || MODE == 4) ? 8u : (MODE == 5) ? 16u : (MODE >= 6) ? 1u : 4u;
  for (int j = 0; j < 8; j++) {
    h ^= h << 13; h ^= h >> 17; h ^= h << 5;
    uint32_t a;
    if (pattern == 0) a = ((lane * width) + (uint32_t)j * 64u * width) % span;                    // conflict-free, aligned
    else if (pattern == 1) a = (h % (span / width)) * width;                                      // random, aligned
    else if (pattern == 2) a = (h % (span - 16u));                                                // random, byte-unaligned
    else if (pattern == 3) a = ((lane * width) + (uint32_t)j * 64u * width + 1u) % (span - 16u);  // contiguous, off by one byte
    else a = ((lane * 52u) + (uint32_t)j * 5u) % (span - 16u);                                    // ~52-byte stride (packed output of ~49 symbols), unaligned
    ad[j] = a;
  }
  uint32_t acc = 0; uint64_t acc64 = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < trips; i++) {
    if (MODE == 0) {
      uint32_t r[8];
      for (int j = 0; j < 8; j++) asm volatile("ds_read_b32 %0, %1" : "=v"(r[j]) : "v"(ad[j]));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
      for (int j = 0; j < 8; j++) acc ^= r[j];
    } else if (MODE == 1) {
      uint64_t r[8];
      for (int j = 0; j < 8; j++) asm volatile("ds_read_b64 %0, %1" : "=v"(r[j]) : "v"(ad[j]));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
      for (int j = 0; j < 8; j++) acc64 ^= r[j];
    } else if (MODE == 2) {
      for (int j = 0; j < 8; j++) asm volatile("ds_write_b32 %0, %1" :: "v"(ad[j]), "v"(acc + j) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 3) {
      for (int j = 0; j < 8; j++) asm volatile("ds_or_b32 %0, %1" :: "v"(ad[j] & ~3u), "v"(acc + j) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 4) {
      for (int j = 0; j < 8; j++) asm volatile("ds_write_b64 %0, %1" :: "v"(ad[j]), "v"(acc64 + j) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 5) {
      typedef uint32_t v4 __attribute__((ext_vector_type(4)));
      v4 r[8];
      for (int j = 0; j < 8; j++) asm volatile("ds_read_b128 %0, %1" : "=v"(r[j]) : "v"(ad[j]));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
      for (int j = 0; j < 8; j++) acc ^= r[j].x ^ r[j].w;
    } else if (MODE == 6) {
      uint32_t r[8];
      for (int j = 0; j < 8; j++) asm volatile("ds_read_u8 %0, %1" : "=v"(r[j]) : "v"(ad[j]));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
      for (int j = 0; j < 8; j++) acc ^= r[j];
    } else if (MODE == 7) {
      for (int j = 0; j < 8; j++) asm volatile("ds_write_b8 %0, %1" :: "v"(ad[j]), "v"(acc + j) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  __syncthreads();
  out[blockIdx.x * blockDim.x + tid] = acc ^ (uint32_t)acc64 ^ (uint32_t)(acc64 >> 32) ^ ((uint32_t*)lds)[tid];
  if (lane == 0) cyc[(blockIdx.x * blockDim.x + tid) >> 6] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------
// dependent decode-like chain: idx = hi >> 21; e = lut[idx]; w <<= e (6 LSBs); acc += e; every 5th step a refill
// (one unaligned ds_read_b64 of the "stream"): what a step costs in latency (1 wave/SIMD) and in throughput (4-8 waves/SIMD)
// ------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(256) void k_chain(uint32_t* out, uint64_t* cyc, int trips) {
  __shared__ uint32_t lut[2048 * 2];
  __shared__ uint32_t strm[4][1024];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t h = 99991u * (tid + 1) + blockIdx.x;
  for (uint32_t i = tid; i < 2048; i += 256) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; const uint32_t nb = 7u + (h % 5u), cnt = 2u + ((h >> 8) % 3u); lut[2 * i] = nb | (cnt << 8); lut[2 * i + 1] = h; }
  for (uint32_t i = tid; i < 4096; i += 256) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; ((uint32_t*)strm)[i] = h; }
  __syncthreads();
  const uint8_t* sb = (const uint8_t*)strm[wave];
  uint32_t* ring = strm[wave];
  uint64_t w = ((uint64_t)h << 32) | tid;
  uint32_t acc = 0, pos = 4000 * 8 - lane * 128;
  uint32_t wp = lane * 52u;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < trips; i++) {
    // refill: 8 bytes below byte ceil(pos/8)
    {
      const uint32_t b = ((pos & 32767u) + 7u) >> 3;
      const uint32_t sh = 8u * b - (pos & 32767u);
      uint64_t x;
      typedef uint64_t __attribute__((aligned(1))) u64u;
      x = *(const u64u*)(sb + ((b + 4088u) & 4095u & ~0u) % 4088u);
      w = x << sh;
    }
#pragma unroll
    for (int s = 0; s < 5; s++) {
      const uint32_t idx = (uint32_t)(w >> 53);
      if (VARIANT == 0) {                       // count-like: meta only
        const uint32_t e = lut[2 * idx];
        w <<= (e & 63u);
        acc += e;
      } else if (VARIANT == 1) {                // write-like: meta + symbols (one 8-byte read), unaligned 4-byte store
        const uint64_t e2 = *(const uint64_t*)&lut[2 * idx];
        const uint32_t e = (uint32_t)e2;
        w <<= (e & 63u);
        acc += e;
        typedef uint32_t __attribute__((aligned(1))) u32u;
        *(u32u*)((uint8_t*)ring + ((wp + (acc >> 8)) & 4091u)) = (uint32_t)(e2 >> 32);
      } else {                                  // today's write step: two reads, 64-bit pack shift, two conditional atomics
        const uint32_t e = lut[2 * idx], sy = lut[2 * idx + 1];
        w <<= (e & 63u);
        acc += e;
        const uint32_t wpos = (wp + (acc >> 8)) & 4091u;
        const uint64_t sp = ((uint64_t)sy) << ((wpos & 3u) << 3);
        uint32_t* d = ring + (wpos >> 2);
        if ((uint32_t)sp) atomicOr(d, (uint32_t)sp);
        if ((uint32_t)(sp >> 32)) atomicOr(d + 1, (uint32_t)(sp >> 32));
      }
    }
    pos -= (acc & 255u); acc &= ~255u;
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  __syncthreads();
  out[blockIdx.x * 256 + tid] = acc ^ (uint32_t)w ^ ring[lane];
  if (lane == 0) cyc[(blockIdx.x * 256 + tid) >> 6] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------
// HBM copy ceilings: W bytes per lane per access, grid-stride, plain or non-temporal
// ------------------------------------------------------------------------------------------------
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
template <typename T, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const T* __restrict__ src, T* __restrict__ dst, size_t n_read, size_t n_write) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = n_read > n_write ? n_read : n_write;
  T acc = {};
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    T v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { const size_t j = i + u * stride; if (j < n_read) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j]; else v[u] = acc; }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) { const size_t j = i + u * stride; if (j < n_write) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; } else acc ^= v[u]; }
  }
  for (; i < n; i += stride) { if (i < n_read) { T v = src[i]; if (i < n_write) dst[i] = v; else acc ^= v; } else dst[i] = acc; }
  if (n_write < n_read) { uint32_t x = ((uint32_t*)&acc)[0]; if (x == 0x12345u) ((uint32_t*)dst)[0] = x; }   // keep the loads
}

static double time_kernel(std::function<void()> f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) f();
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms; }
  (void)sum;
  return best;
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int CUS = pr.multiProcessorCount;
  printf("# device %s, %d CUs, clock %d MHz\n", pr.gcnArchName, CUS, pr.clockRate / 1000);
  uint32_t* d_out; uint64_t* d_cyc; CK(hipMalloc(&d_out, 64 << 20)); CK(hipMalloc(&d_cyc, 8 << 20));
  std::vector<uint64_t> h_cyc(1 << 20);
  auto all = [&](const char* s) { return !strcmp(what, "all") || !strcmp(what, s); };

  if (all("valu")) {
    const char* names[] = {"v_add_u32", "v_lshlrev_b64", "v_perm_b32", "v_bfe_u32", "v_and_or_b32", "v_lshl_add_u32", "v_cndmask_b32", "v_lshrrev_b32",
                           "v_add_u32_dpp", "v_cmp+v_cndmask", "v_alignbit_b32", "v_mov_b32", "v_pk_add_u16", "v_lshrrev_b64", "v_mad_u32_u24", "v_add3_u32"};
    printf("# VALU: cycles per wave-instruction per SIMD (shader clock, from s_memtime inside the kernel), by waves per SIMD\n");
    printf("%-18s %8s %8s %8s %8s\n", "op", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD");
    for (int op = 0; op < 16; op++) {
      printf("%-18s", names[op]);
      for (int wps = 1; wps <= 8; wps *= 2) {
        const int trips = 256; const int blocks = CUS * wps;     // 256-thread blocks: 4 waves = 1 per SIMD
        auto launch = [&]() {
#define L(OP_) case OP_: hipLaunchKernelGGL((k_valu<OP_>), dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, trips, 3u); break;
          switch (op) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) }
#undef L
        };
        launch(); launch(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_cyc.data(), d_cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < blocks * 4; i++) s += (double)h_cyc[i];
        const double per_wave = s / (blocks * 4);
        const double ninstr = (double)trips * 64 * (op == 9 ? 2 : 1);
        printf(" %8.2f", per_wave / ninstr / wps);                 // cycles per instruction per SIMD (wps waves share a SIMD)
      }
      printf("\n");
    }
  }

  if (all("lds")) {
    const char* names[] = {"ds_read_b32", "ds_read_b64", "ds_write_b32", "ds_or_b32", "ds_write_b64", "ds_read_b128", "ds_read_u8", "ds_write_b8"};
    const char* pats[] = {"linear", "random-aligned", "random-unaligned", "linear+1B", "stride52-unaligned"};
    printf("# LDS: cycles per wave-instruction per CU (s_memtime), 8 accesses per wait, span 16 KiB, by waves per CU\n");
    printf("%-14s %-20s %8s %8s %8s\n", "op", "pattern", "4w/CU", "8w/CU", "16w/CU");
    for (int m = 0; m < 8; m++) for (int p = 0; p < 5; p++) {
      if ((m == 3) && (p >= 2)) continue;                          // atomics are dword-aligned by construction
      if ((m == 5) && (p >= 2)) continue;
      printf("%-14s %-20s", names[m], pats[p]);
      for (int bpc = 1; bpc <= 4; bpc *= 2) {
        const int trips = 512; const int blocks = CUS * bpc;
        auto launch = [&]() {
#define L(M_) case M_: hipLaunchKernelGGL((k_lds<M_>), dim3(blocks), dim3(256), 16384, 0, d_out, d_cyc, trips, p, 16384u); break;
          switch (m) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) }
#undef L
        };
        launch(); launch(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_cyc.data(), d_cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < blocks * 4; i++) s += (double)h_cyc[i];
        const double per_wave = s / (blocks * 4);
        printf(" %8.2f", per_wave / (trips * 8.0) / (bpc * 4));    // cycles per wave-instruction per CU
      }
      printf("\n");
    }
  }

  if (all("chain")) {
    const char* names[] = {"count-like (1 read)", "write-like (b64 read + unaligned store)", "today's write (2 reads, 2 cond. atomics)"};
    printf("# decode-like dependent chain: cycles per STEP per wave (latency at 1 wave/SIMD), and per step per SIMD at higher occupancy\n");
    printf("%-44s %10s %10s %10s\n", "variant", "1w/SIMD", "2w/SIMD", "4w/SIMD");
    for (int v = 0; v < 3; v++) {
      printf("%-44s", names[v]);
      for (int wps = 1; wps <= 4; wps *= 2) {
        const int trips = 200; const int blocks = CUS * wps;
        auto launch = [&]() {
          if (v == 0) hipLaunchKernelGGL((k_chain<0>), dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, trips);
          else if (v == 1) hipLaunchKernelGGL((k_chain<1>), dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, trips);
          else hipLaunchKernelGGL((k_chain<2>), dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, trips);
        };
        launch(); launch(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h_cyc.data(), d_cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < blocks * 4; i++) s += (double)h_cyc[i];
        const double per_wave = s / (blocks * 4);
        printf(" %10.1f", per_wave / (trips * 5.0) / wps);
      }
      printf("\n");
    }
  }

  if (all("copy")) {
    const size_t N = (size_t)4 << 30;
    uint8_t *a, *b; CK(hipMalloc(&a, N)); CK(hipMalloc(&b, N));
    CK(hipMemset(a, 1, N)); CK(hipMemset(b, 2, N));
    printf("# HBM ceilings, hand-written grid-stride kernels (best of 10), 4 GiB buffers\n");
    printf("%-52s %10s %10s\n", "kernel", "ms", "GB/s");
    struct Case { const char* name; int w; bool nt; size_t rd, wr; };
    const size_t C = 2844282670ull & ~(size_t)15;       // compressed payload of the 4 GiB config
    Case cases[] = {
      {"copy 16B/lane plain        (4 GiB -> 4 GiB)", 16, false, N, N},
      {"copy 16B/lane non-temporal (4 GiB -> 4 GiB)", 16, true, N, N},
      {"copy  8B/lane non-temporal (4 GiB -> 4 GiB)", 8, true, N, N},
      {"copy  4B/lane non-temporal (4 GiB -> 4 GiB)", 4, true, N, N},
      {"read  16B/lane nt (4 GiB)", 16, true, N, 0},
      {"read   8B/lane nt (4 GiB)", 8, true, N, 0},
      {"read   4B/lane nt (4 GiB)", 4, true, N, 0},
      {"write 16B/lane nt (4 GiB)", 16, true, 0, N},
      {"decode-shaped: read 2.84 GB + write 4.29 GB, 16B nt", 16, true, C, N},
      {"encode-shaped: read 4.29 GB + write 2.84 GB, 16B nt", 16, true, N, C},
    };
    for (auto& c : cases) {
      for (int blocks_per_cu : {8, 16, 32}) {
        const int blocks = CUS * blocks_per_cu;
        auto launch = [&]() {
          if (c.w == 16) { if (c.nt) hipLaunchKernelGGL((k_copy<v4u, true, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, c.rd / 16, c.wr / 16); else hipLaunchKernelGGL((k_copy<v4u, false, 4>), dim3(blocks), dim3(256), 0, 0, (const v4u*)a, (v4u*)b, c.rd / 16, c.wr / 16); }
          else if (c.w == 8) hipLaunchKernelGGL((k_copy<v2u, true, 4>), dim3(blocks), dim3(256), 0, 0, (const v2u*)a, (v2u*)b, c.rd / 8, c.wr / 8);
          else hipLaunchKernelGGL((k_copy<uint32_t, true, 4>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, c.rd / 4, c.wr / 4);
        };
        const double ms = time_kernel(launch, 10);
        char nm[96]; snprintf(nm, sizeof nm, "%s [%d WG/CU]", c.name, blocks_per_cu);
        printf("%-60s %10.3f %10.0f\n", nm, ms, (double)(c.rd + c.wr) / ms / 1e6);
      }
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
