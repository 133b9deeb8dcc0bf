// zn_common.hpp — shared constants and small device helpers for the gfx950 kernels.
//
// Wire format and return conventions follow the reference byte for byte
// (reference csrc/zipnn_core.c:105-244,881-1028).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define ZN_WAVE 64

// huff0 constants (Cyan4973/FiniteStateEntropy lib/huf.h as shipped in zstd 1.4.8)
#define ZN_HUF_BLOCK_MAX (128u * 1024u)
#define ZN_HUF_LOG_MAX 12u
#define ZN_HUF_LOG_DEFAULT 11u
#define ZN_HUF_SYM_MAX 255u
#define ZN_FSE_LOG_MIN 5u
#define ZN_FSE_LOG_MAX 12u
#define ZN_WEIGHT_FSE_LOG 6u

// device-side status bits OR-ed into the per-call status word
#define ZN_DEV_BAD_TYPE 1u
#define ZN_DEV_CORRUPT 2u
#define ZN_DEV_SYNC_TIMEOUT 8u   // a workgroup gave up waiting for another workgroup of the same launch (the merge workgroups of a partial chunk, the one-pass encoder's look-back): not a verdict on the data
#define ZN_DEV_MISSPEC 4u  // the one-pass encoder's layout speculation (every plane but the last stored raw) did not hold: the caller runs the four-kernel encoder

// plane-chunk kinds resolved by the decoder from (type, stored size, plane length)
#define ZN_KIND_RAW 0u   // type 0, or type 1 with csize == plane_len (HUF_decompress memcpy rule)
#define ZN_KIND_RLE 1u   // type 1, csize == 1
#define ZN_KIND_HUF 2u   // type 1, real huff0 block
#define ZN_KIND_HUFS 3u  // huff0 block of a partial last chunk, already decoded into the tail scratch (4 padded streams)
#define ZN_TAIL_SEGPAD 32768u            // stride of a stream inside a tail-scratch slot (a huff0 block is ≤ 128 KiB)
#define ZN_TAIL_SLOT (4u * ZN_TAIL_SEGPAD)
#define ZN_TAIL_WG_MIN_PLANE 512u        // a partial chunk's plane shorter than this is not worth four workgroups: the merge workgroup decodes it serially

__device__ __forceinline__ uint32_t zn_hb32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

// Hand-over of a result between workgroups of ONE launch (MI355X_MICROARCH.md, inter-workgroup visibility): the producer's stores, a workgroup barrier,
// then on one lane an agent-scope release fence, a drained store queue and a relaxed agent-scope update of the flag word; the consumer polls the word with
// relaxed agent-scope loads on one lane (bounded: a launch never hangs on a flag), then an agent-scope acquire fence and a workgroup barrier before its loads.
#if defined(ZN_SIMT_EMULATOR)
#define ZN_FLAG_LOAD32(p) (*(volatile const uint32_t*)(p))
#define ZN_FLAG_ADD32(p, v) (*(volatile uint32_t*)(p) += (v))
#define ZN_FLAG_RELEASE() ((void)0)
#define ZN_FLAG_ACQUIRE() ((void)0)
#define ZN_FLAG_NAP() ((void)0)
#define ZN_FLAG_CLOCK() 0ull
#else
#define ZN_FLAG_LOAD32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ZN_FLAG_ADD32(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ZN_FLAG_RELEASE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define ZN_FLAG_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define ZN_FLAG_NAP() __builtin_amdgcn_s_sleep(8)
#define ZN_FLAG_CLOCK() __builtin_amdgcn_s_memrealtime()      /* 100 MHz */
#endif
// lane 0 of the calling wave waits until *p >= want; false after two seconds (the caller reports it: never a hang)
__device__ __forceinline__ bool zn_flag_wait(const uint32_t* p, uint32_t want) {
  const unsigned long long t0 = ZN_FLAG_CLOCK();
  for (;;) {
    if (ZN_FLAG_LOAD32(p) >= want) return true;
#if defined(ZN_SIMT_EMULATOR)
    return false;                                /* (blocks run one after the other: a flag that is not there yet never comes) */
#else
    if (ZN_FLAG_CLOCK() - t0 > 200000000ull) return false;
    ZN_FLAG_NAP();
#endif
  }
}

// Byte-granular loads for metadata and bit-stream heads/tails, where the address has
// no alignment guarantee (payload offsets are sums of arbitrary compressed sizes).
__device__ __forceinline__ uint32_t zn_ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t zn_ld32(const uint8_t* p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint64_t zn_ld64(const uint8_t* p) {
  return (uint64_t)zn_ld32(p) | ((uint64_t)zn_ld32(p + 4) << 32);
}
__device__ __forceinline__ void zn_st64(uint8_t* p, uint64_t v) {
  for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}

// Sign-bit rotate of the reference (csrc/data_manipulation_dtype16.c:10-20,145-155 and
// data_manipulation_dtype32.c:39-49,275-285), on one 32-bit word.
__device__ __forceinline__ uint32_t zn_rot_fwd16(uint32_t u) {
  return ((u << 1) & 0xFF00FF00u) | ((u >> 8) & 0x00800080u) | (u & 0x007F007Fu);
}
__device__ __forceinline__ uint32_t zn_rot_inv16(uint32_t u) {
  return ((u << 8) & 0x80008000u) | ((u >> 1) & 0x7F807F80u) | (u & 0x007F007Fu);
}
__device__ __forceinline__ uint32_t zn_rot_fwd32(uint32_t u) {
  return ((u << 1) & 0xFF000000u) | ((u >> 8) & 0x00800000u) | (u & 0x007FFFFFu);
}
__device__ __forceinline__ uint32_t zn_rot_inv32(uint32_t u) {
  return ((u << 8) & 0x80000000u) | ((u >> 1) & 0x7F800000u) | (u & 0x007FFFFFu);
}

// Geometry of one frame body, passed by value to every kernel.
// A pointer that comes out of a struct / table is "generic" to the compiler: its loads and stores become FLAT
// operations, which also count on the LDS counter (every LDS wait would then wait for HBM).  These pointers
// are global memory: say so.
// (ZN_LDS_PTR: the same for a pointer into LDS that crossed a real function call.)
#if defined(ZN_SIMT_EMULATOR)
#define ZN_GLOBAL_PTR(T, p) ((T*)(p))
#define ZN_LDS_PTR(T, p) ((T*)(p))
#else
#if defined(__HIP_DEVICE_COMPILE__)
#define ZN_LDS_PTR(T, p) ((T*)(__attribute__((address_space(3))) T*)(unsigned int)(unsigned long long)(p))
#else
#define ZN_LDS_PTR(T, p) ((T*)(p))       /* (host pass of the same source: never executed) */
#endif
#define ZN_GLOBAL_PTR(T, p) ((T*)(__attribute__((address_space(1))) T*)(unsigned long long)(p))   // (via an integer: a plain round trip is folded away)
#endif

struct ZnGeom {
  uint64_t n;         // original length in bytes
  uint64_t chunk;     // origChunkSize
  uint64_t K;         // number of chunks
  uint32_t P;         // planes (num_buf)
  uint32_t rot;       // bits_mode == 1 && P > 1
};

// length of chunk c and of its plane p (reference csrc/zipnn_core.c:1006-1027,
// data_manipulation_dtype16.c:70-75, data_manipulation_dtype32.c:81-90)
__device__ __forceinline__ uint32_t zn_chunk_len(const ZnGeom& g, uint64_t c) {
  return (uint32_t)((c == g.K - 1) ? (g.n - c * g.chunk) : g.chunk);
}
__device__ __forceinline__ uint32_t zn_plane_len(uint32_t chunk_len, uint32_t P, uint32_t p) {
  return chunk_len / P + (p < chunk_len % P ? 1u : 0u);
}

// The one developer-only switch left: ZN_PHASE_TIMERS(_SUB) adds shader-clock timers to the kernels (scripts/phase_profile.py).  It may not reach a product
// build: zipnn_amd/build.py never defines ZN_DEV_BUILD, and without it the macro is an error.  Round 6 removed the A/B switches of rounds 1-5 whose experiments are
// settled (about forty `#ifndef ZN_F_* / ZN_E_* / ZN_OP_*` defaults: they are plain constants now, the losing branches are gone; what each one measured is in
// profiles/r0[1-5]_*.txt and profiles/DESIGN_history_r1_r4.md, and scripts/ab_variants.py rebuilds any of them from the commit that still had it).
#if !defined(ZN_DEV_BUILD) && !defined(ZN_SIMT_EMULATOR)
#if defined(ZN_PHASE_TIMERS) || defined(ZN_PHASE_TIMERS_SUB)
#error "developer-only macro (ZN_PHASE_TIMERS) without ZN_DEV_BUILD: not a product configuration"
#endif
#endif

// ---------------------------------------------------------------------------
// Optional in-kernel phase timers (developer tool: scripts/phase_profile.py builds a second
// library with -DZN_PHASE_TIMERS; the shipped libzipnn_hip.so has none of this).  Thread 0 of
// each workgroup adds the shader-clock cycles spent since the previous mark to slot i.
// ---------------------------------------------------------------------------
#ifdef ZN_PHASE_TIMERS
static __device__ unsigned long long zn_phase_acc[64];   // one copy per translation unit; only the fused decode TU reads it back
// cycles are accumulated in registers and published once (global atomics inside the phases would
// be waited on by the kernel's own vmcnt waits and distort what they measure)
struct ZnPhaseTimer { unsigned long long t0; unsigned int a[24]; };
#define ZN_PT_DECL ZnPhaseTimer zn_pt_; do { for (int i_ = 0; i_ < 24; i_++) zn_pt_.a[i_] = 0; zn_pt_.t0 = __builtin_readcyclecounter(); } while (0)
// (scheduling barriers pin the clock read between the phases it separates)
#define ZN_PT(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); zn_pt_.a[i] += (unsigned int)(t_ - zn_pt_.t0); zn_pt_.t0 = t_; } while (0)
#define ZN_PT_COUNT(i, n) do { zn_pt_.a[i] += (unsigned int)(n); } while (0)
#define ZN_PT_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 24; i_++) if (zn_pt_.a[i_]) atomicAdd(&zn_phase_acc[i_], (unsigned long long)zn_pt_.a[i_]); } while (0)
// a callee that shares its caller's timer: extra parameter / argument / local reference
#define ZN_PT_PARAM , ZnPhaseTimer* zn_ptp_
#define ZN_PT_PASS , &zn_pt_
#define ZN_PT_SHARED ZnPhaseTimer& zn_pt_ = *zn_ptp_
#else
#define ZN_PT_PARAM
#define ZN_PT_PASS
#define ZN_PT_SHARED do { } while (0)
#define ZN_PT_DECL do { } while (0)
#define ZN_PT(i) do { } while (0)
#define ZN_PT_COUNT(i, n) do { } while (0)
#define ZN_PT_FLUSH() do { } while (0)
#endif
