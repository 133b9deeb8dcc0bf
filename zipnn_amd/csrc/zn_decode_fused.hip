// zn_decode_fused.hip — the bandwidth path of decompress: one kernel, one pass over HBM.
//
// One workgroup (4 waves) per GROUP of up to 4 consecutive full chunks whose planes are raw/RLE plus at
// most one huff0 block (what real weights look like: bf16/fp32 exponent plane Huffman-coded, mantissa
// planes stored raw).  Inside a chunk, wave w owns quarter w = stream w of the huff0 block.
//
//   1. one thread per (chunk, plane) parses the metadata; wave j turns chunk j's tree description into
//      the canonical symbol order (zn_huf_wave.hpp: FSE chain on the scalar ALU, everything else with
//      ballots) — four serial jobs side by side; then, chunk by chunk, all 256 threads fill the
//      single-symbol LUT and from it the MULTI-symbol LUT: one 8-byte entry per 11-bit window = the (up to) 4
//      symbols that start in it + a meta word (bits they take, their count, the bit offsets of symbols 2-4).
//   2. each wave decodes its backward bit-stream IN PARALLEL ACROSS ITS 64 LANES.  huff0 has no gap array,
//      so this uses Huffman self-synchronisation, format-transparently: the stream is cut into tiles of 64
//      sub-blocks of D dwords (D from the stream's average code length; D = 4 is the compile-time instance).
//      The register-resident form (zn_decode_chain.hpp) takes a tile like this: lane k guesses a start 22 bits
//      above its sub-block and runs into it ("sync": two whole-group steps + a boundary step on one window
//      refill); then every lane decodes its sub-block ONCE, each step's 4 symbols and count going into named
//      registers (zn_pass1: refill + whole-group steps, closed by boundary steps that take exactly the symbols
//      that start above the sub-block's end); a wave shuffle checks that every lane's exit position is the next
//      lane's start (a mismatch re-runs the pass from the exact positions; the top lane always starts from the
//      true position carried from the previous tile, so the result is exact); a prefix sum of the counts gives
//      each lane its output offset and zn_pass2 shifts the recorded groups into place (v_alignbyte_b32) and ORs
//      them into a small LDS staging buffer (ds_or_b32, aligned dwords only: neighbouring lanes share boundary
//      dwords).  What that form does not take (more steps than it has registers for, a chain four iterations do
//      not close, a tile denser than the staging buffer, a run-time D) is decoded by the looping form: a
//      counting pass and a writing pass over the same chain (zn_fused_run), in lane groups when needed.
//   3. the number of complete output rows is known after the prefix sum, so the raw planes' bytes for
//      exactly those rows are requested from HBM before the compaction and consumed after it: staging
//      bytes + raw bytes get the sign-bit rotate undone at plane level, are byte-interleaved with
//      v_perm_b32 and go out as coalesced 16-byte stores.  The next tile of the stream is prefetched
//      into registers the same way and copied into its LDS buffer inside this flush — behind the flush's one
//      wait, ahead of its stores — so that no wave ever waits for a store to be acknowledged (gfx9 counts loads
//      and stores on one in-order counter).  Decoded symbols never touch HBM; the float stream is written once.
//   4. wave priorities (s_setprio) rise with the progress through a tile: sync 1, decode 2, compaction 3, the rest 0 —
//      the passes are chains of dependent LDS look-ups and should issue ahead of the other waves' flush code.
//
// A launch decodes one tensor or a batch (segment table, zn_internal.hpp); the Huffman planes of partial
// last chunks are decoded by extra workgroups at the front of the same grid (zn_decode_tail_wg).
//
// Algorithmic HBM traffic per chunk: stored bytes in + chunk bytes out.
// Chunks this kernel does not take (partial tail, ≥2 Huffman planes, tableLog 12, odd chunk
// sizes) are left to zn_decode_generic.hip via the per-chunk `done` flag.
//
// Replaces, for those chunks: decompression_chunk_worker (reference csrc/zipnn_core.c:768-861),
// HUF_decompress (:807), combine_buffers_dtype16/32 + revert_all_floats_* (data_manipulation_
// dtype16.c:145-216, data_manipulation_dtype32.c:275-294,391-456).
#include "zn_internal.hpp"
#include <atomic>
#include "zn_huf_wave.hpp"
#include "zn_decode_common.hpp"
#include "zn_decode_rest.hpp"

#define ZN_F_THREADS 256
#define ZN_F_RING_BYTES 4096u            // per wave; multiple of every row size (512 / 1024 symbols)
#define ZN_F_WAVES_PER_SIMD 4            // workgroups per CU the LDS budget allows (drives the VGPR budget)
#define ZN_F_RING_DW (ZN_F_RING_BYTES / 4u)
#define ZN_F_DMAX 6                      // largest sub-block (dwords); sizes the stream-tile buffer and its prefetch registers (LDS: 40.8 KB, still 4 workgroups/CU; dense codes — fp8, fp16 — want long sub-blocks: +22 % on fp8)
// Dense codes (fp16's top byte, fp8: 5-6 bits a symbol) would get sub-blocks of 6 dwords from the density rule, which only
// the looping form decodes.  Capped at the compile-time size they run the register-resident form: fp16 +14 %.  The
// exception is a code that does not re-synchronise — most of its code space at ONE length (fp8 e4m3 weights: 72 % of the
// symbols are 5 bits long, a mis-aligned decoder stays mis-aligned for a median of 35 bits, 12 % beyond 128): nearly every
// tile then needs fix-up passes whatever the run-in, and the looping form with long sub-blocks is the faster one (−13 %
// with the cap).  `dom` = the share of the code space held by the most populated code length (ZnWaveStats).
#define ZN_F_DCAP 4
#define ZN_F_DOM_MAX 154                 // (0.6 · 256)
#define ZN_F_DCONST 4                    // the sub-block size that gets a compile-time instance
// The second compile-time size: the DENSE-code instance (fp8, fp16's top byte: 5-6 bits a symbol want 6-dword sub-blocks).  Round 2
// had it off — 25 + 3 record slots spilled in every instance.  Round 3: a code whose shortest length is ≥ 4 bits puts at most two
// symbols into an 11-bit window, so two steps share a record register (zn_rec_put<I, DENSE>): 14 + 7 registers for 28 steps, fewer
// than the 4-dword instance's 20 + 5, and the tiles of such streams take the register-resident form (no second decode for the
// write pass).  Instantiated for one- and two-plane tensors; 0 = off.
#define ZN_F_DCONST2 6
#define ZN_F_DENSE_LMIN 4                // shortest code length (bits) from which a stream counts as dense
// dwords 0 (look-ahead below the tile) .. 64 D (its top) of a stream tile, + spare
#define ZN_F_IN_DW (64 * ZN_F_DMAX + 4)
#define ZN_F_TF(D) ((D) * 4 + 1)         // whole-group step slots of the register-resident decode for a sub-block of D dwords
#define ZN_F_TLMAX 11u
// the 16-bit single-symbol table (scratch of the LUT build, aliased into the idle staging buffers 0 and 1): one
// dword of padding per 32 — the build reads it at (u << pos) & mask, power-of-two strides across the lanes that
// would otherwise pile 16 or 32 lanes onto one bank
#define ZN_L16(i) ((i) + (((i) >> 6) << 1))
// wave priority (s_setprio) by phase: it rises with the progress through a tile — sync run-in 1, count pass 2, write
// pass 3, everything else (flush, staging, table build) 0.  The passes are chains of dependent LDS look-ups: their next
// instruction issues ahead of the other waves' flush / table-build code, and a wave that is further along goes first.
// Measured: all three passes at 3: +1.6 % over no priorities; graded like this: another +1 % (bf16; fp16 +3 %).
#define ZN_F_PRIO_SYNC 1
#define ZN_F_PRIO_COUNT 2
#define ZN_F_PRIO_WRITE 3
// (the tree-description parse and the LUT fill run at priority 0: raising them measured nothing, profiles/r03_decode_experiments.txt)
#if !defined(ZN_SIMT_EMULATOR)
#define ZN_PRIO(v) __builtin_amdgcn_s_setprio(v)
#else
#define ZN_PRIO(v) do { } while (0)
#endif
#if !defined(ZN_SIMT_EMULATOR)
#define ZN_OPAQUE32(x) asm volatile("" : "+v"(x))
#define ZN_KEEP32(x) asm volatile("" :: "v"(x))        // a use the compiler cannot remove (keeps a loaded-and-ignored value's register until here)
// (a & mask) | (b & ~mask) as ONE v_bfi_b32: left to itself the compiler simplifies the masked shift first and
// then needs and + and + or for the second of the two (seen in the ISA of the un-rotate: 8 VALU per dword pair)
__device__ __forceinline__ uint32_t zn_bfi_(uint32_t mask, uint32_t a, uint32_t b) {
  uint32_t d; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b)); return d;
}
#define ZN_BFI(mask, a, b) zn_bfi_((mask), (a), (b))
#define ZN_NO_IFCVT asm volatile("")     // keeps a rarely-taken block a real (scalar) branch
#define ZN_ASM_MARK(txt) asm volatile("; " txt)   // a comment in the ISA listing (scripts/hot_spills.py looks for spills between the marks)
#else
#define ZN_NO_IFCVT ((void)0)
#define ZN_ASM_MARK(txt) ((void)0)
#define ZN_OPAQUE32(x) ((void)0)
#define ZN_KEEP32(x) ((void)(x))
#define ZN_BFI(mask, a, b) ((((uint32_t)(a)) & (uint32_t)(mask)) | (((uint32_t)(b)) & ~(uint32_t)(mask)))
#endif
// (the next stream tile is staged inside the flush, ahead of its stores — at the top of the tile loop the wait for its registers was a wait for those stores: 19 % of a wave's time)
#define ZN_F_DELTA0 44                   // initial sync run-in (bits): 22 / 33 / 44 = two / three / four whole groups ahead of the boundary step.  Measured
                                         // (profiles/r03_decode_experiments.txt): 44 from the start is neutral on bf16 / fp32 (fewer fix-ups pay for the two
                                         // extra look-ups) and 2 % faster on the dense codes (fp16, fp8), whose streams end up there anyway
#define ZN_F_DELTA0_DENSE 88             // … of the dense-code instance (fp8 +2 %, fp16 +1.3 % over 44; bf16 would lose 4 % to it)
#define ZN_F_DELTA_FAST 44               // longest run-in the unrolled sync handles (beyond it: the looping form)
#define ZN_F_DELTA_MAX 1024              // the run-in never grows beyond this (and never beyond a sub-block)
#define ZN_F_NMIS 3                      // tiles of a stream that needed a fix-up before its run-in is lengthened

// path counters of the emulated build (tests/simt): which form decoded how many tiles — [0] tiles, [1] looping form,
// [2] fix-up iterations, [3] tiles written in several lane groups.  The device build has none of this.
#if defined(ZN_SIMT_EMULATOR)
static unsigned long long zn_dbg_tiles[8];
extern "C" void zn_debug_tile_counters(unsigned long long* out, int reset) { for (int i = 0; i < 8; i++) { out[i] = zn_dbg_tiles[i]; if (reset) zn_dbg_tiles[i] = 0; } }
#define ZN_DBG_COUNT(i) do { if (lane == 0) zn_dbg_tiles[i]++; } while (0)
#else
#define ZN_DBG_COUNT(i) do { } while (0)
#endif

// stream-tile dword i lives at in[ZN_IN_IDX(i)] (a padded layout — one dword per 32 against the 4-way bank conflict of the D = 4 refills —
// was tried: no gain, profiles/r02_decode_experiments.txt)
#define ZN_IN_IDX(i) (i)

// 16-byte output store, non-temporal: the output is written once and never read back by this kernel
#if !defined(ZN_SIMT_EMULATOR)
typedef uint32_t zn_v4u __attribute__((ext_vector_type(4)));
#define ZN_ST128(p, a, b, c, d) __builtin_nontemporal_store((zn_v4u){(a), (b), (c), (d)}, (zn_v4u*)(p))
#else
#define ZN_ST128(p, a, b, c, d) (*(uint4*)(p) = make_uint4((a), (b), (c), (d)))
#endif

#if defined(ZN_SIMT_EMULATOR)
__device__ __forceinline__ uint32_t zn_lane_id() { return threadIdx.x & 63u; }
#else
__device__ __forceinline__ uint32_t zn_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
#endif
// wave-uniform values the compiler cannot see as such (v_readfirstlane → scalar registers)
__device__ __forceinline__ uint32_t zn_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t zn_uniform64(uint64_t v) { return ((uint64_t)zn_uniform((uint32_t)(v >> 32)) << 32) | zn_uniform((uint32_t)v); }
template <typename T> __device__ __forceinline__ T* zn_uniform_ptr(T* p) { return ZN_GLOBAL_PTR(T, zn_uniform64((uint64_t)p)); }

typedef uint64_t __attribute__((aligned(1))) zn_u64u;
typedef uint32_t __attribute__((aligned(1))) zn_u32u;
// the payload is read once: streaming (non-temporal) loads for the raw planes and the stream tiles
#if !defined(ZN_SIMT_EMULATOR)
#define ZN_LD_RAW32(p) __builtin_nontemporal_load((const zn_u32u*)(p))
#define ZN_LD_RAW64(p) __builtin_nontemporal_load((const zn_u64u*)(p))
#define ZN_LD_STREAM32(p) __builtin_nontemporal_load((const uint32_t*)(p))
#else
#define ZN_LD_RAW32(p) (*(const zn_u32u*)(p))
#define ZN_LD_RAW64(p) (*(const zn_u64u*)(p))
#define ZN_LD_STREAM32(p) (*(const uint32_t*)(p))
#endif
// the four-plane rows' stores: non-temporal (each store instruction writes whole 32-byte sectors: ZN_F_SPLIT4)
#if defined(ZN_SIMT_EMULATOR)
#define ZN_ST128_4(p, a, b, c, d) (*(uint4*)(p) = make_uint4((a), (b), (c), (d)))
#else
#define ZN_ST128_4(p, a, b, c, d) ZN_ST128(p, a, b, c, d)
#endif

struct ZnFusedPlane { uint64_t off; uint32_t kind; uint32_t csize; };   // off: body offset (RAW/HUF) or byte value (RLE)

struct __attribute__((aligned(16))) ZnFusedLds {
  uint2 lut[1u << ZN_F_TLMAX];             // multi-symbol decode table: {symbols, meta} per TL-bit window (ZN_E_META)
  uint32_t ring[4][ZN_F_RING_DW];          // per-wave output ring; ring[0] holds the 16-bit LUT while tables are built
  uint32_t in[4][ZN_F_IN_DW];              // per-wave staged stream tile
  uint8_t symlist[4][256];                 // per chunk of the group: symbols in canonical order
  uint32_t rank_start[4][14], sym_start[4][14];
  ZnFusedPlane plane[4][4];                // [chunk of the group][plane]
  ZnWaveStats st[4];
  uint32_t what[4];
};

#include "zn_decode_chain.hpp"

// Decode tables of one huff0 block, by the whole workgroup: the canonical single-symbol LUT (u16, aliased into
// staging buffer 0, idle at this point), then the multi-symbol LUT.  j = which of the group's symbol orders.
// Contains one __syncthreads(); the caller syncs again before the staging buffers are used.
// NT = threads of the workgroup (256: the fused kernel; 1024: the wide kernel of zn_decode_wide.hpp); LDS = its shared-memory struct.
template <int NT = ZN_F_THREADS, typename LDS>
__device__ __forceinline__ void zn_fused_fill_luts(LDS& L, uint32_t tid, uint32_t TL, uint32_t j, uint32_t lmin = 1) {
  constexpr int KE = (int)((1u << ZN_F_TLMAX) / (uint32_t)NT);   // table entries per thread
  uint16_t* lut16 = (uint16_t*)&L.ring[0][0];
  {
    const ZnRankTab rt = zn_load_ranks(L.rank_start[j], L.sym_start[j]);
    for (uint32_t u = tid; u < (1u << TL); u += (uint32_t)NT) lut16[ZN_L16(u)] = (uint16_t)zn_lut_entry(u, TL, L.symlist[j], rt, L.rank_start[j], L.sym_start[j]);
  }
  __syncthreads();
  {
    // 2^TL / NT ≤ KE entries per thread, advanced in lock-step so the dependent LUT16 reads overlap
    const uint32_t mask = (1u << TL) - 1u;
    uint32_t pos[KE], cnt[KE], syms[KE], ef[KE];
    for (int k = 0; k < KE; k++) { pos[k] = 0; cnt[k] = 0; syms[k] = 0; ef[k] = 0; }
    // (a window of TL bits holds at most TL / lmin symbols: a dense code's table needs two rounds of look-ups, not four)
    const int rounds = (int)zn_uniform(lmin ? (TL / lmin > 4u ? 4u : TL / lmin) : 4u);
    for (int step = 0; step < 4; step++) {
      if (step >= rounds) break;
      for (int k = 0; k < KE; k++) {
        const uint32_t u = tid + (uint32_t)k * (uint32_t)NT;
        if (u <= mask && cnt[k] == (uint32_t)step) {
          const uint32_t e = lut16[ZN_L16((u << pos[k]) & mask)]; const uint32_t len = e >> 8;
          if (pos[k] + len <= TL) {             // the window holds this code completely
            if (step > 0) ef[k] |= pos[k] << (16 + 4 * (step - 1));     // E_step = where this symbol starts
            syms[k] |= (e & 0xFFu) << (8 * step);
            pos[k] += len; cnt[k]++;
          }
        }
      }
    }
    for (int k = 0; k < KE; k++) {
      const uint32_t u = tid + (uint32_t)k * (uint32_t)NT;
      for (uint32_t i = 1; i <= 3u; i++) if (i >= cnt[k]) ef[k] |= pos[k] << (16u + 4u * (i - 1u));   // E_i = total for a symbol that does not exist
      if (u <= mask) L.lut[u] = make_uint2(syms[k], ZN_E_META(cnt[k], pos[k], ef[k]));
    }
  }
}

// Everything one wave does for its quarter of the chunk, with the Huffman plane index H known at
// compile time (H = -1: no Huffman plane) so that register arrays are statically indexed.
// DC: the sub-block size when it is known at compile time (the common value gets its own instance,
// which lets the tile loads / staging loops unroll exactly), 0 = run-time Du.
// X: the output is XORed with the delta base `xq` (same offsets as outq; may still be null for a tensor without one).
// xq == outq accumulates into rows already written (every row is loaded before it is stored, once).
template <int P, int H, int DC, bool X = false>
__device__ __forceinline__ bool zn_fused_wave(const ZnGeom& g, const uint8_t* __restrict__ body, const uint8_t* body_end,
                                              uint8_t* outq_, const uint8_t* xq_, const ZnFusedPlane (&pl_)[P], const uint8_t* const (&rawq_)[P],
                                              const uint2* lut, uint32_t* ring, uint32_t* in, uint32_t lane_, uint32_t seg_,
                                              uint32_t TL_, uint32_t Du, const uint8_t* stream_, uint32_t slen_, bool ragged ZN_PT_PARAM) {
  // Everything the caller hands over is the same in all 64 lanes, but most of it came through LDS or was derived from
  // threadIdx, which makes it per-lane data to the compiler: vector registers, and exec-masked control flow around every
  // branch that depends on it (the tile loop, the choice of the decode form).  Say that it is uniform.
  uint32_t lane = lane_;                       // (recomputed at the top of every tile: two v_mbcnt instead of a register that lives — or is spilled — across the whole loop)
  uint8_t* const outq = zn_uniform_ptr(outq_); const uint8_t* const xq = zn_uniform_ptr(xq_);
  const uint8_t* const stream = zn_uniform_ptr(stream_);
  const uint32_t seg = zn_uniform(seg_), TL = zn_uniform(TL_), slen = zn_uniform(slen_);
  ZnFusedPlane pl[P]; const uint8_t* rawq[P];
  for (int p = 0; p < P; p++) {
    pl[p].off = zn_uniform64(pl_[p].off); pl[p].kind = zn_uniform(pl_[p].kind); pl[p].csize = zn_uniform(pl_[p].csize);
    rawq[p] = zn_uniform_ptr(rawq_[p]);
  }
  constexpr int EPL = (P == 1) ? 16 : 8;      // bytes per plane per lane in one flushed row
  constexpr int EW = EPL / 4;                 // … in dwords
  constexpr uint32_t UNIT = 64u * EPL;        // symbols per flushed row (lane row = EPL*P output bytes)
  // rows kept in registers at once (fetched before the write pass, emitted after it).  A 4-plane row holds three
  // raw planes in registers: two rows at a time, or the kernel spills (ZN_F_RB4).
#define ZN_F_RB4 2
  // Four planes: a lane's 8 symbols of a row are 32 output bytes = two 16-byte stores; as ONE run of 8 symbols each store instruction writes 16 bytes at a stride
  // of 32 (half of every 32-byte sector).  ZN_F_SPLIT4: the lane owns two runs of 4 symbols, [4 lane, +4) and [256 + 4 lane, +4) of the row's 512, so that each
  // store instruction writes 1 KB contiguous (whole sectors, whole lines) and can be non-temporal.
#define ZN_F_SPLIT4 1
  constexpr bool SPLIT = (P == 4) && (ZN_F_SPLIT4 != 0);
  // (two planes: 4 rows per batch, a tile's ~6 rows in two batches.  8 rows — one batch — keeps 16 more registers in flight
  //  through the compaction; every build of that variant spilled somewhere in the tile loop and ran between 1.61 and 2.2 ms
  //  on the 4 GiB config depending on where, 4 rows ran 1.60 ms three builds in a row; with the tile loop spill-free 3 rows
  //  — two even batches for the ~6 rows of a tile — run 1.545 against 1.568 (4), 1.588 (2), 1.581 (5), 1.82 (8):
  //  profiles/r02_decode_experiments.txt)
#define ZN_F_RB2 3
  constexpr int RB = (P == 2) ? (int)ZN_F_RB2 : (P == 4) ? ZN_F_RB4 : (int)(ZN_F_RING_BYTES / 1024u);
  // register-resident decode (zn_pass1 / zn_pass2): slots for whole-group steps / boundary steps, unchecked head.
  // A sub-block of D dwords needs about 32 D / 9.6 steps (a step consumes 9-10 of its ≤ 11 window bits on every
  // dtype's data, or 4 symbols when the codes are short); lanes that need more send the tile to the looping form.
  constexpr int TF = DC ? ZN_F_TF(DC) : 0, TB = 3;      // (the run-time-D instance only has the looping form)
  // the second compile-time size is the dense-code instance: its caller guarantees a shortest code of ≥ 4 bits (≤ 2 symbols per
  // window), and its records are packed two steps to a register (zn_rec_put)
  constexpr bool DENSE = (ZN_F_DCONST2 != 0) && (DC == ZN_F_DCONST2);
  constexpr int UF = DC ? (32 * DC - 31) / 11 : 0;
  ZN_PT_SHARED;

  // raw-plane bytes (and, in emit, ring bytes of plane H) for up to RB rows
  uint32_t pre[RB][P][EW];
  // the lane index as the HBM address computations see it: made opaque once per tile (below), so that the compiler cannot
  // hoist `base pointer + lane` out of the tile loop as 64-bit per-lane pointers — loop-invariant, so it did, one pair of
  // registers per stream / plane / output, and then spilled them and reloaded them behind an s_waitcnt vmcnt(0)
  uint32_t lane_v = lane;
  auto fetch_row_to = [&](uint32_t (&pre)[RB][P][EW], uint32_t first_row_sym, int r) {
    for (int p = 0; p < P; p++) if (p != H && pl[p].kind == ZN_KIND_RAW) {
      // (a wave-uniform base + a 32-bit lane offset: the compiler then addresses with a scalar base register and one vector
      //  offset instead of keeping — and spilling — a 64-bit pointer per lane and row)
      const uint8_t* au = rawq[p] + (first_row_sym + (uint32_t)r * UNIT);
      if constexpr (SPLIT) {
        const uint8_t* a = au + 4u * lane_v;
        pre[r][p][0] = ZN_LD_RAW32(a); pre[r][p][1 % EW] = ZN_LD_RAW32(a + UNIT / 2u);
      } else {
        const uint8_t* a = au + (uint32_t)EPL * lane_v;
        for (int k = 0; k < EW / 2; k++) { const uint64_t t = ZN_LD_RAW64(a + 8 * k); pre[r][p][2 * k] = (uint32_t)t; pre[r][p][2 * k + 1] = (uint32_t)(t >> 32); }
      }
    }
  };
  auto fetch_row = [&](uint32_t first_row_sym, int r) { fetch_row_to(pre, first_row_sym, r); };
  auto fetch_rows = [&](uint32_t first_row_sym, int nrows) {
    for (int r = 0; r < RB; r++) if (r < nrows) fetch_row_to(pre, first_row_sym, r);
  };
  // interleave rows (ring bytes for the Huffman plane, fetched bytes for raw planes) and store them.
  // All loads are complete before the first store is issued, so no store latency is ever waited on.
  // `stage_row0` = staging-buffer row that holds symbol `first_row_sym`.
  // `after_wait` runs right behind the wait, before the first store is issued: whatever else has to consume
  // loaded registers (the staging of the next stream tile) does it there, so that nothing ever waits on a STORE
  // (gfx9 has one in-order counter for loads and stores: a wait for a load issued after stores waits for their acks)
  auto emit_rows_from = [&](uint32_t (&pre)[RB][P][EW], bool full_wait, uint32_t first_row_sym, int nrows, uint32_t stage_row0, auto&& after_wait) {
    if (full_wait) __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): every fetched row (and the prefetched tile) has landed
    ZN_PT(10);  // wait for the fetched rows
    after_wait();
    // delta base of these rows, two rows at a time, double-buffered: the next pair is requested before the current
    // pair is stored (more rows in flight would spill: the fetched raw rows are live here too)
    constexpr int XW = X ? (P == 4 ? 8 : 4) : 1, XG = 2;
    uint32_t xd[2][XG][XW];
    auto load_delta = [&](int grp) {
      for (int i = 0; i < XG; i++) {
        const int r = grp * XG + i;
        for (int k = 0; k < XW; k++) xd[grp & 1][i][k] = 0;
        if (xq && r < RB && r < nrows) {
          const uint8_t* a = SPLIT ? xq + (uint64_t)(first_row_sym + (uint32_t)r * UNIT) * P + 16u * lane
                                   : xq + (uint64_t)(first_row_sym + (uint32_t)r * UNIT + (uint32_t)EPL * lane) * P;
          for (int k = 0; k < XW / 4; k++) { const uint4 t = *(const uint4*)(a + (SPLIT ? (UNIT / 2u) * P : 16u) * k); xd[grp & 1][i][4 * k] = t.x; xd[grp & 1][i][4 * k + 1] = t.y; xd[grp & 1][i][4 * k + 2] = t.z; xd[grp & 1][i][4 * k + 3] = t.w; }
        }
      }
    };
    if (X) load_delta(0);
    for (int r = 0; r < RB; r++) if (r < nrows) {
      for (int p = 0; p < P; p++) {
        if (p == H) {
          if constexpr (SPLIT) { const uint32_t i = ((stage_row0 + (uint32_t)r) * UNIT >> 2) + lane; for (int k = 0; k < EW; k++) { pre[r][p][k] = ring[i + (UNIT / 8u) * k]; ring[i + (UNIT / 8u) * k] = 0; } }
          else { const uint32_t i = ((stage_row0 + (uint32_t)r) * UNIT + (uint32_t)EPL * lane) >> 2; for (int k = 0; k < EW; k++) { pre[r][p][k] = ring[i + k]; ring[i + k] = 0; } }
        }
        else if (pl[p].kind == ZN_KIND_RLE) { ZN_NO_IFCVT; for (int k = 0; k < EW; k++) pre[r][p][k] = ((uint32_t)pl[p].off & 0xFFu) * 0x01010101u; }   // (a branch: as selects this cost every row two VALU ops)
      }
    }
    // undo the sign-bit rotate on the two top planes BEFORE the interleave (4 elements per dword):
    // top byte = (low.7) | (top >> 1), low byte = (top.0 << 7) | (low & 0x7F) — two v_bfi per dword pair
    if (P >= 2 && g.rot) {
      for (int r = 0; r < RB; r++) if (r < nrows)
        for (int k = 0; k < EW; k++) {
          const uint32_t hi = pre[r][P - 1][k], lo = pre[r][(P >= 2) ? P - 2 : 0][k];
          pre[r][P - 1][k] = ZN_BFI(0x80808080u, lo, hi >> 1);
          pre[r][(P >= 2) ? P - 2 : 0][k] = ZN_BFI(0x7F7F7F7Fu, lo, hi << 7);
        }
    }
    for (int r = 0; r < RB; r++) if (r < nrows) {
      if (X && r % XG == 0 && r + XG < RB) load_delta(r / XG + 1);
      const uint32_t* xr_ = xd[(r / XG) & 1][r % XG];
      uint8_t* o = (outq + (uint64_t)(first_row_sym + (uint32_t)r * UNIT) * P) + (SPLIT ? 16u : (uint32_t)EPL * (uint32_t)P) * lane_v;   // (uniform base + lane offset)
      if (P == 1) {
        if (X) ZN_ST128(o, pre[r][0][0] ^ xr_[0], pre[r][0][1 % EW] ^ xr_[1 % XW], pre[r][0][2 % EW] ^ xr_[2 % XW], pre[r][0][3 % EW] ^ xr_[3 % XW]);
        else ZN_ST128(o, pre[r][0][0], pre[r][0][1 % EW], pre[r][0][2 % EW], pre[r][0][3 % EW]);
      } else if (P == 2) {
        uint32_t x[4];
        x[0] = __builtin_amdgcn_perm(pre[r][1 % P][0], pre[r][0][0], 0x05010400u); x[1] = __builtin_amdgcn_perm(pre[r][1 % P][0], pre[r][0][0], 0x07030602u);
        x[2] = __builtin_amdgcn_perm(pre[r][1 % P][1 % EW], pre[r][0][1 % EW], 0x05010400u); x[3] = __builtin_amdgcn_perm(pre[r][1 % P][1 % EW], pre[r][0][1 % EW], 0x07030602u);
        if (X) for (int k = 0; k < 4; k++) x[k] ^= xr_[k % XW];
        ZN_ST128(o, x[0], x[1], x[2], x[3]);
      } else {
        for (int half = 0; half < 2; half++) {
          const int k = half % EW;
          const uint32_t ab_lo = __builtin_amdgcn_perm(pre[r][1 % P][k], pre[r][0][k], 0x05010400u), ab_hi = __builtin_amdgcn_perm(pre[r][1 % P][k], pre[r][0][k], 0x07030602u);
          const uint32_t cd_lo = __builtin_amdgcn_perm(pre[r][3 % P][k], pre[r][2 % P][k], 0x05010400u), cd_hi = __builtin_amdgcn_perm(pre[r][3 % P][k], pre[r][2 % P][k], 0x07030602u);
          uint32_t x[4];
          x[0] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u); x[1] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
          x[2] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u); x[3] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
          if (X) for (int k = 0; k < 4; k++) x[k] ^= xr_[(4 * half + k) % XW];
          if constexpr (SPLIT) ZN_ST128_4(o + (UNIT / 2u) * P * half, x[0], x[1], x[2], x[3]);        // (1 KB contiguous per store instruction: whole sectors)
          else *(uint4*)(o + 16 * half) = make_uint4(x[0], x[1], x[2], x[3]);   // (two half-line stores per lane: NOT non-temporal, the L2 merges them — nt cost 40 % here)
        }
      }
    }
  };
  auto emit_rows = [&](uint32_t first_row_sym, int nrows, uint32_t stage_row0, auto&& after_wait) { emit_rows_from(pre, true, first_row_sym, nrows, stage_row0, after_wait); };

  uint32_t JF = 0;                            // symbols flushed to HBM so far
  if (H < 0) {
    // no Huffman plane: the chunk is a pure P-way interleave of raw / RLE planes
    while (JF < seg) {
      const uint32_t left = (seg - JF) / UNIT; const int nr = left < (uint32_t)RB ? (int)left : RB;
      fetch_rows(JF, nr); emit_rows(JF, nr, 0, [] {}); JF += (uint32_t)nr * UNIT;
    }
    ZN_PT(3);
    return true;
  }

  for (uint32_t i = lane; i < ZN_F_RING_DW; i += 64u) ring[i] = 0;
  __builtin_amdgcn_wave_barrier();

  uint32_t J = 0;                             // symbols decoded into the ring so far
  const uint8_t last = stream[slen - 1];
  if (last == 0) return false;
  // (pointer arithmetic, not an integer round trip: keeps the loads in the global address space — flat loads
  //  would tie every later LDS wait to the HBM latency of the tile prefetch)
  const uint32_t mis = (uint32_t)((uint64_t)stream & 3u);
  const uint32_t* gdw = (const uint32_t*)(stream - mis);
  const int32_t b0 = (int32_t)(8u * mis);
  int32_t carry = __builtin_amdgcn_readfirstlane(b0 + (int32_t)(8u * (slen - 1u)) + (int32_t)zn_hb32(last));   // (wave-uniform)
  int32_t hi_dw = (carry + 31) >> 5;
  const int32_t Di = DC ? DC : (int32_t)Du, TD = 64 * Di;             // dwords per sub-block / per tile
  // (a dense code re-synchronises slowly: its streams start with the longer run-in they would grow into anyway)
  int32_t delta = (DENSE ? ZN_F_DELTA0_DENSE : ZN_F_DELTA0) < 32 * Di ? (DENSE ? ZN_F_DELTA0_DENSE : ZN_F_DELTA0) : 32 * Di;
  int nmis = 0;                               // tiles of this stream that needed a fix-up since the run-in was last lengthened

  // stream-tile prefetch registers: dword (lo_dw - 1 + lane + 64 i) of the NEXT tile, i = 0..D-1, and (lane 0) the
  // tile's last dword, in a register of its own so that nothing selects on a loaded value before the staging
  // (the stream's top dword may straddle the end of the buffer: that one dword is assembled from bytes)
  uint32_t nx[ZN_F_DMAX], nx_last = 0;
  const int32_t top_dw = hi_dw - 1;
  const bool top_guard = ((const uint8_t*)(gdw + hi_dw) > body_end);
  auto fetch_tile = [&](int32_t lo_dw_, int32_t hi_dw_) {
    const uint32_t* pu = gdw + (lo_dw_ - 1);               // wave-uniform (scalar registers); the lane index is the vector offset
    if (lo_dw_ >= 0 && !(top_guard && hi_dw_ - 1 == top_dw)) {
      // common case: every dword of the tile exists in the buffer
      for (int i = 0; i < ZN_F_DMAX; i++) nx[i] = (i < Di) ? ZN_LD_STREAM32(pu + (lane_v + 64u * (uint32_t)i)) : 0u;
      nx_last = gdw[__builtin_amdgcn_readfirstlane(hi_dw_) - 1];   // (every lane, the same address: no select on the way into the register)
    } else {
      auto dword_at = [&](int32_t li) -> uint32_t {
        const int32_t gi = lo_dw_ - 1 + li;
        uint32_t x = 0;
        if (li <= TD && gi >= -1 && gi < hi_dw_) {
          if (top_guard && gi == top_dw) { const uint8_t* pa = (const uint8_t*)(gdw + gi); for (int b = 0; b < 4; b++) if (pa + b < body_end) x |= (uint32_t)pa[b] << (8 * b); }
          else x = gdw[gi];
        }
        return x;
      };
      for (int i = 0; i < ZN_F_DMAX; i++) nx[i] = (i < Di) ? dword_at((int32_t)lane + 64 * i) : 0u;
      nx_last = (lane == 0) ? dword_at(TD) : 0u;
    }
  };
  // stage the prefetched tile into LDS.  Done for the first tile here and for every later one from inside the flush
  // of its predecessor (behind the flush's wait, ahead of its stores): at the top of the tile loop the wait for
  // the prefetch registers is a wait for those stores to be acknowledged as well (gfx9 counts loads and stores on
  // one in-order counter) — the phase timers had 19 % of a wave's time there.
  auto stage_tile = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < ZN_F_DMAX; i++) if (i < Di) in[ZN_IN_IDX((int32_t)lane + 64 * i)] = nx[i];
    if (lane == 0) in[ZN_IN_IDX(TD)] = nx_last;
    __builtin_amdgcn_wave_barrier();
  };
  fetch_tile(hi_dw - TD, hi_dw);
  stage_tile();

  // sync: every sub-block except the tile's first guesses a start `delta` bits above itself and runs into it; returns the
  // position the lane's own decode starts from (lane 0: the true position carried over from the previous tile)
  auto sync_run = [&](int32_t base_bit, int32_t hi_k, bool active, bool regular_tile) -> int32_t {
    ZnChain c;
    c.pos = (lane > 0 && active) ? hi_k + delta : hi_k; c.stop = hi_k; c.n = 0; c.wpos = 0;
    if (delta <= ZN_F_DELTA_FAST) {
      // short run-in: G = 2, 3 or 4 whole groups (≤ TL bits each) and the boundary step (a TL-bit look-up), all from at most two
      // window refills (a refill holds ≥ 33 valid bits = three look-ups).  22 bits synchronise all but ≈ 0.2 % of the sub-blocks of a
      // bf16 exponent stream; a tile with one lane that did not is decoded twice (a fix-up re-runs the whole pass: 14+ steps), so a
      // stream whose tiles keep mismatching gets one more group per tile instead (note_mismatch)
      const uint32_t sh = 32u - TL;
      const int G = delta <= 22 ? 2 : delta <= 33 ? 3 : 4;          // (wave-uniform)
      uint64_t w = zn_window(in, c.pos - 1 - base_bit);
      auto group = [&]() { uint2 e = lut[(uint32_t)(w >> 32) >> sh]; if (!(c.pos > c.stop + (int32_t)TL - 1)) { e.x = 0; e.y = 0; } w <<= (e.y & 63u); c.pos -= (int32_t)ZN_M_NB(e.y); };
      group(); group();
      if (G >= 3) { ZN_NO_IFCVT; group(); w = zn_window(in, c.pos - 1 - base_bit); }
      if (G >= 4) { ZN_NO_IFCVT; group(); }
      { uint2 e = lut[(uint32_t)(w >> 32) >> sh]; if (!(c.pos > c.stop)) { e.x = 0; e.y = 0; } c.pos -= (int32_t)ZN_M_NB(zn_trim_group(e, c.pos - c.stop).y); }
      while (__any(c.pos > c.stop)) {
        w = zn_window(in, c.pos - 1 - base_bit);
        uint2 e = lut[(uint32_t)(w >> 32) >> sh]; if (!(c.pos > c.stop)) { e.x = 0; e.y = 0; }
        c.pos -= (int32_t)ZN_M_NB(zn_trim_group(e, c.pos - c.stop).y);
      }
    } else {
      zn_fused_run<0, DENSE>(lut, in, base_bit, TL, c, nullptr, regular_tile ? (delta - 21) / 11 : 0);
    }
    return (lane > 0) ? c.pos : carry;
  };
  // a longer run-in for the rest of the stream once mismatches keep coming (the third tile that needs a fix-up): beyond
  // 22 bits the run-in takes the looping form, which costs every later tile of the stream several steps
  // (22 → 33 → 44 bits stay on the fast path above: one more group each; beyond that the run-in doubles and takes the looping form)
  auto note_mismatch = [&]() { if (++nmis >= ZN_F_NMIS) { nmis = 0; delta = (delta < 22) ? 22 : (delta < ZN_F_DELTA_FAST) ? delta + 11 : 2 * delta; if (delta > 32 * Di) delta = 32 * Di; if (delta > ZN_F_DELTA_MAX) delta = ZN_F_DELTA_MAX; } };
  // the incomplete last row (< UNIT symbols, at staging row `total_rows`) moves to the start of the staging buffer
  auto keep_remainder = [&](uint32_t total_rows) {
    const uint32_t i = (total_rows * UNIT + (uint32_t)EPL * lane) >> 2;
    uint32_t t[EW];
    for (int k = 0; k < EW; k++) { t[k] = ring[i + k]; ring[i + k] = 0; }
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < EW; k++) ring[((uint32_t)EPL * lane >> 2) + k] = t[k];
    __builtin_amdgcn_wave_barrier();
  };

  bool ok = true;
  while (32 * hi_dw > b0) {
    if (DC) ZN_ASM_MARK("ZN_HOT_TILE_BEGIN");
    // the stream position and the counters are wave-uniform; said once per tile, because across the two tile forms and
    // the fix-up loop the compiler takes them for per-lane values: vector registers (the 64-bit stream pointer among them,
    // spilled), and an exec-mask region around every branch that depends on them
    lane = zn_lane_id();
    lane_v = lane; ZN_OPAQUE32(lane_v);
    hi_dw = __builtin_amdgcn_readfirstlane(hi_dw); carry = __builtin_amdgcn_readfirstlane(carry);
    delta = __builtin_amdgcn_readfirstlane(delta); nmis = __builtin_amdgcn_readfirstlane(nmis);
    J = zn_uniform(J); JF = zn_uniform(JF);
    // ---- tile: dwords [lo_dw, hi_dw) of the stream, plus one below for look-ahead ----
    const int32_t lo_dw = hi_dw - TD;
    if (32 * lo_dw > b0) fetch_tile(lo_dw - TD, lo_dw);      // prefetch the next tile while this one is decoded
    // (tried, r02: requesting the rows this tile will PROBABLY complete — predicted from the previous tile's count — at the top of
    //  the tile, a whole decode pass ahead: no gain; a second flush batch's rows requested behind the first batch's wait: slower)
    // (tried, r02: touching the raw-plane lines this tile's flush will need — one dword per 128-byte line, at the top of
    //  the tile, so that the loads proper hit the L2 — made the kernel 8 % SLOWER: profiles/r02_decode_experiments.txt)
    ZN_PT(4);   // stage tile
    const int32_t base_bit = 32 * (lo_dw - 1);
    const int32_t hi_k = 32 * (hi_dw - (int32_t)lane * Di), lo_k = hi_k - 32 * Di;
    const int32_t stop = lo_k > b0 ? lo_k : b0;
    const bool active = hi_k > b0;
    const bool regular = (32 * lo_dw >= b0);             // every lane owns a whole sub-block (stop == lo_k, all active)
    ZN_PT_COUNT(18, 1);                      // tiles
    ZN_DBG_COUNT(0);

    // ---- the register-resident form (zn_decode_chain.hpp): decode once, verify the chain, compact, flush.  It commits
    // nothing before it knows that the tile is its own; what it does not take — more than TF whole-group steps in some
    // lane, a chain that a few fix-up iterations do not close, a tile denser than one staging buffer — is left, untouched,
    // to the looping form below, whose state (and registers) it shares nothing with but the stream position.
    bool done = false;
    if (TF > 0) {
      ZN_PRIO(ZN_F_PRIO_SYNC);
      if (DC) ZN_ASM_MARK("ZN_MARK sync");
      int32_t s = sync_run(base_bit, hi_k, active, regular);
      ZN_PRIO(ZN_F_PRIO_COUNT);
      ZN_PT(5);   // sync run-in
      if (DC) ZN_ASM_MARK("ZN_MARK pass1");
      ZnRec rec;
      int nfull = 0, nbnd = 0;
      uint32_t acc = 0, n = 0; int32_t e = s;
      bool took = true, chained = false;
      for (int it = 0; it < 4; it++) {
        // (one call site per instance: the unchecked-head one for regular tiles, the checked one for a stream's last tile)
        if (regular) took = zn_pass1<TF, TB, UF, DENSE>(lut, in, base_bit, TL, s, stop, true, rec, acc, nfull, nbnd);
        else took = zn_pass1<TF, TB, 0, DENSE>(lut, in, base_bit, TL, s, stop, active, rec, acc, nfull, nbnd);
        if (!took) break;
        e = s - (int32_t)(acc & 0xFFu); n = (acc >> 8) & 0xFFu;
        const int32_t e_prev = __shfl_up(e, 1u);
        const bool mism = active && lane > 0 && e_prev != s;
        if (it == 0) ZN_PT(6); else ZN_PT(7);   // first decode pass / fix-up passes
        if (__builtin_expect(!__any(mism), 1)) { chained = true; break; }
        ZN_PT_COUNT(16, 1);                    // number of fix-up iterations
        ZN_PT_COUNT(17, __popcll(__ballot(mism)));
        ZN_DBG_COUNT(2);
        if (it == 0) note_mismatch();
        if (mism) s = e_prev;                  // (every lane decodes again: the lanes run in lock-step anyway; round 5 tried the mismatching lanes alone, exec-masked,
                                               //  for the dense instance: fp8 1.354 vs 1.202 ms, fp16 0.576 vs 0.559 — the masks cost more than the conflict-free look-ups save)
      }
      if (DC) ZN_ASM_MARK("ZN_MARK scan");
      if (took && chained) {
        if (!active) n = 0;
        uint32_t N = 0;
        const uint32_t o_k = zn_wave_excl_scan(n, lane, &N);
        // (J, JF and everything derived from them are wave-uniform; the compiler loses that across the loop — as per-lane
        //  values every `r < rows` below becomes an exec-mask dance of six scalar instructions instead of one compare-and-branch)
        J = zn_uniform(J); JF = zn_uniform(JF);
        const uint32_t base = J - JF;                            // < UNIT
        if (J + N <= seg && base + N <= ZN_F_RING_BYTES - 4u) {
          const uint32_t nact = (uint32_t)__popcll(__ballot(active));
          carry = __builtin_amdgcn_readlane(e, (int)(nact ? nact - 1u : 0u)); hi_dw = lo_dw;
          // rows that will be complete after this tile: request their raw bytes now, use them after the compaction
          int rows = (int)zn_uniform((base + N) / UNIT);
          const uint32_t total_rows = (uint32_t)rows;
          const int first = rows < RB ? rows : RB;
          constexpr int RH = (RB >= 4) ? RB / 2 : RB;         // rows requested before the compaction; the second half of a batch of four or more in the middle of it, when half of the record registers are free again
          for (int r = 0; r < RH; r++) if (r < first) fetch_row(JF, r);
          ZN_PT(8);   // scans / shuffles / issue loads
          ZN_PRIO(ZN_F_PRIO_WRITE);
          if (DC) ZN_ASM_MARK("ZN_MARK pass2");
          // (uniform — said again here, where they are used: through the fix-up loop the compiler takes the slot counts for
          //  per-lane values and turns the compaction's `t < nfull` tests into a ten-deep nest of exec-mask regions)
          const int nf = __builtin_amdgcn_readfirstlane(nfull), nb_ = __builtin_amdgcn_readfirstlane(nbnd);
          // (lanes without a sub-block hold zero records.)  The raw rows RH.. are requested after step TF/2 - 1: the records of
          // the steps before it are dead by then, so the rows' registers do not add to the peak at the start of the compaction.
          bool late_rows = (RH < RB);
          zn_pass2<TF, TB, DENSE>(ring, base + o_k, rec, nf, nb_, [&](auto I) {
            if constexpr (RH < RB && decltype(I)::v == TF / 2 - 1) { for (int r = RH; r < RB; r++) if (r < first) fetch_row(JF, r); late_rows = false; }
          });
          if (late_rows) for (int r = RH; r < RB; r++) if (r < first) fetch_row(JF, r);     // (a tile of fewer steps than that)
          __builtin_amdgcn_wave_barrier();
          ZN_PRIO(0);
          ZN_PT(9);   // compaction
          if (DC) ZN_ASM_MARK("ZN_MARK flush");
          J += N;
          // (the compaction is over: the stream-tile buffer is free for the next tile, staged behind the flush's wait)
          emit_rows(JF, first, 0, [&] {
            if (32 * hi_dw > b0) stage_tile();
          });
          uint32_t srow = (uint32_t)first;
          JF += (uint32_t)first * UNIT; rows -= first;
          while (rows > 0) { const int nr = rows < RB ? rows : RB; fetch_rows(JF, nr); emit_rows(JF, nr, srow, [] {}); JF += (uint32_t)nr * UNIT; srow += (uint32_t)nr; rows -= nr; }
          if (total_rows > 0 && J > JF) keep_remainder(total_rows);
          ZN_PT_COUNT(20, 1);                    // write groups (== tiles when nothing overflowed)
          if (DC) ZN_ASM_MARK("ZN_MARK fastend");
          done = true;
        }
      }
    }

    // ---- the looping form: count pass (with fix-up iterations), then a second decode that ORs the symbols into the
    // staging buffer, in several lane groups with a flush after each when the tile is denser than the buffer
    if (!done) {
      ZN_NO_IFCVT;
      ZN_PT_COUNT(22, 1);                    // tiles that took the looping form
      ZN_DBG_COUNT(1);
      ZN_PRIO(ZN_F_PRIO_SYNC);
      int32_t s = sync_run(base_bit, hi_k, active, regular);
      ZN_PRIO(ZN_F_PRIO_COUNT);
      ZnChain A; A.wpos = 0;
      // (whole-group steps no lane of a regular tile can take too far: see zn_fused_run)
      const int U_blk = regular ? (int)zn_uniform((uint32_t)((32 * Di - 31) / 11)) : 0;
      int32_t e = s; uint32_t n = 0; bool need = active, chained = false;
      for (int it = 0; it < 66; it++) {
        A.pos = need ? s : stop; A.stop = stop; A.n = 0;
        zn_fused_run<1>(lut, in, base_bit, TL, A, nullptr, U_blk);
        if (need) { e = A.pos; n = A.n; }
        const int32_t e_prev = __shfl_up(e, 1u);
        const bool mism = active && lane > 0 && e_prev != s;
        if (!__any(mism)) { chained = true; break; }
        if (it == 0) note_mismatch();
        need = mism;
        if (mism) s = e_prev;
      }
      if (!active) n = 0;
      uint32_t N = 0;
      const uint32_t o_k = zn_wave_excl_scan(n, lane, &N);
      const uint32_t nact = (uint32_t)__popcll(__ballot(active));
      const int32_t e_last = __builtin_amdgcn_readlane(e, (int)(nact ? nact - 1u : 0u));
      if (!chained || J + N > seg) { ok = false; break; }
      carry = e_last; hi_dw = lo_dw;
      uint32_t lane_lo = 0, wdone = 0;
      do {
        const uint32_t base = J - JF;                            // < UNIT
        const bool fits = o_k + n <= wdone + (ZN_F_RING_BYTES - 4u - base);   // monotone in the lane index
        const uint32_t lane_hi = (uint32_t)__popcll(__ballot(fits));
        if (lane_hi <= lane_lo) { ok = false; break; }           // cannot happen: one sub-block always fits
        if (lane_hi < 64u) ZN_DBG_COUNT(3);
        const uint32_t wend = (lane_hi >= 64u) ? N : (uint32_t)__shfl((int)o_k, (int)(lane_hi & 63u));
        const uint32_t nsub = wend - wdone;
        int rows = (int)((base + nsub) / UNIT);
        const int first = rows < RB ? rows : RB;
        fetch_rows(JF, first);
        ZN_PRIO(ZN_F_PRIO_WRITE);
        const bool mine = active && lane >= lane_lo && lane < lane_hi;
        A.pos = mine ? s : stop; A.stop = stop; A.wpos = mine ? base + o_k - wdone : 0u;
        zn_fused_run<2>(lut, in, base_bit, TL, A, ring, (lane_lo == 0u && lane_hi >= 64u) ? U_blk : 0);
        __builtin_amdgcn_wave_barrier();
        ZN_PRIO(0);
        J += nsub; wdone = wend; lane_lo = lane_hi;
        const uint32_t total_rows = (uint32_t)rows;
        uint32_t srow = 0;
        emit_rows(JF, first, srow, [&] {
          if (lane_hi >= 64u && 32 * hi_dw > b0) stage_tile();
        });
        JF += (uint32_t)first * UNIT; srow += (uint32_t)first; rows -= first;
        while (rows > 0) { const int nr = rows < RB ? rows : RB; fetch_rows(JF, nr); emit_rows(JF, nr, srow, [] {}); JF += (uint32_t)nr * UNIT; srow += (uint32_t)nr; rows -= nr; }
        if (total_rows > 0 && J > JF) keep_remainder(total_rows);
        ZN_PT_COUNT(20, 1);
      } while (lane_lo < 64u);
      if (!ok) break;
    }
    ZN_PT(3);   // flush rows
    if (DC) ZN_ASM_MARK("ZN_HOT_TILE_END");
  }
  ZN_PRIO(0);                                 // (an error exit leaves the loop from inside a pass)
  if (ragged && ok && carry == b0 && J == seg && JF < seg) {
    // a stream of a partial chunk: the last row is incomplete — store it whole (the destination is padded)
    emit_rows(JF, 1, 0, [] {}); JF += UNIT;
  }
  return ok && carry == b0 && J == seg && JF >= seg;
}

// A MERGE workgroup of a tensor's partial last chunk (four waves; index m among the launch's merge workgroups, `merge_per` per tensor): waits until the tensor's
// tail workgroups — lower block indices of the same launch — have all reported (tailsync[2 t]), classifies the chunk's planes (the generic path's own item function,
// every merge workgroup for itself) and merges its share of the chunk's words into the output.  Shared by zn_k_decode_fused (REST) and zn_k_decode_wide.
template <int P>
__device__ __forceinline__ void zn_tail_merge_wg(ZnFusedLds& L, const ZnSeg& one_c, const ZnSeg* __restrict__ segs, uint32_t nseg, uint32_t m, uint32_t merge_per,
                                                 uint32_t* __restrict__ tailsync, ZnPlaneDesc* __restrict__ descs_rest, uint32_t* __restrict__ status,
                                                 uint8_t* __restrict__ tail_done, uint8_t* __restrict__ tail_scratch) {
  const uint32_t tt = m / merge_per, jm = m % merge_per;
  const ZnSeg S = zn_find_seg<3>(one_c, segs, nseg, (uint64_t)tt * (uint32_t)P);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint64_t c = S.g.K - 1u;
  __shared__ uint32_t ok_s, serial_s;
  if (tid == 0) { ok_s = zn_flag_wait(tailsync + 2u * tt, 4u * (uint32_t)P) ? 1u : 0u; serial_s = 0; ZN_FLAG_ACQUIRE(); }      // (one lane's acquire drops this CU's L1 lines: the tail workgroups' bytes come from L2)
  __syncthreads();
  if (!ok_s) { if (tid == 0) atomicOr(status, ZN_DEV_SYNC_TIMEOUT); return; }     // (a slow or preempted device is not a corrupt frame: ZN_E_TIMEOUT)
  static_assert(4u * sizeof(ZnPlanesLds) <= sizeof(ZnFusedLds), "four generic plane decoders fit over the fused kernel's LDS");
  // every merge workgroup classifies the chunk's planes for itself (the same descriptors from all of them: raw / RLE / decoded by the tail workgroups) …
  if (wave < (uint32_t)P) zn_decode_plane_item(reinterpret_cast<ZnPlanesLds*>(&L)[wave], one_c, segs, nseg, S.desc0 + (uint64_t)wave * S.g.K + c, descs_rest, status, tail_done, lane, true, &serial_s);
  __syncthreads();
  if (tid == 0) ZN_FLAG_ACQUIRE();             // (the descriptors were written by other waves of this CU: read them from L2, not from a stale L1 line)
  __syncthreads();
  uint32_t sub = jm, nsub = merge_per;           // this workgroup's share of the chunk's words
  if (serial_s) {
    // … unless a huff0 plane is left for the serial decoder (a tiny plane, tableLog 12, a block the tail workgroup gave up on): ONE workgroup does the chunk
    if (jm != 0u) return;
    if (wave < (uint32_t)P) zn_decode_plane_item(reinterpret_cast<ZnPlanesLds*>(&L)[wave], one_c, segs, nseg, S.desc0 + (uint64_t)wave * S.g.K + c, descs_rest, status, tail_done, lane);
    __threadfence(); __syncthreads();
    sub = 0; nsub = 1;
  }
  zn_merge_chunk_item<P>(one_c, segs, nseg, S.chunk0 + c, sub, descs_rest, tail_scratch, nsub);
}

// The Huffman-coded planes of a PARTIAL last chunk.  Round 6: FOUR workgroups per plane, one per huff0 stream (bt = 4 · plane + stream; the first
// workgroups of zn_k_decode_fused, so that the job overlaps the decode of the full chunks instead of following it).  Each parses the tree description and fills the
// look-up table for itself (13 + 5 µs, side by side on four CUs); then its four waves SHARE the stream the way the small-input kernel's do (zn_wide_chunk, one-stream form:
// in round r wave q takes tile 4 r + q, the tiles' tops guess their start like every sub-block does) — a stream's five or six tiles take two rounds instead of six
// tile times on one wave, which is what a ragged tensor of a few hundred chunks waited for (100 MiB + 250 KB: 108 µs against 77 without the tail).  A stream whose code
// is not of that form's density (dense codes, very short codes) is decoded by wave 0 alone with zn_fused_wave, as all four were until round 5.  Into a padded scratch
// slot: stream w's symbols start at slot + w * ZN_TAIL_SEGPAD, the last, incomplete row of a stream is stored whole; tail_done[4 plane + stream] = 1 where it worked —
// the merge workgroups / the generic merge kernel take a plane whose four flags are set from the scratch, anything else (raw / RLE / tiny planes, tableLog 12,
// malformed blocks) is left to the serial generic code.
__device__ void zn_decode_tail_wg(ZnFusedLds& L_, const ZnSeg& one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint32_t bt,
                                  uint8_t* __restrict__ scratch, uint8_t* __restrict__ tail_done, uint32_t* __restrict__ status, uint32_t* tail0_out);
#include "zn_decode_wide.hpp"

__device__ void zn_decode_tail_wg(ZnFusedLds& L_, const ZnSeg& one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint32_t bt,
                                  uint8_t* __restrict__ scratch, uint8_t* __restrict__ tail_done, uint32_t* __restrict__ status, uint32_t* tail0_out) {
  ZnFusedLds& L = *ZN_LDS_PTR(ZnFusedLds, &L_);     // (this is a real call: keep the LDS accesses DS operations)
  const uint32_t b = bt >> 2, sq = zn_uniform(bt & 3u);     // tail plane of the launch / huff0 stream of its block
  const ZnSeg S = zn_find_seg<3>(one, segs, nseg, b);
  *tail0_out = S.tail0;
  const ZnGeom g = S.g;
  const uint8_t* __restrict__ body = ZN_GLOBAL_PTR(const uint8_t, S.body); const uint64_t body_len = S.body_len;
  const uint8_t* body_end = body + body_len;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // (wave-uniform, and the compiler should know: everything derived from it lives in scalar registers)
  const uint32_t p = b - S.tail0;
  const uint64_t c = g.K - 1u;
  ZN_PT_DECL;
  const ZnPcMeta m = zn_pc_meta(g, body, body_len, p, c);
  if (!(m.ok && m.type == 1u && m.csize > 1u && m.csize < m.plen && m.plen >= ZN_TAIL_WG_MIN_PLANE && m.plen <= 4u * ZN_TAIL_SEGPAD)) return;
  const uint8_t* src = body + m.off;
  if (wave == 0) {
    uint8_t* tmp = (uint8_t*)&L.ring[0][0];
    const ZnWaveStats st = zn_wave_read_stats(src, m.csize, body_end, lane, tmp, L.symlist[0], L.rank_start[0], L.sym_start[0], tmp + 512);
    if (lane == 0) L.st[0] = st;
  }
  __syncthreads();
  const ZnWaveStats st = L.st[0];
  const int hs = st.hs; const uint32_t TL = st.tl;
  if (hs < 0 || TL > ZN_F_TLMAX || (uint32_t)hs >= m.csize || m.csize - (uint32_t)hs < 10u) return;
  __syncthreads();                             // (ring[0] held the parser's scratch)
  zn_fused_fill_luts(L, tid, TL, 0, zn_uniform(st.lmin));
  const uint8_t* js = src + hs; const uint32_t rem = m.csize - (uint32_t)hs;
  const uint32_t l1 = zn_uniform(zn_ld16(js)), l2 = zn_uniform(zn_ld16(js + 2)), l3 = zn_uniform(zn_ld16(js + 4));
  if (l1 + l2 + l3 + 6u > rem) return;
  const uint32_t l4 = rem - 6u - l1 - l2 - l3;
  if (l1 == 0 || l2 == 0 || l3 == 0 || l4 == 0) return;
  const uint32_t seg3 = (m.plen + 3u) / 4u;
  if (3u * seg3 >= m.plen) return;
  const uint32_t segw = (sq < 3u) ? seg3 : m.plen - 3u * seg3;        // this workgroup's stream: its symbols …
  const uint32_t so = 6u + (sq > 0 ? l1 : 0u) + (sq > 1 ? l2 : 0u) + (sq > 2 ? l3 : 0u);
  const uint8_t* stream = js + so; const uint32_t slen = (sq == 0) ? l1 : (sq == 1) ? l2 : (sq == 2) ? l3 : l4;      // … and its bytes
  __syncthreads();                             // lut16 (aliasing ring[0]) is dead from here on

  constexpr uint32_t UNIT1 = 64u * 16u;        // row of the single-plane instance
  ZnFusedPlane pl[1]; pl[0].off = m.off; pl[0].kind = ZN_KIND_HUF; pl[0].csize = m.csize;
  uint8_t* outq = scratch + (uint64_t)b * ZN_TAIL_SLOT + (uint64_t)sq * ZN_TAIL_SEGPAD;
  uint32_t Du = ((ZN_F_RING_BYTES - UNIT1 - 128u) * slen) / (256u * segw);
  Du = Du > ZN_F_DMAX ? ZN_F_DMAX : (Du < 1u ? 1u : Du);
  // the shared form takes the density the small-input kernel takes (the fused kernel's 4-dword sub-blocks, not its dense-code instance; same rule as zn_k_decode_wide)
  // (… measured the way that kernel measures it — against the row of a TWO-plane tensor: its rule is about the code's bits per symbol, ≥ 2.4, and the tile slots of
  //  4-dword sub-blocks; this function's own one-plane rows would make a bf16 exponent stream "3 dwords" — which is also why, until round 5, every bf16 / fp32 tail
  //  ran the looping form: Du == 3 has no compile-time instance)
  const uint32_t Dw = ((ZN_F_RING_BYTES - 512u - 128u) * slen) / (256u * segw);
  const bool dense = Dw > 4u && zn_uniform(st.lmin) >= ZN_F_DENSE_LMIN;
  const bool shared_form = Dw >= 4u && !dense && (Dw == 4u || zn_uniform(st.dom) < ZN_F_DOM_MAX);
  if (ZN_F_DCAP && Du > ZN_F_DCAP && st.dom < ZN_F_DOM_MAX) Du = ZN_F_DCAP;
  Du = (uint32_t)__builtin_amdgcn_readfirstlane((int)Du);
  bool ok = false;
  if (shared_form) {
    ok = zn_wide_chunk<1, 0, 4, ZnTailWideLds, true>(*reinterpret_cast<ZnTailWideLds*>(&L), g, body, body_end, outq, pl, segw, TL, js, l1, l2, l3, l4, sq);
#if defined(ZN_SIMT_EMULATOR)
    if (tid == 0) zn_dbg_tiles[ok ? 7 : 5]++;            // (emulated build: tail streams decoded by the shared form / attempts that gave up)
#endif
  }
  if (!ok) {
    // (one wave, the stream's tiles one after the other: the form of rounds 1-5 — also behind a shared-form attempt that gave up: it has written nothing that this does not overwrite)
    __syncthreads();
    const uint8_t* rawq[1] = {nullptr};
    bool ok1 = true;
    if (wave == 0) {
      ok1 = (Du == ZN_F_DCONST)
        ? zn_fused_wave<1, 0, ZN_F_DCONST>(g, body, body_end, outq, nullptr, pl, rawq, L.lut, L.ring[0], L.in[0], lane, segw, TL, Du, stream, slen, true ZN_PT_PASS)
        : zn_fused_wave<1, 0, 0>(g, body, body_end, outq, nullptr, pl, rawq, L.lut, L.ring[0], L.in[0], lane, segw, TL, Du, stream, slen, true ZN_PT_PASS);
      if (lane == 0) L.what[0] = ok1 ? 1u : 0u;
    }
    __syncthreads();
    ok = L.what[0] != 0u;
  }
  if (tid == 0) {
    if (ok) tail_done[bt] = 1;
    else atomicOr(status, ZN_DEV_CORRUPT);
  }
  ZN_PT_FLUSH();
}


// Further Huffman planes of a chunk whose first one zn_k_decode_fused has just decoded: one more pass per plane.
// The rows are already in the output with zero bytes in that plane; the pass XORs its bytes in (the un-rotate is a
// bit permutation, so it distributes over the XOR) — the delta instance of the wave function with the output itself
// as the base.  The tree description is parsed by wave 0 between the passes.  Returns 1 = ok, 0 = a stream did not
// decode cleanly, -1 = a plane this kernel does not take (the generic path redoes the chunk).
// (Inlined: as a real call it halved the speed of the common one-plane case — measured, profiles/README.md.)
template <int P>
__device__ __forceinline__ int zn_fused_more_passes(ZnFusedLds& L, const ZnGeom& g, const uint8_t* __restrict__ body, const uint8_t* body_end,
                                                    uint8_t* outq, uint32_t j, uint32_t more, uint32_t seg ZN_PT_PARAM) {
  constexpr int EPL = (P == 1) ? 16 : 8;
  constexpr uint32_t UNIT = 64u * EPL;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // (wave-uniform, and the compiler should know: everything derived from it lives in scalar registers)
  uint32_t* ring = L.ring[wave]; uint32_t* in = L.in[wave];
  bool ok = true;
  ZN_PT_SHARED;
  for (int p2 = 1; p2 < P; p2++) {
    if (!((more >> p2) & 1u)) continue;
    __syncthreads();                         // every wave is done with the tables and rings of the previous pass
    const uint64_t off2 = L.plane[j][p2].off; const uint32_t cs2 = L.plane[j][p2].csize;
    if (wave == 0) {
      uint8_t* scratch = (uint8_t*)&L.ring[0][0];
      ZnWaveStats st2 = zn_wave_read_stats(body + off2, cs2, body_end, lane, scratch, L.symlist[j], L.rank_start[j], L.sym_start[j], scratch + 512);
      if (st2.hs < 0 || st2.tl > ZN_F_TLMAX || (uint32_t)st2.hs >= cs2 || cs2 - (uint32_t)st2.hs < 10u) st2.hs = -1;
      if (lane == 0) L.st[j] = st2;
    }
    __syncthreads();
    const ZnWaveStats st2 = L.st[j];
    if (st2.hs < 0) return -1;
    const uint32_t TL2 = st2.tl;
    zn_fused_fill_luts(L, tid, TL2, j);
    const uint8_t* js = body + off2 + st2.hs; const uint32_t rem = cs2 - (uint32_t)st2.hs;
    const uint32_t l1 = zn_ld16(js), l2 = zn_ld16(js + 2), l3 = zn_ld16(js + 4);
    uint32_t l4 = 0; bool bad = false;
    if (l1 + l2 + l3 + 6u > rem) bad = true; else l4 = rem - 6u - l1 - l2 - l3;
    if (l1 == 0 || l2 == 0 || l3 == 0 || l4 == 0) bad = true;
    __syncthreads();                         // lut16 (aliasing ring[0]) is dead from here on
    if (bad) return -1;
    const uint32_t so = 6u + (wave > 0 ? l1 : 0u) + (wave > 1 ? l2 : 0u) + (wave > 2 ? l3 : 0u);
    const uint8_t* stream2 = js + so; const uint32_t slen2 = (wave == 0) ? l1 : (wave == 1) ? l2 : (wave == 2) ? l3 : l4;
    ZnFusedPlane pl2[P]; const uint8_t* rawq[P];
    for (int p = 0; p < P; p++) { pl2[p] = L.plane[j][p]; rawq[p] = body; if (p != p2) { pl2[p].kind = ZN_KIND_RLE; pl2[p].off = 0; } }
    uint32_t Du2 = ((ZN_F_RING_BYTES - UNIT - 128u) * slen2) / (256u * seg);
    Du2 = Du2 > ZN_F_DMAX ? ZN_F_DMAX : (Du2 < 1u ? 1u : Du2);
    Du2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)Du2);
#define ZN_ACC_ARGS g, body, body_end, outq, outq, pl2, rawq, L.lut, ring, in, lane, seg, TL2, Du2, stream2, slen2, false ZN_PT_PASS
    bool ok2;
    if (p2 == 1) ok2 = zn_fused_wave<P, (P >= 2 ? 1 : 0), 0, true>(ZN_ACC_ARGS);
    else if (p2 == 2) ok2 = zn_fused_wave<P, (P >= 4 ? 2 : 0), 0, true>(ZN_ACC_ARGS);
    else ok2 = zn_fused_wave<P, (P >= 4 ? 3 : 0), 0, true>(ZN_ACC_ARGS);
#undef ZN_ACC_ARGS
    ok = ok && ok2;
  }
  return ok ? 1 : 0;
}

// One workgroup decodes a GROUP of up to 4 consecutive chunks.  The tree description of a huff0 block is
// a serial job for one wave (zn_huf_wave.hpp), so the four waves first parse the descriptions of the
// group's four chunks side by side; after that the whole workgroup decodes the chunks one after the other
// (LUT fill by 256 threads, then wave w = stream w).  ncg = chunks per group (1..4, chosen by the host so
// that small inputs still spread over every CU).
// (the delta instance at 3 waves per SIMD — 168 VGPRs, fewer spills — was 50 % slower than at 4: occupancy matters
// more than the spills, measured in profiles/r01z_delta_path.txt; every instance runs at ZN_F_WAVES_PER_SIMD)
static_assert(sizeof(ZnFusedLds) * ZN_F_WAVES_PER_SIMD <= 160u * 1024u, "ZnFusedLds: the LDS budget of ZN_F_WAVES_PER_SIMD workgroups per CU");
// REST (launches without partial chunks and delta bases): a chunk this kernel does not take is decoded right here by the generic
// path's own device functions (zn_decode_rest.hpp: one wave per plane, then the merge) instead of being left to two more launches that would
// return at once in nearly every call; `descs_rest` = the launch's plane descriptors (the generic kernels' workspace).
template <int P, bool X, bool REST = false>
__global__ __launch_bounds__(ZN_F_THREADS, ZN_F_WAVES_PER_SIMD) void zn_k_decode_fused(ZnSeg one, const ZnSeg* __restrict__ segs, uint32_t nseg,
                                                                  uint8_t* __restrict__ done_all, uint8_t* __restrict__ pdone_all,
                                                                  uint32_t* __restrict__ status, uint32_t ntail,
                                                                  uint8_t* __restrict__ tail_scratch, uint8_t* __restrict__ tail_done, uint32_t only_pending,
                                                                  ZnPlaneDesc* __restrict__ descs_rest, uint32_t nchunk_wg, uint32_t merge_per,
                                                                  uint32_t* __restrict__ tailsync) {
  constexpr int EPL = (P == 1) ? 16 : 8;
  constexpr uint32_t UNIT = 64u * EPL;
  __shared__ ZnFusedLds L;

  // Round 5: the partial last chunk of a tensor is FINISHED inside this launch (REST instance, tailsync != null).  Its Huffman planes are decoded by the tail
  // workgroups at the front of the grid, as before; `merge_per` workgroups per such tensor at the END of the grid wait for them (tailsync[2 t]: tail workgroups of
  // tensor t that are through), classify the chunk's planes (the generic path's own item function, every one for itself) and merge its planes into the output
  // (the generic merge's item function, its 64 sub-ranges dealt out) — the two generic launches behind every call of a ragged tensor (20-30 us) are gone.
  // The tail workgroups have the lowest block indices and the merge workgroups are few: the wait is for workgroups that run or have run (and it is bounded).
  if (REST && tailsync && blockIdx.x >= ntail + nchunk_wg) {
    const ZnSeg one_c = one;
    zn_tail_merge_wg<P>(L, one_c, segs, nseg, blockIdx.x - (ntail + nchunk_wg), merge_per, tailsync, descs_rest, status, tail_done, tail_scratch);
    return;
  }

  // (`one` goes to the real functions of the cold paths — the tail workgroups, the rest instance's generic code — as a COPY made on that path: with its own
  //  address escaping, the by-value kernel argument lived in private memory and EVERY thread of EVERY workgroup stored its 96 bytes to scratch in the prologue and
  //  loaded them back — 24 KB of HBM writes per workgroup, 2.4 % of the kernel's writes at 4 GiB, 11 % at 64 MiB; profiles/r04_decode_experiments.txt)
  if (blockIdx.x < ntail) {
    const ZnSeg one_c = one; uint32_t tail0 = 0;
    zn_decode_tail_wg(L, one_c, segs, nseg, blockIdx.x, tail_scratch, tail_done, status, &tail0);
    if (REST && tailsync) {                      // (every tail workgroup of the tensor reports, whatever it found: the merge workgroups count them)
      __syncthreads();
      if (threadIdx.x == 0) { ZN_FLAG_RELEASE(); ZN_FLAG_ADD32(tailsync + 2u * (tail0 / (uint32_t)P), 1u); }
    }
    return;
  }
  const uint32_t wg = blockIdx.x - ntail;      // workgroup index among the full-chunk groups
  // (the segment comes back through private memory — a kernel argument or a table entry, chosen at run time — which makes
  //  every field per-lane data to the compiler: 64-bit pointers in vector registers, spilled and reloaded once per chunk,
  //  0.17 GB of scratch reads per GiB decoded.  They are uniform: scalar registers.)
  const ZnSeg S_ = zn_find_seg<0>(one, segs, nseg, wg);
  ZnSeg S;
  S.g.n = zn_uniform64(S_.g.n); S.g.chunk = zn_uniform64(S_.g.chunk); S.g.K = zn_uniform64(S_.g.K); S.g.P = zn_uniform(S_.g.P); S.g.rot = zn_uniform(S_.g.rot);
  S.body = (const uint8_t*)zn_uniform64((uint64_t)S_.body); S.body_len = zn_uniform64(S_.body_len); S.dst = (uint8_t*)zn_uniform64((uint64_t)S_.dst);
  S.chunk0 = zn_uniform64(S_.chunk0); S.desc0 = zn_uniform64(S_.desc0); S.wg0 = zn_uniform(S_.wg0); S.ncg = zn_uniform(S_.ncg);
  S.tail0 = zn_uniform(S_.tail0); S.has_tail = zn_uniform(S_.has_tail); S.xr = (const uint8_t*)zn_uniform64((uint64_t)S_.xr);
  const ZnGeom g = S.g;
  const uint8_t* __restrict__ body = ZN_GLOBAL_PTR(const uint8_t, S.body); const uint64_t body_len = S.body_len;
  uint8_t* __restrict__ dst = ZN_GLOBAL_PTR(uint8_t, S.dst); uint8_t* __restrict__ done = done_all + S.chunk0;
  uint8_t* __restrict__ pdone = pdone_all + S.desc0;   // the same flag per (plane, chunk)
// (status[1 + q], q = 0/1/2 for 1/2/4 planes: how many chunks of this launch are left to the generic kernels — they
//  return at once when it is zero)
// (REST: the chunk is decoded here and now by the generic path's code — its planes by waves 0 .. P-1, each on an LDS instance of its own laid over
//  this kernel's tables, then the merge by the whole workgroup; the done flag stays 0: "not by the fused kernel")
#define ZN_REST_CHUNK(c_) do { if constexpr (REST) { \
    static_assert(4u * sizeof(ZnPlanesLds) <= sizeof(ZnFusedLds), "four generic plane decoders fit over the fused kernel's LDS"); \
    __syncthreads(); \
    const ZnSeg one_c_ = one; \
    if (wave < (uint32_t)P) zn_decode_plane_item(reinterpret_cast<ZnPlanesLds*>(&L)[wave], one_c_, segs, nseg, S.desc0 + (uint64_t)wave * g.K + (c_), descs_rest, status, nullptr, lane); \
    __threadfence(); __syncthreads(); \
    for (uint32_t sub_ = 0; sub_ < ZN_MERGE_SUB; sub_++) zn_merge_chunk_item<P>(one_c_, segs, nseg, S.chunk0 + (c_), sub_, descs_rest, nullptr); \
    __syncthreads(); } } while (0)
#define ZN_SET_DONE(c_, v_) do { if (tid == 0) { done[c_] = (v_); if (!(v_)) atomicAdd(status + 1 + (P == 1 ? 0 : P == 2 ? 1 : 2), 1u); } if (tid < (uint32_t)P) pdone[(uint64_t)tid * g.K + (c_)] = (v_); } while (0)
  const uint32_t ncg = S.ncg;
  // behind the wide kernel (zn_decode_wide.hpp; one chunk per workgroup there and here): only the chunks it left pending
  if (only_pending && zn_uniform(done[(uint64_t)(wg - S.wg0) * ncg]) != 2u) return;

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // (wave-uniform, and the compiler should know: everything derived from it lives in scalar registers)
  const uint64_t c0 = (uint64_t)(wg - S.wg0) * ncg;
  const uint32_t nc = (g.K - c0 < (uint64_t)ncg) ? (uint32_t)(g.K - c0) : ncg;
  const uint32_t plen = (uint32_t)(g.chunk / P);
  const uint32_t seg = plen / 4u;             // symbols per stream == plane bytes per quarter
  const uint8_t* body_end = body + body_len;
  ZN_PT_DECL;

  // ---- metadata: one thread per (chunk, plane) ----
  if (tid < nc * (uint32_t)P) {
    const uint32_t j = tid / (uint32_t)P, p = tid % (uint32_t)P;
    const ZnPcMeta m = zn_pc_meta(g, body, body_len, p, c0 + j);
    ZnFusedPlane pl; pl.off = m.off; pl.csize = m.csize; pl.kind = 99u;   // 99 = not for this kernel
    if (m.ok && m.type <= 1u && zn_chunk_len(g, c0 + j) == g.chunk) {
      if (m.type == 0u) { if (m.csize >= plen) pl.kind = ZN_KIND_RAW; }
      else if (m.csize == plen) pl.kind = ZN_KIND_RAW;
      else if (m.csize == 1u) { pl.kind = ZN_KIND_RLE; pl.off = body[m.off]; }
      else if (m.csize > 1u && m.csize < plen) pl.kind = ZN_KIND_HUF;
    }
    L.plane[j][p] = pl;
  }
  __syncthreads();
  ZN_PT(0);   // metadata

  // ---- wave j: is chunk j ours, and if it has a Huffman plane, its tree description ----
  if (wave < nc) {
    int h = -1; uint32_t nhuf = 0;
    bool elig = (g.chunk % (4u * P * UNIT)) == 0 && ((((uint64_t)dst) & 15u) == 0);
    if (X) elig = elig && ((((uint64_t)S.xr) & 15u) == 0);   // (the host picks the X instance whenever a tensor has a base)
    for (int p = 0; p < P; p++) {
      const uint32_t kind = L.plane[wave][p].kind;
      if (kind == 99u) elig = false;
      if (kind == ZN_KIND_HUF) { if (h < 0) h = p; nhuf++; }     // h: the FIRST Huffman plane (further ones: extra passes below)
    }
    (void)nhuf;
    ZnWaveStats st; st.hs = 0; st.nsym = 0; st.tl = 0; st.lmin = 1; st.dom = 0;
    if (elig && h >= 0) {
      const uint32_t csize = L.plane[wave][h].csize;
      uint8_t* scratch = (uint8_t*)&L.ring[wave][0];          // weights + FSE cells; the ring is idle until the decode
      st = zn_wave_read_stats(body + L.plane[wave][h].off, csize, body_end, lane, scratch, L.symlist[wave], L.rank_start[wave],
                              L.sym_start[wave], scratch + 512);
      if (st.hs < 0 || st.tl > ZN_F_TLMAX || (uint32_t)st.hs >= csize || csize - (uint32_t)st.hs < 10u) elig = false;
    }
    if (lane == 0) { L.st[wave] = st; L.what[wave] = elig ? (uint32_t)(h + 2) : 0u; }   // 0: not ours, 1: no Huffman plane, 2+h
  }
  __syncthreads();
  ZN_PT(1);   // tree descriptions (one per wave)

  for (uint32_t j = 0; j < nc; j++) {
    const uint64_t c = c0 + j;
    if (j > 0) __syncthreads();              // the previous chunk's tables are no longer in use
    ZN_PT(21);  // wait for the slowest wave of the previous chunk
    const uint32_t what = zn_uniform(L.what[j]);          // (what comes out of LDS or HBM below is wave-uniform: scalar registers, scalar branches)
    if (REST && tailsync && what == 0u && S.has_tail && c == g.K - 1u) {              // the partial last chunk: the tail + merge workgroups of this launch
      if (tid == 0) done[c] = 0; if (tid < (uint32_t)P) pdone[(uint64_t)tid * g.K + c] = 0;
      continue;
    }
    if (what == 0u) { ZN_REST_CHUNK(c); ZN_SET_DONE(c, 0); continue; }
    const int h = (int)what - 2;
    ZnFusedPlane pl[P];
    uint32_t more = 0;                          // further Huffman planes (bit p): decoded by extra passes, zero bytes in this one
    for (int p = 0; p < P; p++) {
      pl[p].off = zn_uniform64(L.plane[j][p].off); pl[p].kind = zn_uniform(L.plane[j][p].kind); pl[p].csize = zn_uniform(L.plane[j][p].csize);
      if (p > h && h >= 0 && pl[p].kind == ZN_KIND_HUF) { more |= 1u << p; pl[p].kind = ZN_KIND_RLE; pl[p].off = 0; }
    }

    uint32_t TL = 0;
    const uint8_t* stream = nullptr; uint32_t slen = 0;
    bool bad = false;
    if (h >= 0) {
      // ---- decode tables ----
      uint64_t h_off = 0; uint32_t csize = 0;
      for (int p = 0; p < P; p++) if (p == h) { h_off = pl[p].off; csize = pl[p].csize; }
      const uint8_t* src = body + h_off;
      const ZnWaveStats st = L.st[j];
      const int hs = (int)zn_uniform((uint32_t)st.hs); TL = zn_uniform(st.tl);
      zn_fused_fill_luts(L, tid, TL, j, zn_uniform(st.lmin));
      // jump table → this wave's stream
      const uint8_t* js = src + hs; const uint32_t rem = csize - (uint32_t)hs;
      const uint32_t l1 = zn_uniform(zn_ld16(js)), l2 = zn_uniform(zn_ld16(js + 2)), l3 = zn_uniform(zn_ld16(js + 4));
      uint32_t l4 = 0;
      if (l1 + l2 + l3 + 6u > rem) bad = true; else l4 = rem - 6u - l1 - l2 - l3;
      if (l1 == 0 || l2 == 0 || l3 == 0 || l4 == 0) bad = true;
      const uint32_t so = 6u + (wave > 0 ? l1 : 0u) + (wave > 1 ? l2 : 0u) + (wave > 2 ? l3 : 0u);
      stream = js + so; slen = (wave == 0) ? l1 : (wave == 1) ? l2 : (wave == 2) ? l3 : l4;
      __syncthreads();                         // lut16 (aliasing ring[0]) is dead from here on
      ZN_PT(2);   // LUT fill
    }
    if (bad) { ZN_REST_CHUNK(c); ZN_SET_DONE(c, 0); continue; }   // malformed jump table: the generic path reports it

    // ---- per-wave: decode the stream tile by tile, flush rows ----
    const uint8_t* rawq[P];
    for (int p = 0; p < P; p++) rawq[p] = body + pl[p].off + (uint64_t)wave * seg;
    uint8_t* outq = dst + c * g.chunk + (uint64_t)wave * (g.chunk / 4u);
    const uint8_t* xq = (X && S.xr) ? ZN_GLOBAL_PTR(const uint8_t, S.xr) + c * g.chunk + (uint64_t)wave * (g.chunk / 4u) : nullptr;
    uint32_t* ring = L.ring[wave]; uint32_t* in = L.in[wave];
    // sub-block size (dwords): a tile of 64 sub-blocks should decode to about one staging buffer minus the
    // carried remainder, at this stream's average code length (8 slen / seg bits per symbol)
    uint32_t Du = ((ZN_F_RING_BYTES - UNIT - 128u) * slen) / (256u * seg);
    Du = Du > ZN_F_DMAX ? ZN_F_DMAX : (Du < 1u ? 1u : Du);
    // (dense code: ≥ 5 dwords wanted and no code shorter than 4 bits — the packed-record instance; otherwise the cap of round 2)
    const bool dense = ZN_F_DCONST2 && !X && P <= 2 && Du > ZN_F_DCAP && zn_uniform(L.st[j].lmin) >= ZN_F_DENSE_LMIN;
    if (dense) Du = ZN_F_DCONST2;
    else if (ZN_F_DCAP && Du > ZN_F_DCAP && zn_uniform(L.st[j].dom) < ZN_F_DOM_MAX) Du = ZN_F_DCAP;
    Du = (uint32_t)__builtin_amdgcn_readfirstlane((int)Du);
    bool ok;
#define ZN_WAVE_ARGS g, body, body_end, outq, xq, pl, rawq, L.lut, ring, in, lane, seg, TL, Du, stream, slen, false ZN_PT_PASS
#define ZN_WAVE_CASE(H_) ok = (Du == ZN_F_DCONST) ? zn_fused_wave<P, H_, ZN_F_DCONST, X>(ZN_WAVE_ARGS) \
                            : (dense) ? zn_fused_wave<P, H_, ((X || P > 2 || !ZN_F_DCONST2) ? 0 : ZN_F_DCONST2), X>(ZN_WAVE_ARGS) \
                            : zn_fused_wave<P, H_, 0, X>(ZN_WAVE_ARGS)
    // (one instance per Huffman plane index that exists for this P — nothing is instantiated twice)
    if (h < 0) ok = zn_fused_wave<P, -1, 0, X>(ZN_WAVE_ARGS);
    else if (P == 1 || h == 0) ZN_WAVE_CASE(0);
    else if (P == 2 || h == 1) ZN_WAVE_CASE((P >= 2 ? 1 : 0));
    else if (h == 2) ZN_WAVE_CASE((P >= 4 ? 2 : 0));
    else ZN_WAVE_CASE((P >= 4 ? 3 : 0));
#undef ZN_WAVE_CASE
#undef ZN_WAVE_ARGS
    // ---- further Huffman planes (deltas, sparse tensors: every plane compresses): one more pass per plane
    if (P >= 2 && __builtin_expect(more != 0u, 0)) {
      const int r2 = zn_fused_more_passes<P>(L, g, body, body_end, outq, j, more, seg ZN_PT_PASS);
      if (r2 < 0) { ZN_REST_CHUNK(c); ZN_SET_DONE(c, 0); continue; }   // a later plane this kernel does not take: the generic path redoes the chunk
      ok = ok && r2 > 0;
    }
    if (!ok) atomicOr(status, ZN_DEV_CORRUPT);
    ZN_SET_DONE(c, 1);
    ZN_PT_COUNT(19, 1);                        // chunks
  }
  ZN_PT_FLUSH();
}

#ifdef ZN_PHASE_TIMERS
extern "C" int zn_debug_phase_read(unsigned long long* out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(zn_phase_acc), sizeof(unsigned long long) * 64) != hipSuccess) return -2;
  if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(zn_phase_acc), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#endif

// chunks per workgroup.  The device runs `slots` workgroups at once (CUs x ZN_F_WAVES_PER_SIMD); a launch of W workgroups takes ceil(W / slots) rounds,
// and a round takes one parse of the group's tree descriptions (side by side on the four waves: ≈ 13 µs) plus its chunks one after the other (≈ 85 µs each
// with the chip full).  Until round 5 the rule was K / slots clamped to 1..4, which put 1 088 chunks into two rounds of one (185 µs; 173 as 544 groups of two),
// 2 304 into two rounds of two (334 µs; 263 as 768 groups of three) and 5 120 into two rounds of four (623 µs; 540 as five rounds of one): the size sweep of
// profiles/r05_decode_group_rule.txt — the cheapest (rounds x (parse + chunks)) of the four group sizes is the measured best one in every row of it.
static std::atomic<int> g_zn_decode_group{0};
uint32_t zn_decode_fused_group(uint64_t K) {
  static std::atomic<int> slots_of[64];          // per device (a node may mix parts); 0 = not asked yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); dev = 0; }
  int slots = slots_of[dev].load(std::memory_order_relaxed);
  if (slots == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); cus = 256; }
    slots = cus * ZN_F_WAVES_PER_SIMD;
    slots_of[dev].store(slots, std::memory_order_relaxed);     // (racing first calls store the same value)
  }
  uint32_t ncg = 1; uint64_t best = ~0ull;
  for (uint32_t m = 1; m <= 4u; m++) {
    const uint64_t wgs = K ? (K + m - 1u) / m : 1u, rounds = (wgs + (uint64_t)slots - 1u) / (uint64_t)slots;
    const uint64_t cost = rounds * (13u + 85u * m);
    if (cost <= best) { best = cost; ncg = m; }     // (a tie: the larger group — fewer parses)
  }
  const int forced = g_zn_decode_group.load(std::memory_order_relaxed);      // zn_set_decode_group (include/zipnn_hip.h): 0 = automatic
  if (forced >= 1 && forced <= 4) ncg = (uint32_t)forced;
  return ncg;
}
// The wide kernel (zn_decode_wide.hpp) takes a call whose full chunks number at most the CUs of the device — below that the fused
// kernel's workgroups leave most of the chip idle — if its tensors are split with the sign rotate (bf16 / fp32: the layouts whose Huffman plane
// is an exponent byte; measured: fp16 / fp8 calls only pay the extra parse, 2-12 us); up to two chunks per CU its 8-wave form, two workgroups per CU.
// Returns the waves per stream (4 / 2) or 0.  zn_set_decode_wide (include/zipnn_hip.h): 0 = never, 1 = automatic, 2 / 3 = always the 16- / 8-wave form.
static std::atomic<int> g_zn_decode_wide{1};
int zn_decode_use_wide(uint64_t K, bool delta, bool weights_like, uint64_t tail_wgs) {     // weights_like: sign-rotated layouts; tail_wgs: the tail workgroups of the call's partial last chunks (they want slots beside the full chunks')
  const int mode = g_zn_decode_wide.load(std::memory_order_relaxed);
  if (mode == 0 || delta || K == 0) return 0;
  if (mode == 2) return 4;
  if (mode == 3) return 2;
  if (!weights_like) return 0;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) { (void)hipGetLastError(); return 0; }
  return K + tail_wgs <= (uint64_t)cus ? 4 : K + tail_wgs <= 2ull * (uint64_t)cus ? 2 : 0;
}
extern "C" int zn_set_decode_wide(int mode) {
  if (mode < 0 || mode > 3) return -1;     // ZN_E_ARG
  g_zn_decode_wide.store(mode, std::memory_order_relaxed);
  return 0;
}
extern "C" int zn_decode_group_for(unsigned long long chunks) { return (int)zn_decode_fused_group((uint64_t)chunks); }
extern "C" int zn_set_decode_group(int chunks_per_workgroup) {
  if (chunks_per_workgroup < 0 || chunks_per_workgroup > 4) return -1;     // ZN_E_ARG
  g_zn_decode_group.store(chunks_per_workgroup, std::memory_order_relaxed);
  return 0;
}

bool zn_launch_decode_fused(int P, const ZnSeg& one, const ZnSeg* d_segs, uint32_t nseg, uint32_t total_wg,
                            uint8_t* d_done, uint8_t* d_pdone, uint32_t* d_status, uint32_t ntail, uint8_t* d_tail_scratch,
                            uint8_t* d_tail_done, bool delta, int wide, bool status_zeroed, ZnPlaneDesc* d_descs_rest, uint32_t* d_tailsync, hipStream_t stream) {
  // the rest instance: every launch without delta bases (its tile loops run as fast as the plain instance's — measured, 160 MiB .. 4 GiB — and the two generic
  // launches behind it are saved: 256 MiB 128.4 -> 125.9 us, 4 GiB 1514.9 -> 1512.9); partial last chunks (ntail > 0) are finished by merge workgroups at the
  // end of the same launch (d_tailsync: two zeroed words per tensor with a partial chunk); behind the wide kernel, whose launch has the tail workgroups, the merge workgroups of this one find their reports in
  if (delta || (ntail != 0 && !d_tailsync)) d_descs_rest = nullptr;
  const uint32_t ntt = ntail / (uint32_t)P;      // tensors with a partial last chunk (ntail: their Huffman-plane slots, P per tensor)
  uint32_t ntail_wg = 4u * ntail;                // … decoded by four workgroups per plane, one per huff0 stream (zn_decode_tail_wg)
  uint32_t merge_per = 0;
  if (d_descs_rest && ntail) { merge_per = 32u; while (merge_per > 1u && (uint64_t)merge_per * ntt > 4096u) merge_per >>= 1; }
  else d_tailsync = nullptr;
  const uint32_t nchunk_wg = total_wg;
  if (total_wg == 0) return false;
  const uint32_t only_pending = wide ? 1u : 0u;
  if (wide) {
    // small inputs: one 16-wave workgroup per chunk first; the fused kernel behind it takes what that one left pending (and the tails)
    const uint32_t zs = (!status_zeroed && ntail == 0) ? 1u : 0u;
#define ZN_GOW(P_, W_) hipLaunchKernelGGL((zn_k_decode_wide<P_, W_>), dim3(total_wg + ntail_wg + merge_per * ntt), dim3(256 * W_), 0, stream, one, d_segs, nseg, d_done, d_pdone, d_status, zs, ntail_wg, d_tail_scratch, d_tail_done, d_tailsync, d_descs_rest, nchunk_wg, merge_per)
    if (wide == 4) { if (P == 1) ZN_GOW(1, 4); else if (P == 2) ZN_GOW(2, 4); else ZN_GOW(4, 4); }
    else { if (P == 1) ZN_GOW(1, 2); else if (P == 2) ZN_GOW(2, 2); else ZN_GOW(4, 2); }
#undef ZN_GOW
    zn_note_kernel(wide == 4 ? (ntail ? (merge_per ? "zn_k_decode_wide+tail+merge" : "zn_k_decode_wide+tail") : "zn_k_decode_wide") : (ntail ? (merge_per ? "zn_k_decode_wide^2+tail+merge" : "zn_k_decode_wide^2+tail") : "zn_k_decode_wide^2"));
    ntail_wg = 0; merge_per = 0; d_tailsync = nullptr;   // (done, merge workgroups included: the launch below has neither — the partial chunks' done flags say "not pending")
  }
  total_wg += ntail_wg;                          // the tail workgroups come first
  total_wg += merge_per * ntt;                   // … and the merge workgroups of the partial chunks last
#define ZN_GO(P_, X_, R_) hipLaunchKernelGGL((zn_k_decode_fused<P_, X_, R_>), dim3(total_wg), dim3(ZN_F_THREADS), 0, stream, one, d_segs, nseg, d_done, d_pdone, d_status, ntail_wg, d_tail_scratch, d_tail_done, only_pending, d_descs_rest, nchunk_wg, merge_per, d_tailsync)
  if (d_descs_rest) { if (P == 1) ZN_GO(1, false, true); else if (P == 2) ZN_GO(2, false, true); else ZN_GO(4, false, true); }
  else if (!delta) { if (P == 1) ZN_GO(1, false, false); else if (P == 2) ZN_GO(2, false, false); else ZN_GO(4, false, false); }
  else { if (P == 1) ZN_GO(1, true, false); else if (P == 2) ZN_GO(2, true, false); else ZN_GO(4, true, false); }
#undef ZN_GO
  zn_note_kernel(d_descs_rest ? (ntail && !wide ? "zn_k_decode_fused^rest+tail+merge" : "zn_k_decode_fused^rest") : wide ? "zn_k_decode_fused^pending" : delta ? (ntail ? "zn_k_decode_fused^delta+tail" : "zn_k_decode_fused^delta") : (ntail ? "zn_k_decode_fused+tail" : "zn_k_decode_fused"));
  return d_descs_rest != nullptr;
}
