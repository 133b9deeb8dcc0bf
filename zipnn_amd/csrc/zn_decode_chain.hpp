// zn_decode_chain.hpp — the per-lane decode chain of the fused decompress kernel (zn_decode_fused.hip): the multi-symbol
// LUT entry format, the looping form of the chain (sync run-in, fallback) and the register-resident form (decode once into
// registers, then compact).  A header of its own so that the developer probe (scripts/ubench/pass_probe.hip) can compile
// and time exactly these functions in isolation.
// Replaces the inner loop of HUF_decompress4X1 (reference csrc/zipnn_core.c:807 → huff0).
#pragma once

#include "zn_common.hpp"

#ifndef ZN_F_TLMAX        /* (the including file may have said it already) */
#define ZN_F_TLMAX 11u
#endif
#ifndef ZN_IN_IDX
#define ZN_IN_IDX(i) (i)
#endif
// a scheduling fence between the unrolled steps of the register-resident decode: a step is one dependent chain, nothing
// is gained by interleaving two of them, and left alone the scheduler postpones the packing of the count bytes, which
// keeps every step's meta word alive (one more register per step)
#if !defined(ZN_SIMT_EMULATOR)
#define ZN_STEP_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ZN_STEP_FENCE() do { } while (0)
#endif

// multi-symbol LUT entry (8 bytes, read with one ds_read_b64) = up to 4 symbols of one TL-bit window.
//   .x  symbols 0-3, one byte each (unused bytes 0)
//   .y  meta:  bits 0-3   nb  = bits consumed when all `cnt` symbols are taken          (bits 4-7 stay 0)
//              bits 8-10  cnt = number of symbols (1..4)                                   (bits 11-15 stay 0)
//              bits 16-27 E1, E2, E3 = bit offset at which symbol 1 / 2 / 3 starts (nb for a symbol that does not exist)
// The layout is chosen for the decode step: a 64-bit shift takes its amount from the low 6 bits of the register, so the
// window is advanced with `w <<= meta`; and `acc += meta` accumulates the consumed bits in byte 0 and the symbol count
// in byte 1 with one add (the E fields only carry upwards).
#define ZN_E_META(cnt, nb, efields) ((nb) | ((cnt) << 8) | (efields))
#define ZN_M_NB(m) ((m) & 15u)
#define ZN_M_CNT(m) (((m) >> 8) & 7u)

// wave-wide exclusive prefix sum on the DPP network (6 v_add_u32 with a dpp modifier, no LDS traffic):
// row_shr 1/2/4/8 scan each row of 16 lanes, row_bcast15 / row_bcast31 carry the row totals forward.
__device__ __forceinline__ uint32_t zn_wave_excl_scan(uint32_t v, uint32_t lane, uint32_t* total) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  *total = (uint32_t)__builtin_amdgcn_readlane(x, 63);
  (void)lane;
  return (uint32_t)x - v;
}

// ------------------------------------------------------------------------------------------------------------------
// The decode chain of one lane = its sub-block of the tile.  The unread bits sit MSB-aligned in a 64-bit window and are
// consumed with one 64-bit shift per step; a refill (two stream-tile dwords from LDS) leaves ≥ 33 valid bits and a step
// consumes ≤ TL ≤ 11, so a refill feeds three steps.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t zn_window(const uint32_t* in, int32_t q) {     // q = (next unread bit) - 1 - base_bit
  q = q < 32 ? 32 : q;                                     // (only lanes that are done can be below; keeps in[j-1] in range)
  const int32_t j = q >> 5; const uint32_t sh = 31u - (uint32_t)(q & 31);
  const uint32_t d1 = in[ZN_IN_IDX(j)], d0 = in[ZN_IN_IDX(j - 1)];
  return (((uint64_t)d1 << 32) | d0) << sh;                // one v_lshlrev_b64
}

// boundary step: of the group in (syms, meta), exactly the symbols that START within the `rem` bits the lane still owns
// (symbol j starts E_j bits below the window top).  Returns the trimmed entry: .x symbols taken, .y = bits | count << 8.
__device__ __forceinline__ uint2 zn_trim_group(uint2 e, int32_t rem) {
  const uint32_t meta = e.y;
  const uint32_t have = ZN_M_CNT(meta);
  const uint32_t k0 = (uint32_t)(rem > 0) + (uint32_t)((int32_t)((meta >> 16) & 15u) < rem) + (uint32_t)((int32_t)((meta >> 20) & 15u) < rem) +
                      (uint32_t)((int32_t)((meta >> 24) & 15u) < rem);
  const uint32_t k = k0 < have ? k0 : have;
  const uint32_t F = ((meta >> 16) & 0xFFFu) | (ZN_M_NB(meta) << 12);        // E1, E2, E3, nb: bits consumed by 1 / 2 / 3 / 4 symbols
  const uint32_t nbk = k ? ((F >> (4u * k - 4u)) & 15u) : 0u;
  uint2 r;
  r.x = e.x & (k >= 4u ? 0xFFFFFFFFu : ~(0xFFFFFFFFu << (8u * k)));
  r.y = nbk | (k << 8);
  return r;
}

// … of a DENSE code's group (shortest code ≥ 4 bits: a TL ≤ 11-bit window holds at most TWO symbols, the second one E1 bits in; E1 = nb when there is none):
// 14 vector instructions where the general form above needs 25 — a dense sub-block closes with two boundary steps in nearly every tile
// (some lane of the 64 has a third symbol starting in its last 10 bits), and its streams are bound by vector-instruction issue (~75 % of the SIMDs' issue cycles: profiles/r05_dense_valu.txt)
__device__ __forceinline__ uint2 zn_trim_group_dense(uint2 e, int32_t rem) {
  const uint32_t meta = e.y;
  const uint32_t nb = ZN_M_NB(meta), e1 = (meta >> 16) & 15u;
  const bool t0 = rem > 0, t1 = (int32_t)e1 < rem && ZN_M_CNT(meta) >= 2u;
  uint2 r;
  r.x = t1 ? e.x : (t0 ? (e.x & 0xFFu) : 0u);
  r.y = t1 ? (nb | (2u << 8)) : (t0 ? (e1 | (1u << 8)) : 0u);
  return r;
}

// ---- the looping form of the chain (sync run-in; the rare tiles the register-resident form below does not take) ----
// MODE 0: advance only; 1: count symbols; 2: OR the symbols into the staging buffer.
struct ZnChain {
  int32_t pos, stop;          // next unread bit / boundary: symbols starting in (stop, pos] belong to this chain
  uint32_t n;                 // MODE 1: symbols counted
  uint32_t wpos;              // MODE 2: byte offset in the staging buffer of the next symbol
};
template <int MODE>
__device__ __forceinline__ void zn_chain_apply(ZnChain& c, uint2 e, uint32_t* stage) {
  if (MODE == 2) {
    const uint64_t sp = (uint64_t)e.x << ((c.wpos & 3u) << 3);
    uint32_t* d = (uint32_t*)((uint8_t*)stage + (c.wpos & ~3u));
    if ((uint32_t)sp) atomicOr(d, (uint32_t)sp);
    if ((uint32_t)(sp >> 32)) atomicOr(d + 1, (uint32_t)(sp >> 32));
    c.wpos += ZN_M_CNT(e.y);
  }
  if (MODE == 1) c.n += ZN_M_CNT(e.y);
  c.pos -= (int32_t)ZN_M_NB(e.y);
}
// U (wave-uniform): whole-group steps that NO lane of the wave can take too far — every lane that matters still owns at least
// 11 U + 10 bits (a step consumes ≤ TL ≤ 11; the caller derives U from the sub-block size or the run-in length, 0 = none).  They run
// without the per-step boundary test and its two selects: 4 instead of 9 vector instructions per step.  Lanes whose result the caller
// discards (fix-up iterations re-run the pass for a few lanes only) may run past their boundary there; MODE 2 callers pass U > 0
// only when every lane of the wave is writing.
// DENSE: the caller guarantees a code without lengths below 4 bits (the boundary steps then use zn_trim_group_dense).
template <int MODE, bool DENSE = false>
__device__ __forceinline__ void zn_fused_run(const uint2* lut, const uint32_t* in, int32_t base_bit, uint32_t TL, ZnChain& c, uint32_t* stage, int U = 0) {
  const uint32_t sh = 32u - TL;
  const int32_t mb = c.stop + (int32_t)TL - 1;
  for (int u = 0; u + 3 <= U; u += 3) {
    uint64_t w = zn_window(in, c.pos - 1 - base_bit);
    for (int s = 0; s < 3; s++) {
      const uint2 e = lut[(uint32_t)(w >> 32) >> sh];
      w <<= (e.y & 63u);
      zn_chain_apply<MODE>(c, e, stage);
    }
  }
  while (__any(c.pos > mb)) {                // whole groups while the group provably starts above `stop`
    uint64_t w = zn_window(in, c.pos - 1 - base_bit);
    for (int s = 0; s < 3; s++) {
      uint2 e = lut[(uint32_t)(w >> 32) >> sh];
      if (!(c.pos > mb)) { e.x = 0; e.y = 0; }
      w <<= (e.y & 63u);
      zn_chain_apply<MODE>(c, e, stage);
    }
  }
  while (__any(c.pos > c.stop)) {            // the boundary step(s): one iteration unless > 4 symbols start in the last TL - 1 bits
    const uint64_t w = zn_window(in, c.pos - 1 - base_bit);
    uint2 e = lut[(uint32_t)(w >> 32) >> sh];
    if (!DENSE && !(c.pos > c.stop)) { e.x = 0; e.y = 0; }
    zn_chain_apply<MODE>(c, DENSE ? zn_trim_group_dense(e, c.pos - c.stop) : zn_trim_group(e, c.pos - c.stop), stage);
  }
}

// ---- the register-resident form: decode ONCE, keep every step's group in registers, then compact --------------------
// The steps of all 64 lanes run in lock-step and the step loops are unrolled at compile time, so step t's LUT entry
// (symbols + meta) stays in a pair of registers of its own: TF whole-group steps + TB boundary steps.  The count pass
// of the classic two-pass scheme is this decode; its write pass — a second chain of dependent LUT look-ups — becomes a
// compaction of the recorded groups into the staging buffer with no table access and no dependence between its steps.
// (The records are NAMED members reached through a compile-time index — and only the count byte of each meta word is
//  kept, four to a register — not an array: an array would be promoted to
//  one 32-register vector value and spilled / reloaded whole around every element update.)
// v_alignbyte_b32: ({hi, lo} >> 8 * (sh & 3)) [31:0]
#if defined(ZN_SIMT_EMULATOR)
__device__ __forceinline__ uint32_t zn_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8u * (sh & 3u))); }
#else
__device__ __forceinline__ uint32_t zn_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
#endif

#define ZN_REC_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) \
                       X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27)
#define ZN_REC_MAX 28
struct ZnRec {
#define X(i) uint32_t s##i;
  ZN_REC_LIST(X)                 // symbols of step i
#undef X
  uint32_t c0, c1, c2, c3, c4, c5, c6;   // byte (i % 4) of c[i / 4] = byte 1 of step i's meta word (its symbol count)
};
template <int I> __device__ __forceinline__ uint32_t& zn_rec_s(ZnRec& r) {
#define X(i) if constexpr (I == i) return r.s##i; else
  ZN_REC_LIST(X) return r.s0;
#undef X
}
template <int G> __device__ __forceinline__ uint32_t& zn_rec_c(ZnRec& r) {
  if constexpr (G == 0) return r.c0; else if constexpr (G == 1) return r.c1; else if constexpr (G == 2) return r.c2;
  else if constexpr (G == 3) return r.c3; else if constexpr (G == 4) return r.c4; else if constexpr (G == 5) return r.c5; else return r.c6;
}
// record step I: symbols, and the count byte of `meta` into its byte of the packed counts (one v_perm_b32).
// DENSE (a code whose shortest length is ≥ 4 bits: an 11-bit window holds at most TWO symbols): two steps share a symbol register,
// step I in half I % 2 of register I / 2 — 14 + 7 registers for 28 steps instead of 28 + 7, which is what lets 6-dword sub-blocks
// (fp8, fp16's top byte) take the register-resident form without spilling.
template <int I, bool DENSE = false> __device__ __forceinline__ void zn_rec_put(ZnRec& r, uint32_t syms, uint32_t meta) {
  if constexpr (DENSE) { uint32_t& sr = zn_rec_s<I / 2>(r); if constexpr (I % 2 == 0) sr = syms; else sr = __builtin_amdgcn_perm(syms, sr, 0x05040100u); }     // (≤ 2 symbols: the upper half of `syms` is zero)
  else zn_rec_s<I>(r) = syms;
  constexpr uint32_t sel = (I % 4 == 0) ? 0x03020105u : (I % 4 == 1) ? 0x03020500u : (I % 4 == 2) ? 0x03050100u : 0x05020100u;
  uint32_t& c = zn_rec_c<I / 4>(r);
  c = __builtin_amdgcn_perm(meta, c, sel);
}
template <int I> __device__ __forceinline__ uint32_t zn_rec_cnt(ZnRec& r) { return (zn_rec_c<I / 4>(r) >> (8 * (I % 4))) & 7u; }
template <int I, bool DENSE> __device__ __forceinline__ uint32_t zn_rec_sym(ZnRec& r) {
  if constexpr (DENSE) return (I % 2 == 0) ? (zn_rec_s<I / 2>(r) & 0xFFFFu) : (zn_rec_s<I / 2>(r) >> 16);
  else return zn_rec_s<I>(r);
}
template <int N> struct ZnIdx { static constexpr int v = N; };
template <int I, int N, typename F> __device__ __forceinline__ void zn_static_for(F&& f) {
  if constexpr (I < N) { f(ZnIdx<I>{}); zn_static_for<I + 1, N>(f); }
}

//   U:  the first U steps run unchecked (the tile is regular — every lane owns a whole sub-block; a step consumes ≤ TL
//       bits, so ⌊(32 D - 31) / 11⌋ steps cannot reach the boundary); U = 0 for the last tile of a stream.
// pos0 / stop: the lane decodes the symbols that start in (stop, pos0]; lanes with own == false sit out (zero records).
// acc (out): byte 0 = bits consumed, byte 1 = symbols.  nfull / nbnd (out, uniform): slots in use.
// Returns false when TF steps did not bring every lane to its boundary (the caller falls back to the looping form).
// A fix-up iteration (some lane started from a wrong position) simply runs the whole pass again from the corrected
// starts: the lanes run in lock-step, so decoding all of them costs what decoding the wrong ones would.
template <int TF, int TB, int U, bool DENSE = false>
__device__ __forceinline__ bool zn_pass1(const uint2* lut, const uint32_t* in, int32_t base_bit, uint32_t TL, int32_t pos0, int32_t stop,
                                         bool own, ZnRec& rec, uint32_t& acc_out, int& nfull, int& nbnd) {
  static_assert(TF + TB <= ZN_REC_MAX, "more record slots than ZnRec has");
  const uint32_t sh = 32u - TL;
  const int32_t R = own ? pos0 - stop : -1000;            // bits this lane owns; lanes that sit out never become active
  const int32_t lim_full = R - (int32_t)TL + 1;            // whole-group steps while consumed < lim_full
  const int32_t qb = pos0 - 1 - base_bit;
  uint64_t w = 0;
  uint32_t acc = 0;
  int t_end = TF; bool live = true;                        // (uniform)
  zn_static_for<0, TF>([&](auto I) {
    constexpr int t = decltype(I)::v;
    constexpr bool checked = t >= U;
    if (live) {
      bool act = true;
      if (checked) {
        act = (int32_t)(acc & 0xFFu) < lim_full;
        if (!__any(act)) { t_end = t; live = false; }
      }
      if (live) {
        if (t % 3 == 0) w = zn_window(in, qb - (int32_t)(acc & 0xFFu));
        uint2 e = lut[(uint32_t)(w >> 32) >> sh];
        if (checked && !act) { e.x = 0; e.y = 0; }
        w <<= (e.y & 63u);
        acc += e.y;
        zn_rec_put<t, DENSE>(rec, e.x, e.y);
        ZN_STEP_FENCE();
      }
    }
  });
  if (live && __any((int32_t)(acc & 0xFFu) < lim_full)) return false;
  int b_end = TB; live = true;
  zn_static_for<0, TB>([&](auto I) {
    constexpr int b = decltype(I)::v;
    if (live) {
      const int32_t rem = R - (int32_t)(acc & 0xFFu);
      if (!__any(rem > 0)) { b_end = b; live = false; }
      if (live) {
        w = zn_window(in, qb - (int32_t)(acc & 0xFFu));
        uint2 e = lut[(uint32_t)(w >> 32) >> sh];
        if (!DENSE && !(rem > 0)) { e.x = 0; e.y = 0; }           // (the dense trim takes nothing from a lane with rem <= 0 by itself)
        e = DENSE ? zn_trim_group_dense(e, rem) : zn_trim_group(e, rem);
        acc += e.y;
        zn_rec_put<TF + b, DENSE>(rec, e.x, e.y);
        ZN_STEP_FENCE();
      }
    }
  });
  nfull = __builtin_amdgcn_readfirstlane(t_end); nbnd = __builtin_amdgcn_readfirstlane(b_end);     // (uniform; the compiler cannot always tell)
  acc_out = acc;
  return true;
}

// compaction: the recorded groups of this lane go to bytes [wpos, wpos + n) of the staging buffer (ds_or_b32: the first
// and the last dword are shared with the neighbouring lanes).  No table reads, no dependent chain: the byte position is
// a running sum of the counts.  Slots a lane did not use hold zeros.
// A group (≤ 4 bytes) lands in two consecutive dwords.  Its byte shift is done with v_alignbyte_b32, which takes the
// byte count from the low two bits of a register as they are: with g = (-wpos) & 3,
//   {syms, 0} >> 8g = syms << 8(4-g)   and   {0, syms} >> 8g = syms >> 8g
// are the two dwords for wpos & 3 = 1, 2, 3; for wpos & 3 == 0 (g = 0) they come out as (0, syms), i.e. one dword late,
// which is exactly right if the pair is stored one dword early: the pair always goes to ((wpos - 1) & ~3).  (No 64-bit
// shift: its source would be a register PAIR per step with a zero upper half, twice the registers.)
// `hook(ZnIdx<t>)` runs after step t (the caller spreads its HBM requests for the flush over the pass there: their
// destination registers come into use as the record registers fall out of it).
// AMASK: byte-offset mask of the pair's first dword — ~3 for a linear staging buffer; (size - 4) for a CIRCULAR one of a power-of-two
// size (zn_decode_wide.hpp), which has one more dword behind its end: the pair that starts in the last dword puts its second half
// there, and the flush ORs that dword into dword 0.
template <int TF, int TB, bool DENSE = false, uint32_t AMASK = ~3u, typename HOOK>
__device__ __forceinline__ void zn_pass2(uint32_t* stage, uint32_t wpos, ZnRec& rec, int nfull, int nbnd, HOOK&& hook) {
  uint32_t wm1 = wpos - 1u;                                // (wpos == 0: the pair starts one dword below the buffer and that dword gets a zero)
  auto put = [&](uint32_t cnt, uint32_t sv) {
    const uint32_t g = ~wm1;                               // low two bits = (-wpos) & 3
    uint32_t* d = (uint32_t*)((uint8_t*)stage + (int32_t)(wm1 & AMASK));
    // (per DWORD: a lane stays out of an atomic whose operand is zero — with wpos == 0 the pair starts one dword BELOW the staging buffer, the neighbouring
    //  wave's last dword or the LUT's, and only this test keeps that zero-valued atomic from being issued.  Measured again in round 6
    //  (profiles/r06_decode_experiments.txt): every lane, every dword — 12 scalar instructions fewer per record — is 6.7 % SLOWER on bf16 (1.602 vs 1.501 ms), 5 % on fp8:
    //  the LDS pipe pays for an atomic whether or not it changes anything)
    const uint32_t lo = zn_alignbyte(sv, 0u, g), hi = zn_alignbyte(0u, sv, g);
    if (lo) atomicOr(d, lo);
    if (hi) atomicOr(d + 1, hi);
    wm1 += cnt;
    ZN_STEP_FENCE();
  };
  if constexpr (DENSE) {      // dense records: the two steps of a register leave together
    // The two steps that share a record register (≤ 2 + 2 bytes) are written as ONE group: the second step's bytes are moved down
    // next to the first's (one v_perm_b32, selector by the first step's count), one pair of dwords instead of two — half the
    // ds_or_b32 of a dense stream's compaction, whose LDS time they are.  A slot that did not run in this pass (beyond nfull /
    // nbnd: its count byte is stale, its half of the register zero or stale) counts zero bytes and is dropped by the selector.
    zn_static_for<0, (TF + TB + 1) / 2>([&](auto R_) {
      constexpr int r = decltype(R_)::v, a = 2 * r, b = 2 * r + 1;
      const bool va = a < TF ? a < nfull : (a - TF) < nbnd;                                       // (uniform)
      const bool vb = b < TF + TB && (b < TF ? b < nfull : (b - TF) < nbnd);
      if (va || vb) {
        uint32_t ca = 0, cb = 0;
        if (va) ca = zn_rec_cnt<a>(rec);
        if constexpr (b < TF + TB) { if (vb) cb = zn_rec_cnt<b>(rec); }
        const uint32_t sr = zn_rec_s<r>(rec);
        const uint32_t sel = (ca >= 2u) ? 0x03020100u : (ca == 1u) ? 0x0c030200u : 0x0c0c0302u;
        put(ca + cb, __builtin_amdgcn_perm(sr, sr, sel));
        if (va) hook(ZnIdx<a>{});
        if constexpr (b < TF + TB) { if (vb) hook(ZnIdx<b>{}); }
      }
    });
  } else {
    zn_static_for<0, TF>([&](auto I) { constexpr int t = decltype(I)::v; if (t < nfull) { put(zn_rec_cnt<t>(rec), zn_rec_sym<t, DENSE>(rec)); hook(I); } });
    zn_static_for<0, TB>([&](auto I) { constexpr int b = decltype(I)::v; if (b < nbnd) { put(zn_rec_cnt<TF + b>(rec), zn_rec_sym<TF + b, DENSE>(rec)); hook(ZnIdx<TF + b>{}); } });
  }
}

