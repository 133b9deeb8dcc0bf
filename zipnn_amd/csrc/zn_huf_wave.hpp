// zn_huf_wave.hpp — huff0 tree description → decode-table layout, by ONE WAVE.
//
// The tree description (HUF_readStats of zstd 1.4.8) is an FSE-coded list of
// ≤255 code weights.  FSE decoding is a serial state chain, so a per-lane implementation
// spends ~100 k cycles per chunk in LDS round trips.  Here the whole wave executes the chain
// redundantly on wave-uniform values: the ≤128 header bytes live in one VGPR spread across the
// lanes (lane i = dword i), the 64-cell FSE decode table in another (lane u = cell u), and every
// table/bit access is a v_readlane with a uniform index — the chain runs on the scalar ALU with
// no memory latency in it.  The FSE table itself, the weight statistics and the canonical symbol
// order are built lane-parallel with ballots.
//
// All 64 lanes of the calling wave must call these functions together (uniform control flow).
#pragma once

#include "zn_common.hpp"

typedef uint32_t __attribute__((aligned(1))) zn_u32u_w;

#ifdef ZN_PHASE_TIMERS_SUB      // (global atomics per sub-phase: distorts the enclosing phase, so separate switch)
#define ZN_WT_DECL unsigned long long zn_wt0_ = __builtin_readcyclecounter()
#define ZN_WT(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&zn_phase_acc[i], t_ - zn_wt0_); zn_wt0_ = t_; } while (0)
#else
#define ZN_WT_DECL do { } while (0)
#define ZN_WT(i) do { } while (0)
#endif

struct ZnWaveHdr { uint32_t v; uint32_t limit_bits; };   // v: lane i holds bytes [4i, 4i+4) of the block

__device__ __forceinline__ uint32_t zn_rl(uint32_t v, uint32_t idx) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)idx); }

// nb ≤ 16 bits at absolute bit position `bitpos` of the staged block (LSB-first), zero past `limit`
__device__ __forceinline__ uint32_t zn_wbits(const ZnWaveHdr& H, uint32_t bitpos, uint32_t nb, uint32_t limit) {
  if (nb == 0 || bitpos >= limit) return 0;
  const uint32_t k = (bitpos >> 5) & 63u;
  const uint64_t w = ((uint64_t)zn_rl(H.v, (k + 1u) & 63u) << 32) | zn_rl(H.v, k);
  uint32_t x = (uint32_t)(w >> (bitpos & 31u)) & ((1u << nb) - 1u);
  const uint32_t valid = limit - bitpos;
  if (valid < nb) x &= (1u << valid) - 1u;
  return x;
}

// FSE-coded weights.  block = tree description incl. its first byte; isz = block[0] (< 128).
// Writes weights to sh_w[0..n) (LDS) and returns n, or -1.  sh_cell: 64 bytes of LDS scratch.
__device__ inline int zn_wave_fse_weights(const ZnWaveHdr& H, uint32_t isz, uint32_t lane, uint8_t* sh_w, uint8_t* sh_cell) {
  const uint32_t F0 = 8u, FEND = 8u * (1u + isz);     // the FSE block occupies bits [F0, FEND)
  if (isz < 2u) return -1;
  ZN_WT_DECL;
  // ---- FSE_readNCount (uniform) ----
  int norm[13];
  for (int i = 0; i < 13; i++) norm[i] = 0;
  uint32_t bitpos = F0, nsym = 0;
  const uint32_t tl = zn_wbits(H, bitpos, 4, FEND) + ZN_FSE_LOG_MIN; bitpos += 4;
  if (tl > ZN_WEIGHT_FSE_LOG) return -1;
  {
    int remaining = (1 << tl) + 1, threshold = 1 << tl, nb_bits = (int)tl + 1, prev0 = 0;
    while (remaining > 1 && nsym <= 12u) {
      if (prev0) {
        uint32_t n0 = nsym;
        while (zn_wbits(H, bitpos, 16, FEND) == 0xFFFFu) { n0 += 24; bitpos += 16; if (bitpos > FEND) return -1; }
        while (zn_wbits(H, bitpos, 2, FEND) == 3u) { n0 += 3; bitpos += 2; if (bitpos > FEND) return -1; }
        n0 += zn_wbits(H, bitpos, 2, FEND); bitpos += 2;
        if (n0 > 12u) return -1;
        nsym = n0;                                     // skipped symbols keep norm 0
      }
      const int mx = (2 * threshold - 1) - remaining; int cval;
      const int lo = (int)zn_wbits(H, bitpos, (uint32_t)nb_bits - 1u, FEND);
      if (lo < mx) { cval = lo; bitpos += (uint32_t)nb_bits - 1u; }
      else { cval = (int)zn_wbits(H, bitpos, (uint32_t)nb_bits, FEND); if (cval >= threshold) cval -= mx; bitpos += (uint32_t)nb_bits; }
      cval--;
      remaining -= cval < 0 ? -cval : cval;
      for (int i = 0; i < 13; i++) if ((uint32_t)i == nsym) norm[i] = cval;   // static indexing keeps norm[] in registers
      nsym++; prev0 = !cval;
      if (remaining < 1) return -1;
      while (remaining < threshold) { nb_bits--; threshold >>= 1; }
    }
    if (remaining != 1 || bitpos > FEND) return -1;
  }
  const uint32_t B0 = (bitpos + 7u) & ~7u;             // backward bit-stream starts at the next byte
  if (B0 >= FEND) return -1;
  ZN_WT(10);   // readNCount

  // ---- FSE decode table, lane = cell ----
  const uint32_t size = 1u << tl, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
  uint32_t nlow = 0;
  for (int s = 0; s < 13; s++) nlow += (norm[s] == -1) ? 1u : 0u;
  const uint32_t high = size - 1u - nlow;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64u - lane));
  {
    const uint32_t p = (lane * step) & mask;
    const bool valid = lane < size && p <= high;
    const uint32_t t = (uint32_t)__popcll(__ballot(valid) & lt);     // occurrence number of this visit
    uint32_t sym = 0, cum = 0;
    for (int s = 0; s < 13; s++) { cum += norm[s] > 0 ? (uint32_t)norm[s] : 0u; sym += (cum <= t) ? 1u : 0u; }
    if (valid && sym < 13u) sh_cell[p] = (uint8_t)sym;
    uint32_t top = size - 1u;
    for (int s = 0; s < 13; s++) if (norm[s] == -1) { if (lane == 0) sh_cell[top] = (uint8_t)s; top--; }
  }
  __builtin_amdgcn_wave_barrier();
  uint32_t entry = 0;
  {
    const uint32_t cs = (lane < size) ? sh_cell[lane] : 255u;
    uint32_t ns = 1;
    for (int s = 0; s < 13; s++) {
      const uint64_t m = __ballot(cs == (uint32_t)s);
      if (cs == (uint32_t)s) ns = (norm[s] == -1 ? 1u : (uint32_t)norm[s]) + (uint32_t)__popcll(m & lt);
    }
    const uint32_t nb = tl - zn_hb32(ns ? ns : 1u);
    entry = (cs & 0xFFu) | (nb << 8) | ((((ns << nb) - size) & 0xFFFFu) << 16);
  }

  ZN_WT(11);   // FSE decode table
  // ---- two interleaved states over the backward stream, all on wave-uniform (scalar) values ----
  // `win` holds the next unread bits top-aligned (MSB = bit pos-1 of the stream), zero below bit 0;
  // decoded weights are packed as nibbles into one VGPR (lane o>>3, nibble o&7) and stored once at the end.
  const uint32_t bn = (FEND - B0) >> 3;
  const uint32_t lastb = zn_wbits(H, FEND - 8u, 8, FEND);
  if (lastb == 0) return -1;
  int32_t pos = (int32_t)(8u * (bn - 1u)) + (int32_t)zn_hb32(lastb);
  uint64_t win = 0; int32_t avail = 0;
  uint32_t wv = 0;                                     // 8 four-bit weights per lane (weights are ≤ 12)
#define ZN_WREFILL() do { \
    const uint32_t abs_ = B0 + (uint32_t)pos - 1u; const uint32_t k_ = abs_ >> 5, r_ = abs_ & 31u; \
    const uint64_t hi_ = ((uint64_t)zn_rl(H.v, k_ & 63u) << 32) | (k_ >= 1u ? zn_rl(H.v, (k_ - 1u) & 63u) : 0u); \
    const uint64_t lo_ = (k_ >= 2u) ? zn_rl(H.v, (k_ - 2u) & 63u) : 0u; \
    win = (hi_ << (31u - r_)) | (lo_ >> (r_ + 1u)); \
    avail = pos < 64 ? pos : 64; \
    if (pos < 64) win &= ~0ull << (64 - pos); } while (0)
#define ZN_WTAKE(nb_, out_) do { const uint32_t n_ = (nb_); \
    if (avail < (int32_t)n_ && pos > avail) ZN_WREFILL(); \
    out_ = n_ ? (uint32_t)(win >> (64u - n_)) : 0u; win = n_ ? (win << n_) : win; avail -= (int32_t)n_; pos -= (int32_t)n_; } while (0)
#define ZN_WPUT(sym_) do { const uint32_t sv_ = ((sym_) & 0xFu) << (4u * ((uint32_t)o & 7u)); \
    wv |= (lane == ((uint32_t)o >> 3)) ? sv_ : 0u; o++; } while (0)          /* weight o → nibble o&7 of lane o>>3 */
  if (pos > 0) ZN_WREFILL();
  uint32_t s1, s2;
  ZN_WTAKE(tl, s1);
  ZN_WTAKE(tl, s2);
  if (pos < 0) return -1;
  int o = 0;
  // Fast part: eight weights a round while the stream still holds the 48 bits they can take at most (no end tests inside; the two
  // states' look-ups are independent and issue back to back; the eight nibbles are collected in a scalar and enter the vector
  // register once).  The tail below finishes with the end tests of FSE_decompress_usingDTable.
  while (pos >= 48 && o + 8 <= 254) {
    if (avail < 48) ZN_WREFILL();
    uint32_t sacc = 0, used = 0;
#define ZN_WFAST2(k_) do { const uint32_t e1_ = zn_rl(entry, s1 & 63u), e2_ = zn_rl(entry, s2 & 63u); \
      sacc |= ((e1_ & 0xFu) << (8 * (k_))) | ((e2_ & 0xFu) << (8 * (k_) + 4)); \
      const uint32_t n1_ = (e1_ >> 8) & 0xFFu, n2_ = (e2_ >> 8) & 0xFFu; \
      const uint32_t v1_ = (uint32_t)((win >> 1) >> (63u - n1_)); win <<= n1_; \
      const uint32_t v2_ = (uint32_t)((win >> 1) >> (63u - n2_)); win <<= n2_; \
      s1 = (e1_ >> 16) + v1_; s2 = (e2_ >> 16) + v2_; used += n1_ + n2_; } while (0)
    ZN_WFAST2(0); ZN_WFAST2(1); ZN_WFAST2(2); ZN_WFAST2(3);
#undef ZN_WFAST2
    wv = (lane == ((uint32_t)o >> 3)) ? sacc : wv;
    avail -= (int32_t)used; pos -= (int32_t)used; o += 8;
  }
  for (;;) {
    if (o >= 254) return -1;
    { const uint32_t e = zn_rl(entry, s1 & 63u); ZN_WPUT(e); uint32_t v; ZN_WTAKE((e >> 8) & 0xFFu, v); s1 = (e >> 16) + v; }
    if (pos < 0) { const uint32_t e = zn_rl(entry, s2 & 63u); ZN_WPUT(e); break; }
    if (o >= 254) return -1;
    { const uint32_t e = zn_rl(entry, s2 & 63u); ZN_WPUT(e); uint32_t v; ZN_WTAKE((e >> 8) & 0xFFu, v); s2 = (e >> 16) + v; }
    if (pos < 0) { const uint32_t e = zn_rl(entry, s1 & 63u); ZN_WPUT(e); break; }
  }
#undef ZN_WREFILL
#undef ZN_WTAKE
#undef ZN_WPUT
  ZN_WT(12);   // FSE state chain
  if (lane < 32u)                                      // unpack: lane l holds weights 8l … 8l+7
    for (uint32_t k = 0; k < 8u; k++) sh_w[8u * lane + k] = (uint8_t)((wv >> (4u * k)) & 0xFu);
  return o;
}

// What the table builders need to know about one huff0 block.
struct ZnWaveStats { int hs; uint32_t nsym, tl, lmin, dom; };   // dom: code space of the most populated code length, in 1/256 of the whole

// HUF_readStats + canonical ordering, by one wave.
//   src/csize: the huff0 block in the body (any alignment); body_end bounds the staging reads.
//   sh_w[256]: weights (LDS, out); sh_symlist[256]: symbols ordered by (weight, symbol) (out);
//   sh_rank_start[14]: first LUT cell of each weight class, [13] = total cells;
//   sh_sym_start[14]: first symlist index of each weight class; sh_cell: 64 bytes scratch.
// Returns hs < 0 on malformed input.
// (A real function: its LDS pointers arrive as generic pointers — FLAT loads and stores.  Measured in round 3: telling it that they are
//  LDS (ds_ operations, which queue behind the decode passes of the CU's other workgroups) costs 1 % on bf16; inlining it gains 0.4 %.)
__device__ inline ZnWaveStats zn_wave_read_stats(const uint8_t* src, uint32_t csize, const uint8_t* body_end, uint32_t lane,
                                                 uint8_t* sh_w, uint8_t* sh_symlist, uint32_t* sh_rank_start,
                                                 uint32_t* sh_sym_start, uint8_t* sh_cell) {
  ZnWaveStats R; R.hs = -1; R.nsym = 0; R.tl = 0; R.lmin = 1; R.dom = 0;
  if (csize == 0) return R;
  ZN_WT_DECL;
  // stage the first 256 bytes of the block across the lanes
  ZnWaveHdr H; H.v = 0; H.limit_bits = 0;
  {
    const uint8_t* a = src + 4u * lane;
    if (a + 4 <= body_end) H.v = *(const zn_u32u_w*)a;
    else for (int b = 0; b < 4; b++) if (a + b < body_end) H.v |= (uint32_t)a[b] << (8 * b);
  }
  const uint32_t h0 = zn_rl(H.v, 0) & 0xFFu;
  uint32_t isz, osz;
  if (h0 >= 128u) {                                   // raw 4-bit weights
    osz = h0 - 127u; isz = (osz + 1u) / 2u;
    if (isz + 1u > csize) return R;
    for (uint32_t q = 0; q < 128u; q += 64u) {          // uniform trip count: the shuffle is a wave collective
      const uint32_t n = q + lane, byte_i = 1u + n / 2u;
      const uint32_t b = (__shfl(H.v, (int)(byte_i >> 2)) >> (8u * (byte_i & 3u))) & 0xFFu;
      if (n < osz) sh_w[n] = (uint8_t)((n & 1u) ? (b & 15u) : (b >> 4));
    }
  } else {
    isz = h0;
    if (isz + 1u > csize) return R;
    const int r = zn_wave_fse_weights(H, isz, lane, sh_w, sh_cell);
    if (r < 0) return R;
    osz = (uint32_t)r;
  }
  __builtin_amdgcn_wave_barrier();
  ZN_WT(13);   // stage + weights (incl. 10-12)

  // ---- weight statistics (ballots) ----
  uint32_t cnt[13]; bool bad = false;
  for (int v = 0; v < 13; v++) cnt[v] = 0;
  for (uint32_t q = 0; q < 256u; q += 64u) {
    const uint32_t s = q + lane; const uint32_t w = (s < osz) ? sh_w[s] : 0u;
    if (__any(w >= ZN_HUF_LOG_MAX)) bad = true;
    for (uint32_t v = 1; v < 12; v++) cnt[v] += (uint32_t)__popcll(__ballot(w == v));
  }
  if (bad) return R;
  uint32_t total = 0;
  for (uint32_t v = 1; v < 12; v++) total += cnt[v] << (v - 1u);
  if (total == 0) return R;
  const uint32_t tl = zn_hb32(total) + 1u;
  if (tl > ZN_HUF_LOG_MAX) return R;
  const uint32_t rest = (1u << tl) - total;
  if ((1u << zn_hb32(rest)) != rest) return R;
  const uint32_t last = zn_hb32(rest) + 1u;            // implied weight of the last symbol (may be 12 when tl == 12)
  if (lane == 0) sh_w[osz] = (uint8_t)last;
  for (uint32_t v = 1; v < 13; v++) if (v == last) cnt[v]++;
  if (cnt[1] < 2u || (cnt[1] & 1u)) return R;
  const uint32_t nsym = osz + 1u;
  __builtin_amdgcn_wave_barrier();
  ZN_WT(14);   // weight statistics

  // ---- canonical order: by weight ascending (longest codes first), symbol ascending inside ----
  uint32_t rs[14], ss[14]; uint32_t cells = 0, syms = 0, vmax = 1;
  rs[0] = 0; ss[0] = 0;
  uint32_t dom_cells = 0;
  for (uint32_t v = 1; v <= 12; v++) { rs[v] = cells; ss[v] = syms; cells += cnt[v] << (v - 1u); syms += cnt[v]; if (cnt[v]) vmax = v; if ((cnt[v] << (v - 1u)) > dom_cells) dom_cells = cnt[v] << (v - 1u); }
  rs[13] = cells; ss[13] = syms;
  if (cells != (1u << tl)) return R;
  for (uint32_t i = 0; i < 14; i++) if (lane == i) { sh_rank_start[i] = rs[i]; sh_sym_start[i] = ss[i]; }
  uint32_t run[13];
  for (int v = 0; v < 13; v++) run[v] = 0;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64u - lane));
  for (uint32_t q = 0; q < 256u; q += 64u) {
    const uint32_t s = q + lane; const uint32_t w = (s < nsym) ? sh_w[s] : 0u;
    uint32_t at = 0;                                   // (the position is selected class by class, the byte stored once: one LDS store a round, not twelve predicated ones)
    for (uint32_t v = 1; v <= 12; v++) {
      const uint64_t m = __ballot(w == v);
      at = (w == v) ? ss[v] + run[v] + (uint32_t)__popcll(m & lt) : at;
      run[v] += (uint32_t)__popcll(m);
    }
    if (w != 0u) sh_symlist[at] = (uint8_t)s;
  }
  __builtin_amdgcn_wave_barrier();
  ZN_WT(15);   // canonical order
  R.hs = (int)(isz + 1u); R.nsym = nsym; R.tl = tl; R.lmin = tl + 1u - vmax; R.dom = (dom_cells << 8) >> tl;
  return R;
}
