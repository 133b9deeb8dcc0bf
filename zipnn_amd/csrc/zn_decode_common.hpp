// zn_decode_common.hpp — pieces shared by the generic and the fused decode kernels:
// per-(plane, chunk) metadata parse (reference csrc/zipnn_core.c:929-1028) and the canonical
// huff0 decode-table layout (HUF_readDTableX1 of zstd 1.4.8).
#pragma once

#include "zn_common.hpp"

// ---------------------------------------------------------------------------
// metadata of one (plane, chunk), computed identically by every lane
// ---------------------------------------------------------------------------
struct ZnPcMeta { uint64_t off; uint32_t csize; uint32_t plen; uint32_t type; uint32_t ok; };

__device__ inline ZnPcMeta zn_pc_meta(const ZnGeom& g, const uint8_t* body, uint64_t body_len, uint32_t p, uint64_t c) {
  ZnPcMeta m;
  const uint64_t PK = (uint64_t)g.P * g.K;
  const uint8_t* cum = body + PK;                       // u64 [P][K], inclusive, unaligned
  uint64_t base = 9u * PK;                              // payload start
  for (uint32_t q = 0; q < p; q++) base += zn_ld64(cum + 8u * ((uint64_t)q * g.K + g.K - 1));
  const uint64_t hi = zn_ld64(cum + 8u * ((uint64_t)p * g.K + c));
  const uint64_t lo = c ? zn_ld64(cum + 8u * ((uint64_t)p * g.K + c - 1)) : 0;
  m.type = body[(uint64_t)p * g.K + c];
  m.plen = zn_plane_len(zn_chunk_len(g, c), g.P, p);
  m.ok = (hi >= lo) && (hi - lo <= 0xFFFFFFFFull) && (base + hi <= body_len);
  m.csize = (uint32_t)(hi - lo);
  m.off = base + lo;
  return m;
}

// ---------------------------------------------------------------------------
// single-symbol decode LUT, filled by all lanes of the calling wave(s)
// ---------------------------------------------------------------------------
// weights -> symbols ordered by (weight, symbol) + per-weight start cells.  Wave 0 only.
// sh_symlist[256], sh_rank_start[14] (cells), sh_sym_start[14] (index into symlist).
__device__ inline void zn_order_symbols(const uint8_t* weights, uint32_t nsym, uint32_t tl, uint8_t* sh_symlist,
                                        uint32_t* sh_rank_start, uint32_t* sh_sym_start, uint32_t lane) {
  uint32_t cnt[13];
  for (int v = 0; v < 13; v++) cnt[v] = 0;
  // pass 1: per-weight totals
  for (uint32_t q = 0; q < 256; q += ZN_WAVE) {
    const uint32_t s = q + lane; const uint32_t w = (s < nsym) ? weights[s] : 0u;
    for (uint32_t v = 1; v <= 12; v++) cnt[v] += (uint32_t)__popcll(__ballot(w == v));
  }
  uint32_t rs[14], ss[14]; uint32_t cells = 0, syms = 0;
  rs[0] = 0; ss[0] = 0;
  for (uint32_t v = 1; v <= 12; v++) { rs[v] = cells; ss[v] = syms; cells += cnt[v] << (v - 1); syms += cnt[v]; }
  rs[13] = cells; ss[13] = syms;
  if (lane < 14) { sh_rank_start[lane] = rs[lane]; sh_sym_start[lane] = ss[lane]; }
  // pass 2: rank of every symbol inside its weight class
  uint32_t run[13];
  for (int v = 0; v < 13; v++) run[v] = 0;
  for (uint32_t q = 0; q < 256; q += ZN_WAVE) {
    const uint32_t s = q + lane; const uint32_t w = (s < nsym) ? weights[s] : 0u;
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t v = 1; v <= 12; v++) {
      const uint64_t m = __ballot(w == v);
      if (w == v) sh_symlist[ss[v] + run[v] + (uint32_t)__popcll(m & lt)] = (uint8_t)s;
      run[v] += (uint32_t)__popcll(m);
    }
  }
  (void)tl;
}

// cell u -> (symbol | nbBits << 8); cells of weight w (code length tl+1-w) are contiguous
__device__ inline uint32_t zn_lut_entry(uint32_t u, uint32_t tl, const uint8_t* sh_symlist, const uint32_t* sh_rank_start,
                                        const uint32_t* sh_sym_start) {
  uint32_t w = 1;
  for (uint32_t v = 2; v <= 12; v++) w += (u >= sh_rank_start[v]) ? 1u : 0u;   // rank_start is non-decreasing
  const uint32_t j = (u - sh_rank_start[w]) >> (w - 1);
  return (uint32_t)sh_symlist[sh_sym_start[w] + j] | ((tl + 1u - w) << 8);
}

