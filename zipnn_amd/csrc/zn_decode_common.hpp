// zn_decode_common.hpp — pieces shared by the generic and the fused decode kernels:
// per-(plane, chunk) metadata parse (reference csrc/zipnn_core.c:929-1028) and the canonical
// huff0 decode-table layout (HUF_readDTableX1 of zstd 1.4.8).
#pragma once

#include "zn_common.hpp"
#include "zn_internal.hpp"

// ---------------------------------------------------------------------------
// which tensor of a batched launch does grid index `b` belong to?  KEY: 0 = fused workgroup, 1 = (plane,
// chunk) index, 2 = chunk index.  Wave-uniform (scalar loads); the last segment with key ≤ b.
// ---------------------------------------------------------------------------
template <int KEY>
__device__ __forceinline__ uint64_t zn_seg_key(const ZnSeg& s) { return KEY == 0 ? (uint64_t)s.wg0 : KEY == 1 ? s.desc0 : KEY == 2 ? s.chunk0 : (uint64_t)s.tail0; }
template <int KEY>
__device__ __forceinline__ ZnSeg zn_find_seg(const ZnSeg& one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint64_t b) {
  // (a struct VALUE that is overwritten, not a choice between two struct ADDRESSES: returning `one` or `segs[lo]` made the compiler keep the
  //  by-value kernel argument in private memory — every thread of every workgroup wrote its 96 bytes to scratch and read them back: 24 KB per
  //  workgroup of the fused kernel, 96 KB per workgroup of the wide one, +37 % HBM writes on a 64 MiB decode; profiles/r04_decode_experiments.txt)
  ZnSeg r = one;
  if (segs != nullptr) {
    uint32_t lo = 0, hi = nseg;               // invariant: key(lo) ≤ b, key(hi) > b (hi == nseg: past the end)
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (zn_seg_key<KEY>(segs[mid]) <= b) lo = mid; else hi = mid; }
    r = segs[lo];
  }
  return r;
}

// ---------------------------------------------------------------------------
// metadata of one (plane, chunk), computed identically by every lane
// ---------------------------------------------------------------------------
struct ZnPcMeta { uint64_t off; uint32_t csize; uint32_t plen; uint32_t type; uint32_t ok; };

// Everything here comes out of an untrusted frame: every sum is checked against what is left of the body BEFORE it
// is formed, so that no crafted cumSizes entry can wrap the 64-bit arithmetic and move `off` outside [0, body_len)
// (a plane total of 2^64 - 9PK - 4096 used to give the next plane a base 4096 bytes before the body).
__device__ inline ZnPcMeta zn_pc_meta(const ZnGeom& g, const uint8_t* body, uint64_t body_len, uint32_t p, uint64_t c) {
  ZnPcMeta m;
  const uint64_t PK = (uint64_t)g.P * g.K;
  const uint8_t* cum = body + PK;                       // u64 [P][K], inclusive, unaligned
  bool ok = (PK <= body_len / 9u);                      // types + cumSizes fit
  uint64_t base = 9u * PK;                              // payload start
  for (uint32_t q = 0; q < p && ok; q++) {
    const uint64_t t = zn_ld64(cum + 8u * ((uint64_t)q * g.K + g.K - 1));   // total of plane q
    if (t > body_len - base) ok = false; else base += t;
  }
  uint64_t hi = 0, lo = 0;
  if (ok) {
    hi = zn_ld64(cum + 8u * ((uint64_t)p * g.K + c));
    lo = c ? zn_ld64(cum + 8u * ((uint64_t)p * g.K + c - 1)) : 0;
    ok = (hi >= lo) && (hi - lo <= 0xFFFFFFFFull) && (hi <= body_len - base);
  }
  m.type = ok ? body[(uint64_t)p * g.K + c] : 0xFFu;
  m.plen = zn_plane_len(zn_chunk_len(g, c), g.P, p);
  m.ok = ok ? 1u : 0u;
  m.csize = ok ? (uint32_t)(hi - lo) : 0u;
  m.off = ok ? base + lo : 0u;
  return m;
}

// ---------------------------------------------------------------------------
// single-symbol decode LUT, filled by all lanes of the calling wave(s)
// ---------------------------------------------------------------------------
// cell u -> (symbol | nbBits << 8); cells of weight w (code length tl+1-w) are contiguous
struct ZnRankTab { uint32_t rs[14]; };   // wave-uniform copy of rank_start (scalar registers)
__device__ __forceinline__ ZnRankTab zn_load_ranks(const uint32_t* sh_rank_start, const uint32_t* sh_sym_start) {
  ZnRankTab t;
  for (int i = 0; i < 14; i++) t.rs[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_rank_start[i]);
  (void)sh_sym_start;
  return t;
}
__device__ __forceinline__ uint32_t zn_lut_entry(uint32_t u, uint32_t tl, const uint8_t* sh_symlist, const ZnRankTab& t,
                                                 const uint32_t* sh_rank_start, const uint32_t* sh_sym_start) {
  uint32_t w = 1;
  for (int v = 2; v <= 12; v++) w += (u >= t.rs[v]) ? 1u : 0u;   // rank_start is non-decreasing
  const uint32_t j = (u - sh_rank_start[w]) >> (w - 1u);
  return (uint32_t)sh_symlist[sh_sym_start[w] + j] | ((tl + 1u - w) << 8);
}

