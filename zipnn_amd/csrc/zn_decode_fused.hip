// zn_decode_fused.hip — the bandwidth path of decompress: one kernel, one pass over HBM.
//
// One workgroup (4 waves) per full chunk whose planes are raw/RLE plus at most one huff0
// block (what real weights look like: bf16/fp32 exponent plane Huffman-coded, mantissa
// planes stored raw).  Wave w owns quarter w of the chunk = stream w of the huff0 block.
//
//   1. threads < P parse the chunk's metadata; the workgroup builds the decode table in LDS:
//      tree description (FSE-coded weights) → canonical single-symbol LUT → MULTI-symbol LUT
//      (one 64-bit entry per 11-bit window: up to 4 symbols, total length, first length).
//   2. each wave decodes its backward bit-stream IN PARALLEL ACROSS ITS 64 LANES.  huff0 has no
//      gap array, so this uses Huffman self-synchronisation, format-transparently: the stream
//      is cut into tiles of 64 sub-blocks of D dwords; lane k guesses a start a few dozen bits
//      above its sub-block, decodes until it crosses into it ("sync"), then decodes its
//      sub-block counting symbols; a wave shuffle checks that every lane's exit position is the
//      next lane's start (mismatching lanes restart from the exact position until the chain is
//      consistent — the top lane always starts from the true position carried from the previous
//      tile); a prefix sum of the counts gives each lane its output offset and a second decode
//      writes the symbols with LDS atomic-OR into a small ring.
//   3. as soon as the ring holds a row of 64 × (16/P) symbols the wave flushes it: ring bytes +
//      the raw planes' bytes (unaligned vector loads straight from the body) are byte-interleaved
//      with v_perm_b32, the sign-bit rotate is undone and 16 bytes per lane go out in one
//      coalesced store.  Decoded symbols never touch HBM; the float stream is written once.
//
// Algorithmic HBM traffic per chunk: stored bytes in + chunk bytes out.
// Chunks this kernel does not take (partial tail, ≥2 Huffman planes, tableLog 12, odd chunk
// sizes) are left to zn_decode_generic.hip via the per-chunk `done` flag.
//
// Replaces, for those chunks: decompression_chunk_worker (reference csrc/zipnn_core.c:768-861),
// HUF_decompress (:807), combine_buffers_dtype16/32 + revert_all_floats_* (data_manipulation_
// dtype16.c:145-216, data_manipulation_dtype32.c:275-294,391-456).
#include "zn_internal.hpp"
#include "zn_huf_tables.hpp"
#include "zn_decode_common.hpp"

#define ZN_F_THREADS 256
#define ZN_F_RING_BYTES 8192u
#define ZN_F_RING_DW (ZN_F_RING_BYTES / 4u)
#define ZN_F_DMAX 8
#define ZN_F_IN_DW (64 * ZN_F_DMAX + 4)
#define ZN_F_TLMAX 11u
#define ZN_F_DELTA 96

typedef uint64_t __attribute__((aligned(1))) zn_u64u;
typedef uint32_t __attribute__((aligned(1))) zn_u32u;

struct ZnFusedPlane { uint64_t off; uint32_t kind; uint32_t csize; };   // off: body offset (RAW/HUF) or byte value (RLE)

struct ZnFusedLds {
  uint64_t lut[1u << ZN_F_TLMAX];          // multi-symbol decode table
  uint32_t ring[4][ZN_F_RING_DW];          // per-wave output ring; ring[0] holds the 16-bit LUT while tables are built
  uint32_t in[4][ZN_F_IN_DW];              // per-wave staged stream tile
  ZnTabScratch S;
  uint8_t symlist[256];
  uint32_t rank_start[14], sym_start[14];
  ZnFusedPlane plane[4];
  int hs; uint32_t nsym, tl, fail;
};

// wave-wide exclusive prefix sum (all 64 lanes participate)
__device__ __forceinline__ uint32_t zn_wave_excl_scan(uint32_t v, uint32_t lane, uint32_t* total) {
  uint32_t x = v;
  for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
  *total = __shfl(x, 63);
  return x - v;
}

// Decode the symbols whose codes start at bit positions in (stop, pos] of the staged tile.
// MODE 0: just advance (sync run-in); 1: count symbols; 2: write symbols into the ring at symbol
// index `wbyte` (LDS atomic OR: neighbouring lanes share boundary dwords).
// base_bit = absolute bit position of bit 0 of in[0].
template <int MODE>
__device__ __forceinline__ int32_t zn_fused_run(const uint64_t* lut, const uint32_t* in, int32_t base_bit, uint32_t TL,
                                                int32_t pos, int32_t stop, uint32_t* count, uint32_t* ring, uint32_t wbyte) {
  uint64_t win = 0; int32_t avail = 0;
  uint64_t acc = 0; uint32_t fill = wbyte & 3u; uint32_t dw = (wbyte >> 2) & (ZN_F_RING_DW - 1u);
  uint32_t n = 0;
  while (pos > stop) {
    if (avail < (int32_t)TL) {
      const int32_t q = pos - 1 - base_bit;                 // ≥ 32 here: in[0] is one dword below the tile
      const int32_t j = q >> 5, r = q & 31;
      win = ((((uint64_t)in[j]) << 32) | in[j - 1]) << (31 - r);
      avail = 33 + r;
    }
    const uint64_t e = lut[(uint32_t)(win >> (64u - TL))];
    const uint32_t hi = (uint32_t)(e >> 32);
    uint32_t nb, cnt, syms;
    if (pos - stop >= (int32_t)TL) { nb = (hi >> 4) & 15u; cnt = hi & 7u; syms = (uint32_t)e; }   // every symbol of the group starts above `stop`
    else { nb = (hi >> 8) & 15u; cnt = 1u; syms = (uint32_t)e & 0xFFu; }                           // near the boundary: one symbol at a time
    win <<= nb; avail -= (int32_t)nb; pos -= (int32_t)nb;
    if (MODE == 1) n += cnt;
    if (MODE == 2) {
      acc |= (uint64_t)syms << (8u * fill); fill += cnt;
      if (fill >= 4u) { atomicOr(&ring[dw], (uint32_t)acc); acc >>= 32; fill -= 4u; dw = (dw + 1u) & (ZN_F_RING_DW - 1u); }
    }
  }
  if (MODE == 2 && fill > 0u) atomicOr(&ring[dw], (uint32_t)acc);
  if (MODE == 1) *count = n;
  return pos;
}

// EPL = 16/P bytes of plane p for this lane: from the ring (Huffman plane), a splat (RLE) or the body (raw).
template <int EPL>
__device__ __forceinline__ void zn_fused_plane_bytes(uint32_t* v, const ZnFusedPlane& pl, bool is_huf, uint32_t* ring,
                                                     const uint8_t* raw_q, uint32_t sym_index) {
  if (is_huf) {
    const uint32_t i = (sym_index >> 2) & (ZN_F_RING_DW - 1u);
    for (int k = 0; k < EPL / 4; k++) { v[k] = ring[i + k]; ring[i + k] = 0; }
  } else if (pl.kind == ZN_KIND_RLE) {
    for (int k = 0; k < EPL / 4; k++) v[k] = ((uint32_t)pl.off & 0xFFu) * 0x01010101u;
  } else {
    const uint8_t* a = raw_q + sym_index;
    if (EPL == 4) v[0] = *(const zn_u32u*)a;
    else for (int k = 0; k < EPL / 8; k++) { const uint64_t t = *(const zn_u64u*)(a + 8 * k); v[2 * k] = (uint32_t)t; v[2 * k + 1] = (uint32_t)(t >> 32); }
  }
}

template <int P>
__global__ __launch_bounds__(ZN_F_THREADS) void zn_k_decode_fused(ZnGeom g, const uint8_t* __restrict__ body, uint64_t body_len,
                                                                  uint8_t* __restrict__ dst, uint8_t* __restrict__ done,
                                                                  uint32_t* __restrict__ status) {
  constexpr int EPL = 16 / P;                 // bytes per plane per lane in one flushed row
  constexpr uint32_t UNIT = 64u * EPL;        // symbols per flushed row
  __shared__ ZnFusedLds L;

  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t c = blockIdx.x;
  const uint32_t clen = zn_chunk_len(g, c);
  const uint32_t plen = (uint32_t)(g.chunk / P);

  // ---- metadata: one thread per plane ----
  if (tid < (uint32_t)P) {
    const ZnPcMeta m = zn_pc_meta(g, body, body_len, tid, c);
    ZnFusedPlane pl; pl.off = m.off; pl.csize = m.csize; pl.kind = 99u;   // 99 = not for this kernel
    if (m.ok && m.type <= 1u && clen == g.chunk) {
      if (m.type == 0u) { if (m.csize >= plen) pl.kind = ZN_KIND_RAW; }
      else if (m.csize == plen) pl.kind = ZN_KIND_RAW;
      else if (m.csize == 1u) { pl.kind = ZN_KIND_RLE; pl.off = body[m.off]; }
      else if (m.csize > 1u && m.csize < plen) pl.kind = ZN_KIND_HUF;
    }
    L.plane[tid] = pl;
  }
  if (tid == 0) L.fail = 0;
  __syncthreads();

  int h = -1; uint32_t nhuf = 0; bool elig = (g.chunk % 4096u) == 0 && ((((uint64_t)dst) & 15u) == 0);
  for (int p = 0; p < P; p++) {
    const uint32_t k = L.plane[p].kind;
    if (k == 99u) elig = false;
    if (k == ZN_KIND_HUF) { h = p; nhuf++; }
  }
  if (!elig || nhuf > 1u) { if (tid == 0) done[c] = 0; return; }

  const uint32_t seg = plen / 4u;             // symbols per stream == plane bytes per quarter
  uint32_t TL = 0, D = ZN_F_DMAX;
  const uint8_t* stream = nullptr; uint32_t slen = 0;

  if (h >= 0) {
    // ---- decode table ----
    const uint8_t* src = body + L.plane[h].off; const uint32_t csize = L.plane[h].csize;
    for (uint32_t i = tid; i < 160u; i += ZN_F_THREADS) L.S.hdr[i] = (i < csize) ? src[i] : 0;
    __syncthreads();
    if (tid == 0) {
      uint32_t nsym = 0, tl = 0;
      L.hs = zn_read_stats(&L.S, L.S.hdr, csize < 160u ? csize : 160u, &nsym, &tl);
      L.nsym = nsym; L.tl = tl;
    }
    __syncthreads();
    const int hs = L.hs; TL = L.tl;
    if (hs < 0 || TL > ZN_F_TLMAX || (uint32_t)hs >= csize || csize - (uint32_t)hs < 10u) { if (tid == 0) done[c] = 0; return; }
    if (wave == 0) zn_order_symbols(L.S.weights, L.nsym, TL, L.symlist, L.rank_start, L.sym_start, lane);
    __syncthreads();
    if (L.rank_start[13] != (1u << TL)) { if (tid == 0) done[c] = 0; return; }
    uint16_t* lut16 = (uint16_t*)&L.ring[0][0];
    for (uint32_t u = tid; u < (1u << TL); u += ZN_F_THREADS)
      lut16[u] = (uint16_t)zn_lut_entry(u, TL, L.symlist, L.rank_start, L.sym_start);
    __syncthreads();
    {
      const uint32_t mask = (1u << TL) - 1u;
      for (uint32_t u = tid; u < (1u << TL); u += ZN_F_THREADS) {
        uint32_t pos = 0, cnt = 0, syms = 0, len0 = 0;
        while (cnt < 4u) {
          const uint32_t e = lut16[(u << pos) & mask]; const uint32_t len = e >> 8;
          if (pos + len > TL) break;        // the window does not hold this code completely
          if (cnt == 0) len0 = len;
          syms |= (e & 0xFFu) << (8u * cnt); pos += len; cnt++;
        }
        L.lut[u] = (uint64_t)syms | ((uint64_t)(cnt | (pos << 4) | (len0 << 8)) << 32);
      }
    }
    // shortest code length → how many symbols a tile can hold → sub-block size D (dwords)
    uint32_t vmax = 1;
    for (uint32_t v = 1; v <= 12; v++) if (L.rank_start[v + 1] > L.rank_start[v]) vmax = v;
    const uint32_t lmin = TL + 1u - vmax;
    D = ((ZN_F_RING_BYTES - UNIT) * lmin) / 2048u;
    if (D > ZN_F_DMAX) D = ZN_F_DMAX;
    if (D < 1u) D = 1u;
    // jump table → this wave's stream
    const uint8_t* js = src + hs; const uint32_t rem = csize - (uint32_t)hs;
    const uint32_t l1 = zn_ld16(js), l2 = zn_ld16(js + 2), l3 = zn_ld16(js + 4);
    if (l1 + l2 + l3 + 6u > rem) { if (tid == 0) done[c] = 0; return; }
    const uint32_t l4 = rem - 6u - l1 - l2 - l3;
    if (l1 == 0 || l2 == 0 || l3 == 0 || l4 == 0) { if (tid == 0) done[c] = 0; return; }
    const uint32_t so = 6u + (wave > 0 ? l1 : 0u) + (wave > 1 ? l2 : 0u) + (wave > 2 ? l3 : 0u);
    stream = js + so; slen = (wave == 0) ? l1 : (wave == 1) ? l2 : (wave == 2) ? l3 : l4;
    __syncthreads();                         // lut16 (aliasing ring[0]) is dead from here on
  }

  // ---- per-wave: zero the ring, decode the stream tile by tile, flush rows ----
  uint32_t* ring = L.ring[wave];
  uint32_t* in = L.in[wave];
  for (uint32_t i = lane; i < ZN_F_RING_DW; i += 64u) ring[i] = 0;
  __builtin_amdgcn_wave_barrier();

  const uint8_t* rawq[P]; ZnFusedPlane pl[P];
  for (int p = 0; p < P; p++) { pl[p] = L.plane[p]; rawq[p] = body + pl[p].off + (uint64_t)wave * seg; }
  uint8_t* outq = dst + c * g.chunk + (uint64_t)wave * (g.chunk / 4u);
  const uint8_t* body_end = body + body_len;

  uint32_t J = (h >= 0) ? 0u : seg;           // symbols decoded into the ring so far
  uint32_t JF = 0;                            // symbols flushed to HBM so far
  bool ok = true;
  int32_t b0 = 0, carry = 0, hi_dw = 0; const uint32_t* gdw = nullptr;
  if (h >= 0) {
    const uint8_t last = stream[slen - 1];
    if (last == 0) ok = false;
    const uint64_t a = (uint64_t)stream;
    gdw = (const uint32_t*)(a & ~(uint64_t)3); b0 = (int32_t)(8u * (uint32_t)(a & 3u));
    carry = b0 + (int32_t)(8u * (slen - 1u)) + (int32_t)zn_hb32(last ? last : 1u);
    hi_dw = (carry + 31) >> 5;
  }
  const int32_t Di = (int32_t)D;
  const int32_t delta = (ZN_F_DELTA < 32 * Di) ? ZN_F_DELTA : 32 * Di;

  for (;;) {
    // flush every complete row the ring holds
    while (J - JF >= UNIT) {
      uint32_t v[P][EPL / 4 > 0 ? EPL / 4 : 1];
      const uint32_t si = JF + (uint32_t)EPL * lane;
      for (int p = 0; p < P; p++) zn_fused_plane_bytes<EPL>(v[p], pl[p], p == h, ring, rawq[p], si);
      uint32_t o[4];
      if (P == 1) { o[0] = v[0][0]; o[1] = v[0][1 % (EPL / 4)]; o[2] = v[0][2 % (EPL / 4)]; o[3] = v[0][3 % (EPL / 4)]; }
      else if (P == 2) {
        o[0] = __builtin_amdgcn_perm(v[1 % P][0], v[0][0], 0x05010400u); o[1] = __builtin_amdgcn_perm(v[1 % P][0], v[0][0], 0x07030602u);
        o[2] = __builtin_amdgcn_perm(v[1 % P][1 % (EPL / 4)], v[0][1 % (EPL / 4)], 0x05010400u);
        o[3] = __builtin_amdgcn_perm(v[1 % P][1 % (EPL / 4)], v[0][1 % (EPL / 4)], 0x07030602u);
        if (g.rot) for (int k = 0; k < 4; k++) o[k] = zn_rot_inv16(o[k]);
      } else {
        const uint32_t ab_lo = __builtin_amdgcn_perm(v[1 % P][0], v[0][0], 0x05010400u), ab_hi = __builtin_amdgcn_perm(v[1 % P][0], v[0][0], 0x07030602u);
        const uint32_t cd_lo = __builtin_amdgcn_perm(v[3 % P][0], v[2 % P][0], 0x05010400u), cd_hi = __builtin_amdgcn_perm(v[3 % P][0], v[2 % P][0], 0x07030602u);
        o[0] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u); o[1] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
        o[2] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u); o[3] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
        if (g.rot) for (int k = 0; k < 4; k++) o[k] = zn_rot_inv32(o[k]);
      }
      *(uint4*)(outq + (uint64_t)JF * P + 16u * lane) = make_uint4(o[0], o[1], o[2], o[3]);
      JF += UNIT;
    }
    if (h < 0 || !ok || 32 * hi_dw <= b0) break;

    // ---- next tile: dwords [lo_dw, hi_dw) of the stream, plus one below for look-ahead ----
    const int32_t lo_dw = hi_dw - 64 * Di;
    __builtin_amdgcn_wave_barrier();
    for (int32_t i = (int32_t)lane; i <= 64 * Di; i += 64) {
      const int32_t gi = lo_dw - 1 + i;
      uint32_t x = 0;
      if (gi >= -1 && gi < hi_dw) {
        const uint8_t* pa = (const uint8_t*)(gdw + gi);
        if (pa + 4 <= body_end) x = gdw[gi];
        else for (int b = 0; b < 4; b++) if (pa + b < body_end) x |= (uint32_t)pa[b] << (8 * b);
      }
      in[i] = x;
    }
    __builtin_amdgcn_wave_barrier();
    const int32_t base_bit = 32 * (lo_dw - 1);
    const int32_t hi_k = 32 * (hi_dw - (int32_t)lane * Di), lo_k = hi_k - 32 * Di;
    const int32_t lo_eff = lo_k > b0 ? lo_k : b0;
    const bool active = hi_k > b0;

    // sync: lanes > 0 guess a start `delta` bits above their sub-block and run into it
    int32_t s = carry;
    if (lane > 0 && active) s = zn_fused_run<0>(L.lut, in, base_bit, TL, hi_k + delta, hi_k, nullptr, nullptr, 0);

    // count, and verify that the lanes form one consistent chain below the true start of lane 0
    uint32_t n = 0; int32_t e = s; bool need = active, chained = false;
    for (int it = 0; it < 66; it++) {
      if (need) e = zn_fused_run<1>(L.lut, in, base_bit, TL, s, lo_eff, &n, nullptr, 0);
      const int32_t e_prev = __shfl_up(e, 1u);
      const bool mism = active && lane > 0 && e_prev != s;
      if (!__any(mism)) { chained = true; break; }
      need = mism; if (mism) s = e_prev;
    }
    if (!active) n = 0;
    uint32_t N = 0;
    const uint32_t o_k = zn_wave_excl_scan(n, lane, &N);
    const uint32_t nact = (uint32_t)__popcll(__ballot(active));
    const int32_t e_last = __shfl(e, (int)(nact ? nact - 1u : 0u));
    if (!chained || J + N > seg || J + N - JF > ZN_F_RING_BYTES) { ok = false; break; }

    // write: second decode of the same sub-block, symbols OR-ed into the ring at their final index
    if (active) zn_fused_run<2>(L.lut, in, base_bit, TL, s, lo_eff, nullptr, ring, J + o_k);
    __builtin_amdgcn_wave_barrier();
    J += N; carry = e_last; hi_dw = lo_dw;
  }

  if (h >= 0 && (!ok || carry != b0 || J != seg || JF != seg)) atomicOr(status, ZN_DEV_CORRUPT);
  if (tid == 0) done[c] = 1;
}

void zn_launch_decode_fused(const ZnGeom& g, const uint8_t* d_body, uint64_t body_len, uint8_t* d_dst, uint8_t* d_done,
                            uint32_t* d_status, hipStream_t stream) {
  if (g.K == 0) return;
  if (g.P == 1) hipLaunchKernelGGL(zn_k_decode_fused<1>, dim3((uint32_t)g.K), dim3(ZN_F_THREADS), 0, stream, g, d_body, body_len, d_dst, d_done, d_status);
  else if (g.P == 2) hipLaunchKernelGGL(zn_k_decode_fused<2>, dim3((uint32_t)g.K), dim3(ZN_F_THREADS), 0, stream, g, d_body, body_len, d_dst, d_done, d_status);
  else hipLaunchKernelGGL(zn_k_decode_fused<4>, dim3((uint32_t)g.K), dim3(ZN_F_THREADS), 0, stream, g, d_body, body_len, d_dst, d_done, d_status);
  zn_note_kernel("zn_k_decode_fused");
}
