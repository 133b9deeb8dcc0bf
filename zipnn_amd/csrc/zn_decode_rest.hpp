// zn_decode_rest.hpp — the generic decode of ONE (plane, chunk) and the merge of ONE chunk, as device functions: the bodies of
// zn_k_decode_planes / zn_k_merge_planes (zn_decode_generic.hip), also called by the `rest` instance of the fused kernel
// (zn_decode_fused.hip), which decodes what it cannot take itself with exactly this code instead of leaving it to two more launches.
// Correctness-first: any dtype, any plane mix, tableLog up to 12, the error bits of a malformed body.
#pragma once

#include "zn_internal.hpp"
#include "zn_huf_wave.hpp"
#include "zn_decode_common.hpp"

// ---------------------------------------------------------------------------
// one backward stream decoded by one lane (byte-granular; the generic path only)
// ---------------------------------------------------------------------------
__device__ inline uint64_t zn_window(const uint8_t* base, int32_t bitpos) {
  // 64-bit container whose top bit is the next unread bit; zero-filled below bit 0
  const int32_t k = bitpos >> 3, r = bitpos & 7;
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) {
    const int32_t idx = k - 7 + i;
    const uint32_t b = (idx >= 0 && (idx < k || r > 0)) ? base[idx] : 0u;
    v |= (uint64_t)b << (8 * i);
  }
  return v << (8 - r);
}

// returns 0 when the stream decodes to exactly `nout` symbols and is fully consumed
// (symbol i goes to out[i * stride])
__device__ inline int zn_decode_stream_serial(const uint8_t* src, uint32_t len, uint8_t* out, uint32_t stride, uint32_t nout,
                                              const uint16_t* lut, uint32_t tl) {
  if (len == 0 || src[len - 1] == 0) return 1;
  int32_t bitpos = (int32_t)(len - 1u) * 8 + (int32_t)zn_hb32(src[len - 1]);
  uint32_t produced = 0;
  while (produced < nout) {
    if (bitpos < -64) return 1;
    uint64_t cont = zn_window(src, bitpos);
    int32_t avail = ((bitpos >> 3) >= 7) ? 56 + (bitpos & 7) : (1 << 30);
    while (produced < nout && avail >= (int32_t)tl) {
      const uint32_t e = lut[(uint32_t)(cont >> (64 - tl))];
      const uint32_t nb = e >> 8;
      out[(uint64_t)produced * stride] = (uint8_t)e; produced++;
      cont <<= nb; bitpos -= (int32_t)nb; avail -= (int32_t)nb;
    }
  }
  return bitpos != 0;
}

// ---------------------------------------------------------------------------
// classify + decode the huff0 planes straight into their (strided) byte positions of the output
// ---------------------------------------------------------------------------
struct ZnPlanesLds {
  uint16_t lut[1u << ZN_HUF_LOG_MAX];
  uint8_t sh_w[256], sh_symlist[256], sh_cell[64];
  uint32_t sh_rank_start[14], sh_sym_start[14];
};

// one (plane, chunk) = launch-wide desc index b, by ONE WAVE (lane = its lane index; the only synchronisation inside is wave-level, so that
// several waves of a workgroup may each run an item of their own on an LDS instance of their own)
__device__ inline void zn_decode_plane_item(ZnPlanesLds& L, const ZnSeg& one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint64_t b,
                                            ZnPlaneDesc* __restrict__ descs_all, uint32_t* __restrict__ status,
                                            const uint8_t* __restrict__ tail_done, uint32_t lane, bool classify_only = false, uint32_t* wants_serial = nullptr) {
  uint16_t* lut = L.lut; uint8_t* sh_w = L.sh_w; uint8_t* sh_symlist = L.sh_symlist; uint8_t* sh_cell = L.sh_cell;
  uint32_t* sh_rank_start = L.sh_rank_start; uint32_t* sh_sym_start = L.sh_sym_start;
  const ZnSeg S = zn_find_seg<1>(one, segs, nseg, b);
  const ZnGeom g = S.g;
  const uint8_t* __restrict__ body = ZN_GLOBAL_PTR(const uint8_t, S.body); const uint64_t body_len = S.body_len;
  uint8_t* __restrict__ dst = ZN_GLOBAL_PTR(uint8_t, S.dst);
  ZnPlaneDesc* __restrict__ descs = descs_all + S.desc0;

  const uint64_t pc = b - S.desc0;
  const uint32_t p = (uint32_t)(pc / g.K);
  const uint64_t c = pc % g.K;
  const ZnPcMeta m = zn_pc_meta(g, body, body_len, p, c);
  ZnPlaneDesc d; d.off = 0; d.kind = ZN_KIND_RAW; d.len = m.plen;

  uint32_t bad = 0;
  if (!m.ok) bad = ZN_DEV_CORRUPT;
  else if (m.type > 1u) bad = ZN_DEV_BAD_TYPE;
  else if (m.type == 0u) { if (m.csize < m.plen) bad = ZN_DEV_CORRUPT; d.off = m.off; }
  else {  // HUF_decompress conventions
    if (m.plen == 0 || m.csize > m.plen || m.csize == 0) bad = ZN_DEV_CORRUPT;
    else if (m.csize == m.plen) { d.off = m.off; }
    else if (m.csize == 1u) { d.kind = ZN_KIND_RLE; d.off = body[m.off]; }
    else { d.kind = ZN_KIND_HUF; d.off = 0; }
  }
  if (bad) { d.kind = ZN_KIND_RLE; d.off = 0; }   // keep the merge kernel in bounds; output is discarded by the caller
  // (four tail workgroups per plane, one per huff0 stream, each with a flag byte of its own: the plane is in the scratch when all four are set)
  if (!bad && d.kind == ZN_KIND_HUF && S.has_tail && c == g.K - 1u && tail_done &&
      (tail_done[4u * (S.tail0 + p)] & tail_done[4u * (S.tail0 + p) + 1u] & tail_done[4u * (S.tail0 + p) + 2u] & tail_done[4u * (S.tail0 + p) + 3u]) != 0) {
    d.kind = ZN_KIND_HUFS; d.off = (uint64_t)(S.tail0 + p) * ZN_TAIL_SLOT;       // already decoded by the tail workgroups of zn_k_decode_fused
  }

  // classify_only (the merge workgroups of a partial last chunk, zn_k_decode_fused: every one of them classifies for itself): a huff0 block that is still to be
  // decoded serially is reported instead (*wants_serial, LDS: one of the workgroups then does the whole chunk) and nothing is written
  if (classify_only && !bad && d.kind == ZN_KIND_HUF) { if (lane == 0) atomicOr(wants_serial, 1u); return; }
  if (!bad && d.kind == ZN_KIND_HUF) {
    const uint8_t* src = body + m.off;
    const ZnWaveStats st = zn_wave_read_stats(src, m.csize, body + body_len, lane, sh_w, sh_symlist, sh_rank_start, sh_sym_start, sh_cell);
    const int hs = st.hs; const uint32_t tl = st.tl;
    if (hs < 0 || (uint32_t)hs >= m.csize || m.csize - (uint32_t)hs < 10u) bad = ZN_DEV_CORRUPT;
    else {
      {
        const ZnRankTab rt = zn_load_ranks(sh_rank_start, sh_sym_start);
        for (uint32_t u = lane; u < (1u << tl); u += ZN_WAVE) lut[u] = (uint16_t)zn_lut_entry(u, tl, sh_symlist, rt, sh_rank_start, sh_sym_start);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (the LUT is read by other lanes of this wave: LDS runs a wave's operations in order)
      __builtin_amdgcn_wave_barrier();
      if (!bad) {
        const uint8_t* js = src + hs; const uint32_t rem = m.csize - (uint32_t)hs;
        const uint32_t l1 = zn_ld16(js), l2 = zn_ld16(js + 2), l3 = zn_ld16(js + 4);
        const uint32_t seg = (m.plen + 3u) / 4u;
        if (l1 + l2 + l3 + 6u > rem || 3u * seg > m.plen) bad = ZN_DEV_CORRUPT;
        else if (lane < 4) {
          const uint32_t lens[4] = {l1, l2, l3, rem - 6u - l1 - l2 - l3};
          uint32_t so = 6; for (uint32_t k = 0; k < lane; k++) so += lens[k];
          const uint32_t nout = (lane < 3) ? seg : m.plen - 3u * seg;
          // plane byte i of this chunk is output byte i * P + p
          uint8_t* o = dst + c * g.chunk + ((uint64_t)lane * seg) * g.P + p;
          if (zn_decode_stream_serial(js + so, lens[lane], o, g.P, nout, lut, tl))
            bad = ZN_DEV_CORRUPT;
        }
      }
    }
  }
  if (bad) atomicOr(status, bad);
  if (lane == 0) descs[pc] = d;
}

// ---------------------------------------------------------------------------
// merge the planes of one chunk into the output
// ---------------------------------------------------------------------------
// byte j of the chunk = byte j / P of plane j % P; a Huffman-decoded plane already sits in `out`
__device__ __forceinline__ uint32_t zn_plane_byte(const ZnPlaneDesc& d, const uint8_t* body, const uint8_t* out, const uint8_t* tails,
                                                  uint32_t j, uint32_t i) {
  if (d.kind == ZN_KIND_RLE) return (uint32_t)d.off & 0xFFu;
  if (d.kind == ZN_KIND_HUF) return out[j];
  if (d.kind == ZN_KIND_HUFS) { const uint32_t seg3 = (d.len + 3u) / 4u, w = i / seg3; return tails[d.off + (uint64_t)w * ZN_TAIL_SEGPAD + (i - w * seg3)]; }
  return body[d.off + i];
}

template <int P>
#define ZN_MERGE_SUB 64u      // a chunk is merged by 64 workgroup-items (one not-done chunk = a partial tail: 250 µs by one workgroup, 45 µs by 16, four byte-gathering iterations per thread by 64)
__device__ __forceinline__ void zn_merge_chunk_item(const ZnSeg& one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint64_t b, uint32_t sub,
                                                    const ZnPlaneDesc* __restrict__ descs_all, const uint8_t* __restrict__ tails, uint32_t nsub = ZN_MERGE_SUB) {
  const ZnSeg S = zn_find_seg<2>(one, segs, nseg, b);
  const ZnGeom g = S.g;
  const uint8_t* __restrict__ body = ZN_GLOBAL_PTR(const uint8_t, S.body);
  uint8_t* dst = ZN_GLOBAL_PTR(uint8_t, S.dst);
  const ZnPlaneDesc* __restrict__ descs = descs_all + S.desc0;
  const uint64_t c = b - S.chunk0;
  const uint32_t clen = zn_chunk_len(g, c);
  uint8_t* out = dst + c * g.chunk;
  const uint8_t* xo = S.xr ? ZN_GLOBAL_PTR(const uint8_t, S.xr) + c * g.chunk : nullptr;   // delta base of this chunk
  ZnPlaneDesc d[P];
  for (int p = 0; p < P; p++) d[p] = descs[(uint64_t)p * g.K + c];
  const uint32_t nwords = clen / 4u;
  // whole 32-bit words: gather P-way, undo the rotate (applies to clen/4 words — all of them)
  const uint32_t w_lo = (uint32_t)(((uint64_t)nwords * sub) / nsub), w_hi = (uint32_t)(((uint64_t)nwords * (sub + 1u)) / nsub);      // (sub-range `sub` of `nsub`)
  for (uint32_t wi = w_lo + threadIdx.x; wi < w_hi; wi += blockDim.x) {
    uint32_t w = 0;
    for (uint32_t t = 0; t < 4; t++) {
      const uint32_t j = 4u * wi + t;
      w |= zn_plane_byte(d[j % P], body, out, tails, j, j / P) << (8 * t);
    }
    if (g.rot) w = (P == 2) ? zn_rot_inv16(w) : zn_rot_inv32(w);
    if (xo) for (uint32_t t = 0; t < 4; t++) w ^= (uint32_t)xo[4ull * wi + t] << (8 * t);
    const uint64_t a = (uint64_t)(out + 4ull * wi);
    if ((a & 3u) == 0) *(uint32_t*)(out + 4ull * wi) = w;
    else for (uint32_t t = 0; t < 4; t++) out[4ull * wi + t] = (uint8_t)(w >> (8 * t));
  }
  // trailing clen % 4 bytes are never rotated (reference rotates len/4 words only)
  if (sub == nsub - 1u && threadIdx.x < (clen & 3u)) {
    const uint32_t j = 4u * nwords + threadIdx.x;
    out[j] = (uint8_t)(zn_plane_byte(d[j % P], body, out, tails, j, j / P) ^ (xo ? (uint32_t)xo[j] : 0u));
  }
}

