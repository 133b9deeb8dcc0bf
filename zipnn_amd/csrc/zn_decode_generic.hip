// zn_decode_generic.hip — generic decode path: handles every dtype (1/2/4 planes), partial
// last chunks, RLE planes, and any mix of raw/Huffman planes.  Two kernels:
//
//   zn_k_decode_planes   one wave per (plane, chunk): parse its metadata, classify it
//                        (raw / RLE / huff0) and, for huff0 blocks, build the decode LUT in
//                        LDS and decode the four backward streams straight into the plane's
//                        (strided) byte positions of the output — no scratch memory.
//   zn_k_merge_planes    one workgroup per chunk: fill in the raw / RLE planes from the body,
//                        undo the sign-bit rotate (in place, word by word), XOR with the delta base if there is one.
//
// Replaces: decompression_chunk_worker (reference csrc/zipnn_core.c:768-861), the metadata
// parse of py_combine_dtype (:929-1028), HUF_decompress (call site :807) and
// combine_buffers_dtype16/32 + revert_all_floats_* (data_manipulation_dtype16.c:145-216,
// data_manipulation_dtype32.c:275-294,391-456).
//
// This is the correctness-first path; the bandwidth path for full chunks with one
// Huffman-coded plane is zn_decode_fused.hip.
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
// kernel 1: classify + decode huff0 planes into scratch
// ---------------------------------------------------------------------------
struct ZnPlanesLds {
  uint16_t lut[1u << ZN_HUF_LOG_MAX];
  uint8_t sh_w[256], sh_symlist[256], sh_cell[64];
  uint32_t sh_rank_start[14], sh_sym_start[14];
};

// one (plane, chunk) = launch-wide desc index b, by one wave
__device__ void zn_decode_plane_item(ZnPlanesLds& L, const ZnSeg& one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint64_t b,
                                     ZnPlaneDesc* __restrict__ descs_all, uint32_t* __restrict__ status,
                                     const uint8_t* __restrict__ tail_done) {
  uint16_t* lut = L.lut; uint8_t* sh_w = L.sh_w; uint8_t* sh_symlist = L.sh_symlist; uint8_t* sh_cell = L.sh_cell;
  uint32_t* sh_rank_start = L.sh_rank_start; uint32_t* sh_sym_start = L.sh_sym_start;
  const ZnSeg S = zn_find_seg<1>(one, segs, nseg, b);
  const ZnGeom g = S.g;
  const uint8_t* __restrict__ body = ZN_GLOBAL_PTR(const uint8_t, S.body); const uint64_t body_len = S.body_len;
  uint8_t* __restrict__ dst = ZN_GLOBAL_PTR(uint8_t, S.dst);
  ZnPlaneDesc* __restrict__ descs = descs_all + S.desc0;

  const uint32_t lane = threadIdx.x;
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
  if (!bad && d.kind == ZN_KIND_HUF && S.has_tail && c == g.K - 1u && tail_done && tail_done[S.tail0 + p]) {
    d.kind = ZN_KIND_HUFS; d.off = (uint64_t)(S.tail0 + p) * ZN_TAIL_SLOT;       // already decoded by the tail workgroups of zn_k_decode_fused
  }

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
      __syncthreads();
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

// Grid-stride over the launch's (plane, chunk) entries; pdone[b] != 0: the fused kernel already wrote that chunk
// (checked before anything else, so a launch where everything is done costs a few microseconds).
__global__ __launch_bounds__(ZN_WAVE) void zn_k_decode_planes(ZnSeg one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint64_t total,
                                                              ZnPlaneDesc* __restrict__ descs_all, uint32_t* __restrict__ status,
                                                              const uint8_t* __restrict__ pdone, const uint8_t* __restrict__ tail_done,
                                                              const uint32_t* __restrict__ left) {
  if (left && *left == 0u) return;             // the fused kernel took every chunk of the launch
  __shared__ ZnPlanesLds L;
  __shared__ uint32_t n_todo; __shared__ uint16_t todo[ZN_WAVE];
  // this wave's items are b = blockIdx.x + i * gridDim.x; their flags are read 64 at a time (one latency)
  const uint64_t nit = (total > blockIdx.x) ? (total - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0;
  for (uint64_t base = 0; base < nit; base += ZN_WAVE) {
    if (threadIdx.x == 0) n_todo = 0;
    __syncthreads();
    const uint64_t i = base + threadIdx.x, b = blockIdx.x + i * gridDim.x;
    if (i < nit && !(pdone && pdone[b])) todo[atomicAdd(&n_todo, 1u)] = (uint16_t)threadIdx.x;
    __syncthreads();
    const uint32_t n = n_todo;
    for (uint32_t k = 0; k < n; k++) {
      const uint64_t bb = blockIdx.x + (base + todo[k]) * gridDim.x;
      __syncthreads();                         // the tables of the previous item are no longer in use
      zn_decode_plane_item(L, one, segs, nseg, bb, descs_all, status, tail_done);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// kernel 2: merge planes of one chunk into the output
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
                                                    const ZnPlaneDesc* __restrict__ descs_all, const uint8_t* __restrict__ tails) {
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
  const uint32_t w_lo = (uint32_t)(((uint64_t)nwords * sub) / ZN_MERGE_SUB), w_hi = (uint32_t)(((uint64_t)nwords * (sub + 1u)) / ZN_MERGE_SUB);
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
  if (sub == ZN_MERGE_SUB - 1u && threadIdx.x < (clen & 3u)) {
    const uint32_t j = 4u * nwords + threadIdx.x;
    out[j] = (uint8_t)(zn_plane_byte(d[j % P], body, out, tails, j, j / P) ^ (xo ? (uint32_t)xo[j] : 0u));
  }
}

template <int P>
__global__ __launch_bounds__(256) void zn_k_merge_planes(ZnSeg one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint64_t total,
                                                         const ZnPlaneDesc* __restrict__ descs_all, const uint8_t* __restrict__ done_all,
                                                         const uint8_t* __restrict__ tails, const uint32_t* __restrict__ left) {
  if (left && *left == 0u) return;             // the fused kernel took every chunk of the launch
  __shared__ uint32_t n_todo; __shared__ uint16_t todo[256];
  // items = (chunk, sixteenth); this workgroup's items are it = blockIdx.x + i * gridDim.x; the chunks' flags are
  // read 256 at a time (one latency)
  const uint64_t items = total * ZN_MERGE_SUB;
  const uint64_t nit = (items > blockIdx.x) ? (items - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0;
  for (uint64_t base = 0; base < nit; base += 256u) {
    if (threadIdx.x == 0) n_todo = 0;
    __syncthreads();
    const uint64_t i = base + threadIdx.x, it = blockIdx.x + i * gridDim.x;
    if (i < nit && !(done_all && done_all[it / ZN_MERGE_SUB])) todo[atomicAdd(&n_todo, 1u)] = (uint16_t)threadIdx.x;   // not written by the fused kernel
    __syncthreads();
    const uint32_t n = n_todo;
    for (uint32_t k = 0; k < n; k++) {
      const uint64_t it2 = blockIdx.x + (base + todo[k]) * gridDim.x;
      zn_merge_chunk_item<P>(one, segs, nseg, it2 / ZN_MERGE_SUB, (uint32_t)(it2 % ZN_MERGE_SUB), descs_all, tails);
    }
    __syncthreads();
  }
}

void zn_launch_decode_generic(int P, const ZnSeg& one, const ZnSeg* d_segs, uint32_t nseg, uint64_t total_pk, uint64_t total_k,
                              ZnPlaneDesc* d_descs, uint32_t* d_status, const uint8_t* d_done, const uint8_t* d_pdone,
                              const uint8_t* d_tail_scratch, const uint8_t* d_tail_done, hipStream_t stream) {
  if (total_k == 0) return;
  // d_done != null: the fused kernel ran before us and counted the chunks it left in d_status[1 + q]
  const uint32_t* left = d_done ? d_status + 1 + (P == 1 ? 0 : P == 2 ? 1 : 2) : nullptr;
  const uint32_t gp = (uint32_t)(total_pk < 2048u ? total_pk : 2048u), gm = (uint32_t)(total_k * ZN_MERGE_SUB < 1024u ? total_k * ZN_MERGE_SUB : 1024u);
  hipLaunchKernelGGL(zn_k_decode_planes, dim3(gp), dim3(ZN_WAVE), 0, stream, one, d_segs, nseg, total_pk, d_descs, d_status, d_pdone, d_tail_done, left);
  zn_note_kernel("zn_k_decode_planes");
  if (P == 1) hipLaunchKernelGGL(zn_k_merge_planes<1>, dim3(gm), dim3(256), 0, stream, one, d_segs, nseg, total_k, d_descs, d_done, d_tail_scratch, left);
  else if (P == 2) hipLaunchKernelGGL(zn_k_merge_planes<2>, dim3(gm), dim3(256), 0, stream, one, d_segs, nseg, total_k, d_descs, d_done, d_tail_scratch, left);
  else hipLaunchKernelGGL(zn_k_merge_planes<4>, dim3(gm), dim3(256), 0, stream, one, d_segs, nseg, total_k, d_descs, d_done, d_tail_scratch, left);
  zn_note_kernel("zn_k_merge_planes");
}
