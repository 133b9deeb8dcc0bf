// zn_encode_generic.hip — generic encode path (any dtype, any tail).  Four kernels:
//
//   zn_k_split_planes    one workgroup per chunk: sign-bit rotate + P-way byte de-interleave
//                        into scratch planes (the caller's input is never modified).
//   zn_k_encode_planes   one wave per (plane, chunk): 256-bin histogram, huff0 decisions,
//                        code construction, tree description, 4 backward bit-streams; emits
//                        (type, stored size) and the huff0 block into an encode slot.
//   zn_k_scan_sizes      one workgroup: per-plane inclusive scan of stored sizes -> types,
//                        cumSizes (wire format), payload offsets, total body length.
//   zn_k_gather_payload  one workgroup per (plane, chunk): copy the stored bytes (huff0 block
//                        or raw plane) to their plane-major position in the body.
//
// Replaces: compression_worker (reference csrc/zipnn_core.c:294-390), HUF_compress (call
// site :366), split_bytearray_dtype8/16/32 (data_manipulation_dtype16.c:33-138,
// data_manipulation_dtype32.c:78-133,219-268), prepare_python_return_buffer and
// copy_compressed_data_interleaved (zipnn_core.c:56-86,105-244).
#include "zn_internal.hpp"
#include "zn_huf_tables.hpp"

// ---------------------------------------------------------------------------
// kernel 1: rotate + split
// ---------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void zn_k_split_planes(ZnESeg one, const ZnESeg* __restrict__ segs, uint32_t nseg,
                                                         uint8_t* __restrict__ planes_all, uint64_t slot) {
  const ZnESeg S = zn_efind_tail(one, segs, nseg, blockIdx.x);
  const ZnGeom g = S.g; const uint64_t c0 = S.nfull;
  const uint8_t* __restrict__ src = ZN_GLOBAL_PTR(const uint8_t, S.src);
  uint8_t* __restrict__ planes = planes_all + S.slot0 * slot;
  const uint64_t c = c0 + (blockIdx.x - S.tail0), KL = g.K - c0;          // scratch slots are indexed relative to c0
  const uint32_t clen = zn_chunk_len(g, c);
  const uint8_t* in = src + c * g.chunk;
  const uint8_t* xin = S.xr ? ZN_GLOBAL_PTR(const uint8_t, S.xr) + c * g.chunk : nullptr;   // delta base: the encoder sees in ^ xin
  const uint32_t nwords = clen / 4u;
  uint8_t* pl[P];
  for (int p = 0; p < P; p++) pl[p] = planes + ((uint64_t)p * KL + (c - c0)) * slot;
  const bool aligned = (((uint64_t)in) & 3u) == 0;
  // gridDim.y workgroups share a chunk (a lone partial chunk should not take 80 µs on one CU)
  const uint32_t w_lo = (uint32_t)(((uint64_t)nwords * blockIdx.y) / gridDim.y), w_hi = (uint32_t)(((uint64_t)nwords * (blockIdx.y + 1u)) / gridDim.y);
  for (uint32_t wi = w_lo + threadIdx.x; wi < w_hi; wi += blockDim.x) {
    uint32_t w = aligned ? *(const uint32_t*)(in + 4ull * wi) : zn_ld32(in + 4ull * wi);
    if (xin) w ^= zn_ld32(xin + 4ull * wi);
    if (g.rot) w = (P == 2) ? zn_rot_fwd16(w) : zn_rot_fwd32(w);
    for (uint32_t t = 0; t < 4; t++) { const uint32_t j = 4u * wi + t; pl[j % P][j / P] = (uint8_t)(w >> (8 * t)); }
  }
  if (blockIdx.y == gridDim.y - 1u && threadIdx.x < (clen & 3u)) { const uint32_t j = 4u * nwords + threadIdx.x; pl[j % P][j / P] = (uint8_t)(in[j] ^ (xin ? xin[j] : 0)); }
}

// ---------------------------------------------------------------------------
// kernel 2: per-plane huff0 encode
// ---------------------------------------------------------------------------
typedef struct __attribute__((aligned(1))) { uint32_t x, y, z, w; } zn_g128u;
typedef uint32_t __attribute__((aligned(1))) zn_g32u;

// total code bits of src[0..n) (without the end mark), by the whole wave
__device__ inline uint32_t zn_stream_bits_wave(const uint8_t* src, uint32_t n, const uint8_t* nbits, uint32_t lane) {
  uint32_t t = 0;
  const uint32_t nv = n / 16u;
  for (uint32_t v = lane; v < nv; v += ZN_WAVE) {
    const zn_g128u x = *(const zn_g128u*)(src + 16u * v);
    const uint32_t d[4] = {x.x, x.y, x.z, x.w};
    for (int k = 0; k < 4; k++) for (int b = 0; b < 4; b++) t += nbits[(d[k] >> (8 * b)) & 0xFFu];
  }
  for (uint32_t i = 16u * nv + lane; i < n; i += ZN_WAVE) t += nbits[src[i]];
  for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
  return t;
}

// wave-wide exclusive prefix sum (DPP), *total = sum
__device__ __forceinline__ uint32_t zn_gen_excl_scan(uint32_t v, uint32_t* total) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  *total = (uint32_t)__builtin_amdgcn_readlane(x, 63);
  return (uint32_t)x - v;
}

// One huff0 stream, by the whole wave: codes of src[n-1] .. src[0], end mark, zero pad, LSB-first into dst
// (nbytes = the stream size, known in advance).  Tiles of 2048 symbols from the END of the segment; a lane packs
// 32 consecutive symbols (pairs, then quads of ≤ 44 bits), a prefix sum of the bit counts places the lanes (lane 63
// lowest), ds_or merges them in the tile buffer `buf` (≥ 772 dwords of LDS), whole dwords go out.
#define ZN_G_BUF_DW 772
__device__ inline void zn_encode_stream_wave(uint8_t* dst, uint32_t nbytes, const uint8_t* src, uint32_t n, const uint8_t* nbits,
                                             const uint16_t* vals, uint32_t* buf, uint32_t lane) {
  for (uint32_t i = lane; i < ZN_G_BUF_DW; i += ZN_WAVE) buf[i] = 0;
  __syncthreads();
  uint32_t carry = 0, written = 0;
  const uint32_t ntiles = (n + 2047u) / 2048u;
  for (uint32_t t = 0; t < ntiles; t++) {
    // this lane's symbols: indices first .. first + 31, first may be negative in the segment's first tile
    const int32_t first = (int32_t)n - 2048 * (int32_t)(t + 1u) + 32 * (int32_t)lane;
    uint8_t sy[32];
    if (first >= 0) {
      const zn_g128u a = *(const zn_g128u*)(src + first), b = *(const zn_g128u*)(src + first + 16);
      const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      for (int e = 0; e < 32; e++) sy[e] = (uint8_t)(d[e >> 2] >> (8 * (e & 3)));
    } else {
      for (int e = 0; e < 32; e++) { const int32_t i = first + e; sy[e] = (i >= 0) ? src[i] : 0; }
    }
    uint64_t qv[8]; uint32_t qn[8];
    for (int i = 0; i < 8; i++) {
      uint32_t pv[2], pn[2];
      for (int h = 0; h < 2; h++) {
        uint32_t v[2], l[2];
        for (int u = 0; u < 2; u++) {
          const int e = 4 * i + 2 * h + u;
          const bool valid = first + e >= 0;
          v[u] = valid ? (uint32_t)vals[sy[e]] : 0u; l[u] = valid ? (uint32_t)nbits[sy[e]] : 0u;
        }
        pv[h] = (v[0] << l[1]) | v[1]; pn[h] = l[0] + l[1];      // the later symbol takes the lower bits
      }
      qv[i] = ((uint64_t)pv[0] << pn[1]) | pv[1]; qn[i] = pn[0] + pn[1];
    }
    uint32_t qo[8]; uint32_t T = 0;
    for (int i = 7; i >= 0; i--) { qo[i] = T; T += qn[i]; }
    uint32_t total = 0;
    const uint32_t excl = zn_gen_excl_scan(T, &total);
    const uint32_t bpos = carry + (total - excl - T);
    for (int i = 0; i < 8; i++) {
      const uint32_t bp = bpos + qo[i], idx = bp >> 5, sh = bp & 31u;
      const uint64_t lo = qv[i] << sh;
      atomicOr(&buf[idx], (uint32_t)lo);
      atomicOr(&buf[idx + 1u], (uint32_t)(lo >> 32));
      atomicOr(&buf[idx + 2u], (uint32_t)(((qv[i] >> 32) << sh) >> 32));
    }
    __syncthreads();
    const uint32_t bits = carry + total, nd = bits >> 5;
    uint32_t tail = 0;
    for (uint32_t i = lane; i <= nd; i += ZN_WAVE) {
      const uint32_t x = buf[i];
      if (i < nd && written + 4u * i + 4u <= nbytes) *(zn_g32u*)(dst + written + 4u * i) = x;
      if (i == nd) tail = x;
      buf[i] = 0;
    }
    tail = __shfl(tail, (int)(nd & 63u));
    __syncthreads();
    if (lane == 0) buf[0] = tail;
    __syncthreads();
    written += 4u * nd; carry = bits & 31u;
  }
  if (lane == 0) {
    const uint32_t x = buf[0] | (1u << carry);             // end mark; the stream ends in a non-zero byte
    const uint32_t nb = (carry + 1u + 7u) >> 3;
    for (uint32_t k = 0; k < nb && written + k < nbytes; k++) dst[written + k] = (uint8_t)(x >> (8 * k));
  }
  __syncthreads();
}

__global__ __launch_bounds__(ZN_WAVE) void zn_k_encode_planes(ZnESeg one, const ZnESeg* __restrict__ segs, uint32_t nseg,
                                                              const uint8_t* __restrict__ planes_all, uint8_t* __restrict__ enc_all, uint64_t slot,
                                                              uint32_t* __restrict__ csize_all, uint8_t* __restrict__ type_all) {
  const ZnESeg SG = zn_efind_ptail(one, segs, nseg, blockIdx.x);
  const ZnGeom g = SG.g; const uint64_t c0 = SG.nfull; const float threshold = SG.threshold;
  const uint8_t* __restrict__ planes = planes_all + SG.slot0 * slot; uint8_t* __restrict__ enc = enc_all + SG.slot0 * slot;
  uint32_t* __restrict__ csize_out = csize_all + SG.pc0; uint8_t* __restrict__ type_out = type_all + SG.pc0;
  __shared__ ZnTabScratch S;
  __shared__ uint32_t S_count[256];
  if (threadIdx.x == 0) S.count = ZN_LDS_PTR(uint32_t, S_count);
  __syncthreads();
  __shared__ ZnHNode nodes[513];
  __shared__ uint32_t sh_hdr, sh_go, sh_bits[4];
  __shared__ uint32_t sh_buf[ZN_G_BUF_DW];

  const uint32_t lane = threadIdx.x;
  const uint64_t KL = g.K - c0, pcl = blockIdx.x - SG.ptail0;  // local (scratch) index
  const uint32_t p = (uint32_t)(pcl / KL);
  const uint64_t c = c0 + pcl % KL;
  const uint64_t pc = (uint64_t)p * g.K + c;                   // global index
  const uint32_t n = zn_plane_len(zn_chunk_len(g, c), g.P, p);
  const uint8_t* src = planes + pcl * slot;
  uint8_t* dst = enc + pcl * slot;
  const uint64_t cap = g.chunk;   // HUF_compress dstCapacity at the call site (zipnn_core.c:366-368)

  for (uint32_t i = lane; i < 256u; i += ZN_WAVE) S_count[i] = 0;
  __syncthreads();
  {
    const uint32_t nv = n / 16u;                       // (scratch planes start 16-byte aligned)
    for (uint32_t v = lane; v < nv; v += ZN_WAVE) {
      const uint4 x = *(const uint4*)(src + 16u * v);
      const uint32_t d[4] = {x.x, x.y, x.z, x.w};
      for (int k = 0; k < 4; k++) for (int b = 0; b < 4; b++) atomicAdd(&S_count[(d[k] >> (8 * b)) & 0xFFu], 1u);
    }
    for (uint32_t i = 16u * nv + lane; i < n; i += ZN_WAVE) atomicAdd(&S_count[src[i]], 1u);
  }
  __syncthreads();

  // HUF_compress_internal control flow, lane 0
  if (lane == 0) {
    uint32_t cs = 0;          // HUF_compress return value, truncated to 32 bits like the reference does
    uint32_t go = 0, hdr = 0;
    if (n == 0) cs = 0;
    else if (n > ZN_HUF_BLOCK_MAX) cs = 0xFFFFFFB8u;   // (size_t)-72 "srcSize_wrong" → fails the threshold test → raw
    else {
      uint32_t max_sv = 255, largest = 0;
      while (S_count[max_sv] == 0) max_sv--;
      for (uint32_t i = 0; i <= max_sv; i++) if (S_count[i] > largest) largest = S_count[i];
      if (largest == n) { dst[0] = src[0]; cs = 1; }
      else if (largest <= (n >> 7) + 4u) cs = 0;
      else {
        uint32_t huff_log = zn_optimal_table_log(ZN_HUF_LOG_DEFAULT, n, max_sv, 1);
        huff_log = zn_huf_build_ctable(&S, nodes, max_sv, huff_log);
        const int h = zn_huf_write_ctable(&S, max_sv, huff_log);
        if (h < 0) cs = 0xFFFFFFFFu;                  // huff0 error code → raw
        else if ((uint32_t)h + 12u >= n) cs = 0;
        else if (cap - (uint32_t)h < 6u + 1u + 1u + 1u + 8u || n < 12u) cs = 0;
        else { hdr = (uint32_t)h; go = 1; }
      }
    }
    sh_hdr = hdr; sh_go = go;
    if (!go) { csize_out[pc] = cs; }
  }
  __syncthreads();

  if (sh_go) {
    const uint32_t hdr = sh_hdr;
    const uint32_t seg = (n + 3u) / 4u;
    for (uint32_t q = 0; q < 4u; q++) {
      const uint32_t len = (q < 3u) ? seg : n - 3u * seg;
      const uint32_t t = zn_stream_bits_wave(src + q * seg, len, S.nbits, lane);
      if (lane == 0) sh_bits[q] = t + 1u;                                    // + end mark
    }
    __syncthreads();
    // sizes, capacity rule of BIT_closeCStream, and the final "did it shrink" test
    uint32_t sz[4], start[4]; uint32_t pos = hdr + 6u; bool fail = false;
    for (int k = 0; k < 4; k++) {
      const uint64_t cap_rem = cap - pos;
      if (cap_rem <= 8u || (uint64_t)(sh_bits[k] >> 3) >= cap_rem - 8u) { fail = true; break; }
      sz[k] = (sh_bits[k] + 7u) >> 3; start[k] = pos; pos += sz[k];
    }
    uint32_t cs = fail ? 0u : pos;
    if (!fail && pos >= n - 1u) cs = 0;
    const bool keep = cs != 0 && (double)cs < (double)n * (double)threshold;
    if (keep) {
      for (uint32_t i = lane; i < hdr; i += ZN_WAVE) dst[i] = S.hdr[i];
      if (lane < 3) { dst[hdr + 2u * lane] = (uint8_t)sz[lane]; dst[hdr + 2u * lane + 1u] = (uint8_t)(sz[lane] >> 8); }
      for (uint32_t q = 0; q < 4u; q++) {
        const uint32_t len = (q < 3u) ? seg : n - 3u * seg;
        zn_encode_stream_wave(dst + start[q], sz[q], src + q * seg, len, S.nbits, S.vals, sh_buf, lane);
      }
    }
    if (lane == 0) csize_out[pc] = cs;
  }
  __syncthreads();
  if (lane == 0) {
    // threshold rule of compression_worker (zipnn_core.c:371-385)
    const uint32_t cs = csize_out[pc];
    const bool huf = cs != 0 && (double)cs < (double)n * (double)threshold;
    type_out[pc] = huf ? 1 : 0;
    csize_out[pc] = huf ? cs : n;
  }
}

// ---------------------------------------------------------------------------
// kernel 3: sizes -> wire-format metadata + payload offsets
// ---------------------------------------------------------------------------
// Plane-major order is index order (i = p·K + c), so ONE exclusive scan over all P·K stored sizes gives
// every payload offset; the wire format's per-plane inclusive cumSizes are that scan minus its value at
// the plane's first index.  Every workgroup owns a contiguous block of T entries (T a multiple of 256):
// it first sums everything in front of its block by plane (coalesced reads of an L2-resident array — no
// inter-workgroup dependency), then scans its block in tiles of 256 with coalesced loads and stores.
#define ZN_SCAN_THREADS 256
__device__ __forceinline__ uint64_t zn_wave_sum64(uint64_t v) {
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, d), hi = __shfl_xor((uint32_t)(v >> 32), d);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}
__global__ __launch_bounds__(ZN_SCAN_THREADS) void zn_k_scan_sizes(ZnESeg one, const ZnESeg* __restrict__ segs, uint32_t nseg,
                                                                   const uint32_t* __restrict__ csize_all, const uint8_t* __restrict__ type_all,
                                                                   uint64_t* __restrict__ offs_all, uint64_t* __restrict__ total_all) {
  const ZnESeg S = zn_efind_scan(one, segs, nseg, blockIdx.x);
  const ZnGeom g = S.g; const uint64_t T = S.T;
  const uint32_t* __restrict__ csize = csize_all + S.pc0; const uint8_t* __restrict__ type = type_all + S.pc0;
  uint64_t* __restrict__ offs = offs_all + S.pc0; uint64_t* __restrict__ total = total_all + S.total_idx;
  uint8_t* __restrict__ body = ZN_GLOBAL_PTR(uint8_t, S.body);
  const uint32_t blk = blockIdx.x - S.scan0;
  __shared__ uint64_t red[ZN_SCAN_THREADS / 64][4];
  __shared__ uint32_t wtot[ZN_SCAN_THREADS / 64];
  __shared__ uint64_t pb_s[4];                 // scan value at a plane start that lies inside this block
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint64_t PK = (uint64_t)g.P * g.K, K = g.K;
  if (K == 0) { if (t == 0 && blk == 0) *total = 0; return; }       // empty input: empty body
  uint8_t* cum = body + PK;
  const uint64_t i0 = (uint64_t)blk * T, i1 = (i0 + T < PK) ? i0 + T : PK;
  if (i0 >= PK) return;

  // everything in front of the block, by plane
  uint64_t sp[4] = {0, 0, 0, 0};
  for (uint64_t j = t; j < i0; j += 8u * ZN_SCAN_THREADS) {      // i0 is a multiple of 256; 8 loads in flight
    uint32_t v[8];
    for (uint32_t u = 0; u < 8u; u++) { const uint64_t i = j + (uint64_t)u * ZN_SCAN_THREADS; v[u] = (i < i0) ? csize[i] : 0u; }
    for (uint32_t u = 0; u < 8u; u++) {
      const uint64_t i = j + (uint64_t)u * ZN_SCAN_THREADS;
      const uint32_t p = (uint32_t)(i >= K) + (uint32_t)(i >= 2u * K) + (uint32_t)(i >= 3u * K);
      for (uint32_t q = 0; q < 4u; q++) sp[q] += (p == q) ? (uint64_t)v[u] : 0u;
    }
  }
  for (uint32_t q = 0; q < 4u; q++) { const uint64_t r = zn_wave_sum64(sp[q]); if (lane == 0) red[wave][q] = r; }
  __syncthreads();
  uint64_t known[4], run = 0;                  // known[p]: scan value at p·K when that lies at or before i0
  for (uint32_t q = 0; q < 4u; q++) { known[q] = run; for (uint32_t w = 0; w < ZN_SCAN_THREADS / 64u; w++) run += red[w][q]; }

  const uint64_t base = 9u * PK;
  const bool aligned = ((((uint64_t)cum) & 7u) == 0);
  for (uint64_t b0 = i0; b0 < i1; b0 += ZN_SCAN_THREADS) {
    const uint64_t i = b0 + t;
    const bool in = i < i1;
    const uint32_t v = in ? csize[i] : 0u;
    const uint8_t ty = in ? type[i] : 0;
    uint32_t incl = v;                          // a tile sums to < 2^32 (256 sizes ≤ 128 KiB)
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
    if (lane == 63u) wtot[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, ttot = 0;
    for (uint32_t w = 0; w < ZN_SCAN_THREADS / 64u; w++) { const uint32_t x = wtot[w]; if (w < wave) woff += x; ttot += x; }
    const uint64_t x = run + woff + (incl - v);  // exclusive scan at i
    const uint32_t p = (uint32_t)(i >= K) + (uint32_t)(i >= 2u * K) + (uint32_t)(i >= 3u * K);
    if (in && i == (uint64_t)p * K) pb_s[p] = x;
    __syncthreads();
    if (in) {
      uint64_t pb = 0;
      for (uint32_t q = 0; q < 4u; q++) if (q == p) pb = known[q];
      if ((uint64_t)p * K >= i0) pb = pb_s[p];
      offs[i] = base + x;
      const uint64_t c = x + v - pb;
      if (aligned) *(uint64_t*)(cum + 8u * i) = c; else zn_st64(cum + 8u * i, c);
      body[i] = ty;
    }
    run += ttot;
  }
  if (i1 == PK && t == 0) *total = base + run;
}

// ---------------------------------------------------------------------------
// kernel 4: payload gather
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void zn_k_gather_payload(ZnESeg one, const ZnESeg* __restrict__ segs, uint32_t nseg,
                                                           const uint8_t* __restrict__ planes_all, const uint8_t* __restrict__ enc_all, uint64_t slot,
                                                           const uint32_t* __restrict__ csize_all, const uint8_t* __restrict__ type_all,
                                                           const uint64_t* __restrict__ offs_all) {
  const ZnESeg S = zn_efind_ptail(one, segs, nseg, blockIdx.x);
  const ZnGeom g = S.g; const uint64_t c0 = S.nfull;
  const uint8_t* __restrict__ planes = planes_all + S.slot0 * slot; const uint8_t* __restrict__ enc = enc_all + S.slot0 * slot;
  const uint32_t* __restrict__ csize = csize_all + S.pc0; const uint8_t* __restrict__ type = type_all + S.pc0;
  const uint64_t* __restrict__ offs = offs_all + S.pc0; uint8_t* __restrict__ body = ZN_GLOBAL_PTR(uint8_t, S.body);
  const uint64_t KL = g.K - c0, pcl = blockIdx.x - S.ptail0;
  const uint64_t pc = (pcl / KL) * g.K + c0 + pcl % KL;
  const uint8_t* s = (type[pc] ? enc : planes) + pcl * slot;
  uint8_t* d = body + offs[pc];
  const uint32_t n = csize[pc];
  const uint32_t lo = (uint32_t)(((uint64_t)n * blockIdx.y) / gridDim.y), hi = (uint32_t)(((uint64_t)n * (blockIdx.y + 1u)) / gridDim.y);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) d[i] = s[i];
}

void zn_launch_encode_generic_stats(int P, const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_tails, uint32_t total_ptails,
                                    uint8_t* d_planes, uint8_t* d_enc, uint64_t slot, uint32_t* d_csize, uint8_t* d_type, hipStream_t stream) {
  if (total_tails == 0) return;
  if (P == 1) hipLaunchKernelGGL(zn_k_split_planes<1>, dim3(total_tails, 16), dim3(256), 0, stream, one, d_segs, nseg, d_planes, slot);
  else if (P == 2) hipLaunchKernelGGL(zn_k_split_planes<2>, dim3(total_tails, 16), dim3(256), 0, stream, one, d_segs, nseg, d_planes, slot);
  else hipLaunchKernelGGL(zn_k_split_planes<4>, dim3(total_tails, 16), dim3(256), 0, stream, one, d_segs, nseg, d_planes, slot);
  zn_note_kernel("zn_k_split_planes");
  hipLaunchKernelGGL(zn_k_encode_planes, dim3(total_ptails), dim3(ZN_WAVE), 0, stream, one, d_segs, nseg, d_planes, d_enc, slot, d_csize, d_type);
  zn_note_kernel("zn_k_encode_planes");
}

// ≤ 256 blocks of T entries each, T a multiple of the tile size
void zn_scan_geometry(uint64_t PK, uint64_t* T_out, uint32_t* blocks) {
  uint64_t nb = (PK + 511u) / 512u; if (nb > 256u) nb = 256u; if (nb < 1u) nb = 1u;
  uint64_t T = (PK + nb - 1u) / nb; T = (T + ZN_SCAN_THREADS - 1u) / ZN_SCAN_THREADS * ZN_SCAN_THREADS; if (T == 0) T = ZN_SCAN_THREADS;
  *T_out = T; *blocks = PK ? (uint32_t)((PK + T - 1u) / T) : 1u;
}

void zn_launch_scan_sizes(const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_blocks, const uint32_t* d_csize,
                          const uint8_t* d_type, uint64_t* d_offs, uint64_t* d_total, hipStream_t stream) {
  if (total_blocks == 0) return;
  hipLaunchKernelGGL(zn_k_scan_sizes, dim3(total_blocks), dim3(ZN_SCAN_THREADS), 0, stream, one, d_segs, nseg, d_csize, d_type, d_offs, d_total);
  zn_note_kernel("zn_k_scan_sizes");
}

void zn_launch_encode_generic_gather(const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_ptails, const uint8_t* d_planes,
                                     const uint8_t* d_enc, uint64_t slot, const uint32_t* d_csize, const uint8_t* d_type,
                                     const uint64_t* d_offs, hipStream_t stream) {
  if (total_ptails == 0) return;
  hipLaunchKernelGGL(zn_k_gather_payload, dim3(total_ptails, 16), dim3(256), 0, stream, one, d_segs, nseg, d_planes, d_enc, slot, d_csize, d_type, d_offs);
  zn_note_kernel("zn_k_gather_payload");
}
