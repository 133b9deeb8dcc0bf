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
__global__ __launch_bounds__(256) void zn_k_split_planes(ZnGeom g, uint64_t c0, const uint8_t* __restrict__ src,
                                                         uint8_t* __restrict__ planes, uint64_t slot) {
  const uint64_t c = c0 + blockIdx.x, KL = g.K - c0;          // scratch slots are indexed relative to c0
  const uint32_t clen = zn_chunk_len(g, c);
  const uint8_t* in = src + c * g.chunk;
  const uint32_t nwords = clen / 4u;
  uint8_t* pl[P];
  for (int p = 0; p < P; p++) pl[p] = planes + ((uint64_t)p * KL + (c - c0)) * slot;
  const bool aligned = (((uint64_t)in) & 3u) == 0;
  for (uint32_t wi = threadIdx.x; wi < nwords; wi += blockDim.x) {
    uint32_t w = aligned ? *(const uint32_t*)(in + 4ull * wi) : zn_ld32(in + 4ull * wi);
    if (g.rot) w = (P == 2) ? zn_rot_fwd16(w) : zn_rot_fwd32(w);
    for (uint32_t t = 0; t < 4; t++) { const uint32_t j = 4u * wi + t; pl[j % P][j / P] = (uint8_t)(w >> (8 * t)); }
  }
  if (threadIdx.x < (clen & 3u)) { const uint32_t j = 4u * nwords + threadIdx.x; pl[j % P][j / P] = in[j]; }
}

// ---------------------------------------------------------------------------
// kernel 2: per-plane huff0 encode
// ---------------------------------------------------------------------------
// total code bits of src[0..n) (without the end mark)
__device__ inline uint32_t zn_stream_bits(const uint8_t* src, uint32_t n, const uint8_t* nbits) {
  uint32_t t = 0;
  for (uint32_t i = 0; i < n; i++) t += nbits[src[i]];
  return t;
}
// codes of src[n-1] .. src[0], end mark, zero pad, LSB-first into dst (nbytes known in advance)
__device__ inline void zn_encode_stream_serial(uint8_t* dst, const uint8_t* src, uint32_t n, const uint8_t* nbits,
                                               const uint16_t* vals) {
  uint64_t acc = 0; uint32_t nacc = 0, o = 0;
  for (uint32_t i = n; i-- > 0;) {
    const uint32_t s = src[i];
    acc |= (uint64_t)vals[s] << nacc; nacc += nbits[s];
    while (nacc >= 8) { dst[o++] = (uint8_t)acc; acc >>= 8; nacc -= 8; }
  }
  acc |= 1ull << nacc; nacc += 1;
  while (nacc >= 8) { dst[o++] = (uint8_t)acc; acc >>= 8; nacc -= 8; }
  if (nacc) dst[o++] = (uint8_t)acc;
}

__global__ __launch_bounds__(ZN_WAVE) void zn_k_encode_planes(ZnGeom g, uint64_t c0, const uint8_t* __restrict__ planes,
                                                              uint8_t* __restrict__ enc, uint64_t slot, float threshold,
                                                              uint32_t* __restrict__ csize_out, uint8_t* __restrict__ type_out) {
  __shared__ ZnTabScratch S;
  __shared__ ZnHNode nodes[513];
  __shared__ uint32_t sh_hdr, sh_go, sh_bits[4];

  const uint32_t lane = threadIdx.x;
  const uint64_t KL = g.K - c0, pcl = blockIdx.x;              // local (scratch) index
  const uint32_t p = (uint32_t)(pcl / KL);
  const uint64_t c = c0 + pcl % KL;
  const uint64_t pc = (uint64_t)p * g.K + c;                   // global index
  const uint32_t n = zn_plane_len(zn_chunk_len(g, c), g.P, p);
  const uint8_t* src = planes + pcl * slot;
  uint8_t* dst = enc + pcl * slot;
  const uint64_t cap = g.chunk;   // HUF_compress dstCapacity at the call site (zipnn_core.c:366-368)

  for (uint32_t i = lane; i < 256u; i += ZN_WAVE) S.count[i] = 0;
  __syncthreads();
  for (uint32_t i = lane; i < n; i += ZN_WAVE) atomicAdd(&S.count[src[i]], 1u);
  __syncthreads();

  // HUF_compress_internal control flow, lane 0
  if (lane == 0) {
    uint32_t cs = 0;          // HUF_compress return value, truncated to 32 bits like the reference does
    uint32_t go = 0, hdr = 0;
    if (n == 0) cs = 0;
    else if (n > ZN_HUF_BLOCK_MAX) cs = 0xFFFFFFB8u;   // (size_t)-72 "srcSize_wrong" → fails the threshold test → raw
    else {
      uint32_t max_sv = 255, largest = 0;
      while (S.count[max_sv] == 0) max_sv--;
      for (uint32_t i = 0; i <= max_sv; i++) if (S.count[i] > largest) largest = S.count[i];
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
    if (lane < 4) {
      const uint32_t len = (lane < 3) ? seg : n - 3u * seg;
      sh_bits[lane] = zn_stream_bits(src + lane * seg, len, S.nbits) + 1u;   // + end mark
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
      if (lane < 4) {
        const uint32_t len = (lane < 3) ? seg : n - 3u * seg;
        zn_encode_stream_serial(dst + start[lane], src + lane * seg, len, S.nbits, S.vals);
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
__global__ __launch_bounds__(ZN_SCAN_THREADS) void zn_k_scan_sizes(ZnGeom g, uint64_t T, const uint32_t* __restrict__ csize,
                                                                   const uint8_t* __restrict__ type, uint64_t* __restrict__ offs,
                                                                   uint64_t* __restrict__ total, uint8_t* __restrict__ body) {
  __shared__ uint64_t red[ZN_SCAN_THREADS / 64][4];
  __shared__ uint32_t wtot[ZN_SCAN_THREADS / 64];
  __shared__ uint64_t pb_s[4];                 // scan value at a plane start that lies inside this block
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint64_t PK = (uint64_t)g.P * g.K, K = g.K;
  if (K == 0) { if (t == 0 && blockIdx.x == 0) *total = 0; return; }       // empty input: empty body
  uint8_t* cum = body + PK;
  const uint64_t i0 = (uint64_t)blockIdx.x * T, i1 = (i0 + T < PK) ? i0 + T : PK;
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
__global__ __launch_bounds__(256) void zn_k_gather_payload(ZnGeom g, uint64_t c0, const uint8_t* __restrict__ planes,
                                                           const uint8_t* __restrict__ enc, uint64_t slot,
                                                           const uint32_t* __restrict__ csize, const uint8_t* __restrict__ type,
                                                           const uint64_t* __restrict__ offs, uint8_t* __restrict__ body) {
  const uint64_t KL = g.K - c0, pcl = blockIdx.x;
  const uint64_t pc = (pcl / KL) * g.K + c0 + pcl % KL;
  const uint8_t* s = (type[pc] ? enc : planes) + pcl * slot;
  uint8_t* d = body + offs[pc];
  const uint32_t n = csize[pc];
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
}

void zn_launch_encode_generic_stats(const ZnGeom& g, uint64_t c0, const uint8_t* d_src, float threshold, uint8_t* d_planes,
                                    uint8_t* d_enc, uint32_t* d_csize, uint8_t* d_type, hipStream_t stream) {
  if (c0 >= g.K) return;
  const uint64_t slot = zn_plane_slot(g.chunk, (int)g.P);
  const uint32_t KL = (uint32_t)(g.K - c0), PKL = (uint32_t)g.P * KL;
  if (g.P == 1) hipLaunchKernelGGL(zn_k_split_planes<1>, dim3(KL), dim3(256), 0, stream, g, c0, d_src, d_planes, slot);
  else if (g.P == 2) hipLaunchKernelGGL(zn_k_split_planes<2>, dim3(KL), dim3(256), 0, stream, g, c0, d_src, d_planes, slot);
  else hipLaunchKernelGGL(zn_k_split_planes<4>, dim3(KL), dim3(256), 0, stream, g, c0, d_src, d_planes, slot);
  zn_note_kernel("zn_k_split_planes");
  hipLaunchKernelGGL(zn_k_encode_planes, dim3(PKL), dim3(ZN_WAVE), 0, stream, g, c0, d_planes, d_enc, slot, threshold, d_csize, d_type);
  zn_note_kernel("zn_k_encode_planes");
}

void zn_launch_scan_sizes(const ZnGeom& g, const uint32_t* d_csize, const uint8_t* d_type, uint64_t* d_offs,
                          uint64_t* d_total, uint8_t* d_body, hipStream_t stream) {
  // ≤ 256 blocks of T entries each, T a multiple of the tile size
  const uint64_t PK = (uint64_t)g.P * g.K;
  uint64_t nb = (PK + 511u) / 512u; if (nb > 256u) nb = 256u; if (nb < 1u) nb = 1u;
  uint64_t T = (PK + nb - 1u) / nb; T = (T + ZN_SCAN_THREADS - 1u) / ZN_SCAN_THREADS * ZN_SCAN_THREADS; if (T == 0) T = ZN_SCAN_THREADS;
  const uint32_t grid = PK ? (uint32_t)((PK + T - 1u) / T) : 1u;
  hipLaunchKernelGGL(zn_k_scan_sizes, dim3(grid), dim3(ZN_SCAN_THREADS), 0, stream, g, T, d_csize, d_type, d_offs, d_total, d_body);
  zn_note_kernel("zn_k_scan_sizes");
}

void zn_launch_encode_generic_gather(const ZnGeom& g, uint64_t c0, const uint8_t* d_planes, const uint8_t* d_enc,
                                     const uint32_t* d_csize, const uint8_t* d_type, const uint64_t* d_offs, uint8_t* d_body,
                                     hipStream_t stream) {
  if (c0 >= g.K) return;
  const uint64_t slot = zn_plane_slot(g.chunk, (int)g.P);
  const uint32_t PKL = (uint32_t)(g.P * (g.K - c0));
  hipLaunchKernelGGL(zn_k_gather_payload, dim3(PKL), dim3(256), 0, stream, g, c0, d_planes, d_enc, slot, d_csize, d_type, d_offs, d_body);
  zn_note_kernel("zn_k_gather_payload");
}
