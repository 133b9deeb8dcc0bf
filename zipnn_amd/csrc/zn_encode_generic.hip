// zn_encode_generic.hip — the size scan of compress: per-plane inclusive scan of the stored sizes -> types, cumSizes (wire format),
// payload offsets, total body length.  (Until round 4 this file also held a second, generic encoder — split / encode / gather kernels
// for partial last chunks and odd geometries; those planes are coded by the tail workgroups of the fused launches now,
// zn_encode_fused.hip.)
//
// Replaces: prepare_python_return_buffer (reference csrc/zipnn_core.c:105-244).
#include "zn_internal.hpp"
#include "zn_huf_tables.hpp"

// ---------------------------------------------------------------------------
// sizes -> wire-format metadata + payload offsets
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
                                                                   uint64_t* __restrict__ offs_all, uint64_t* __restrict__ total_all, uint32_t* __restrict__ spec_status) {
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
      // the one-pass encoder placed the last plane behind P - 1 planes stored raw in every chunk: a stored size is never larger than its plane, so the
      // sum in front of the last plane is the speculated one exactly when no plane in front of it was stored any other way
      if (spec_status && S.nfull && g.P > 1u && i == (uint64_t)(g.P - 1u) * K) {
        const uint32_t last_len = (uint32_t)(g.n - (K - 1u) * g.chunk);
        uint64_t want = 0;
        for (uint32_t q = 0; q + 1u < g.P; q++) want += (K - 1u) * (g.chunk / g.P) + zn_plane_len(last_len, g.P, q);
        if (x != want) atomicOr(spec_status, ZN_DEV_MISSPEC);
      }
      const uint64_t c = x + v - pb;
      if (aligned) *(uint64_t*)(cum + 8u * i) = c; else zn_st64(cum + 8u * i, c);
      body[i] = ty;
    }
    run += ttot;
  }
  if (i1 == PK && t == 0) *total = base + run;
}

// ≤ 256 blocks of T entries each, T a multiple of the tile size
void zn_scan_geometry(uint64_t PK, uint64_t* T_out, uint32_t* blocks) {
  uint64_t nb = (PK + 511u) / 512u; if (nb > 256u) nb = 256u; if (nb < 1u) nb = 1u;
  uint64_t T = (PK + nb - 1u) / nb; T = (T + ZN_SCAN_THREADS - 1u) / ZN_SCAN_THREADS * ZN_SCAN_THREADS; if (T == 0) T = ZN_SCAN_THREADS;
  *T_out = T; *blocks = PK ? (uint32_t)((PK + T - 1u) / T) : 1u;
}

void zn_launch_scan_sizes(const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_blocks, const uint32_t* d_csize,
                          const uint8_t* d_type, uint64_t* d_offs, uint64_t* d_total, uint32_t* d_spec_status, hipStream_t stream) {
  if (total_blocks == 0) return;
  hipLaunchKernelGGL(zn_k_scan_sizes, dim3(total_blocks), dim3(ZN_SCAN_THREADS), 0, stream, one, d_segs, nseg, d_csize, d_type, d_offs, d_total, d_spec_status);
  zn_note_kernel("zn_k_scan_sizes");
}

