// zn_encode_fused.hip — the bandwidth path of compress: three kernels and one tiny scan, no scratch
// planes, every compressed byte written once, directly at its final position in the frame body.
//
// The wire format is plane-major (all chunks of plane 0, then plane 1, …), so where a chunk's bytes go
// depends on the stored sizes of ALL chunks.  huff0's output size, however, is known without encoding:
// header size + Σ_streams ceil((Σ_{symbols of the stream} len[sym] + 1) / 8).  Hence:
//
//   zn_k_encode_stats<P>   one workgroup per full chunk, wave w = quarter w = stream w:
//                          rotate + split on the fly, byte histograms with bank-conflict-free LDS atomics
//                          (packed 16-bit counters, one column per lane of a half-wave), quarter by quarter —
//                          which yields the per-stream symbol counts; the cheap exits of HUF_compress (RLE,
//                          "not compressible") decided here, the other planes marked for a code table.
//   zn_k_encode_tables     one wave per marked (plane, chunk): HUF_sort as a rank sort over the lanes, the serial
//                          part of HUF_compress on lane 0 (tree, length limiting, tree description incl. its FSE
//                          coding: zn_huf_tables.hpp), symbol-parallel the rest; stream sizes = per-quarter counts ·
//                          code lengths, capacity and threshold rules → type + stored size; kept planes leave their
//                          code table + tree description in a small descriptor.
//   zn_k_scan_sizes        (zn_encode_generic.hip) per-plane inclusive scan → types, cumSizes, offsets, total.
//   zn_k_encode_emit<P>    one workgroup per full chunk: re-reads the chunk, writes raw
//                          planes straight to their payload position and bit-packs the 4 streams of a Huffman
//                          plane in place: each lane packs the codes of 32 consecutive symbols in registers,
//                          a wave prefix sum of the bit counts gives its bit offset in the tile, `ds_or_b32`
//                          merges the lanes' words in an LDS tile buffer, full dwords go out coalesced.
//
// Algorithmic HBM bytes: N in + C out; this design reads N twice — 2 N + C, which is what the plane-major layout costs
// any encoder of a tensor larger than the Infinity Cache (no byte of plane 1 can be placed before the
// sizes of ALL chunks of plane 0 are known).
// Chunks these kernels do not take (the partial tail; or everything when chunk % (8192·P) ≠ 0 or planes
// exceed 128 KiB) are handled by zn_encode_generic.hip — the host splits the chunk range.
//
// Replaces: compression_worker + HUF_compress call (reference csrc/zipnn_core.c:294-390), split_bytearray_*
// (data_manipulation_dtype16.c:33-138, dtype32.c:78-133), prepare_python_return_buffer (:105-244).
#include "zn_internal.hpp"
#include "zn_huf_tables.hpp"

#define ZN_E_THREADS 256
#define ZN_E_SPL 32                        // symbols per lane per tile
#define ZN_E_TILE (64 * ZN_E_SPL)          // symbols per tile
#define ZN_E_BUF_DW 768                    // tile bit buffer: 2048 symbols × ≤12 bits = 768 dwords, + carry

// 16-byte input loads.  Measured (4 GiB bf16): non-temporal loads help the histogram pass (one pure streaming
// read, -3 % on compress) and hurt the emit pass (its four strided loads per tile share cache lines).
#if !defined(ZN_SIMT_EMULATOR)
typedef uint32_t zn_ev4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 zn_ldnt128(const void* p) { const zn_ev4u v = __builtin_nontemporal_load((const zn_ev4u*)p); return make_uint4(v.x, v.y, v.z, v.w); }
#else
__device__ __forceinline__ uint4 zn_ldnt128(const void* p) { return *(const uint4*)p; }
#endif
#define ZN_LD_STATS(p) zn_ldnt128(p)
// (measured and gone, profiles/r02_decode_experiments.txt: the emit pass in reverse chunk order — ±0 at 4 GiB: a non-temporal first read leaves nothing in the
//  Infinity Cache —; the stats pass requesting step i + 1 before it counts step i — ±0: what it adds to a pure read are LDS atomics, not exposed latency)
#define ZN_LD_EMIT(p) (*(const uint4*)(p))

typedef uint64_t __attribute__((aligned(1))) zn_eu64u;
typedef uint32_t __attribute__((aligned(1))) zn_eu32u;
struct __attribute__((aligned(1))) zn_eu128u { uint32_t x, y, z, w; };

// wave-wide exclusive prefix sum on the DPP network (row_shr 1/2/4/8 inside rows of 16 lanes, row_bcast15 /
// row_bcast31 across rows); *total = sum over the wave
__device__ __forceinline__ uint32_t zn_wave_excl_scan_u32(uint32_t v, uint32_t* total) {
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

__device__ __forceinline__ uint64_t zn_wave_sum64_e(uint64_t v) {
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, d), hi = __shfl_xor((uint32_t)(v >> 32), d);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

// Four consecutive elements (P dwords) -> one dword per byte plane, the reference's sign-bit rotate applied at PLANE level: byte p of dword-pair / dword
// element e goes to byte e of pl[p]; with `rot` the two top planes exchange one bit per byte (data_manipulation_dtype16.c:10-20, dtype32.c:39-49:
// top' = top << 1 | next.7, next' = top.7 << 7 | next & 0x7F) — two v_perm + four bit-field ops for four bf16 elements, where rotating every dword
// and then picking bytes cost 14 + 10 (the encoder is VALU-bound: profiles/r05_encoder_pmc.txt).
#if !defined(ZN_SIMT_EMULATOR)
__device__ __forceinline__ uint32_t zn_ebfi(uint32_t mask, uint32_t a, uint32_t b) { uint32_t d; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b)); return d; }
#else
__device__ __forceinline__ uint32_t zn_ebfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }
#endif
template <int P>
__device__ __forceinline__ void zn_split4(const uint32_t* d, uint32_t rot, uint32_t (&pl)[P]) {
  if (P == 1) { pl[0] = d[0]; return; }
  if (P == 2) {
    pl[0] = __builtin_amdgcn_perm(d[1], d[0], 0x06040200u);        // the four low bytes
    pl[1 % P] = __builtin_amdgcn_perm(d[1], d[0], 0x07050301u);    // the four high bytes
  } else {
    const uint32_t a = __builtin_amdgcn_perm(d[1], d[0], 0x05010400u), b = __builtin_amdgcn_perm(d[1], d[0], 0x07030602u);      // bytes 0,1 / 2,3 of elements 0,1
    const uint32_t c = __builtin_amdgcn_perm(d[3 % (4 * P / 4)], d[2 % P], 0x05010400u), e = __builtin_amdgcn_perm(d[3 % P], d[2 % P], 0x07030602u);   // … of elements 2,3
    pl[0] = __builtin_amdgcn_perm(c, a, 0x05040100u); pl[1 % P] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    pl[2 % P] = __builtin_amdgcn_perm(e, b, 0x05040100u); pl[3 % P] = __builtin_amdgcn_perm(e, b, 0x07060302u);
  }
  if (rot) {
    const uint32_t top = pl[P - 1], nxt = pl[(P >= 2) ? P - 2 : 0];
    pl[P - 1] = zn_ebfi(0xFEFEFEFEu, top << 1, nxt >> 7);
    pl[(P >= 2) ? P - 2 : 0] = zn_ebfi(0x7F7F7F7Fu, nxt, top);
  }
}

// rotated dword → the reference's forward bit reorder for this plane count
template <int P> __device__ __forceinline__ uint32_t zn_rot_fwd(uint32_t u, uint32_t rot) {
  if (!rot) return u;
  return (P == 2) ? zn_rot_fwd16(u) : (P == 4) ? zn_rot_fwd32(u) : u;
}

// ---------------------------------------------------------------------------
// kernel A: histograms → RLE / raw decisions, per-quarter symbol counts of the planes that need a table
// ---------------------------------------------------------------------------
template <int P>
struct ZnStatsLds {
  // workgroup-shared histograms, one column per lane of a half-wave so that a wave's 32 concurrent LDS
  // atomics always hit 32 different banks.  Two planes share a counter dword (plane 2i in the low, plane
  // 2i+1 in the high 16 bits), so the increment of a byte is a constant (1 or 65536) and its address is
  // just its value: counter(pair, bin, col) = pair[bin * COLS + col].  A column sees ≤ 8192 symbols of a
  // plane, so the low half never carries into the high one.  (P = 1: one plane, plain 32-bit counters.)
  static constexpr int COLS = (P == 4) ? 16 : 32;
  static constexpr int PAIRS = (P + 1) / 2;
  uint32_t hist[PAIRS][256 * COLS];      // 32 KiB for P = 2: five workgroups per CU — the cross-wave reduction below reuses its head
};

// The chunk is histogrammed quarter by quarter (a quarter = the symbols of one huff0 stream); after each
// quarter the columns are summed (thread = bin), which yields the per-stream symbol counts that turn code
// lengths into stream sizes later without a second pass over the data.
// X: the encoder sees src ^ xr (delta base; the host picks this instance when some tensor of the launch has one).
// ---- RAGGED planes (the partial last chunk of a tensor; every chunk of a geometry the fused kernels do not take) ----
// Extra workgroups BEHIND the full chunks of the same three launches (stats, tables, emit) — the encoder's counterpart of the decoder's
// tail workgroups — so that a tensor's 250 KB tail is coded while its bulk is, instead of by three further kernels behind it (split /
// encode / gather: 0.34 ms for one chunk, as long as 600 MB of bulk).  Stats: FOUR workgroups per (plane, chunk), one per huff0 stream
// (a stream is ceil(n / 4) symbols, the last one what is left): each de-interleaves its quarter of the plane into the scratch slot (the
// rotate of the reference, its un-rotated trailing bytes included) and counts it; the plane's decisions (RLE / raw / table) are taken
// by its table job, which sees the four quarters' counts.  (One workgroup per plane took 80-130 µs for a 250 KB chunk — the whole stats
// pass of a 100 MiB tensor takes 36.)  Reference: compression_worker csrc/zipnn_core.c:294-390 codes every chunk the same way.
template <bool X>
__device__ __forceinline__ void zn_encode_tail_stats(uint32_t* hist /* LDS: [256][32] */, const ZnESeg& S, uint64_t pcl, uint32_t q,
                                                     uint8_t* __restrict__ planes_all, uint64_t slot, uint32_t* __restrict__ csize_all,
                                                     uint8_t* __restrict__ type_all, ZnEncDesc* __restrict__ descs_all) {
  const ZnGeom g = S.g; const uint64_t c0 = S.nfull, KL = g.K - c0;
  const uint32_t P = g.P, p = (uint32_t)(pcl / KL);
  const uint64_t c = c0 + pcl % KL, pc = (uint64_t)p * g.K + c;
  const uint32_t clen = zn_chunk_len(g, c), n = zn_plane_len(clen, P, p);
  const uint8_t* in = ZN_GLOBAL_PTR(const uint8_t, S.src) + c * g.chunk;
  const uint8_t* xin = (X && S.xr) ? ZN_GLOBAL_PTR(const uint8_t, S.xr) + c * g.chunk : nullptr;
  uint8_t* pl = planes_all + (S.slot0 + pcl) * slot;
  ZnEncDesc* D = descs_all + S.pc0 + pc;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const bool countable = n != 0u && n <= ZN_HUF_BLOCK_MAX;
  if (!countable && q == 0u && tid == 0) { type_all[S.pc0 + pc] = 0; csize_all[S.pc0 + pc] = n; }   // empty, or larger than a huff0 block (HUF_compress: srcSize_wrong -> fails the threshold test): raw
  const uint32_t seg = (n + 3u) / 4u;
  // this workgroup's plane indices [i_lo, i_hi)
  const uint32_t i_lo = (q * seg < n) ? q * seg : n, i_hi = (q == 3u) ? n : (((q + 1u) * seg < n) ? (q + 1u) * seg : n);
  constexpr uint32_t TC = 32u;
  uint32_t* hcol = hist + (lane & (TC - 1u));
  for (uint32_t i = tid; i < 256u * TC; i += ZN_E_THREADS) hist[i] = 0;
  __syncthreads();
  const uint32_t nwords = clen / 4u;
  auto one_byte = [&](uint32_t i) {              // plane byte i by itself: its source word, rotated — or a trailing byte, which keeps its place
    const uint32_t j = i * P + p, wi = j >> 2;
    uint32_t b;
    if (wi < nwords) {
      uint32_t w = zn_ld32(in + 4ull * wi);
      if (xin) w ^= zn_ld32(xin + 4ull * wi);
      if (g.rot) w = (P == 2) ? zn_rot_fwd16(w) : (P == 4) ? zn_rot_fwd32(w) : w;
      b = (w >> (8u * (j & 3u))) & 0xFFu;
    } else b = (uint32_t)(in[j] ^ (xin ? xin[j] : 0));
    pl[i] = (uint8_t)b;
    if (countable) atomicAdd(hcol + b * TC, 1u);
  };
  // whole 16-byte source vectors (16 / P consecutive bytes of this plane: P divides 4, so byte t of every word belongs to plane t mod P):
  // four loads in flight per thread, ONE store per vector; the edges of the range, and everything when the source is not 16-byte aligned,
  // byte by byte
  const uint32_t per = 16u / P;
  const bool aligned = ((((uint64_t)in) | ((uint64_t)xin)) & 15u) == 0;
  uint32_t v_lo = (i_lo + per - 1u) / per, v_hi = i_hi / per;
  if (v_hi > nwords / 4u) v_hi = nwords / 4u;
  if (!aligned || v_hi < v_lo) { v_lo = 0; v_hi = 0; }
  const uint32_t e_lo = (v_hi > v_lo) ? per * v_lo : i_hi, e_hi = (v_hi > v_lo) ? per * v_hi : i_hi;      // [i_lo, e_lo) and [e_hi, i_hi): the edges
  for (uint32_t v0 = v_lo + tid; v0 < v_hi; v0 += 4u * ZN_E_THREADS) {
    uint4 xs[4];
    for (int u = 0; u < 4; u++) { const uint32_t v = v0 + ZN_E_THREADS * (uint32_t)u; xs[u] = (v < v_hi) ? ZN_LD_STATS(in + 16ull * v) : make_uint4(0, 0, 0, 0); }
    if (xin) for (int u = 0; u < 4; u++) { const uint32_t v = v0 + ZN_E_THREADS * (uint32_t)u; if (v < v_hi) { const uint4 t = ZN_LD_STATS(xin + 16ull * v); xs[u].x ^= t.x; xs[u].y ^= t.y; xs[u].z ^= t.z; xs[u].w ^= t.w; } }
    for (int u = 0; u < 4; u++) {
      const uint32_t v = v0 + ZN_E_THREADS * (uint32_t)u;
      if (v >= v_hi) continue;
      uint32_t d[4] = {xs[u].x, xs[u].y, xs[u].z, xs[u].w};
      if (g.rot) for (int k = 0; k < 4; k++) d[k] = (P == 2) ? zn_rot_fwd16(d[k]) : (P == 4) ? zn_rot_fwd32(d[k]) : d[k];
      uint32_t o[4] = {0, 0, 0, 0};
      if (P == 1) { for (int k = 0; k < 4; k++) o[k] = d[k]; }
      else if (P == 2) { for (int k = 0; k < 4; k++) { const uint32_t h = ((d[k] >> (8u * p)) & 0xFFu) | (((d[k] >> (8u * p + 16u)) & 0xFFu) << 8); o[k >> 1] |= h << (16 * (k & 1)); } }
      else { for (int k = 0; k < 4; k++) o[0] |= ((d[k] >> (8u * p)) & 0xFFu) << (8 * k); }
      const uint32_t i0 = per * v;
      if (P == 1) *(uint4*)(pl + i0) = make_uint4(o[0], o[1], o[2], o[3]);
      else if (P == 2) *(uint2*)(pl + i0) = make_uint2(o[0], o[1]);
      else *(uint32_t*)(pl + i0) = o[0];
      if (countable) for (uint32_t e = 0; e < per; e++) atomicAdd(hcol + ((o[e >> 2] >> (8u * (e & 3u))) & 0xFFu) * TC, 1u);
    }
  }
  for (uint32_t i = i_lo + tid; i < e_lo; i += ZN_E_THREADS) one_byte(i);
  for (uint32_t i = e_hi + tid; i < i_hi; i += ZN_E_THREADS) one_byte(i);
  __syncthreads();
  if (!countable) return;
  uint32_t cum = 0;
  for (uint32_t r = 0; r < TC; r++) cum += hist[tid * TC + ((r + tid) & (TC - 1u))];
  D->qcount[q][tid] = (uint16_t)cum;             // (a quarter of a huff0 block: ≤ 32 768 symbols)
}

#if !defined(ZN_SIMT_EMULATOR)
#define ZN_STATS_FENCE(on) do { if (on) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ZN_STATS_FENCE(on) do { } while (0)
#endif
// The histogram pass over one full chunk: thread = bin afterwards (tot[p] = count of byte value `tid` in plane p, qc[p][q] = … in quarter q).
// NT: non-temporal loads (the stats kernel: one pure streaming read, -3 % on compress) — or plain ones (the one-pass encoder, whose second read of the
// chunk is to be served by the Infinity Cache: a non-temporal first read leaves nothing there, profiles/r05_mall_reread.txt).
// WHOLE: a quarter is a whole number of workgroup steps (4 x 256 vectors: every chunk size that is a multiple of 64 KiB, the default 256 KiB among them) — no vector of a
// step is out of range, and the loop body loses its per-vector range tests (an exec-mask region around every half vector: a tenth of its instructions)
template <int P, bool X, bool NT, bool WHOLE>
__device__ __forceinline__ void zn_stats_count_impl(ZnStatsLds<P>& L, const ZnGeom& g, const uint8_t* cs0, const uint8_t* xcs0, uint32_t tid, uint32_t lane,
                                                    uint32_t (&tot)[P], uint32_t (&qc)[P][4]) {
  constexpr uint32_t COLS = ZnStatsLds<P>::COLS;
  constexpr uint32_t PAIRS = ZnStatsLds<P>::PAIRS;
  for (int p = 0; p < P; p++) tot[p] = 0;
  const uint32_t nvec = (uint32_t)(g.chunk / 4u) / 16u;       // 16-byte vectors per quarter (a multiple of 256)
  uint32_t* hbase = &L.hist[0][lane & (COLS - 1u)];
  const uint32_t qbytes = (uint32_t)(g.chunk / 4u);
  auto ld = [&](const uint8_t* a) -> uint4 { return NT ? ZN_LD_STATS(a) : *(const uint4*)a; };
  auto fetch = [&](uint4 (&xs)[4], uint32_t q, uint32_t v0) {
    const uint8_t* qs = cs0 + (uint64_t)q * qbytes;
    for (int u = 0; u < 4; u++) { const uint32_t v = v0 + ZN_E_THREADS * (uint32_t)u; xs[u] = (WHOLE || v < nvec) ? ld(qs + 16ull * v) : make_uint4(0, 0, 0, 0); }
    if (X && xcs0) {
      const uint8_t* xqs = xcs0 + (uint64_t)q * qbytes;
      for (int u = 0; u < 4; u++) { const uint32_t v = v0 + ZN_E_THREADS * (uint32_t)u; if (WHOLE || v < nvec) { const uint4 t = ld(xqs + 16ull * v); xs[u].x ^= t.x; xs[u].y ^= t.y; xs[u].z ^= t.z; xs[u].w ^= t.w; } }
    }
  };
  uint4 nx[4];
  for (uint32_t i = tid; i < PAIRS * 256u * COLS; i += ZN_E_THREADS) (&L.hist[0][0])[i] = 0;
  __syncthreads();
  for (int q = 0; q < 4; q++) {
    // 4 independent 16-byte loads in flight per thread per step
    for (uint32_t v0 = tid; v0 < nvec; v0 += 4u * ZN_E_THREADS) {
      uint4 xs[4];
      fetch(nx, (uint32_t)q, v0);
      for (int u = 0; u < 4; u++) xs[u] = nx[u];
      for (int u = 0; u < 4; u++) if (WHOLE || v0 + ZN_E_THREADS * (uint32_t)u < nvec) {
        // (the vector's 16 / P elements plane by plane: zn_split4 per four elements, then one counter per byte)
        const uint32_t d[4] = {xs[u].x, xs[u].y, xs[u].z, xs[u].w};
        for (int k0 = 0; k0 < 4; k0 += P) {
          uint32_t pl[P];
          zn_split4<P>(d + k0, g.rot, pl);
          for (int p = 0; p < P; p++)
            for (int t = 0; t < 4; t++) {
              const uint32_t b = (pl[p] >> (8 * t)) & 0xFFu;
              atomicAdd(hbase + (uint32_t)(p >> 1) * (256u * COLS) + b * COLS, (p & 1) ? 65536u : 1u);
            }
        }
        ZN_STATS_FENCE(!NT);                     // (the one-pass kernel lives within 128 registers: one vector's sixteen counters at a time)
      }
    }
    __syncthreads();
    // sum the columns (column order staggered per thread so that the lanes of a wave read different banks)
    // (two planes share a counter: one read serves both)
    uint32_t cum[P];
    for (int p = 0; p < P; p++) cum[p] = 0;
    // (the thread index made opaque per quarter: the 32 staggered column addresses are loop-invariant, and the compiler kept all of them — 32 registers —
    //  alive across the counting loops above: 164 registers for two planes, three workgroups per CU instead of the five the histogram's LDS allows)
    uint32_t t_ = tid;
#if !defined(ZN_SIMT_EMULATOR)
    asm volatile("" : "+v"(t_));
#endif
    for (int pr = 0; pr < (int)PAIRS; pr++) {
      for (uint32_t r0 = 0; r0 < COLS; r0 += 8u) {
        uint32_t x[8];
        for (uint32_t r = 0; r < 8u; r++) x[r] = L.hist[pr][t_ * COLS + ((r0 + r + t_) & (COLS - 1u))];
        for (uint32_t r = 0; r < 8u; r++) {
          if (P == 1) cum[0] += x[r];
          else { cum[2 * pr] += x[r] & 0xFFFFu; cum[(2 * pr + 1) % P] += x[r] >> 16; }
        }
        ZN_STATS_FENCE(!NT);                     // (the one-pass kernel lives within 128 registers: eight columns at a time)
      }
    }
    for (int p = 0; p < P; p++) { qc[p][q] = cum[p] - tot[p]; tot[p] = cum[p]; }
    __syncthreads();
  }
}
template <int P, bool X, bool NT>
__device__ __forceinline__ void zn_stats_count(ZnStatsLds<P>& L, const ZnGeom& g, const uint8_t* cs0, const uint8_t* xcs0, uint32_t tid, uint32_t lane,
                                               uint32_t (&tot)[P], uint32_t (&qc)[P][4]) {
  // (the one-pass kernel only: 4 GiB bf16 2.179 -> 2.141 ms; the stats kernel's own loop got slower with it — fp32 1 GiB 0.650 -> 0.674 ms, fp8 0.689 -> 0.705)
  if (!NT && (g.chunk % (64ull * 4u * ZN_E_THREADS)) == 0) zn_stats_count_impl<P, X, NT, true>(L, g, cs0, xcs0, tid, lane, tot, qc);      // (wave-uniform: the geometry is a kernel argument)
  else zn_stats_count_impl<P, X, NT, false>(L, g, cs0, xcs0, tid, lane, tot, qc);
}

template <int P, bool X>
__global__ __launch_bounds__(ZN_E_THREADS) void zn_k_encode_stats(ZnESeg one, const ZnESeg* __restrict__ segs, uint32_t nseg,
                                                                  uint32_t* __restrict__ csize_all, uint8_t* __restrict__ type_all,
                                                                  ZnEncDesc* __restrict__ descs_all, uint32_t nchunks,
                                                                  uint8_t* __restrict__ planes_all, uint64_t slot) {
  __shared__ ZnStatsLds<P> L;
  if (blockIdx.x >= nchunks) {                   // a ragged plane (behind the full chunks)
    const uint32_t tb = (blockIdx.x - nchunks) / 4u, tq = (blockIdx.x - nchunks) % 4u;       // four workgroups per ragged plane: one per stream
    const ZnESeg S = zn_efind_ptail(one, segs, nseg, tb);
    static_assert(sizeof(L.hist) >= 256u * 32u * sizeof(uint32_t), "a ragged quarter's counters lie over the histogram columns");
    zn_encode_tail_stats<X>(&L.hist[0][0], S, tb - S.ptail0, tq, planes_all, slot, csize_all, type_all, descs_all);
    return;
  }
  const uint32_t sbid = blockIdx.x;
  const ZnESeg S = zn_efind_chunk(one, segs, nseg, sbid);
  const ZnGeom g = S.g;
  const uint8_t* __restrict__ src = ZN_GLOBAL_PTR(const uint8_t, S.src); const float threshold = S.threshold;
  uint32_t* __restrict__ csize_out = csize_all + S.pc0; uint8_t* __restrict__ type_out = type_all + S.pc0;
  ZnEncDesc* __restrict__ descs = descs_all + S.pc0;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t c = sbid - S.chunk0;
  const uint32_t n = (uint32_t)(g.chunk / P);                 // plane length of a full chunk (the host launches full, eligible chunks only)
  ZN_PT_DECL;

  uint32_t tot[P], qc[P][4];                                  // thread = bin: count so far / per quarter
  const uint8_t* cs0 = src + c * g.chunk;
  const uint8_t* xcs0 = (X && S.xr) ? ZN_GLOBAL_PTR(const uint8_t, S.xr) + c * g.chunk : nullptr;
  zn_stats_count<P, X, true>(L, g, cs0, xcs0, tid, lane, tot, qc);
  ZN_PT(0);   // zero + histograms

  // ---- per plane: largest count, highest symbol, and the cheap exits of HUF_compress ----
  // (the histogram is dead from here on — the last column sum ended with a barrier —: its first 8·P dwords hold the
  //  per-wave maxima, which keeps the structure at 32 KiB)
  uint32_t (*red_mx)[4] = (uint32_t (*)[4])&L.hist[0][0];
  uint32_t (*red_hi)[4] = red_mx + P;
  for (int p = 0; p < P; p++) {
    uint32_t mx = tot[p], hi = tot[p] ? tid : 0u;
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t m2 = __shfl_xor(mx, d), h2 = __shfl_xor(hi, d); if (m2 > mx) mx = m2; if (h2 > hi) hi = h2; }
    if (lane == 0) { red_mx[p][wave] = mx; red_hi[p][wave] = hi; }
  }
  __syncthreads();
  for (int p = 0; p < P; p++) {
    uint32_t mx = 0, hi = 0;
    for (int w = 0; w < 4; w++) { if (red_mx[p][w] > mx) mx = red_mx[p][w]; if (red_hi[p][w] > hi) hi = red_hi[p][w]; }
    const uint64_t pc = (uint64_t)p * g.K + c;
    if (mx == n) {                               // RLE: HUF_compress returns 1; threshold rule of compression_worker (zipnn_core.c:371-385)
      const bool keep = 1.0 < (double)n * (double)threshold;
      if (tid == 0) { type_out[pc] = keep ? 1 : 0; csize_out[pc] = keep ? 1u : n; if (keep) descs[pc].hdr[0] = (uint8_t)hi; }
    } else if (mx <= (n >> 7) + 4u) {            // "probably not compressible": stored raw
      if (tid == 0) { type_out[pc] = 0; csize_out[pc] = n; }
    } else {                                     // needs a code table: zn_k_encode_tables takes over
      ZnEncDesc* D = descs + pc;
      for (int q = 0; q < 4; q++) D->qcount[q][tid] = (uint16_t)qc[p][q];
      if (tid == 0) { type_out[pc] = 2; csize_out[pc] = n; }
    }
  }
  ZN_PT(1);   // decisions + counts out
  ZN_PT_COUNT(19, 1);
  ZN_PT_FLUSH();
}

// ---------------------------------------------------------------------------
// kernel B: code tables.  One wave per (plane, chunk) that kernel A marked: HUF_sort (rank sort over the
// lanes), then the serial part of HUF_compress on lane 0 — tree, length limiting, canonical values, tree
// description (zn_huf_tables.hpp) — then stream sizes from the per-quarter counts, the capacity and
// threshold rules, and the descriptor for the emit kernel.  Serial work per plane, but thousands of
// planes: every SIMD holds several of these waves.
// ---------------------------------------------------------------------------
struct ZnTablesLds {
  ZnTabScratch S;
  ZnHNode nodes[513];
  uint32_t hl;
};

// One table job by ONE WAVE: the per-quarter symbol counts of a plane (qcount[q][symbol]: global memory — the stats kernel's — or LDS — the
// one-pass encoder's) -> the plane's type and stored size and, for a plane that is kept, its code table, tree description and stream sizes in *D.
// WGSYNC: the wave is the whole workgroup (the table kernel: __syncthreads) — or one wave of a larger workgroup whose other waves wait at a later
// barrier (wave-level ordering only: the LDS executes a wave's operations in program order).  `ragged`: the cheap exits of HUF_compress are taken
// here (a full chunk's histogram pass has taken them already).  All 64 lanes call it together; type_o / cs_o are wave-uniform.
#define ZN_TJ_SYNC() do { if (WGSYNC) __syncthreads(); else __builtin_amdgcn_wave_barrier(); } while (0)
template <bool WGSYNC>
__device__ __forceinline__ void zn_table_job(ZnTablesLds& L, const uint16_t (*qcount)[256], uint32_t n, uint64_t cap, float threshold, bool legacy, bool ragged,
                                             ZnEncDesc* D, uint32_t lane, uint32_t& type_o, uint32_t& cs_o ZN_PT_PARAM) {
  ZN_PT_SHARED;
  // symbol counts (lane handles symbols lane + 64 k), highest symbol
  uint32_t qv[4][4], cnt[4], max_sv = 0;
  for (int k = 0; k < 4; k++) {
    cnt[k] = 0;
    for (int q = 0; q < 4; q++) { qv[q][k] = qcount[q][lane + 64u * (uint32_t)k]; cnt[k] += qv[q][k]; }
    const uint64_t m = __ballot(cnt[k] != 0);
    if (m) max_sv = 64u * (uint32_t)k + 63u - (uint32_t)__builtin_clzll(m);
  }
  if (ragged) {
    // a ragged plane's decisions — the cheap exits of HUF_compress — are taken here, where the four quarters' counts meet (a full chunk's
    // stats workgroup has taken them already)
    uint32_t mx = 0;
    for (int k = 0; k < 4; k++) mx = cnt[k] > mx ? cnt[k] : mx;
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t m2 = __shfl_xor(mx, d); if (m2 > mx) mx = m2; }
    if (mx == n) {                               // RLE (threshold rule of compression_worker, zipnn_core.c:371-385)
      const bool keep = 1.0 < (double)n * (double)threshold;
      if (lane == 0 && keep) D->hdr[0] = (uint8_t)max_sv;
      type_o = keep ? 1u : 0u; cs_o = keep ? 1u : n;
      return;
    }
    if (mx <= (n >> 7) + 4u) { type_o = 0u; cs_o = n; return; }
  }
  for (uint32_t i = lane; i < 513u; i += 64u) { ZnHNode z; z.count = 0; z.parent = 0; z.byte = 0; z.nb = 0; L.nodes[i] = z; }
  __builtin_amdgcn_wave_barrier();
  ZN_TJ_SYNC();
  // HUF_sort: a symbol's position is the number of symbols that sort before it (larger count, or equal
  // count and smaller symbol value).  Only symbols that occur can sort before one that occurs, so the
  // candidates are walked through the ballot masks (wave-uniform broadcast with v_readlane); symbols that
  // do not occur follow in symbol order.
  uint32_t nz_total = 0;                         // symbols that occur
  {
    uint64_t nzm[4];
    for (int k = 0; k < 4; k++) { nzm[k] = __ballot(cnt[k] != 0); nz_total += (uint32_t)__popcll(nzm[k]); }
    uint32_t rank[4] = {0, 0, 0, 0};
    // one key per symbol — count << 8 | 255 - symbol (a plane of a fused chunk has ≤ 2^17 bytes) —: "u sorts before sym" is key(u) > key(sym),
    // one compare and one add-with-carry per pair
    uint32_t key[4];
    for (int k = 0; k < 4; k++) key[k] = (cnt[k] << 8) | (255u - (lane + 64u * (uint32_t)k));
    for (int k2 = 0; k2 < 4; k2++) {
      uint64_t m = nzm[k2];
      while (m) {
        const uint32_t j = (uint32_t)__builtin_ctzll(m); m &= m - 1;
        const uint32_t ku = (uint32_t)__builtin_amdgcn_readlane((int)key[k2], (int)j);
        for (int k = 0; k < 4; k++) rank[k] += (ku > key[k]) ? 1u : 0u;
      }
    }
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64u - lane));
    uint32_t nz_before = 0;                      // symbols < sym that occur
    for (int k = 0; k < 4; k++) {
      const uint32_t sym = lane + 64u * (uint32_t)k;
      const uint32_t nzb = nz_before + (uint32_t)__popcll(nzm[k] & lt);
      const uint32_t r = cnt[k] ? rank[k] : nz_total + (sym - nzb);
      if (sym <= max_sv) { ZnHNode z; z.count = cnt[k]; z.parent = 0; z.byte = (uint8_t)sym; z.nb = 0; L.nodes[1u + r] = z; }
      nz_before += (uint32_t)__popcll(nzm[k]);
    }
  }
  ZN_TJ_SYNC();
  ZN_PT(2);   // counts + sort
  // the tree over the sorted leaves (serial on lane 0: the merge and the internal depths), code lengths and the first code value of
  // every length (lane-parallel): zn_wave_tree_from_sorted
  const uint32_t huff_log = zn_wave_tree_from_sorted(&L.S, L.nodes, (int)nz_total - 1, zn_optimal_table_log(ZN_HUF_LOG_DEFAULT, n, max_sv, 1), lane);
  ZN_TJ_SYNC();
  ZN_PT(5);   // tree + code lengths
  // parallel: code length of every symbol (sorted position → symbol), zero beyond the highest symbol
  for (int k = 0; k < 4; k++) {
    const uint32_t i = lane + 64u * (uint32_t)k;
    if (i <= max_sv) L.S.nbits[L.nodes[1u + i].byte] = L.nodes[1u + i].nb;
    else { L.S.nbits[i] = 0; L.S.vals[i] = 0; }
  }
  ZN_TJ_SYNC();
  // parallel: canonical values (symbols of one length in symbol order), huff0 weights and their histogram
  uint32_t wc[13];
  {
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64u - lane));
    uint32_t run[13];
    for (int v = 0; v < 13; v++) { run[v] = 0; wc[v] = 0; }
    for (int k = 0; k < 4; k++) {
      const uint32_t sym = lane + 64u * (uint32_t)k;
      const uint32_t nb = (sym <= max_sv) ? (uint32_t)L.S.nbits[sym] : 0xFFu;
      const uint32_t w = (sym < max_sv) ? (nb ? huff_log + 1u - nb : 0u) : 0xFFu;     // the last symbol's weight is implied
      if (sym < max_sv) L.S.weights[sym] = (uint8_t)w;
      uint32_t val = 0;
      for (uint32_t v = 0; v < 13u; v++) {
        const uint64_t m = __ballot(nb == v);
        val = (nb == v) ? (uint32_t)L.S.val_rank[v] + run[v] + (uint32_t)__popcll(m & lt) : val;
        run[v] += (uint32_t)__popcll(m);
        wc[v] += (uint32_t)__popcll(__ballot(w == v));
      }
      if (nb <= 12u) L.S.vals[sym] = (uint16_t)val;
    }
    for (uint32_t v = 0; v < 13u; v++) if (lane == v) L.S.wcount[v] = wc[v];
    if (lane == 13u) L.S.wcount[13] = 0;
  }
  ZN_TJ_SYNC();
  ZN_PT(9);   // values + weights
  // the tree description: the whole wave, its serial chain on wave-uniform values (zn_wave_write_ctable)
  uint32_t cs = 0, hdr_len = 0; bool go = false;
  {
    const int h = zn_wave_write_ctable(&L.S, max_sv, wc, lane, legacy ? -1 : 1);
    if (h < 0) cs = 0xFFFFFFFFu;               // huff0 error → fails the threshold test → raw
    else if ((uint32_t)h + 12u >= n) cs = 0;
    else if (cap - (uint32_t)h < 6u + 1u + 1u + 1u + 8u || n < 12u) cs = 0;      // (HUF_compress4X: a source of fewer than 12 bytes is not coded)
    else go = true;
    hdr_len = (uint32_t)(h > 0 ? h : 0);
  }
  ZN_TJ_SYNC();
  ZN_PT(6);   // tree description
  uint32_t sz[4] = {0, 0, 0, 0};
  if (go) {
    // stream q's bit count = Σ_symbols count_q[s] · len[s], + 1 for the end mark
    uint32_t bits[4];
    for (int q = 0; q < 4; q++) {
      uint32_t b = 0;
      for (int k = 0; k < 4; k++) b += qv[q][k] * (uint32_t)L.S.nbits[lane + 64u * (uint32_t)k];
      for (int d = 32; d >= 1; d >>= 1) b += __shfl_xor(b, d);
      bits[q] = b + 1u;
    }
    uint32_t pos = hdr_len + 6u; bool fail = false;
    for (int k = 0; k < 4; k++) {              // BIT_closeCStream's capacity rule, stream by stream
      const uint64_t cap_rem = cap - pos;
      if (cap_rem <= 8u || (uint64_t)(bits[k] >> 3) >= cap_rem - 8u) { fail = true; sz[k] = 0; break; }
      sz[k] = (bits[k] + 7u) >> 3; pos += sz[k];
    }
    cs = fail ? 0u : pos;
    if (!fail && pos >= n - 1u) cs = 0;
  }
  ZN_PT(8);   // stream sizes
  // threshold rule of compression_worker (zipnn_core.c:371-385)
  const bool keep = cs != 0 && (double)cs < (double)n * (double)threshold;
  if (keep) {
    for (int k = 0; k < 4; k++) { const uint32_t s = lane + 64u * (uint32_t)k; D->code[s] = (uint32_t)L.S.vals[s] | ((uint32_t)L.S.nbits[s] << 16); }
    for (uint32_t i = lane; i < 136u; i += 64u) D->hdr[i] = (i < hdr_len) ? L.S.hdr[i] : 0;
    if (lane == 0) { D->hdr_len = hdr_len; for (int k = 0; k < 4; k++) D->ssize[k] = sz[k]; }
  }
  type_o = keep ? 1u : 0u; cs_o = keep ? cs : n;
}
#undef ZN_TJ_SYNC

__global__ __launch_bounds__(64) void zn_k_encode_tables(ZnESeg one, const ZnESeg* __restrict__ segs, uint32_t nseg,
                                                         uint32_t* __restrict__ csize_all, uint8_t* __restrict__ type_all,
                                                         ZnEncDesc* __restrict__ descs_all, uint32_t njobs, uint32_t* __restrict__ status_zero) {
  __shared__ ZnTablesLds L;
  // (the call's status word — set by the emit kernels, behind this launch — starts at zero: one memset node less in front of every compress call)
  if (status_zero && blockIdx.x == 0 && threadIdx.x == 0) *status_zero = 0;
  const bool ragged = blockIdx.x >= njobs;       // the jobs of the ragged planes come behind those of the full chunks
  const ZnESeg S = ragged ? zn_efind_ptail(one, segs, nseg, blockIdx.x - njobs) : zn_efind_job(one, segs, nseg, blockIdx.x);
  const ZnGeom g = S.g; const uint64_t nfull = S.nfull; const float threshold = S.threshold;
  uint32_t* __restrict__ csize_out = csize_all + S.pc0; uint8_t* __restrict__ type_out = type_all + S.pc0;
  ZnEncDesc* __restrict__ descs = descs_all + S.pc0;
  const uint32_t lane = threadIdx.x;
  uint32_t p; uint64_t c;
  if (ragged) { const uint64_t KL = g.K - nfull, pcl = (uint64_t)(blockIdx.x - njobs) - S.ptail0; p = (uint32_t)(pcl / KL); c = nfull + pcl % KL; }
  else { const uint64_t job = blockIdx.x - S.job0; p = (uint32_t)(job / nfull); c = job % nfull; }
  const uint64_t pc = (uint64_t)p * g.K + c;
  if (!ragged && type_out[pc] != 2) return;
  const uint32_t n = ragged ? zn_plane_len(zn_chunk_len(g, c), g.P, p) : (uint32_t)(g.chunk / g.P);
  if (ragged && (n == 0u || n > ZN_HUF_BLOCK_MAX)) return;       // stored raw by its stats workgroups
  const uint64_t cap = g.chunk;                  // HUF_compress dstCapacity at the call site (zipnn_core.c:366-368)
  ZnEncDesc* D = descs + pc;
  ZN_PT_DECL;

  uint32_t ty = 0, cs = n;
  zn_table_job<true>(L, D->qcount, n, cap, threshold, S.legacy_weights != 0, ragged, D, lane, ty, cs ZN_PT_PASS);
  if (lane == 0) { type_out[pc] = (uint8_t)ty; csize_out[pc] = cs; }
  ZN_PT(3);   // sizes + descriptor
  ZN_PT_COUNT(18, 1);
  ZN_PT_FLUSH();
}

// ---------------------------------------------------------------------------
// kernel C: emit
// ---------------------------------------------------------------------------
template <int P>
struct ZnEmitLds {
  uint32_t code[256];
  uint32_t buf[4][ZN_E_BUF_DW + 4];
};

// H = plane to Huffman-encode in this pass (or -1: raw planes only); raw planes are written when `do_raw`.
// NT: non-temporal loads of the chunk (the one-pass encoder's second read: what the histogram pass left in the Infinity Cache is used once and should not
// be re-allocated; the emit kernel's plain loads are the measured better choice there — its four strided loads per tile share cache lines)
template <int P, bool X, bool NT = false>
__device__ __forceinline__ bool zn_emit_pass(const ZnGeom& g, const uint8_t* __restrict__ chunk_src, const uint8_t* __restrict__ chunk_xr, uint8_t* __restrict__ body,
                                             const uint64_t (&off)[P], const uint32_t (&kind)[P], int H, bool do_raw,
                                             const ZnEncDesc* D, const uint32_t* code, uint32_t* buf, uint32_t lane, uint32_t wave) {
  const uint32_t n = (uint32_t)(g.chunk / P), seg = n / 4u;
  const uint8_t* qsrc = chunk_src + (uint64_t)wave * (g.chunk / 4u);       // source bytes of this quarter
  uint8_t* sdst = nullptr; uint32_t ssize = 0;
  if (H >= 0) {
    uint32_t so = D->hdr_len + 6u;
    for (uint32_t k = 0; k < wave; k++) so += D->ssize[k];
    ssize = D->ssize[wave];
    uint64_t offH = 0;
    for (int p = 0; p < P; p++) if (p == H) offH = off[p];
    sdst = body + offH + so;
    for (uint32_t i = lane; i < ZN_E_BUF_DW + 4u; i += 64u) buf[i] = 0;
    __builtin_amdgcn_wave_barrier();
  }
  uint32_t carry = 0;          // bits already in buf[0] from the previous tile (< 32)
  uint32_t written = 0;        // stream bytes already stored
  // tiles from the END of the segment to its start: huff0 packs the last symbol first
  // this lane's 32 elements of a tile = 32·P source bytes.  ZN_E_SPLIT: TWO runs of 16 — elements [16 l, +16) of the tile's lower half (d[0 .. 4P)) and of its upper half
  // (d[4P .. 8P)) — instead of one run of 32: a raw plane's 32 bytes per lane then leave as two 16-byte stores that each write 1 KB contiguous across the wave
  // (whole 32-byte sectors); as one run, each non-temporal store wrote half of every sector (the emit kernel's 8 % write surplus: profiles/r04_decode_experiments.txt)
#define ZN_E_SPLIT 1
#define ZN_E_PREFETCH 1                   // the NEXT tile's source vectors are requested before this tile is packed (the emit kernel; the one-pass kernel has no registers for them)
  constexpr uint32_t RUN = ZN_E_SPLIT ? ZN_E_SPL / 2u : ZN_E_SPL;          // consecutive elements of a run
  constexpr uint32_t HALF = ZN_E_SPLIT ? ZN_E_TILE / 2u : 0u;              // element distance between the lane's two runs
#define ZN_E_PREFETCH_NT 1                // … in the one-pass kernel as well (its four-workgroups-per-CU build: 128 registers; 4 GiB bf16 2.223 -> 2.185 ms against five workgroups without)
  constexpr bool AHEAD = (ZN_E_PREFETCH != 0) && !X && (!NT || ZN_E_PREFETCH_NT != 0);
  auto load_tile = [&](uint32_t (&d)[8 * P], int32_t base) {
    const uint8_t* a = qsrc + (uint64_t)P * ((uint32_t)base + RUN * lane);
    for (int k = 0; k < 2 * P; k++) {
      const uint8_t* ak = ZN_E_SPLIT ? a + (k >= P ? (uint64_t)P * HALF + 16 * (k - P) : 16 * k) : a + 16 * k;
      const uint4 x = NT ? zn_ldnt128(ak) : ZN_LD_EMIT(ak); d[4 * k] = x.x; d[4 * k + 1] = x.y; d[4 * k + 2] = x.z; d[4 * k + 3] = x.w;
    }
    if (X && chunk_xr) {
      const uint8_t* xa = chunk_xr + (a - chunk_src);
      for (int k = 0; k < 2 * P; k++) {
        const uint8_t* xk = ZN_E_SPLIT ? xa + (k >= P ? (uint64_t)P * HALF + 16 * (k - P) : 16 * k) : xa + 16 * k;
        const uint4 x = NT ? zn_ldnt128(xk) : ZN_LD_EMIT(xk); d[4 * k] ^= x.x; d[4 * k + 1] ^= x.y; d[4 * k + 2] ^= x.z; d[4 * k + 3] ^= x.w;
      }
    }
  };
  uint32_t dnext[8 * P];
  if (AHEAD && (int32_t)seg - ZN_E_TILE >= 0) load_tile(dnext, (int32_t)seg - ZN_E_TILE);
  for (int32_t base = (int32_t)seg - ZN_E_TILE; base >= 0; base -= ZN_E_TILE) {
    uint32_t d[8 * P];
    if (AHEAD) { for (int k = 0; k < 8 * P; k++) d[k] = dnext[k]; if (base - ZN_E_TILE >= 0) load_tile(dnext, base - ZN_E_TILE); }
    else load_tile(d, base);
    // the lane's 32 elements plane by plane: pl[p][j] = byte p of elements 4 j .. 4 j + 3 (rotate at plane level, zn_split4)
    uint32_t pl[P][8];
    for (int j = 0; j < 8; j++) {
      uint32_t t4[P];
      zn_split4<P>(d + P * j, g.rot, t4);
      for (int p = 0; p < P; p++) pl[p][j] = t4[p];
    }
    // raw planes: 32 bytes per lane, contiguous across the wave
    if (do_raw) {
      for (int p = 0; p < P; p++) if (kind[p] == 0u) {
        const uint32_t* o = pl[p];
        uint8_t* r = body + off[p] + (uint64_t)wave * seg + (uint32_t)base + RUN * lane;
        uint8_t* r2 = ZN_E_SPLIT ? r + HALF : r + 16;                  // (the second run's 16 bytes / the second half of the one run)
#if !defined(ZN_SIMT_EMULATOR)                  // non-temporal: the payload is written once and not read back here
        typedef uint32_t zn_ev4u_u __attribute__((ext_vector_type(4), aligned(1)));
        __builtin_nontemporal_store((zn_ev4u_u){o[0], o[1], o[2], o[3]}, (zn_ev4u_u*)r);
        __builtin_nontemporal_store((zn_ev4u_u){o[4], o[5], o[6], o[7]}, (zn_ev4u_u*)r2);
#else
        zn_eu128u s0 = {o[0], o[1], o[2], o[3]}, s1 = {o[4], o[5], o[6], o[7]};
        *(zn_eu128u*)r = s0; *(zn_eu128u*)r2 = s1;
#endif
      }
    }
    if (H < 0) continue;
    // Codes of this lane's 32 symbols (val | len << 16), merged branch-free: pairs (≤ 22 bits), then quads
    // (≤ 44 bits, 64-bit).  The stream is written backwards, so inside a group the LATER symbol takes the
    // lower bits.
    uint64_t qv[8]; uint32_t qn[8];
    for (int i = 0; i < 8; i++) {
      uint32_t pv[2], pn[2];
      for (int h = 0; h < 2; h++) {
        uint32_t cw[2];
        for (int t = 0; t < 2; t++) {
          const int e = 4 * i + 2 * h + t;
          uint32_t sym = 0;
          for (int p = 0; p < P; p++) if (p == H) sym = (pl[p][e >> 2] >> (8 * (e & 3))) & 0xFFu;
          cw[t] = code[sym];
        }
        pv[h] = ((cw[0] & 0xFFFFu) << (cw[1] >> 16)) | (cw[1] & 0xFFFFu);
        pn[h] = (cw[0] >> 16) + (cw[1] >> 16);
      }
      qv[i] = ((uint64_t)pv[0] << pn[1]) | pv[1];
      qn[i] = pn[0] + pn[1];
    }
    // bit offset of every quad inside its run (the run's last quad lowest), and the run totals
    uint32_t qo[8]; uint32_t T = 0, TB = 0;
    if (ZN_E_SPLIT) { for (int i = 7; i >= 4; i--) { qo[i] = TB; TB += qn[i]; } for (int i = 3; i >= 0; i--) { qo[i] = T; T += qn[i]; } }
    else for (int i = 7; i >= 0; i--) { qo[i] = T; T += qn[i]; }
    // bit offset of this lane's run(s) in the tile: the stream is written backwards, so the tile's upper half comes first, and in a half the lanes are packed
    // from lane 63 down to lane 0
    uint32_t total = 0, totalB = 0, bB = 0;
    if (ZN_E_SPLIT) { const uint32_t exclB = zn_wave_excl_scan_u32(TB, &totalB); bB = carry + (totalB - exclB - TB); }
    const uint32_t excl = zn_wave_excl_scan_u32(T, &total);
    const uint32_t b = carry + totalB + (total - excl - T);
    total += totalB;
    for (int i = 0; i < 8; i++) {
      const uint32_t bp = ((ZN_E_SPLIT && i >= 4) ? bB : b) + qo[i], idx = bp >> 5, sh = bp & 31u;
      const uint64_t lo = qv[i] << sh;                                  // bits 0-63 of the shifted quad
      const uint32_t w2 = (uint32_t)(((qv[i] >> 32) << sh) >> 32);      // bits 64-75 (quad < 2^44)
      atomicOr(&buf[idx], (uint32_t)lo);
      atomicOr(&buf[idx + 1u], (uint32_t)(lo >> 32));
      atomicOr(&buf[idx + 2u], w2);
    }
    __builtin_amdgcn_wave_barrier();
    // flush whole dwords, keep the remainder (< 32 bits) at buf[0]
    const uint32_t bits = carry + total, nd = bits >> 5;
    if (written + 4u * nd > ssize) return false;
    uint32_t tail = 0;
    for (uint32_t i = lane; i <= nd; i += 64u) {
      const uint32_t x = buf[i];
      if (i < nd) { *(zn_eu32u*)(sdst + written + 4u * i) = x; }
      if (i == nd) tail = x;
      buf[i] = 0;
    }
    tail = __shfl(tail, (int)(nd & 63u));
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) buf[0] = tail;
    __builtin_amdgcn_wave_barrier();
    written += 4u * nd; carry = bits & 31u;
  }
  if (H >= 0) {
    // end mark + zero padding: the stream ends in a non-zero byte
    if (lane == 0) {
      const uint32_t x = buf[0] | (1u << carry);
      const uint32_t nbytes = (carry + 1u + 7u) >> 3;
      if (written + nbytes != ssize) { buf[1] = 0xDEAD; }
      else for (uint32_t k = 0; k < nbytes; k++) sdst[written + k] = (uint8_t)(x >> (8 * k));
    }
    __builtin_amdgcn_wave_barrier();
    if (buf[1] == 0xDEAD) return false;
  }
  return true;
}

// One huff0 stream of a RAGGED plane, by one wave: codes of src[n-1] .. src[0], end mark, zero pad, LSB-first into dst (nbytes = the
// stream size, known from the counts).  Tiles of 2048 symbols from the END of the segment — a lane packs 32 consecutive symbols, the
// segment's first tile is the partial one —, otherwise the scheme of zn_emit_pass: pairs, quads of ≤ 44 bits, a prefix sum of the bit
// counts places the lanes (lane 63 lowest), ds_or merges them in `buf` (ZN_E_BUF_DW + 4 dwords of LDS), whole dwords go out.
__device__ __forceinline__ bool zn_pack_stream_ragged(uint8_t* dst, uint32_t nbytes, const uint8_t* src, uint32_t n, const uint32_t* code, uint32_t* buf, uint32_t lane) {
  for (uint32_t i = lane; i < ZN_E_BUF_DW + 4u; i += 64u) buf[i] = 0;
  __builtin_amdgcn_wave_barrier();
  uint32_t carry = 0, written = 0;
  const uint32_t ntiles = (n + ZN_E_TILE - 1u) / ZN_E_TILE;
  for (uint32_t t = 0; t < ntiles; t++) {
    const int32_t first = (int32_t)n - (int32_t)ZN_E_TILE * (int32_t)(t + 1u) + (int32_t)(ZN_E_SPL * lane);      // this lane's symbols: first .. first + 31 (negative: before the segment)
    uint32_t cw[32];
    if (first >= 0) {
      const zn_eu128u a = *(const zn_eu128u*)(src + first), b2 = *(const zn_eu128u*)(src + first + 16);
      const uint32_t dd[8] = {a.x, a.y, a.z, a.w, b2.x, b2.y, b2.z, b2.w};
      for (int e = 0; e < 32; e++) cw[e] = code[(dd[e >> 2] >> (8 * (e & 3))) & 0xFFu];
    } else {
      for (int e = 0; e < 32; e++) { const int32_t i = first + e; cw[e] = (i >= 0) ? code[src[i]] : 0u; }
    }
    uint64_t qv[8]; uint32_t qn[8];
    for (int i = 0; i < 8; i++) {
      uint32_t pv[2], pn[2];
      for (int h = 0; h < 2; h++) {
        const uint32_t c0 = cw[4 * i + 2 * h], c1 = cw[4 * i + 2 * h + 1];
        pv[h] = ((c0 & 0xFFFFu) << (c1 >> 16)) | (c1 & 0xFFFFu);                 // the later symbol takes the lower bits
        pn[h] = (c0 >> 16) + (c1 >> 16);
      }
      qv[i] = ((uint64_t)pv[0] << pn[1]) | pv[1];
      qn[i] = pn[0] + pn[1];
    }
    uint32_t qo[8]; uint32_t T = 0;
    for (int i = 7; i >= 0; i--) { qo[i] = T; T += qn[i]; }
    uint32_t total = 0;
    const uint32_t excl = zn_wave_excl_scan_u32(T, &total);
    const uint32_t b = carry + (total - excl - T);
    for (int i = 0; i < 8; i++) {
      const uint32_t bp = b + qo[i], idx = bp >> 5, sh = bp & 31u;
      const uint64_t lo = qv[i] << sh;
      const uint32_t w2 = (uint32_t)(((qv[i] >> 32) << sh) >> 32);
      if ((uint32_t)lo) atomicOr(&buf[idx], (uint32_t)lo);
      if ((uint32_t)(lo >> 32)) atomicOr(&buf[idx + 1u], (uint32_t)(lo >> 32));
      if (w2) atomicOr(&buf[idx + 2u], w2);
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t bits = carry + total, nd = bits >> 5;
    if (written + 4u * nd > nbytes) return false;
    uint32_t tail = 0;
    for (uint32_t i = lane; i <= nd; i += 64u) {
      const uint32_t x = buf[i];
      if (i < nd) *(zn_eu32u*)(dst + written + 4u * i) = x;
      if (i == nd) tail = x;
      buf[i] = 0;
    }
    tail = __shfl(tail, (int)(nd & 63u));
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) buf[0] = tail;
    __builtin_amdgcn_wave_barrier();
    written += 4u * nd; carry = bits & 31u;
  }
  if (lane == 0) {
    const uint32_t x = buf[0] | (1u << carry);             // end mark; the stream ends in a non-zero byte
    const uint32_t nb = (carry + 1u + 7u) >> 3;
    if (written + nb != nbytes) buf[1] = 0xDEAD;
    else for (uint32_t k = 0; k < nb; k++) dst[written + k] = (uint8_t)(x >> (8 * k));
  }
  __builtin_amdgcn_wave_barrier();
  return buf[1] != 0xDEAD;
}

// A ragged plane's bytes to their place in the body: the scratch plane as it is (raw), its one byte (RLE), or the huff0 block —
// tree description, jump table, and wave w packs stream w.
__device__ __forceinline__ void zn_encode_tail_emit(uint32_t* code /* LDS [256] */, uint32_t (*buf)[ZN_E_BUF_DW + 4], const ZnESeg& S, uint64_t pcl,
                                    const uint8_t* __restrict__ planes_all, uint64_t slot, const uint32_t* __restrict__ csize_all,
                                    const uint8_t* __restrict__ type_all, const uint64_t* __restrict__ offs_all,
                                    const ZnEncDesc* __restrict__ descs_all, uint32_t* __restrict__ status) {
  const ZnGeom g = S.g; const uint64_t c0 = S.nfull, KL = g.K - c0;
  const uint32_t p = (uint32_t)(pcl / KL);
  const uint64_t c = c0 + pcl % KL, pc = (uint64_t)p * g.K + c;
  const uint32_t n = zn_plane_len(zn_chunk_len(g, c), g.P, p);
  const uint8_t* pl = planes_all + (S.slot0 + pcl) * slot;
  uint8_t* d = ZN_GLOBAL_PTR(uint8_t, S.body) + offs_all[S.pc0 + pc];
  const uint32_t ty = type_all[S.pc0 + pc], cs = csize_all[S.pc0 + pc];
  const ZnEncDesc* D = descs_all + S.pc0 + pc;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (!ty) {                                     // raw: 16 bytes per thread where source and destination allow it (the slot is 16-byte aligned)
    const uint32_t head = (uint32_t)((16u - ((uint64_t)d & 15u)) & 15u) < n ? (uint32_t)((16u - ((uint64_t)d & 15u)) & 15u) : n;
    for (uint32_t i = tid; i < head; i += ZN_E_THREADS) d[i] = pl[i];
    const uint32_t nv = (n - head) / 16u;
    for (uint32_t v0 = tid; v0 < nv; v0 += 4u * ZN_E_THREADS) {           // four vectors in flight per thread
      zn_eu128u x[4];
      for (int u = 0; u < 4; u++) { const uint32_t v = v0 + ZN_E_THREADS * (uint32_t)u; if (v < nv) x[u] = *(const zn_eu128u*)(pl + head + 16u * v); }
      for (int u = 0; u < 4; u++) { const uint32_t v = v0 + ZN_E_THREADS * (uint32_t)u; if (v < nv) *(uint4*)(d + head + 16u * v) = make_uint4(x[u].x, x[u].y, x[u].z, x[u].w); }
    }
    for (uint32_t i = head + 16u * nv + tid; i < n; i += ZN_E_THREADS) d[i] = pl[i];
    return;
  }
  if (cs == 1u) { if (tid == 0) d[0] = D->hdr[0]; return; }
  const uint32_t hl = D->hdr_len;
  for (uint32_t i = tid; i < hl; i += ZN_E_THREADS) d[i] = D->hdr[i];
  if (tid < 3u) { const uint32_t s = D->ssize[tid]; d[hl + 2u * tid] = (uint8_t)s; d[hl + 2u * tid + 1u] = (uint8_t)(s >> 8); }
  code[tid] = D->code[tid];
  __syncthreads();
  const uint32_t seg = (n + 3u) / 4u;
  uint32_t so = hl + 6u;
  for (uint32_t k = 0; k < wave; k++) so += D->ssize[k];
  const uint32_t len = (wave < 3u) ? seg : n - 3u * seg;
  if (!zn_pack_stream_ragged(d + so, D->ssize[wave], pl + wave * seg, len, code, buf[wave], lane)) { if (lane == 0) atomicOr(status, ZN_DEV_CORRUPT); }
}

template <int P, bool X>
__global__ __launch_bounds__(ZN_E_THREADS) void zn_k_encode_emit(ZnESeg one, const ZnESeg* __restrict__ segs, uint32_t nseg,
                                                                 const uint32_t* __restrict__ csize_all, const uint8_t* __restrict__ type_all,
                                                                 const uint64_t* __restrict__ offs_all, const ZnEncDesc* __restrict__ descs_all,
                                                                 uint32_t* __restrict__ status, uint32_t nchunks,
                                                                 const uint8_t* __restrict__ planes_all, uint64_t slot) {
  __shared__ ZnEmitLds<P> L;
  if (blockIdx.x >= nchunks) {                   // a ragged plane (behind the full chunks)
    const uint32_t tb = blockIdx.x - nchunks;
    const ZnESeg S = zn_efind_ptail(one, segs, nseg, tb);
    zn_encode_tail_emit(L.code, L.buf, S, tb - S.ptail0, planes_all, slot, csize_all, type_all, offs_all, descs_all, status);
    return;
  }
  const uint32_t bid = blockIdx.x;
  const ZnESeg S = zn_efind_chunk(one, segs, nseg, bid);
  const ZnGeom g = S.g;
  const uint8_t* __restrict__ src = ZN_GLOBAL_PTR(const uint8_t, S.src); uint8_t* __restrict__ body = ZN_GLOBAL_PTR(uint8_t, S.body);
  const uint32_t* __restrict__ csize = csize_all + S.pc0; const uint8_t* __restrict__ type = type_all + S.pc0;
  const uint64_t* __restrict__ offs = offs_all + S.pc0; const ZnEncDesc* __restrict__ descs = descs_all + S.pc0;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t c = bid - S.chunk0;
  const uint8_t* chunk_src = src + c * g.chunk;
  const uint8_t* chunk_xr = (X && S.xr) ? ZN_GLOBAL_PTR(const uint8_t, S.xr) + c * g.chunk : nullptr;
  uint64_t off[P]; uint32_t kind[P]; int nhuf = 0;           // kind: 0 raw, 1 RLE, 2 huff0
  for (int p = 0; p < P; p++) {
    const uint64_t pc = (uint64_t)p * g.K + c;
    off[p] = offs[pc];
    kind[p] = !type[pc] ? 0u : (csize[pc] == 1u ? 1u : 2u);
    nhuf += kind[p] == 2u;
  }
  for (int p = 0; p < P; p++) {
    const uint64_t pc = (uint64_t)p * g.K + c;
    if (kind[p] == 1u && tid == 0) body[off[p]] = descs[pc].hdr[0];
    if (kind[p] == 2u) {                                       // tree description + jump table
      const ZnEncDesc* D = descs + pc;
      const uint32_t hl = D->hdr_len;
      for (uint32_t i = tid; i < hl; i += ZN_E_THREADS) body[off[p] + i] = D->hdr[i];
      if (tid < 3u) { const uint32_t s = D->ssize[tid]; body[off[p] + hl + 2u * tid] = (uint8_t)s; body[off[p] + hl + 2u * tid + 1u] = (uint8_t)(s >> 8); }
    }
  }
  bool ok = true, raw_done = false;
  if (nhuf == 0) ok = zn_emit_pass<P, X>(g, chunk_src, chunk_xr, body, off, kind, -1, true, nullptr, L.code, L.buf[wave], lane, wave);
  else {
    for (int p = 0; p < P; p++) if (kind[p] == 2u) {
      const ZnEncDesc* D = descs + ((uint64_t)p * g.K + c);
      __syncthreads();
      L.code[tid] = D->code[tid];
      __syncthreads();
      ok = zn_emit_pass<P, X>(g, chunk_src, chunk_xr, body, off, kind, p, !raw_done, D, L.code, L.buf[wave], lane, wave) && ok;
      raw_done = true;
    }
  }
  if (!ok) atomicOr(status, ZN_DEV_CORRUPT);
}

// ---------------------------------------------------------------------------
// The ONE-PASS encoder (round 5): histogram, code table, placement and emit of a chunk by ONE workgroup — N + C of HBM traffic instead of 2 N + C.
//
// The plane-major wire format puts a chunk's bytes behind the stored sizes of ALL chunks of the earlier planes and of the earlier chunks of its own
// plane, which is why the four-kernel encoder above reads the tensor twice (sizes first, bytes second).  Two observations remove the second HBM read:
//   * what weights look like: every plane but the LAST is stored raw in every chunk (bf16 / fp16: the low byte; fp32: the three low bytes; fp8 has one
//     plane), so the base of plane p is K · planeLen · p and only the last plane needs a running sum — over the chunks BEFORE it, which a workgroup gets
//     by decoupled look-back: every workgroup publishes {generation | state | size} as ONE 64-bit word (a relaxed agent-scope store: flag and value travel
//     together, no fence, no hand-over of anything else), first its own size, then — once it has summed its predecessors' words — the inclusive prefix.
//     Chunks are handed out by a ticket counter, so a workgroup only ever waits for workgroups that started before it.  The speculation is CHECKED: a
//     chunk whose earlier planes are not raw sets ZN_DEV_MISSPEC (and the size scan compares the last plane's base with the speculated one); the caller
//     then runs the four-kernel encoder, which overwrites everything.
//   * the second read of the chunk, 40-100 us after the first, is served by the 256 MB Infinity Cache IF the first read allocates there (plain loads)
//     and the second does not (non-temporal loads): measured with the access pattern alone, scripts/ubench/mallbench.hip, profiles/r05_mall_reread.txt —
//     N read + N re-read + 0.66 N written in 1.58 ms at 4 workgroups per CU and 40 us between the reads, against 2.12 ms when the second read comes from HBM
//     (with twice the cache's size in flight — 8 workgroups per CU — the saving falls from 3/4 to 1/4 as the gap grows to 120 us).
// One workgroup = the stats kernel's histogram pass (zn_stats_count), the table kernel's job on wave 0 (zn_table_job), the look-back on wave 0, the emit
// kernel's pass (zn_emit_pass).  Types and stored sizes go to the same arrays; zn_k_scan_sizes writes the wire-format tables from them as before.
// Replaces compression_worker + prepare_python_return_buffer for full chunks (reference csrc/zipnn_core.c:294-390, :105-244).
// ---------------------------------------------------------------------------
#if defined(ZN_SIMT_EMULATOR)
#define ZN_LB_LOAD(p) (*(volatile const uint64_t*)(p))
#define ZN_LB_STORE(p, v) (*(volatile uint64_t*)(p) = (v))
#define ZN_LB_SLEEP() ((void)0)
#else
#define ZN_LB_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ZN_LB_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ZN_LB_SLEEP() __builtin_amdgcn_s_sleep(4)
#endif
#define ZN_LB_VBITS 40u                   // look-back word: generation (22 bits) | state (2 bits: 1 = own size, 2 = inclusive prefix) | value (40 bits)
#define ZN_LB_VMASK ((1ull << ZN_LB_VBITS) - 1ull)

#define ZN_OP_WGS 4                      // one-pass workgroups per CU: the histogram's 32 KiB of LDS allow five (96 registers a wave); the time is the same from two to five
                                         // (profiles/r05_encoder_onepass.txt), and four leave the registers for the emit pass's next tile (ZN_E_PREFETCH_NT)
#define ZN_OP_NT2 1                      // the second read non-temporal (0: plain — it then allocates in the caches like the first)
#define ZN_OP_NT1 0                      // (developer A/B: 1 = the first read non-temporal as well — nothing is left in the Infinity Cache for the second)
template <int P>
struct ZnOnePassLds {
  union {
    ZnStatsLds<P> st;                                                   // phase 1: the histogram columns (32 KiB) — and the ticket, in its first word, before that
    struct { ZnEncDesc D; uint32_t pad_; uint16_t qc[P][4][256];       // then: the last plane's descriptor, the planes' per-quarter counts,
             union { ZnTablesLds tab; ZnEmitLds<P> em; } u;             //       the table job's scratch / the emit pass's buffers,
             uint32_t kind[4], csz[4], rle[4], need[4];                 //       per plane: 0 raw / 1 RLE / 2 huff0, stored size, the RLE byte, "wants a code table"
             uint64_t excl; uint32_t excl_bad; } b;
  };
};

template <int P, bool X>
__global__ __launch_bounds__(ZN_E_THREADS, ZN_OP_WGS) void zn_k_encode_onepass(ZnESeg one, const ZnESeg* __restrict__ segs, uint32_t nseg,
                                                                    uint32_t* __restrict__ csize_all, uint8_t* __restrict__ type_all,
                                                                    uint64_t* __restrict__ lb_all, uint32_t* __restrict__ ticket,
                                                                    uint32_t* __restrict__ status, uint32_t gen) {
  __shared__ ZnOnePassLds<P> L;
  static_assert(sizeof(ZnOnePassLds<P>) * ZN_OP_WGS <= 160u * 1024u, "ZN_OP_WGS one-pass workgroups per CU");
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) L.st.hist[0][0] = atomicAdd(ticket, 1u);  // chunks in ticket order: whoever a workgroup waits for below has started before it
  __syncthreads();
  const uint32_t sbid = L.st.hist[0][0];
  __syncthreads();                                        // (the histogram pass zeroes that word)
  const ZnESeg S = zn_efind_chunk(one, segs, nseg, sbid);
  const ZnGeom g = S.g;
  const uint8_t* __restrict__ src = ZN_GLOBAL_PTR(const uint8_t, S.src); uint8_t* __restrict__ body = ZN_GLOBAL_PTR(uint8_t, S.body);
  const float threshold = S.threshold;
  uint32_t* __restrict__ csize_out = csize_all + S.pc0; uint8_t* __restrict__ type_out = type_all + S.pc0;
  const uint64_t c = sbid - S.chunk0;
  const uint32_t n = (uint32_t)(g.chunk / P);
  const uint8_t* chunk_src = src + c * g.chunk;
  const uint8_t* chunk_xr = (X && S.xr) ? ZN_GLOBAL_PTR(const uint8_t, S.xr) + c * g.chunk : nullptr;
  ZN_PT_DECL;
  ZN_PT(10);  // ticket + segment

  // ---- 1. histograms (plain loads: the lines stay in the Infinity Cache for step 4), 2. per plane: the cheap exits of HUF_compress ----
  {
    uint32_t tot[P], qc[P][4];
    zn_stats_count<P, X, ZN_OP_NT1 != 0>(L.st, g, chunk_src, chunk_xr, tid, lane, tot, qc);
    ZN_PT(11);  // histograms
    uint32_t (*red_mx)[4] = (uint32_t (*)[4])&L.st.hist[0][0];
    uint32_t (*red_hi)[4] = red_mx + P;
    for (int p = 0; p < P; p++) {
      uint32_t mx = tot[p], hi = tot[p] ? tid : 0u;
      for (int d = 32; d >= 1; d >>= 1) { const uint32_t m2 = __shfl_xor(mx, d), h2 = __shfl_xor(hi, d); if (m2 > mx) mx = m2; if (h2 > hi) hi = h2; }
      if (lane == 0) { red_mx[p][wave] = mx; red_hi[p][wave] = hi; }
    }
    __syncthreads();
    if (tid < (uint32_t)P) {
      const uint32_t p = tid;
      uint32_t mx = 0, hi = 0;
      for (int w = 0; w < 4; w++) { if (red_mx[p][w] > mx) mx = red_mx[p][w]; if (red_hi[p][w] > hi) hi = red_hi[p][w]; }
      uint32_t kind = 0, cs = n, need = 0;
      if (mx == n) { if (1.0 < (double)n * (double)threshold) { kind = 1; cs = 1u; } }            // RLE (threshold rule of compression_worker, zipnn_core.c:371-385)
      else if (mx > (n >> 7) + 4u) need = 1;                                                      // (else: "probably not compressible", stored raw)
      L.b.kind[p] = kind; L.b.csz[p] = cs; L.b.rle[p] = hi; L.b.need[p] = need;
    }
    __syncthreads();                             // (the maxima are read: the histogram's memory is free)
    for (int p = 0; p < P; p++) for (int q = 0; q < 4; q++) L.b.qc[p][q][tid] = (uint16_t)qc[p][q];
    __syncthreads();
  }
  ZN_PT(12);  // decisions + counts to LDS
  // ---- … or a code table: one job after the other on wave 0 (nearly always the last plane's alone) ----
  if (wave == 0) {
    bool failed = false;                         // the speculation: a plane in front of the last one is kept
#pragma nounroll
    for (int p = 0; p < P; p++) {
      if (p < P - 1 && L.b.kind[p]) failed = true;
      if (!L.b.need[p] || failed) continue;        // (after a failure the work is thrown away anyway)
      uint32_t ty = 0, cs = n;
      zn_table_job<false>(L.b.u.tab, L.b.qc[p], n, g.chunk, threshold, S.legacy_weights != 0, false, &L.b.D, lane, ty, cs ZN_PT_PASS);
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) { L.b.kind[p] = ty ? 2u : 0u; L.b.csz[p] = cs; }
      __builtin_amdgcn_wave_barrier();
      if (p < P - 1 && ty) failed = true;
    }
    if (lane < (uint32_t)P) { const uint64_t pc = (uint64_t)lane * g.K + c; type_out[pc] = L.b.kind[lane] ? 1 : 0; csize_out[pc] = L.b.csz[lane]; }
  }
  __syncthreads();
  ZN_PT(13);  // rest of the table job + barrier
  uint32_t kind[P];
  bool spec_ok = true;
  for (int p = 0; p < P; p++) { kind[p] = L.b.kind[p]; if (p < P - 1 && kind[p]) spec_ok = false; }
  const uint32_t csz_last = L.b.csz[P - 1];
  if (!spec_ok && tid == 0) atomicOr(status, ZN_DEV_MISSPEC);

  // ---- 3. where the last plane's bytes go: decoupled look-back over the chunks before this one (wave 0) ----
  // (published on the failed path as well: the workgroups behind this one wait for the word)
  if (wave == 0) {
    uint64_t* lb = lb_all + sbid;
    const uint64_t tag = (uint64_t)gen << (ZN_LB_VBITS + 2u);
    const uint64_t mine = csz_last;
    uint64_t excl = 0;
    bool timed_out = false;
    if (c == 0) { if (lane == 0) ZN_LB_STORE(lb, tag | (2ull << ZN_LB_VBITS) | mine); }
    else {
      if (lane == 0) ZN_LB_STORE(lb, tag | (1ull << ZN_LB_VBITS) | mine);
      int64_t j0 = (int64_t)c - 1;               // lane l looks at chunk j0 - l (of this tensor)
      // (bounded like zn_flag_wait: a predecessor that never publishes — a device fault, a kernel stopped by a debugger — must not hang the launch and the
      //  host behind it.  After two seconds the workgroup reports ZN_DEV_SYNC_TIMEOUT, publishes an inclusive word all the same — its successors stop
      //  waiting at once — and leaves; the host then runs the four-kernel encoder, which waits for nobody.  ADVICE r5)
      const unsigned long long t0 = ZN_FLAG_CLOCK();
      for (;;) {
        const int64_t j = j0 - (int64_t)lane;
        uint32_t fi;
        uint64_t v;
        for (;;) {
          v = (j >= 0) ? ZN_LB_LOAD(lb - (c - (uint64_t)j)) : (tag | (2ull << ZN_LB_VBITS));       // in front of the tensor's first chunk: an inclusive prefix of zero
          const bool ready = (v >> (ZN_LB_VBITS + 2u)) == (uint64_t)gen && ((v >> ZN_LB_VBITS) & 3ull) != 0ull;
          const bool incl = ready && ((v >> ZN_LB_VBITS) & 3ull) == 2ull;
          const uint64_t im = __ballot(incl), nm = __ballot(!ready);
          fi = im ? (uint32_t)__builtin_ctzll(im) : 64u;                    // the nearest predecessor with an inclusive prefix
          const uint64_t upto = fi >= 63u ? ~0ull : ((2ull << fi) - 1ull);     // lanes 0 .. fi
          if (!(nm & upto)) break;               // everything up to it is published
          if (ZN_FLAG_CLOCK() - t0 > 200000000ull) { timed_out = true; break; }      // (wave-uniform: one scalar clock)
          ZN_LB_SLEEP();
        }
        if (timed_out) break;
        excl += zn_wave_sum64_e((lane <= fi) ? (v & ZN_LB_VMASK) : 0ull);
        if (fi < 64u) break;
        j0 -= 64;
      }
      if (lane == 0) ZN_LB_STORE(lb, tag | (2ull << ZN_LB_VBITS) | ((excl + mine) & ZN_LB_VMASK));
      if (timed_out && lane == 0) atomicOr(status, ZN_DEV_SYNC_TIMEOUT);
    }
    if (lane == 0) { L.b.excl = excl; L.b.excl_bad = timed_out ? 1u : 0u; }
  }
  __syncthreads();
  ZN_PT(14);  // look-back
  if (!spec_ok || L.b.excl_bad) return;

  // ---- 4. emit: the chunk again (Infinity Cache), raw planes to their speculated places, the last plane behind its predecessors ----
  uint64_t off[P];
  {
    const uint64_t last_len = g.n - (g.K - 1u) * g.chunk;           // (K >= 1: this tensor has a full chunk)
    uint64_t base = 9ull * P * g.K;
    for (int p = 0; p < P; p++) {
      off[p] = base + ((p < P - 1) ? c * (uint64_t)n : L.b.excl);
      base += (g.K - 1u) * (uint64_t)n + zn_plane_len((uint32_t)last_len, (uint32_t)P, (uint32_t)p);
    }
  }
  const int H = (kind[P - 1] == 2u) ? P - 1 : -1;
  if (kind[P - 1] == 1u && tid == 0) body[off[P - 1]] = (uint8_t)L.b.rle[P - 1];
  if (H >= 0) {                                  // tree description + jump table, the code table for the pass
    const uint32_t hl = L.b.D.hdr_len;
    for (uint32_t i = tid; i < hl; i += ZN_E_THREADS) body[off[P - 1] + i] = L.b.D.hdr[i];
    if (tid < 3u) { const uint32_t s = L.b.D.ssize[tid]; body[off[P - 1] + hl + 2u * tid] = (uint8_t)s; body[off[P - 1] + hl + 2u * tid + 1u] = (uint8_t)(s >> 8); }
    L.b.u.em.code[tid] = L.b.D.code[tid];        // (the table job's scratch is dead: the barrier behind the look-back)
    __syncthreads();
  }
  const bool ok = zn_emit_pass<P, X, ZN_OP_NT2 != 0>(g, chunk_src, chunk_xr, body, off, kind, H, true, &L.b.D, L.b.u.em.code, L.b.u.em.buf[wave], lane, wave);
  if (!ok) atomicOr(status, ZN_DEV_CORRUPT);
  ZN_PT(15);  // emit
  ZN_PT_COUNT(17, 1);
  ZN_PT_FLUSH();
}

#ifdef ZN_PHASE_TIMERS
extern "C" int zn_debug_phase_read_enc(unsigned long long* out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(zn_phase_acc), sizeof(unsigned long long) * 64) != hipSuccess) return -2;
  if (reset) { unsigned long long z[64] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(zn_phase_acc), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#endif

// The fused encoder takes chunks [0, nfull): full chunks of a geometry zn_encode_fused_ok() accepted.
bool zn_encode_fused_ok(const ZnGeom& g, const void* d_src, const void* d_xr) {
  if ((((uint64_t)d_xr) & 15u) != 0) return false;
  const uint64_t n = g.chunk / g.P;
  // quarters are read 256 vectors of 16 bytes at a time; streams are packed in tiles of 2048 symbols
  return (g.chunk % 16384ull) == 0 && (g.chunk % (8192ull * g.P)) == 0 && n <= ZN_HUF_BLOCK_MAX && ((((uint64_t)d_src) & 15u) == 0);
}

bool zn_launch_encode_fused_stats(int P, const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_chunks, uint32_t total_jobs,
                                  uint32_t total_ptails, uint8_t* d_planes, uint64_t slot,
                                  uint32_t* d_csize, uint8_t* d_type, ZnEncDesc* d_descs, bool delta, uint32_t* d_status_zero, hipStream_t stream) {
  if (total_chunks + total_ptails == 0) return false;
  // (the ragged planes — total_ptails of them — are further workgroups of the same launches, behind the full chunks / their table jobs)
#define ZN_GO(P_, X_) hipLaunchKernelGGL((zn_k_encode_stats<P_, X_>), dim3(total_chunks + 4u * total_ptails), dim3(ZN_E_THREADS), 0, stream, one, d_segs, nseg, d_csize, d_type, d_descs, total_chunks, d_planes, slot)
  // (the table kernel is serial-latency bound — ≈65 µs per job, ≈5 600 jobs on the chip at once, 0.21 ms for the 16 384 chunks of
  //  4 GiB.  Building the tables of one slab of chunks on a second stream while the stats kernel reads the next was tried in round 2
  //  and removed in round 3: 2.431 ms per call without, 2.440 / 2.464 / 2.548 ms with 2 / 4 / 8 slabs — the cross-stream event
  //  hand-overs cost more than the table build they hide: profiles/r02_decode_experiments.txt)
  if (!delta) { if (P == 1) ZN_GO(1, false); else if (P == 2) ZN_GO(2, false); else ZN_GO(4, false); }
  else { if (P == 1) ZN_GO(1, true); else if (P == 2) ZN_GO(2, true); else ZN_GO(4, true); }
#undef ZN_GO
  zn_note_kernel(delta ? (total_ptails ? "zn_k_encode_stats^delta+tail" : "zn_k_encode_stats^delta") : (total_ptails ? "zn_k_encode_stats+tail" : "zn_k_encode_stats"));
  hipLaunchKernelGGL(zn_k_encode_tables, dim3(total_jobs + total_ptails), dim3(64), 0, stream, one, d_segs, nseg, d_csize, d_type, d_descs, total_jobs, d_status_zero);
  zn_note_kernel("zn_k_encode_tables");
  return true;                                   // (the table kernel has zeroed *d_status_zero)
}

void zn_launch_encode_fused_emit(int P, const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_chunks, uint32_t total_ptails,
                                 const uint8_t* d_planes, uint64_t slot,
                                 const uint32_t* d_csize, const uint8_t* d_type, const uint64_t* d_offs, const ZnEncDesc* d_descs,
                                 uint32_t* d_status, bool delta, hipStream_t stream) {
  if (total_chunks + total_ptails == 0) return;
#define ZN_GO(P_, X_) hipLaunchKernelGGL((zn_k_encode_emit<P_, X_>), dim3(total_chunks + total_ptails), dim3(ZN_E_THREADS), 0, stream, one, d_segs, nseg, d_csize, d_type, d_offs, d_descs, d_status, total_chunks, d_planes, slot)
  if (!delta) { if (P == 1) ZN_GO(1, false); else if (P == 2) ZN_GO(2, false); else ZN_GO(4, false); }
  else { if (P == 1) ZN_GO(1, true); else if (P == 2) ZN_GO(2, true); else ZN_GO(4, true); }
#undef ZN_GO
  zn_note_kernel(delta ? (total_ptails ? "zn_k_encode_emit^delta+tail" : "zn_k_encode_emit^delta") : (total_ptails ? "zn_k_encode_emit+tail" : "zn_k_encode_emit"));
}

void zn_launch_encode_onepass(int P, const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_chunks, uint32_t* d_csize, uint8_t* d_type,
                              uint64_t* d_lb, uint32_t* d_ticket, uint32_t* d_status, uint32_t gen, bool delta, hipStream_t stream) {
  if (total_chunks == 0) return;
#define ZN_GO(P_, X_) hipLaunchKernelGGL((zn_k_encode_onepass<P_, X_>), dim3(total_chunks), dim3(ZN_E_THREADS), 0, stream, one, d_segs, nseg, d_csize, d_type, d_lb, d_ticket, d_status, gen)
  if (!delta) { if (P == 1) ZN_GO(1, false); else if (P == 2) ZN_GO(2, false); else ZN_GO(4, false); }
  else { if (P == 1) ZN_GO(1, true); else if (P == 2) ZN_GO(2, true); else ZN_GO(4, true); }
#undef ZN_GO
  zn_note_kernel(delta ? "zn_k_encode_onepass^delta" : "zn_k_encode_onepass");
}
