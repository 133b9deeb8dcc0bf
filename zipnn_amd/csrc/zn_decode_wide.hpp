// zn_decode_wide.hpp — decompress of SMALL inputs: one workgroup of 16 (or 8) waves per chunk.  (Included by zn_decode_fused.hip.)
//
// The fused kernel gives a chunk four waves, one per huff0 stream, and a stream's ~11 tiles are decoded one after the other:
// ≈ 80 µs per chunk whatever the input size, which a tensor of a few hundred chunks (≤ 64 MiB of bf16: fewer chunks than the
// device has CUs) pays in full while most of the chip idles.  Here FOUR waves share a stream: in round r wave q takes tile
// 4 r + q.  Only the first tile of a stream knows where its first code starts; the other three guess — every sub-block of the
// fused kernel's tiles does that already (a run-in of 44 bits above the sub-block, Huffman self-synchronisation), here the
// tile's top sub-block does it too.  After the decode pass (zn_pass1, the same register-resident records) the waves publish
// (first start, last exit, symbols) in LDS; each checks the chain across the four tiles, a wave whose guess was wrong decodes
// its tile again from the true position (rare: the run-in fails for ≈ 0.5 % of the sub-blocks), prefix sums give every tile
// its place, the records are compacted into ONE circular staging buffer per stream (zn_pass2 with an address mask) and the
// stream's complete rows are interleaved with the raw planes and stored, a row per wave in turn.
//
// Takes the common shape only — full chunks, exactly one Huffman plane, a code of bf16 / fp32-exponent density (sub-blocks of 4
// dwords), no delta base.  Everything else is marked pending (done flag 2) and decoded by the fused kernel, which the host
// launches behind this one in its `only pending` mode.  Results are the same bytes either way; this is a latency form.
//
// Two sizes (template WPS = waves per stream).  Four: 16 waves, LDS 16 KB LUT + 4 × 16 KB staging + 16 × 1 KB stream tiles ≈ 99 KB, one workgroup
// per CU — for calls of at most one chunk per CU.  Two: 8 waves, 16 + 4 × 8 + 8 ≈ 58 KB, two workgroups per CU (one parses its tree description
// while the other decodes) — for calls of at most two chunks per CU, where four-wave workgroups of the fused kernel would still leave half the chip's
// wave slots empty for the ≈ 80 µs a chunk takes them.
#pragma once

#define ZN_W_D 4                         // sub-block dwords
#define ZN_W_TD (64 * ZN_W_D)
#define ZN_W_IN_DW (ZN_W_TD + 8)         // dwords lo_dw - 1 .. hi_dw + 1 of the tile (look-ahead below, run-in of the top sub-block above)
#define ZN_W_DELTA 44                    // run-in (bits)
#define ZN_W_MAXFIX 6                    // cross-wave fix-up iterations per round before the chunk is handed to the fused kernel

// WPS = waves per huff0 stream: 4 (a 16-wave workgroup, one per CU: calls of at most one chunk per CU) or 2 (8 waves, 58 KB of LDS: two
// workgroups per CU, one's tree description under the other's tiles: calls of at most two chunks per CU).
template <int WPS>
struct __attribute__((aligned(16))) ZnWideLds {
  static constexpr uint32_t RING = 4096u * WPS;      // staging bytes per stream (power of two): a round's WPS tiles (≤ ~3.5 KB each) + the carried remainder
  static constexpr uint32_t THREADS = 256u * WPS;
  uint2 lut[1u << ZN_F_TLMAX];
  uint32_t ring[4][RING / 4 + 4];        // per stream, circular; dword RING / 4 = the second half of a pair that starts in the last dword
  uint32_t in[4 * WPS][ZN_W_IN_DW];      // per wave
  uint8_t symlist[1][256];
  uint32_t rank_start[1][14], sym_start[1][14];
  ZnFusedPlane plane[4];
  ZnWaveStats st;
  uint32_t what;
  int32_t x_start[4 * WPS], x_exit[4 * WPS];   // per wave and round: where its tile's decode started / ended (bit positions)
  uint32_t x_n[4 * WPS], x_flag[4 * WPS];      // symbols of the tile; 0 = no tile this round, 1 = decoded, 2 = failed
  uint32_t again[ZN_W_MAXFIX + 2];       // [i]: some wave decoded again in fix-up iteration i
  uint32_t rounds, fail;
};
// ONE stream by four waves (round 6): the layout of a TAIL workgroup — a partial last chunk's Huffman plane gets four of them, one per huff0 stream, at the front of
// the fused launch —, laid over the fused kernel's own LDS allocation (its look-up table sits at the same offset; what follows it is dead once that table is filled)
struct __attribute__((aligned(16))) ZnTailWideLds {
  static constexpr uint32_t RING = 16384u;           // the stream's circular staging buffer: a round's four tiles (≤ ~3.5 KB each at this density) + the carried remainder
  static constexpr uint32_t THREADS = 256u;
  uint2 lut[1u << ZN_F_TLMAX];
  uint32_t ring[1][RING / 4 + 4];
  uint32_t in[4][ZN_W_IN_DW];
  int32_t x_start[4], x_exit[4];
  uint32_t x_n[4], x_flag[4];
  uint32_t again[ZN_W_MAXFIX + 2];
  uint32_t rounds, fail;
};
static_assert(sizeof(ZnTailWideLds) <= sizeof(ZnFusedLds), "the one-stream wide form lives in the fused kernel's LDS allocation");
static_assert(sizeof(ZnWideLds<4>) <= 160u * 1024u, "ZnWideLds<4>: one workgroup per CU");
static_assert(sizeof(ZnWideLds<2>) <= 80u * 1024u, "ZnWideLds<2>: two workgroups per CU");

// One chunk, all 16 waves.  Returns (workgroup-uniform) whether the chunk was decoded; false = nothing usable was produced.
// ONE (LDS = ZnTailWideLds, four waves): the workgroup decodes huff0 stream `s_one` alone — `seg` symbols, to outc (a padded scratch: the stream's last,
// incomplete row is stored whole) — instead of the chunk's four streams side by side; one plane, no raw planes.
template <int P, int H, int WPS, typename LDS = ZnWideLds<WPS>, bool ONE = false>
__device__ __forceinline__ bool zn_wide_chunk(LDS& L, const ZnGeom& g, const uint8_t* __restrict__ body, const uint8_t* body_end,
                                              uint8_t* outc, const ZnFusedPlane (&pl)[P], uint32_t seg, uint32_t TL,
                                              const uint8_t* js, uint32_t l1, uint32_t l2, uint32_t l3, uint32_t l4, uint32_t s_one = 0) {
  static_assert(!ONE || (P == 1 && H == 0 && WPS == 4), "the one-stream form: a single Huffman plane by four waves");
  constexpr int EPL = (P == 1) ? 16 : 8, EW = EPL / 4;
  constexpr uint32_t UNIT = 64u * EPL;
  constexpr bool SPLIT = (P == 4) && (ZN_F_SPLIT4 != 0);      // four planes: a lane owns two runs of 4 symbols of a row (whole-sector stores; see zn_fused_wave)
  constexpr int RB = (P == 2) ? 8 : 4;      // rows of one flush batch: a wave's share of a round (the staging buffer holds ≤ 31 rows of 512 symbols, ≤ 15 of 1024) in one batch (two for 4 planes)
  constexpr int TF = ZN_F_TF(ZN_W_D), TB = 3, UF = (32 * ZN_W_D - 31) / 11;
  constexpr int32_t TD = ZN_W_TD;
  constexpr uint32_t ZN_W_RING = LDS::RING, ZN_W_THREADS = LDS::THREADS;
  const uint32_t tid = threadIdx.x, wave = zn_uniform(tid >> 6), s_id = wave / (uint32_t)WPS, q = wave % (uint32_t)WPS;      // (ONE: four waves, s_id = 0: LDS indices only)
  const uint32_t ss = ONE ? zn_uniform(s_one) : s_id;                  // which huff0 stream of the block
  uint32_t lane = zn_lane_id();
  uint32_t lane_v = lane; ZN_OPAQUE32(lane_v);
  const uint32_t so = 6u + (ss > 0 ? l1 : 0u) + (ss > 1 ? l2 : 0u) + (ss > 2 ? l3 : 0u);
  const uint8_t* const stream = zn_uniform_ptr(js + so);
  const uint32_t slen = (ss == 0) ? l1 : (ss == 1) ? l2 : (ss == 2) ? l3 : l4;
  uint8_t* const outq = zn_uniform_ptr(ONE ? outc : outc + (uint64_t)s_id * (g.chunk / 4u));
  const uint8_t* rawq[P];
  for (int p = 0; p < P; p++) rawq[p] = zn_uniform_ptr(body + pl[p].off + (uint64_t)s_id * seg);
  uint32_t* const ring = L.ring[s_id]; uint32_t* const in = L.in[wave];
  const uint2* const lut = L.lut;

  // (staging buffers 1-3 were zeroed while the tree description was parsed; buffer 0 held the parser's scratch and the 16-bit LUT)
  for (uint32_t i = tid; i < ZN_W_RING / 4u + 4u; i += ZN_W_THREADS) L.ring[0][i] = 0;
  if (tid == 0) { L.rounds = 0; L.fail = 0; }
  if (tid < ZN_W_MAXFIX + 2) L.again[tid] = 0;
  __syncthreads();

  const uint8_t last = stream[slen - 1];
  bool sok = last != 0;                                 // (stream-uniform: the same in the stream's four waves)
  const uint32_t mis = (uint32_t)((uint64_t)stream & 3u);
  const uint32_t* gdw = (const uint32_t*)(stream - mis);
  const int32_t b0 = (int32_t)(8u * mis);
  const int32_t carry0 = __builtin_amdgcn_readfirstlane(b0 + (int32_t)(8u * (slen - 1u)) + (int32_t)zn_hb32(last | 1u));
  const int32_t hi_dw0 = (carry0 + 31) >> 5, top_dw = hi_dw0 - 1;
  const bool top_guard = ((const uint8_t*)(gdw + hi_dw0) > body_end);
  {
    const uint32_t ntiles = (uint32_t)((32 * hi_dw0 - b0 + 32 * TD - 1) / (32 * TD));
    if (lane == 0 && q == 0) atomicMax(&L.rounds, (ntiles + (uint32_t)WPS - 1u) / (uint32_t)WPS);
  }
  // tile k of the stream = dwords [hi_dw0 - (k + 1) TD, hi_dw0 - k TD); its dwords -1 .. TD + 2 (relative to lo_dw) go to `in`
  uint32_t nx[5];
  auto fetch_tile = [&](int32_t hi_dw_) {
    const int32_t lo_dw_ = hi_dw_ - TD;
    for (int i = 0; i < 5; i++) {
      const int32_t li = (int32_t)lane + 64 * i, gi = lo_dw_ - 1 + li;
      uint32_t x = 0;
      if (li < TD + 3 && gi >= -1 && gi < hi_dw0) {
        if (top_guard && gi == top_dw) { const uint8_t* pa = (const uint8_t*)(gdw + gi); for (int b = 0; b < 4; b++) if (pa + b < body_end) x |= (uint32_t)pa[b] << (8 * b); }
        else x = ZN_LD_STREAM32(gdw + gi);
      }
      nx[i] = x;
    }
  };
  auto stage_tile = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < 5; i++) { const uint32_t li = lane + 64u * (uint32_t)i; if (li < (uint32_t)TD + 3u) in[li] = nx[i]; }
    __builtin_amdgcn_wave_barrier();
  };
  auto tile_hi = [&](uint32_t r) -> int32_t { return hi_dw0 - (int32_t)((uint32_t)WPS * r + q) * TD; };
  if (sok && 32 * tile_hi(0) > b0) { fetch_tile(tile_hi(0)); stage_tile(); }
  __syncthreads();
  const uint32_t rounds = zn_uniform(L.rounds);

  uint32_t J = 0, JF = 0;                                // symbols in the staging buffer / flushed (stream-uniform)
  int32_t carry = carry0;                                // true position at the top of the round's first tile

  for (uint32_t r = 0; r < rounds; r++) {
    lane = zn_lane_id(); lane_v = lane; ZN_OPAQUE32(lane_v);
    J = zn_uniform(J); JF = zn_uniform(JF); carry = __builtin_amdgcn_readfirstlane(carry);
    const int32_t hi_dw = tile_hi(r), lo_dw = hi_dw - TD;
    const bool exists = sok && 32 * hi_dw > b0;
    const bool have_next = sok && 32 * tile_hi(r + 1) > b0;
    if (have_next) fetch_tile(tile_hi(r + 1));
    const int32_t base_bit = 32 * (lo_dw - 1);
    const int32_t hi_k = 32 * (hi_dw - (int32_t)lane * ZN_W_D), lo_k = hi_k - 32 * ZN_W_D;
    const int32_t stop = lo_k > b0 ? lo_k : b0;
    const bool active = exists && hi_k > b0;
    const bool regular = (32 * lo_dw >= b0);
    const bool first_tile = (r == 0 && q == 0);

    ZnRec rec;
    int nfull = 0, nbnd = 0;
    uint32_t n = 0, N = 0, o_k = 0;
    int32_t s = hi_k, e = hi_k;
    // decode the tile from the starts in `s`; closes the chain INSIDE the tile (fix-up passes); false = not this form's tile
    auto decode_tile = [&]() -> bool {
      uint32_t acc = 0;
      for (int it = 0; it < 5; it++) {
        bool took;
        if (regular) took = zn_pass1<TF, TB, UF, false>(lut, in, base_bit, TL, s, stop, true, rec, acc, nfull, nbnd);
        else took = zn_pass1<TF, TB, 0, false>(lut, in, base_bit, TL, s, stop, active, rec, acc, nfull, nbnd);
        if (!took) return false;
        e = s - (int32_t)(acc & 0xFFu); n = active ? ((acc >> 8) & 0xFFu) : 0u;
        const int32_t e_prev = __shfl_up(e, 1u);
        const bool mism = active && lane > 0 && e_prev != s;
        if (__builtin_expect(!__any(mism), 1)) { o_k = zn_wave_excl_scan(n, lane, &N); return true; }
        if (mism) s = e_prev;
      }
      return false;
    };
    auto publish = [&](bool good) {
      const uint32_t nact = (uint32_t)__popcll(__ballot(active));
      const int32_t e_last = __builtin_amdgcn_readlane(e, (int)(nact ? nact - 1u : 0u));
      const int32_t s_first = __builtin_amdgcn_readfirstlane(s);
      if (lane == 0) { L.x_start[wave] = s_first; L.x_exit[wave] = e_last; L.x_n[wave] = N; L.x_flag[wave] = good ? 1u : 2u; }
    };
    if (exists) {
      ZN_PRIO(ZN_F_PRIO_SYNC);
      // run-in: every sub-block but the stream's very first guesses a start ZN_W_DELTA bits above itself
      {
        ZnChain c;
        const bool guess = active && !(first_tile && lane == 0);
        c.pos = guess ? hi_k + ZN_W_DELTA : hi_k; c.stop = hi_k; c.n = 0; c.wpos = 0;
        const uint32_t sh = 32u - TL;
        uint64_t w = zn_window(in, c.pos - 1 - base_bit);
        auto group = [&]() { uint2 en = lut[(uint32_t)(w >> 32) >> sh]; if (!(c.pos > c.stop + (int32_t)TL - 1)) { en.x = 0; en.y = 0; } w <<= (en.y & 63u); c.pos -= (int32_t)ZN_M_NB(en.y); };
        group(); group(); group();
        w = zn_window(in, c.pos - 1 - base_bit);
        group();
        { uint2 en = lut[(uint32_t)(w >> 32) >> sh]; if (!(c.pos > c.stop)) { en.x = 0; en.y = 0; } c.pos -= (int32_t)ZN_M_NB(zn_trim_group(en, c.pos - c.stop).y); }
        while (__any(c.pos > c.stop)) {
          w = zn_window(in, c.pos - 1 - base_bit);
          uint2 en = lut[(uint32_t)(w >> 32) >> sh]; if (!(c.pos > c.stop)) { en.x = 0; en.y = 0; }
          c.pos -= (int32_t)ZN_M_NB(zn_trim_group(en, c.pos - c.stop).y);
        }
        s = (first_tile && lane == 0) ? carry0 : c.pos;
      }
      ZN_PRIO(ZN_F_PRIO_COUNT);
      const bool good = decode_tile();
      ZN_PRIO(0);
      publish(good);
    } else if (lane == 0) L.x_flag[wave] = 0;
    __syncthreads();

    // ---- the chain across the stream's four tiles: the first wave whose start is not its predecessor's exit decodes again ----
    int32_t expect = carry; bool chained = false;
    for (int it = 0; it <= ZN_W_MAXFIX; it++) {
      expect = carry; int bad = -1; bool broken = false;
      for (uint32_t t = 0; t < (uint32_t)WPS; t++) {
        const uint32_t f = zn_uniform(L.x_flag[(uint32_t)WPS * s_id + t]);
        if (f == 0u) break;
        if (f == 2u) { broken = true; break; }
        if ((int32_t)zn_uniform((uint32_t)L.x_start[(uint32_t)WPS * s_id + t]) != expect) { bad = (int)t; break; }
        expect = (int32_t)zn_uniform((uint32_t)L.x_exit[(uint32_t)WPS * s_id + t]);
      }
      if (broken) sok = false;
      const bool redo = sok && bad == (int)q && it < ZN_W_MAXFIX;
      if (sok && bad >= 0 && it == ZN_W_MAXFIX) sok = false;
      __syncthreads();                               // every wave has read this iteration's values
      if (redo) {
        ZN_NO_IFCVT;
        ZN_DBG_COUNT(6);                               // (emulated build: tiles decoded again because the guess at their top was wrong)
        if (lane == 0) s = expect;
        ZN_PRIO(ZN_F_PRIO_COUNT);
        const bool good = decode_tile();
        ZN_PRIO(0);
        publish(good);
        if (lane == 0) L.again[it] = 1u;
      }
      __syncthreads();
      if (zn_uniform(L.again[it]) == 0u) { chained = sok && bad < 0; break; }
    }
    if (sok && !chained) sok = false;
    // (expect = the exit of the round's last tile)
    uint32_t off = 0, Nr = 0;
    for (uint32_t t = 0; t < (uint32_t)WPS; t++) { const uint32_t f = zn_uniform(L.x_flag[(uint32_t)WPS * s_id + t]); const uint32_t nt = (f == 1u) ? zn_uniform(L.x_n[(uint32_t)WPS * s_id + t]) : 0u; if (t < q) off += nt; Nr += nt; }
    if (sok && (J + Nr > seg || (J - JF) + Nr > ZN_W_RING - 8u)) sok = false;
    if (!sok) { if (lane == 0) L.fail = 1u; Nr = 0; }

    // ---- the stream's complete rows after this round: wave q flushes rows q, q + 4, …; their raw-plane bytes are requested
    // now and used after the compaction ----
    const uint32_t rows_total = sok ? (J + Nr - JF) / UNIT : 0u;
    uint32_t pre[RB][P][EW];
    auto request_rows = [&](uint32_t m0) {
      for (int rr = 0; rr < RB; rr++) {
        const uint32_t i = m0 + (uint32_t)WPS * (uint32_t)rr;
        if (i < rows_total) {
          const uint32_t a = JF + i * UNIT;
          for (int p = 0; p < P; p++) if (p != H && pl[p].kind == ZN_KIND_RAW) {
            const uint8_t* au = rawq[p] + a;
            if constexpr (SPLIT) { const uint8_t* ap = au + 4u * lane_v; pre[rr][p][0] = ZN_LD_RAW32(ap); pre[rr][p][1 % EW] = ZN_LD_RAW32(ap + UNIT / 2u); }
            else { const uint8_t* ap = au + (uint32_t)EPL * lane_v; for (int k = 0; k < EW / 2; k++) { const uint64_t t = ZN_LD_RAW64(ap + 8 * k); pre[rr][p][2 * k] = (uint32_t)t; pre[rr][p][2 * k + 1] = (uint32_t)(t >> 32); } }
          }
        }
      }
    };
    request_rows(q);

    // ---- compaction into the stream's circular staging buffer ----
    if (exists && sok) {
      ZN_PRIO(ZN_F_PRIO_WRITE);
      const int nf = __builtin_amdgcn_readfirstlane(nfull), nb_ = __builtin_amdgcn_readfirstlane(nbnd);
      zn_pass2<TF, TB, false, ZN_W_RING - 4u>(ring, J + off + o_k, rec, nf, nb_, [](auto) {});
      ZN_PRIO(0);
    }
    if (tid < ZN_W_MAXFIX + 2) L.again[tid] = 0;      // (read for the last time before the barrier above)
    __syncthreads();

    // ---- flush; the next tile is staged behind the first wait, ahead of the stores ----
    for (uint32_t m0 = q; ; m0 += (uint32_t)WPS * RB) {
      if (m0 != q) request_rows(m0);
      __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
      if (m0 == q && have_next) stage_tile();
      for (int rr = 0; rr < RB; rr++) {
        const uint32_t i = m0 + (uint32_t)WPS * (uint32_t)rr;
        if (i < rows_total) {
          const uint32_t a = JF + i * UNIT, ra = a & (ZN_W_RING - 1u);
          for (int p = 0; p < P; p++) {
            if (p == H) {
              if constexpr (SPLIT) { const uint32_t ix = (ra >> 2) + lane; for (int k = 0; k < EW; k++) { pre[rr][p][k] = ring[ix + (UNIT / 8u) * k]; ring[ix + (UNIT / 8u) * k] = 0; } }
              else { const uint32_t ix = (ra + (uint32_t)EPL * lane) >> 2; for (int k = 0; k < EW; k++) { pre[rr][p][k] = ring[ix + k]; ring[ix + k] = 0; } }
              if (ra == 0u && lane == 0) { pre[rr][p][0] |= ring[ZN_W_RING / 4u]; ring[ZN_W_RING / 4u] = 0; }
            } else if (pl[p].kind == ZN_KIND_RLE) { for (int k = 0; k < EW; k++) pre[rr][p][k] = ((uint32_t)pl[p].off & 0xFFu) * 0x01010101u; }
          }
          if (P >= 2 && g.rot) {
            for (int k = 0; k < EW; k++) {
              const uint32_t hi = pre[rr][P - 1][k], lo = pre[rr][(P >= 2) ? P - 2 : 0][k];
              pre[rr][P - 1][k] = ZN_BFI(0x80808080u, lo, hi >> 1);
              pre[rr][(P >= 2) ? P - 2 : 0][k] = ZN_BFI(0x7F7F7F7Fu, lo, hi << 7);
            }
          }
          uint8_t* o = (outq + (uint64_t)a * P) + (SPLIT ? 16u : (uint32_t)EPL * (uint32_t)P) * lane_v;
          if (P == 1) {
            ZN_ST128(o, pre[rr][0][0], pre[rr][0][1 % EW], pre[rr][0][2 % EW], pre[rr][0][3 % EW]);
          } else if (P == 2) {
            const uint32_t x0 = __builtin_amdgcn_perm(pre[rr][1 % P][0], pre[rr][0][0], 0x05010400u), x1 = __builtin_amdgcn_perm(pre[rr][1 % P][0], pre[rr][0][0], 0x07030602u);
            const uint32_t x2 = __builtin_amdgcn_perm(pre[rr][1 % P][1 % EW], pre[rr][0][1 % EW], 0x05010400u), x3 = __builtin_amdgcn_perm(pre[rr][1 % P][1 % EW], pre[rr][0][1 % EW], 0x07030602u);
            ZN_ST128(o, x0, x1, x2, x3);
          } else {
            for (int half = 0; half < 2; half++) {
              const int k = half % EW;
              const uint32_t ab_lo = __builtin_amdgcn_perm(pre[rr][1 % P][k], pre[rr][0][k], 0x05010400u), ab_hi = __builtin_amdgcn_perm(pre[rr][1 % P][k], pre[rr][0][k], 0x07030602u);
              const uint32_t cd_lo = __builtin_amdgcn_perm(pre[rr][3 % P][k], pre[rr][2 % P][k], 0x05010400u), cd_hi = __builtin_amdgcn_perm(pre[rr][3 % P][k], pre[rr][2 % P][k], 0x07030602u);
              const uint32_t y0 = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u), y1 = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
              const uint32_t y2 = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u), y3 = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
              if constexpr (SPLIT) ZN_ST128_4(o + (UNIT / 2u) * P * half, y0, y1, y2, y3);
              else *(uint4*)(o + 16 * half) = make_uint4(y0, y1, y2, y3);
            }
          }
        }
      }
      if (m0 + (uint32_t)WPS * RB >= rows_total) break;
    }
    if (sok) { J += Nr; JF += rows_total * UNIT; carry = expect; }
  }
  if constexpr (ONE) {
    // the stream's last row is incomplete (a partial chunk's streams are not whole rows): stored whole by wave 0 — the destination is padded, the ring beyond the last
    // symbol holds zeros
    if (sok && carry == b0 && J == seg && JF < seg && J - JF < UNIT) {
      if (q == 0) {
        const uint32_t ra = JF & (ZN_W_RING - 1u);
        const uint32_t ix = (ra + (uint32_t)EPL * lane) >> 2;
        uint32_t t[EW];
        for (int k = 0; k < EW; k++) t[k] = ring[ix + k];
        if (ra == 0u && lane == 0) t[0] |= ring[ZN_W_RING / 4u];
        uint8_t* o = (outq + (uint64_t)JF * P) + (uint32_t)EPL * (uint32_t)P * lane_v;
        ZN_ST128(o, t[0], t[1 % EW], t[2 % EW], t[3 % EW]);
      }
      JF = seg;
    }
  }
  if (!(sok && carry == b0 && J == seg && JF == seg)) { if (lane == 0) L.fail = 1u; }
  __syncthreads();
  return zn_uniform(L.fail) == 0u;
}

// zero_status: this is the first launch of the call (and it has no tail workgroups, which report through the status word).
// grid: one workgroup per full chunk of the launch (segment table with ncg == 1).  done flag: 1 = decoded here, 2 = pending
// (the fused kernel, launched behind this one with only_pending set, takes it).
template <int P, int WPS>
__global__ __launch_bounds__(256 * WPS, 4) void zn_k_decode_wide(ZnSeg one, const ZnSeg* __restrict__ segs, uint32_t nseg,
                                                                    uint8_t* __restrict__ done_all, uint8_t* __restrict__ pdone_all,
                                                                    uint32_t* __restrict__ status, uint32_t zero_status, uint32_t ntail,
                                                                    uint8_t* __restrict__ tail_scratch, uint8_t* __restrict__ tail_done,
                                                                    uint32_t* __restrict__ tailsync, ZnPlaneDesc* __restrict__ descs_rest, uint32_t nchunk_wg, uint32_t merge_per) {
  constexpr int EPL = (P == 1) ? 16 : 8;
  constexpr uint32_t UNIT = 64u * EPL;
  __shared__ ZnWideLds<WPS> L;
  constexpr uint32_t ZN_W_RING = ZnWideLds<WPS>::RING, ZN_W_THREADS = ZnWideLds<WPS>::THREADS;
  static_assert(sizeof(ZnFusedLds) <= sizeof(ZnWideLds<WPS>), "the tail workgroups use the fused kernel's layout in the same allocation");
  // the Huffman planes of partial last chunks: the fused kernel's tail workgroups (four waves), at the front of this grid too, so that
  // they run beside the full chunks and not behind them
  if (blockIdx.x < ntail) {
    if (threadIdx.x >= ZN_F_THREADS) return;
    const ZnSeg one_c = one;                       // (a copy on this path only: see zn_k_decode_fused)
    uint32_t tail0 = 0;
    zn_decode_tail_wg(*reinterpret_cast<ZnFusedLds*>(&L), one_c, segs, nseg, blockIdx.x, tail_scratch, tail_done, status, &tail0);
    if (tailsync) {                              // (the merge workgroups at the end of this grid count the reports, as they do inside a fused launch)
      __syncthreads();
      if (threadIdx.x == 0) { ZN_FLAG_RELEASE(); ZN_FLAG_ADD32(tailsync + 2u * (tail0 / (uint32_t)P), 1u); }
    }
    return;
  }
  // … and their merge workgroups (round 6; the fused kernel's own: zn_tail_merge_wg) at its END: a ragged call is finished inside this launch, beside its
  // full chunks — in a launch of its own behind this one the merge cost a ragged call of 400 chunks 20 µs
  if (tailsync && blockIdx.x >= ntail + nchunk_wg) {       // (all of this workgroup's waves: the plane items are per wave, the merge strides by the workgroup's size)
    const ZnSeg one_c = one;
    zn_tail_merge_wg<P>(*reinterpret_cast<ZnFusedLds*>(&L), one_c, segs, nseg, blockIdx.x - (ntail + nchunk_wg), merge_per, tailsync, descs_rest, status, tail_done, tail_scratch);
    return;
  }
  const uint32_t wg = blockIdx.x - ntail;
  // (the call's first launch: the status word and the three "left to the generic kernels" counters start at zero — nothing before the
  //  kernels behind this one touches them, so this replaces a memset node in front of a call whose whole time is a few launches)
  if (zero_status && wg == 0 && threadIdx.x < 4u) status[threadIdx.x] = 0;
  const ZnSeg S_ = zn_find_seg<0>(one, segs, nseg, wg);
  ZnGeom g;
  g.n = zn_uniform64(S_.g.n); g.chunk = zn_uniform64(S_.g.chunk); g.K = zn_uniform64(S_.g.K); g.P = zn_uniform(S_.g.P); g.rot = zn_uniform(S_.g.rot);
  const uint8_t* __restrict__ body = ZN_GLOBAL_PTR(const uint8_t, zn_uniform64((uint64_t)S_.body)); const uint64_t body_len = zn_uniform64(S_.body_len);
  uint8_t* __restrict__ dst = ZN_GLOBAL_PTR(uint8_t, zn_uniform64((uint64_t)S_.dst));
  uint8_t* __restrict__ done = done_all + zn_uniform64(S_.chunk0);
  uint8_t* __restrict__ pdone = pdone_all + zn_uniform64(S_.desc0);
  const uint64_t c = (uint64_t)(wg - zn_uniform(S_.wg0));               // (ncg == 1)
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = zn_uniform(tid >> 6);
  const uint32_t plen = (uint32_t)(g.chunk / P), seg = plen / 4u;
  const uint8_t* body_end = body + body_len;
  if (c >= g.K) return;
  if (tailsync && zn_uniform(S_.has_tail) && c == g.K - 1u) {          // the partial last chunk: the tail + merge workgroups of this launch ("not by the fused kernels", and not pending)
    if (tid == 0) done[c] = 0; if (tid < (uint32_t)P) pdone[(uint64_t)tid * g.K + c] = 0;
    return;
  }

  if (tid < (uint32_t)P) {
    const uint32_t p = tid;
    const ZnPcMeta m = zn_pc_meta(g, body, body_len, p, c);
    ZnFusedPlane pl; pl.off = m.off; pl.csize = m.csize; pl.kind = 99u;
    if (m.ok && m.type <= 1u && zn_chunk_len(g, c) == g.chunk) {
      if (m.type == 0u) { if (m.csize >= plen) pl.kind = ZN_KIND_RAW; }
      else if (m.csize == plen) pl.kind = ZN_KIND_RAW;
      else if (m.csize == 1u) { pl.kind = ZN_KIND_RLE; pl.off = body[m.off]; }
      else if (m.csize > 1u && m.csize < plen) pl.kind = ZN_KIND_HUF;
    }
    L.plane[p] = pl;
  }
  __syncthreads();
  if (wave == 0) {
    int h = -1; uint32_t nhuf = 0;
    bool elig = (g.chunk % (4u * P * UNIT)) == 0 && ((((uint64_t)dst) & 15u) == 0) && S_.xr == nullptr;
    for (int p = 0; p < P; p++) {
      const uint32_t kind = L.plane[p].kind;
      if (kind == 99u) elig = false;
      if (kind == ZN_KIND_HUF) { if (h < 0) h = p; nhuf++; }
    }
    if (nhuf != 1u) elig = false;
    ZnWaveStats st; st.hs = 0; st.nsym = 0; st.tl = 0; st.lmin = 1; st.dom = 0;
    if (elig) {
      const uint32_t csize = L.plane[h].csize;
      uint8_t* scratch = (uint8_t*)&L.ring[0][0];
      st = zn_wave_read_stats(body + L.plane[h].off, csize, body_end, lane, scratch, L.symlist[0], L.rank_start[0], L.sym_start[0], scratch + 512);
      if (st.hs < 0 || st.tl > ZN_F_TLMAX || (uint32_t)st.hs >= csize || csize - (uint32_t)st.hs < 10u) elig = false;
    }
    if (lane == 0) { L.st = st; L.what = elig ? (uint32_t)(h + 2) : 0u; }
  } else {
    // the other fifteen waves, while wave 0 parses: zero the staging buffers 1-3 and pull the lines the first round will ask for —
    // the top four tiles of every stream, the first rows of the raw planes — towards the L2 (loads whose values nobody uses)
    for (uint32_t i = tid - 64u; i < 3u * (ZN_W_RING / 4u + 4u); i += ZN_W_THREADS - 64u) (&L.ring[1][0])[i] = 0;
    int h = -1;
    for (int p = 0; p < P; p++) if (L.plane[p].kind == ZN_KIND_HUF && h < 0) h = p;
    if (h >= 0) {
      const uint8_t* src = body + L.plane[h].off; const uint32_t csize = L.plane[h].csize;
      const uint32_t s_id = wave & 3u, part = wave >> 2;
      if (src < body_end && src + csize <= body_end && csize > 16u) {
        const uint32_t h0 = src[0];
        const uint32_t hs = h0 >= 128u ? 1u + (h0 - 126u) / 2u : 1u + h0;
        if (hs + 6u < csize) {
          const uint8_t* js = src + hs;
          const uint32_t l1 = zn_ld16(js), l2 = zn_ld16(js + 2), l3 = zn_ld16(js + 4);
          const uint32_t end_s = 6u + l1 + (s_id > 0 ? l2 : 0u) + (s_id > 1 ? l3 : 0u);        // end of stream s_id (s_id == 3: of the block)
          const uint8_t* top = (s_id == 3u) ? src + csize : js + end_s;
          const uint8_t* a = top - 1024u * part - 128u * (lane & 7u) - 4u;
          uint32_t t0 = 0;
          if (lane < 8u && a >= src && a + 4 <= body_end && top <= src + csize) t0 = *(const zn_u32u*)a;
          ZN_KEEP32(t0);
        }
      }
      // raw planes: this wave's share of the first ~14 KB of each stream's quarter, one dword per 128-byte line
      for (int p = 0; p < P; p++) if (p != h && L.plane[p].kind == ZN_KIND_RAW) {
        const uint8_t* a = body + L.plane[p].off + (uint64_t)s_id * seg + 4096u * part + 128u * (lane & 31u);
        uint32_t t1 = 0;
        if (lane < 32u && 4096u * part + 128u * (lane & 31u) < seg && a + 4 <= body_end) t1 = *(const zn_u32u*)a;
        ZN_KEEP32(t1);
      }
    }
  }
  __syncthreads();
  uint32_t what = zn_uniform(L.what);
  const int h = (int)what - 2;
  ZnFusedPlane pl[P];
  for (int p = 0; p < P; p++) { pl[p].off = zn_uniform64(L.plane[p].off); pl[p].kind = zn_uniform(L.plane[p].kind); pl[p].csize = zn_uniform(L.plane[p].csize); }
  bool ok = false;
  if (what != 0u) {
    uint64_t h_off = 0; uint32_t csize = 0;
    for (int p = 0; p < P; p++) if (p == h) { h_off = pl[p].off; csize = pl[p].csize; }
    const ZnWaveStats st = L.st;
    const int hs = (int)zn_uniform((uint32_t)st.hs); const uint32_t TL = zn_uniform(st.tl);
    const uint8_t* js = body + h_off + hs; const uint32_t rem = csize - (uint32_t)hs;
    const uint32_t l1 = zn_uniform(zn_ld16(js)), l2 = zn_uniform(zn_ld16(js + 2)), l3 = zn_uniform(zn_ld16(js + 4));
    uint32_t l4 = 0; bool bad = false;
    if (l1 + l2 + l3 + 6u > rem) bad = true; else l4 = rem - 6u - l1 - l2 - l3;
    if (l1 == 0 || l2 == 0 || l3 == 0 || l4 == 0) bad = true;
    // this form's codes: the density that gives the fused kernel 4-dword sub-blocks (the same rule), and not its dense-code instance
    uint32_t Dmin = 99u;
    if (!bad) {
      for (uint32_t t = 0; t < 4u; t++) {
        const uint32_t sl = t == 0 ? l1 : t == 1 ? l2 : t == 2 ? l3 : l4;
        const uint32_t Du = ((ZN_F_RING_BYTES - UNIT - 128u) * sl) / (256u * seg);
        Dmin = Du < Dmin ? Du : Dmin;
      }
    }
    const bool dense = Dmin > 4u && zn_uniform(st.lmin) >= ZN_F_DENSE_LMIN;
    if (!bad && Dmin >= 4u && !dense && (Dmin == 4u || zn_uniform(st.dom) < ZN_F_DOM_MAX)) {
      zn_fused_fill_luts<(int)ZN_W_THREADS>(L, tid, TL, 0, zn_uniform(st.lmin));
      __syncthreads();
      uint8_t* outc = dst + c * g.chunk;
      if (P == 1 || h == 0) ok = zn_wide_chunk<P, 0, WPS>(L, g, body, body_end, outc, pl, seg, TL, js, l1, l2, l3, l4);
      else if (P == 2 || h == 1) ok = zn_wide_chunk<P, (P >= 2 ? 1 : 0), WPS>(L, g, body, body_end, outc, pl, seg, TL, js, l1, l2, l3, l4);
      else if (h == 2) ok = zn_wide_chunk<P, (P >= 4 ? 2 : 0), WPS>(L, g, body, body_end, outc, pl, seg, TL, js, l1, l2, l3, l4);
      else ok = zn_wide_chunk<P, (P >= 4 ? 3 : 0), WPS>(L, g, body, body_end, outc, pl, seg, TL, js, l1, l2, l3, l4);
    }
  }
  if (tid == 0) done[c] = ok ? 1 : 2;
#if defined(ZN_SIMT_EMULATOR)
  if (tid == 0) zn_dbg_tiles[ok ? 4 : 5]++;            // (emulated build: chunks decoded here / left pending)
#endif
  if (ok && tid < (uint32_t)P) pdone[(uint64_t)tid * g.K + c] = 1;
}
