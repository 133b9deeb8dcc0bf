// zn_huf_tables.hpp — huff0 ENCODER table construction on the device (one lane, LDS scratch):
// histogram → code lengths (exact tie-breaks) → canonical codes → tree description.
// ≤256 symbols per plane-chunk; runs on lane 0 while the other lanes wait at a barrier; the
// data-parallel parts (histogram, bit packing) live in the kernels.  The decoder side is
// zn_huf_wave.hpp.  Everything follows the huff0 format of zstd 1.4.8 exactly, including
// tie-breaks, because compressed bytes must equal the CPU reference's
// (reference call sites: csrc/zipnn_core.c:366 HUF_compress, :807 HUF_decompress).
// Spec: SURVEY.md Appendix B.
#pragma once

#include "zn_common.hpp"

// ---------------------------------------------------------------------------
// LDS scratch shared by the table builders of one workgroup
// ---------------------------------------------------------------------------
struct ZnHNode { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nb; };

struct ZnTabScratch {
  uint32_t* count;           // byte histogram, 256 entries, OWNED BY THE CALLER (the generic encoder's; the fused table kernel sorts from registers and has none:
                             // 1 KB of LDS less per table job = 26 instead of 22 jobs resident per CU)
  uint8_t  weights[256];     // huff0 weights written into the tree description
  uint8_t  nbits[256];       // code length per symbol (encoder)
  uint16_t vals[256];        // code value per symbol (encoder)
  uint8_t  hdr[160];         // tree description bytes (≤ 1 + 128)
  // FSE over the weight alphabet (≤13 symbols, table log ≤ 6)
  int16_t  norm[16];
  uint16_t next[16];
  uint32_t tt_nb[16];
  int32_t  tt_fs[16];
  uint8_t  cell[64];
  uint16_t state[64];
  // small work arrays of the serial builders (kept in LDS: dynamically indexed private arrays would
  // otherwise live in scratch memory)
  uint32_t rk_base[33], rk_cur[33], rank_last[ZN_HUF_LOG_MAX + 2];
  uint16_t per_rank[ZN_HUF_LOG_MAX + 2], val_rank[ZN_HUF_LOG_MAX + 2];
  uint32_t wcount[ZN_HUF_LOG_MAX + 2], cumul[ZN_HUF_LOG_MAX + 4];
};

// FSE_optimalTableLog_internal: 32-bit unsigned arithmetic, wrap-around included
__device__ inline uint32_t zn_optimal_table_log(uint32_t max_log, uint32_t src_size, uint32_t max_sv, uint32_t minus) {
  uint32_t max_bits_src = zn_hb32(src_size - 1u) - minus;
  uint32_t t = max_log;
  uint32_t a = zn_hb32(src_size) + 1u, b = zn_hb32(max_sv) + 2u;
  uint32_t min_bits = a < b ? a : b;
  if (max_bits_src < t) t = max_bits_src;
  if (min_bits > t) t = min_bits;
  if (t < ZN_FSE_LOG_MIN) t = ZN_FSE_LOG_MIN;
  if (t > ZN_FSE_LOG_MAX) t = ZN_FSE_LOG_MAX;
  return t;
}

// ---------------------------------------------------------------------------
// encoder side: histogram -> code lengths/values -> tree description
// ---------------------------------------------------------------------------
// order: count descending, equal counts keep ascending symbol order (HUF_sort)
__device__ inline void zn_huf_sort(ZnTabScratch* S, ZnHNode* node, const uint32_t* count, uint32_t max_sv) {
  uint32_t* base = S->rk_base; uint32_t* cur = S->rk_cur;
  for (int i = 0; i < 33; i++) base[i] = 0;
  for (uint32_t n = 0; n <= max_sv; n++) base[zn_hb32(count[n] + 1u)]++;
  for (int n = 30; n > 0; n--) base[n - 1] += base[n];
  for (int i = 0; i < 33; i++) cur[i] = base[i];
  for (uint32_t n = 0; n <= max_sv; n++) {
    const uint32_t c = count[n], r = zn_hb32(c + 1u) + 1u;
    uint32_t pos = cur[r]++;
    while (pos > base[r] && c > node[pos - 1].count) { node[pos] = node[pos - 1]; pos--; }
    node[pos].count = c; node[pos].byte = (uint8_t)n;
  }
}

// HUF_setMaxHeight
__device__ inline uint32_t zn_huf_limit_height(ZnTabScratch* S, ZnHNode* node, uint32_t last, uint32_t max_nb) {
  const uint32_t largest = node[last].nb;
  if (largest <= max_nb) return largest;
  int cost = 0; const uint32_t base_cost = 1u << (largest - max_nb);
  int n = (int)last;
  while (node[n].nb > max_nb) { cost += (int)(base_cost - (1u << (largest - node[n].nb))); node[n].nb = (uint8_t)max_nb; n--; }
  while (node[n].nb == max_nb) n--;
  cost >>= (largest - max_nb);
  const uint32_t NONE = 0xF0F0F0F0u;
  uint32_t* rank_last = S->rank_last;
  for (uint32_t i = 0; i < ZN_HUF_LOG_MAX + 2; i++) rank_last[i] = NONE;
  { uint32_t cur = max_nb;
    for (int pos = n; pos >= 0; pos--) { if (node[pos].nb >= cur) continue; cur = node[pos].nb; rank_last[max_nb - cur] = (uint32_t)pos; } }
  while (cost > 0) {
    uint32_t d = zn_hb32((uint32_t)cost) + 1u;
    for (; d > 1; d--) {
      const uint32_t hp = rank_last[d], lp = rank_last[d - 1];
      if (hp == NONE) continue;
      if (lp == NONE) break;
      if (node[hp].count <= 2u * node[lp].count) break;
    }
    while (d <= ZN_HUF_LOG_MAX && rank_last[d] == NONE) d++;
    cost -= 1 << (d - 1);
    if (rank_last[d - 1] == NONE) rank_last[d - 1] = rank_last[d];
    node[rank_last[d]].nb++;
    if (rank_last[d] == 0) rank_last[d] = NONE;
    else { rank_last[d]--; if (node[rank_last[d]].nb != max_nb - d) rank_last[d] = NONE; }
  }
  while (cost < 0) {
    if (rank_last[1] == NONE) {
      while (node[n].nb == max_nb) n--;
      node[n + 1].nb--; rank_last[1] = (uint32_t)(n + 1); cost++; continue;
    }
    node[rank_last[1] + 1].nb--; rank_last[1]++; cost++;
  }
  return max_nb;
}

// Code lengths and canonical values from symbols ALREADY sorted into tab0[1..] (count descending, equal
// counts in ascending symbol order; every other field of the 513 nodes zero): the part of HUF_buildCTable
// after HUF_sort.  Returns the maximum code length.
__device__ inline uint32_t zn_huf_build_from_sorted(ZnTabScratch* S, ZnHNode* tab0, uint32_t max_sv, uint32_t max_nb_bits);

// HUF_buildCTable: S->count[0..max_sv] -> S->nbits/S->vals; tab0 = 513 nodes of LDS.
// Returns the maximum code length.
__device__ inline uint32_t zn_huf_build_ctable(ZnTabScratch* S, ZnHNode* tab0, uint32_t max_sv, uint32_t max_nb_bits) {
  for (int i = 0; i < 513; i++) { tab0[i].count = 0; tab0[i].parent = 0; tab0[i].byte = 0; tab0[i].nb = 0; }
  zn_huf_sort(S, tab0 + 1, S->count, max_sv);
  return zn_huf_build_from_sorted(S, tab0, max_sv, max_nb_bits);
}

// The serial core: tree over the leaves tab0[1 .. 1 + non_null] (sorted, counts > 0), code lengths (height-
// limited), and the first code value of every length in S->val_rank.  Returns the maximum code length.
__device__ inline uint32_t zn_huf_tree_from_sorted(ZnTabScratch* S, ZnHNode* tab0, int non_null, uint32_t max_nb_bits) {
  const int START = 256;
  ZnHNode* node = tab0 + 1;
  int low_s = non_null, node_nb = START, low_n = START;
  const int root = node_nb + low_s - 1;
  node[node_nb].count = node[low_s].count + node[low_s - 1].count;
  node[low_s].parent = node[low_s - 1].parent = (uint16_t)node_nb;
  node_nb++; low_s -= 2;
  for (int n = node_nb; n <= root; n++) node[n].count = 1u << 30;
  tab0[0].count = 1u << 31;   // node[-1]: barrier below the smallest leaf
  while (node_nb <= root) {
    const int n1 = (node[low_s].count < node[low_n].count) ? low_s-- : low_n++;
    const int n2 = (node[low_s].count < node[low_n].count) ? low_s-- : low_n++;
    node[node_nb].count = node[n1].count + node[n2].count;
    node[n1].parent = node[n2].parent = (uint16_t)node_nb;
    node_nb++;
  }
  node[root].nb = 0;
  for (int n = root - 1; n >= START; n--) node[n].nb = (uint8_t)(node[node[n].parent].nb + 1);
  for (int n = 0; n <= non_null; n++) node[n].nb = (uint8_t)(node[node[n].parent].nb + 1);
  max_nb_bits = zn_huf_limit_height(S, node, (uint32_t)non_null, max_nb_bits);
  uint16_t* per_rank = S->per_rank; uint16_t* val_rank = S->val_rank;
  for (uint32_t i = 0; i <= ZN_HUF_LOG_MAX; i++) { per_rank[i] = 0; val_rank[i] = 0; }
  for (int n = 0; n <= non_null; n++) per_rank[node[n].nb]++;
  { uint16_t mn = 0; for (int n = (int)max_nb_bits; n > 0; n--) { val_rank[n] = mn; mn = (uint16_t)(mn + per_rank[n]); mn >>= 1; } }
  return max_nb_bits;
}

__device__ inline uint32_t zn_huf_build_from_sorted(ZnTabScratch* S, ZnHNode* tab0, uint32_t max_sv, uint32_t max_nb_bits) {
  ZnHNode* node = tab0 + 1;
  int non_null = (int)max_sv;
  while (node[non_null].count == 0) non_null--;
  max_nb_bits = zn_huf_tree_from_sorted(S, tab0, non_null, max_nb_bits);
  uint16_t* val_rank = S->val_rank;
  for (int n = 0; n <= (int)max_sv; n++) S->nbits[node[n].byte] = node[n].nb;      // (symbols that do not occur: nb 0)
  for (int n = 0; n <= (int)max_sv; n++) S->vals[n] = val_rank[S->nbits[n]]++;
  return max_nb_bits;
}

// tiny LSB-first bit writer into S->hdr (tree descriptions are ≤ 129 bytes)
struct ZnSmallBitW { uint8_t* out; uint32_t cap; uint64_t acc; uint32_t nacc; uint32_t nbytes; };
__device__ inline void zn_sbw_add(ZnSmallBitW* w, uint32_t v, uint32_t nb) {
  if (nb == 0) return;
  w->acc |= ((uint64_t)(v & ((1u << nb) - 1u))) << w->nacc; w->nacc += nb;
  while (w->nacc >= 8) { if (w->nbytes < w->cap) w->out[w->nbytes] = (uint8_t)w->acc; w->nbytes++; w->acc >>= 8; w->nacc -= 8; }
}

// FSE_normalizeCount (+ FSE_normalizeM2) for the weight histogram.  low_prob: what a count that rounds below one table cell
// becomes — +1 (huff0 of zstd ≥ 1.4.7: HUF_compressWeights passes useLowProbCount = 0; the default here and what the oracle's pin
// writes) or -1 (the huff0 of the FiniteStateEntropy library the reference's PyPI wheels bundle, /root/reference/setup.py:23-28:
// the "less than one" marker — the symbol gets the table's top cell; zn_set_legacy_tree_descriptions).  Both decode everywhere.
// Returns 0 ok, 1 = rle, -1 = error.
__device__ inline int zn_fse_normalize(int16_t* norm, uint32_t tl, const uint32_t* count, uint32_t total_in, uint32_t max_sv, int low_prob = 1) {
#define ZN_RTB(p_) ((p_) == 0 ? 0u : (p_) == 1 ? 473195u : (p_) == 2 ? 504333u : (p_) == 3 ? 520860u : (p_) == 4 ? 550000u : (p_) == 5 ? 700000u : (p_) == 6 ? 750000u : 830000u)
  uint64_t total = total_in;
  const uint64_t scale = 62 - tl, step = (1ULL << 62) / (uint32_t)total, vstep = 1ULL << (scale - 20);
  int still = 1 << tl; uint32_t largest = 0; int16_t largest_p = 0;
  const uint32_t low_thr = (uint32_t)(total >> tl);
  for (uint32_t s = 0; s <= max_sv; s++) {
    if (count[s] == total) return 1;
    if (count[s] == 0) { norm[s] = 0; continue; }
    if (count[s] <= low_thr) { norm[s] = (int16_t)low_prob; still--; continue; }
    int16_t p = (int16_t)(((uint64_t)count[s] * step) >> scale);
    if (p < 8) { const uint64_t beat = vstep * ZN_RTB(p); p += ((uint64_t)count[s] * step) - ((uint64_t)p << scale) > beat; }
    if (p > largest_p) { largest_p = p; largest = s; }
    norm[s] = p; still -= p;
  }
  if (-still < (norm[largest] >> 1)) { norm[largest] += (int16_t)still; return 0; }
  // secondary normalisation
  const int16_t UNSET = -2;
  uint32_t distributed = 0, to_dist, low_one = (uint32_t)((total * 3) >> (tl + 1));
  for (uint32_t s = 0; s <= max_sv; s++) {
    if (count[s] == 0) { norm[s] = 0; continue; }
    if (count[s] <= low_thr) { norm[s] = (int16_t)low_prob; distributed++; total -= count[s]; continue; }
    if (count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; continue; }
    norm[s] = UNSET;
  }
  to_dist = (1u << tl) - distributed;
  if (to_dist == 0) return 0;
  if ((total / to_dist) > low_one) {
    low_one = (uint32_t)((total * 3) / (to_dist * 2));
    for (uint32_t s = 0; s <= max_sv; s++)
      if (norm[s] == UNSET && count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; }
    to_dist = (1u << tl) - distributed;
  }
  if (distributed == max_sv + 1) {
    uint32_t best = 0, best_c = 0;
    for (uint32_t s = 0; s <= max_sv; s++) if (count[s] > best_c) { best = s; best_c = count[s]; }
    norm[best] += (int16_t)to_dist;
    return 0;
  }
  if (total == 0) {
    for (uint32_t s = 0; to_dist > 0; s = (s + 1) % (max_sv + 1))
      if (norm[s] > 0) { to_dist--; norm[s]++; }
    return 0;
  }
  {
    const uint64_t vlog = 62 - tl, mid = (1ULL << (vlog - 1)) - 1;
    const uint64_t rstep = (((1ULL << vlog) * to_dist) + mid) / (uint32_t)total;
    uint64_t run = mid;
    for (uint32_t s = 0; s <= max_sv; s++) {
      if (norm[s] != UNSET) continue;
      const uint64_t end = run + (uint64_t)count[s] * rstep;
      const uint32_t wgt = (uint32_t)(end >> vlog) - (uint32_t)(run >> vlog);
      if (wgt < 1) return -1;
      norm[s] = (int16_t)wgt; run = end;
    }
  }
  return 0;
}

// HUF_compressWeights: weights w[0..nw) -> dst.  0 = not compressible, 1 = single value,
// >1 = size (1000 = "too long to be kept"), -1 = error.
// pre: S->wcount already holds the histogram of w[0..nw) (filled in parallel by the caller).
__device__ inline int zn_huf_compress_weights(ZnTabScratch* S, uint8_t* dst, uint32_t cap, const uint8_t* w, uint32_t nw, bool pre = false, int low_prob = 1) {
  uint32_t* count = S->wcount;
  uint32_t max_sv = ZN_HUF_LOG_MAX, max_c = 0;
  if (nw <= 1) return 0;
  if (!pre) {
    for (uint32_t i = 0; i <= ZN_HUF_LOG_MAX; i++) count[i] = 0;
    for (uint32_t i = 0; i < nw; i++) count[w[i]]++;
  }
  while (count[max_sv] == 0) max_sv--;
  for (uint32_t i = 0; i <= max_sv; i++) if (count[i] > max_c) max_c = count[i];
  if (max_c == nw) return 1;
  if (max_c == 1) return 0;
  const uint32_t tl = zn_optimal_table_log(ZN_WEIGHT_FSE_LOG, nw, max_sv, 2);
  { int r = zn_fse_normalize(S->norm, tl, count, nw, max_sv, low_prob); if (r != 0) return r == 1 ? 0 : -1; }
  uint32_t off = 0;
  // FSE_writeNCount
  {
    const int table_size = 1 << tl;
    int nb_bits = (int)tl + 1, remaining = table_size + 1, threshold = table_size;
    uint32_t bits = 0; int nbit = 0; uint32_t sym = 0; const uint32_t alpha = max_sv + 1; int prev0 = 0;
    bits += (tl - ZN_FSE_LOG_MIN) << nbit; nbit += 4;
    while (sym < alpha && remaining > 1) {
      if (prev0) {
        uint32_t start = sym;
        while (sym < alpha && !S->norm[sym]) sym++;
        if (sym == alpha) break;
        while (sym >= start + 24) { start += 24; bits += 0xFFFFu << nbit; dst[off] = (uint8_t)bits; dst[off + 1] = (uint8_t)(bits >> 8); off += 2; bits >>= 16; }
        while (sym >= start + 3) { start += 3; bits += 3u << nbit; nbit += 2; }
        bits += (sym - start) << nbit; nbit += 2;
        if (nbit > 16) { dst[off] = (uint8_t)bits; dst[off + 1] = (uint8_t)(bits >> 8); off += 2; bits >>= 16; nbit -= 16; }
      }
      {
        int c = S->norm[sym++];
        const int mx = (2 * threshold - 1) - remaining;
        remaining -= c < 0 ? -c : c;
        c++;
        if (c >= threshold) c += mx;
        bits += (uint32_t)c << nbit; nbit += nb_bits; nbit -= (c < mx);
        prev0 = (c == 1);
        if (remaining < 1) return -1;
        while (remaining < threshold) { nb_bits--; threshold >>= 1; }
      }
      if (nbit > 16) { dst[off] = (uint8_t)bits; dst[off + 1] = (uint8_t)(bits >> 8); off += 2; bits >>= 16; nbit -= 16; }
    }
    if (remaining != 1) return -1;
    dst[off] = (uint8_t)bits; dst[off + 1] = (uint8_t)(bits >> 8);
    off += (uint32_t)(nbit + 7) / 8u;
  }
  // FSE_buildCTable
  {
    const uint32_t size = 1u << tl, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    uint32_t* cumul = S->cumul;
    uint32_t high = size - 1u;
    cumul[0] = 0;
    for (uint32_t u = 1; u <= max_sv + 1; u++) {
      if (S->norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; S->cell[high--] = (uint8_t)(u - 1); }
      else cumul[u] = cumul[u - 1] + (uint32_t)S->norm[u - 1];
    }
    cumul[max_sv + 1] = size + 1;
    { uint32_t pos = 0;
      for (uint32_t s = 0; s <= max_sv; s++)
        for (int i = 0; i < S->norm[s]; i++) { S->cell[pos] = (uint8_t)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; } }
    for (uint32_t u = 0; u < size; u++) { const uint32_t s = S->cell[u]; S->state[cumul[s]++] = (uint16_t)(size + u); }
    int total = 0;
    for (uint32_t s = 0; s <= max_sv; s++) {
      const int f = S->norm[s];
      if (f == 0) { S->tt_nb[s] = ((tl + 1) << 16) - (1u << tl); S->tt_fs[s] = 0; }
      else if (f == -1 || f == 1) { S->tt_nb[s] = (tl << 16) - (1u << tl); S->tt_fs[s] = total - 1; total++; }
      else {
        const uint32_t max_bits_out = tl - zn_hb32((uint32_t)f - 1u);
        S->tt_nb[s] = (max_bits_out << 16) - ((uint32_t)f << max_bits_out);
        S->tt_fs[s] = total - f; total += f;
      }
    }
  }
  // FSE_compress_usingCTable: two states, symbols walked backwards
  {
    if (nw <= 2) return 0;
    ZnSmallBitW bw; bw.out = dst + off; bw.cap = cap - off; bw.acc = 0; bw.nacc = 0; bw.nbytes = 0;
    uint32_t i = nw, s1, s2;
#define ZN_FSE_INIT(sym_) ({ const uint32_t y_ = (sym_); const uint32_t nb_ = (S->tt_nb[y_] + (1u << 15)) >> 16; \
                            const uint32_t v_ = (nb_ << 16) - S->tt_nb[y_]; (uint32_t)S->state[(int)(v_ >> nb_) + S->tt_fs[y_]]; })
#define ZN_FSE_ENC(st_, sym_) ({ const uint32_t y_ = (sym_); const uint32_t nb_ = ((st_) + S->tt_nb[y_]) >> 16; \
                                 zn_sbw_add(&bw, (st_), nb_); (uint32_t)S->state[(int)((st_) >> nb_) + S->tt_fs[y_]]; })
    if (nw & 1) { s1 = ZN_FSE_INIT(w[--i]); s2 = ZN_FSE_INIT(w[--i]); s1 = ZN_FSE_ENC(s1, w[--i]); }
    else        { s2 = ZN_FSE_INIT(w[--i]); s1 = ZN_FSE_INIT(w[--i]); }
    while (i > 0) { s2 = ZN_FSE_ENC(s2, w[--i]); s1 = ZN_FSE_ENC(s1, w[--i]); }
#undef ZN_FSE_INIT
#undef ZN_FSE_ENC
    zn_sbw_add(&bw, s2, tl); zn_sbw_add(&bw, s1, tl);
    zn_sbw_add(&bw, 1, 1);
    uint32_t c = bw.nbytes;
    if (bw.nacc) { if (bw.nbytes < bw.cap) bw.out[bw.nbytes] = (uint8_t)bw.acc; c++; }
    if (c > bw.cap) return 1000;   // longer than any description huff0 would keep (needs < maxSV/2 ≤ 127)
    off += c;
  }
  return (int)off;
}

// HUF_writeCTable: S->nbits -> S->hdr.  Returns header size, or -1 (caller stores the plane raw).
// pre: S->weights[0..max_sv) and their histogram S->wcount are already filled (in parallel, by the caller).
__device__ inline int zn_huf_write_ctable(ZnTabScratch* S, uint32_t max_sv, uint32_t huff_log, bool pre = false) {
  uint8_t* w = S->weights; uint8_t* op = S->hdr;
  if (!pre) for (uint32_t n = 0; n < max_sv; n++) w[n] = S->nbits[n] ? (uint8_t)(huff_log + 1u - S->nbits[n]) : 0;
  {
    // an FSE description is only kept when shorter than maxSV/2 ≤ 127 bytes
    const int h = zn_huf_compress_weights(S, op + 1, 140, w, max_sv, pre);
    if (h < 0) return -1;
    if (h > 1 && (uint32_t)h < max_sv / 2u) { op[0] = (uint8_t)h; return h + 1; }
  }
  if (max_sv > 128u) return -1;
  op[0] = (uint8_t)(128u + (max_sv - 1u));
  w[max_sv] = 0;
  for (uint32_t n = 0; n < max_sv; n += 2) op[n / 2 + 1] = (uint8_t)((w[n] << 4) + w[n + 1]);
  return (int)(((max_sv + 1u) / 2u) + 1u);
}

// ---------------------------------------------------------------------------
// zn_huf_tree_from_sorted by ONE WAVE (the fused table kernel; round 3): what is serial in it — the two-queue merge and the depths
// of the internal nodes, one step per symbol each — stays on lane 0; the three other walks over the symbols (sentinel counts of the
// internal nodes, depths of the leaves, the histogram of code lengths) are lane-parallel.  For fp8 / fp16 planes (100-250 symbols
// that occur) the serial tree was the largest part of a table job.  Same results as the serial function (tests: frames == oracle).
// All 64 lanes call it together; `tab0` = the 513 nodes in LDS with the sorted leaves in tab0[1 ..]; returns the maximum code length.
// ---------------------------------------------------------------------------
__device__ inline uint32_t zn_wave_tree_from_sorted(ZnTabScratch* S, ZnHNode* tab0, int non_null, uint32_t max_nb_bits, uint32_t lane) {
  const int START = 256;
  ZnHNode* node = tab0 + 1;
  const int root = START + non_null - 1;
  for (int n = START + (int)lane; n <= root; n += 64) node[n].count = 1u << 30;      // internal nodes not made yet: larger than any sum
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    int low_s = non_null, node_nb = START, low_n = START;
    node[node_nb].count = node[low_s].count + node[low_s - 1].count;
    node[low_s].parent = node[low_s - 1].parent = (uint16_t)node_nb;
    node_nb++; low_s -= 2;
    tab0[0].count = 1u << 31;   // node[-1]: barrier below the smallest leaf
    while (node_nb <= root) {
      const int n1 = (node[low_s].count < node[low_n].count) ? low_s-- : low_n++;
      const int n2 = (node[low_s].count < node[low_n].count) ? low_s-- : low_n++;
      node[node_nb].count = node[n1].count + node[n2].count;
      node[n1].parent = node[n2].parent = (uint16_t)node_nb;
      node_nb++;
    }
    node[root].nb = 0;
    for (int n = root - 1; n >= START; n--) node[n].nb = (uint8_t)(node[node[n].parent].nb + 1);
  }
  __builtin_amdgcn_wave_barrier();
  for (int n = (int)lane; n <= non_null; n += 64) node[n].nb = (uint8_t)(node[node[n].parent].nb + 1);
  __builtin_amdgcn_wave_barrier();
  uint32_t largest = node[non_null].nb;                                // (the smallest count has the longest code)
  __builtin_amdgcn_wave_barrier();                                     // (every lane has read it before lane 0 may change it)
  if (largest > max_nb_bits) {                                         // HUF_setMaxHeight: rare, serial
    if (lane == 0) S->rk_base[0] = zn_huf_limit_height(S, node, (uint32_t)non_null, max_nb_bits);
    __builtin_amdgcn_wave_barrier();
    largest = S->rk_base[0];
  }
  // code lengths' histogram (ballots) -> first code value of every length
  uint32_t per_rank[ZN_HUF_LOG_MAX + 1];
  for (uint32_t v = 0; v <= ZN_HUF_LOG_MAX; v++) per_rank[v] = 0;
  for (int q = 0; q < 256; q += 64) {
    const int n = q + (int)lane;
    const uint32_t nb = (n <= non_null) ? (uint32_t)node[n].nb : 0xFFu;
    for (uint32_t v = 0; v <= ZN_HUF_LOG_MAX; v++) per_rank[v] += (uint32_t)__popcll(__ballot(nb == v));
  }
  {
    uint32_t mn = 0, mine = 0, pr = per_rank[0];
    for (int n = (int)ZN_HUF_LOG_MAX; n > 0; n--) {
      const uint32_t vr = ((uint32_t)n <= largest) ? mn : 0u;
      if ((uint32_t)n <= largest) { mn = (mn + per_rank[n]) & 0xFFFFu; mn >>= 1; }
      mine = (lane == (uint32_t)n) ? vr : mine;
      pr = (lane == (uint32_t)n) ? per_rank[n] : pr;
    }
    if (lane <= ZN_HUF_LOG_MAX) { S->val_rank[lane] = (uint16_t)mine; S->per_rank[lane] = (uint16_t)pr; }
  }
  __builtin_amdgcn_wave_barrier();
  return largest;
}

// ---------------------------------------------------------------------------
// The same tree description by ONE WAVE (the fused table kernel; round 3).
//
// HUF_compressWeights is a serial state chain over ≤255 weights; on one lane every step is three dependent LDS round
// trips (weight → its transform → next state), 300 cycles a weight, 75 k cycles a table — more than the rest of the
// table job together.  Here the whole wave runs the chain on wave-uniform values, the mirror image of the decoder's
// parser (zn_huf_wave.hpp): the FSE state table lives in one VGPR (lane u = entry u), the per-symbol transforms in two
// (lane s = weight value s), the weights themselves as nibbles in a fourth (lane l = weights 8l … 8l+7), the output
// bytes in a fifth (lane d = dword d); every access is a v_readlane with a uniform index, so the chain is scalar-ALU
// code with no memory latency in it.  The normalised counts, the symbol spread, the state table and the transforms are
// built lane-parallel.  Same bytes and return codes as zn_huf_compress_weights / zn_huf_write_ctable above (the generic
// encoder keeps those; tests compare both against the oracle).
//
// All 64 lanes must call these together (uniform control flow).  S->weights[0..nw) and S->wcount[0..13] are filled.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zn_trl(uint32_t v, uint32_t idx) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)idx); }

__device__ inline int zn_wave_compress_weights(ZnTabScratch* S, uint8_t* dst, uint32_t cap, const uint32_t (&wc)[13], uint32_t nw, uint32_t lane, int low_prob = 1) {
  if (nw <= 1) return 0;
  uint32_t max_sv = 0, max_c = 0;
  for (uint32_t v = 0; v < 13u; v++) { if (wc[v]) max_sv = v; if (wc[v] > max_c) max_c = wc[v]; }
  if (max_c == nw) return 1;
  if (max_c == 1) return 0;
  const uint32_t tl = zn_optimal_table_log(ZN_WEIGHT_FSE_LOG, nw, max_sv, 2);
  const uint32_t size = 1u << tl;
  if (low_prob < 0) {
    // legacy tree descriptions (-1 markers): only a description in which some weight value is rare enough to get one differs from the
    // default form — that one is written by the serial builder on one lane (its table build knows the marker's top cells); every other
    // description takes the wave's path below, whose bytes are the same in both forms
    bool rare = false;
    for (uint32_t v = 0; v <= max_sv; v++) if (wc[v] != 0u && wc[v] <= (nw >> tl)) rare = true;
    if (rare) {
      if (lane == 0) { const int h = zn_huf_compress_weights(S, dst, cap, S->weights, nw, true, -1); S->cumul[14] = (uint32_t)h; }
      __builtin_amdgcn_wave_barrier();
      return (int)S->cumul[14];
    }
  }

  // ---- FSE_normalizeCount, lane s = weight value s; the correction of the largest count in symbol order on uniform values ----
  uint32_t normv = 0;
  {
    uint32_t cl = 0;
    for (uint32_t v = 0; v < 13u; v++) cl = (lane == v) ? wc[v] : cl;
    const uint64_t scale = 62 - tl, step = (1ULL << 62) / nw, vstep = 1ULL << (scale - 20);
    const uint32_t low_thr = nw >> tl;
    uint32_t p = 0, pn = 0;                         // pn: p of the counts that went through the proportional rule
    if (lane <= max_sv && cl != 0) {
      if (cl <= low_thr) p = 1;
      else {
        p = (uint32_t)(((uint64_t)cl * step) >> scale);
        if (p < 8u) { const uint64_t beat = vstep * ZN_RTB(p); p += ((uint64_t)cl * step) - ((uint64_t)p << scale) > beat ? 1u : 0u; }
        pn = p;
      }
    }
    int still = (int)size; uint32_t largest = 0, largest_p = 0;
    for (uint32_t s = 0; s < 13u; s++) {
      still -= (int)zn_trl(p, s);
      const uint32_t q = zn_trl(pn, s);
      if (q > largest_p) { largest_p = q; largest = s; }
    }
    const int nl = (int)zn_trl(p, largest);
    if (-still < (nl >> 1)) normv = (lane == largest) ? (uint32_t)(nl + still) : p;
    else {
      // secondary normalisation (rare): the serial builder on one lane
      if (lane == 0) { const int r = zn_fse_normalize(S->norm, tl, S->wcount, nw, max_sv, 1); S->cumul[0] = (uint32_t)r; }      // (no rare count on this path in either form: see above)
      __builtin_amdgcn_wave_barrier();
      const int r = (int)S->cumul[0];
      if (r != 0) return r == 1 ? 0 : -1;
      normv = (lane <= max_sv) ? (uint32_t)(int)S->norm[lane] : 0u;
    }
  }

  // ---- FSE_writeNCount (uniform; the counts are never -1 here) ----
  uint32_t off = 0;
  {
    const int table_size = 1 << tl;
    int nb_bits = (int)tl + 1, remaining = table_size + 1, threshold = table_size;
    uint32_t bits = 0; int nbit = 0; uint32_t sym = 0; const uint32_t alpha = max_sv + 1; int prev0 = 0;
    bits += (tl - ZN_FSE_LOG_MIN) << nbit; nbit += 4;
#define ZN_NC_FLUSH16() do { if (lane == 0) { dst[off] = (uint8_t)bits; dst[off + 1] = (uint8_t)(bits >> 8); } off += 2; bits >>= 16; } while (0)
    while (sym < alpha && remaining > 1) {
      if (prev0) {
        uint32_t start = sym;
        while (sym < alpha && !zn_trl(normv, sym)) sym++;
        if (sym == alpha) break;
        while (sym >= start + 24) { start += 24; bits += 0xFFFFu << nbit; ZN_NC_FLUSH16(); }
        while (sym >= start + 3) { start += 3; bits += 3u << nbit; nbit += 2; }
        bits += (sym - start) << nbit; nbit += 2;
        if (nbit > 16) { ZN_NC_FLUSH16(); nbit -= 16; }
      }
      {
        int c = (int)zn_trl(normv, sym); sym++;
        const int mx = (2 * threshold - 1) - remaining;
        remaining -= c;
        c++;
        if (c >= threshold) c += mx;
        bits += (uint32_t)c << nbit; nbit += nb_bits; nbit -= (c < mx);
        prev0 = (c == 1);
        if (remaining < 1) return -1;
        while (remaining < threshold) { nb_bits--; threshold >>= 1; }
      }
      if (nbit > 16) { ZN_NC_FLUSH16(); nbit -= 16; }
    }
    if (remaining != 1) return -1;
    if (lane == 0) { dst[off] = (uint8_t)bits; dst[off + 1] = (uint8_t)(bits >> 8); }
    off += (uint32_t)(nbit + 7) / 8u;
#undef ZN_NC_FLUSH16
  }
  if (nw <= 2) return 0;

  // ---- FSE_buildCTable, lane-parallel: lane = table position (spread, state table) and lane = weight value (transforms) ----
  uint32_t statev, ttnb, ttfs;
  {
    const uint32_t mask = size - 1u, stepc = (size >> 1) + (size >> 3) + 3u;
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64u - lane));
    uint32_t cum = 0, run = 0, symj = 0;            // cum: Σ norm of the weight values below this lane's
    for (uint32_t s = 0; s < 13u; s++) { const uint32_t ns = zn_trl(normv, s); cum += (lane > s) ? ns : 0u; run += ns; symj += (run <= lane) ? 1u : 0u; }
    if (lane < size) S->cell[(lane * stepc) & mask] = (uint8_t)symj;       // the j-th visited cell belongs to the j-th entry in symbol order
    __builtin_amdgcn_wave_barrier();
    const uint32_t cs = (lane < size) ? (uint32_t)S->cell[lane] : 255u;
    uint32_t idx = 0;
    for (uint32_t s = 0; s < 13u; s++) {
      const uint64_t m = __ballot(cs == s);
      const uint32_t c0 = zn_trl(cum, s);
      if (cs == s) idx = c0 + (uint32_t)__popcll(m & lt);
    }
    if (lane < size) S->state[idx] = (uint16_t)(size + lane);
    __builtin_amdgcn_wave_barrier();
    statev = (lane < size) ? (uint32_t)S->state[lane] : 0u;
    const uint32_t f = normv;
    if (f == 0) { ttnb = ((tl + 1u) << 16) - size; ttfs = 0; }
    else if (f == 1) { ttnb = (tl << 16) - size; ttfs = cum - 1u; }
    else { const uint32_t mbo = tl - zn_hb32(f - 1u); ttnb = (mbo << 16) - (f << mbo); ttfs = cum - f; }
  }

  // ---- FSE_compress_usingCTable: two states, weights walked backwards, everything wave-uniform ----
  uint32_t pk = 0;                                  // lane l < 32: weights 8l … 8l+7 as nibbles
  if (lane < 32u) for (uint32_t k = 0; k < 8u; k++) pk |= ((uint32_t)S->weights[8u * lane + k] & 15u) << (4u * k);
  uint64_t acc = 0; uint32_t nacc = 0, widx = 0, outv = 0;
#define ZN_TW(i_) ((zn_trl(pk, (i_) >> 3) >> (4u * ((i_) & 7u))) & 15u)
#define ZN_TEMIT(v_, nb_) do { const uint32_t n_ = (nb_); acc |= (uint64_t)((v_) & ((1u << n_) - 1u)) << nacc; nacc += n_; \
    if (nacc >= 32u) { outv = (lane == widx) ? (uint32_t)acc : outv; widx++; acc >>= 32; nacc -= 32u; } } while (0)
#define ZN_TINIT(y_) ({ const uint32_t t_ = zn_trl(ttnb, (y_)); const uint32_t nb_ = (t_ + (1u << 15)) >> 16; const uint32_t v_ = (nb_ << 16) - t_; \
    zn_trl(statev, (uint32_t)((int)(v_ >> nb_) + (int)zn_trl(ttfs, (y_))) & 63u); })
#define ZN_TENC(st_, y_) do { const uint32_t nb_ = ((st_) + zn_trl(ttnb, (y_))) >> 16; ZN_TEMIT((st_), nb_); \
    st_ = zn_trl(statev, (uint32_t)((int)((st_) >> nb_) + (int)zn_trl(ttfs, (y_))) & 63u); } while (0)
  uint32_t i = nw, s1, s2;
  if (nw & 1u) { i--; s1 = ZN_TINIT(ZN_TW(i)); i--; s2 = ZN_TINIT(ZN_TW(i)); i--; { const uint32_t y = ZN_TW(i); ZN_TENC(s1, y); } }
  else         { i--; s2 = ZN_TINIT(ZN_TW(i)); i--; s1 = ZN_TINIT(ZN_TW(i)); }
  while (i > 0) {
    i--; { const uint32_t y = ZN_TW(i); ZN_TENC(s2, y); }
    i--; { const uint32_t y = ZN_TW(i); ZN_TENC(s1, y); }
  }
  ZN_TEMIT(s2, tl); ZN_TEMIT(s1, tl);
  ZN_TEMIT(1u, 1u);
  const uint32_t c = 4u * widx + (nacc + 7u) / 8u;
  if (nacc) outv = (lane == widx) ? (uint32_t)acc : outv;
#undef ZN_TW
#undef ZN_TEMIT
#undef ZN_TINIT
#undef ZN_TENC
  if (c > cap - off) return 1000;                   // longer than any description huff0 would keep (needs < maxSV/2 ≤ 127)
  for (uint32_t b = 0; b < 4u; b++) if (4u * lane + b < c) dst[off + 4u * lane + b] = (uint8_t)(outv >> (8u * b));
  return (int)(off + c);
}

// HUF_writeCTable by one wave: S->weights / S->wcount (+ the same histogram, wave-uniform, in wc) -> S->hdr.
// Returns the header size, or -1 (the caller stores the plane raw).  S->hdr is complete after the caller's next barrier.
__device__ inline int zn_wave_write_ctable(ZnTabScratch* S, uint32_t max_sv, const uint32_t (&wc)[13], uint32_t lane, int low_prob = 1) {
  const int h = zn_wave_compress_weights(S, S->hdr + 1, 140, wc, max_sv, lane, low_prob);
  if (h < 0) return -1;
  if (h > 1 && (uint32_t)h < max_sv / 2u) { if (lane == 0) S->hdr[0] = (uint8_t)h; return h + 1; }
  if (max_sv > 128u) return -1;
  if (lane == 0) S->hdr[0] = (uint8_t)(128u + (max_sv - 1u));
  for (uint32_t n = 2u * lane; n < max_sv; n += 128u) {
    const uint32_t w0 = S->weights[n], w1 = (n + 1u < max_sv) ? (uint32_t)S->weights[n + 1u] : 0u;
    S->hdr[n / 2u + 1u] = (uint8_t)((w0 << 4) + w1);
  }
  return (int)(((max_sv + 1u) / 2u) + 1u);
}
