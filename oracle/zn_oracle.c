/*
 * zn_oracle.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY (see zn_oracle.h).
 *
 * Scalar restatement of the ZipNN compress/decompress hot path.  Nothing here is
 * shipped or measured as the product; the product path is zipnn_amd/csrc (HIP).
 *
 * huff0 sections restate the algorithm of Cyan4973/FiniteStateEntropy lib/
 * (huf_compress.c, huf_decompress.c, fse_compress.c, fse_decompress.c,
 * entropy_common.c, hist.c) as shipped inside zstd 1.4.8; the reference calls it at
 * csrc/zipnn_core.c:366 (HUF_compress), :807 (HUF_decompress), :813 (HUF_isError).
 * Spec and validation notes: SURVEY.md Appendix B / E.
 */
#include "zn_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* small helpers                                                              */
/* ------------------------------------------------------------------------- */
#define HUF_BLOCK_MAX    (128u * 1024u)
#define HUF_LOG_MAX      12u
#define HUF_LOG_DEFAULT  11u
#define HUF_SYM_MAX      255u
#define FSE_LOG_MIN      5u
#define FSE_LOG_MAX      12u
#define WEIGHT_FSE_LOG   6u /* MAX_FSE_TABLELOG_FOR_HUFF_HEADER */

static unsigned hb32(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }

unsigned zo_huf_is_error(size_t code) { return code > (size_t)-120; }

/* LSB-first bit writer; mirrors BIT_CStream_t's overflow rule: the stream is
 * declared overflowed iff floor(total_bits/8) >= capacity - 8. */
typedef struct {
  uint8_t* out;
  size_t cap;
  uint64_t acc;
  unsigned nacc;   /* bits pending in acc (< 8 after a drain) */
  size_t nbytes;   /* whole bytes already emitted */
  size_t total_bits;
} bitw_t;

static int bitw_init(bitw_t* w, uint8_t* out, size_t cap) {
  w->out = out; w->cap = cap; w->acc = 0; w->nacc = 0; w->nbytes = 0; w->total_bits = 0;
  return cap > 8;
}
static void bitw_drain(bitw_t* w) {
  while (w->nacc >= 8) {
    if (w->nbytes < w->cap) w->out[w->nbytes] = (uint8_t)w->acc;
    w->nbytes++; w->acc >>= 8; w->nacc -= 8;
  }
}
static void bitw_add(bitw_t* w, uint32_t v, unsigned nb) {
  if (nb == 0) return;
  w->acc |= ((uint64_t)(v & ((1u << nb) - 1u))) << w->nacc;
  w->nacc += nb; w->total_bits += nb;
  bitw_drain(w);
}
/* end mark + zero pad; returns stream bytes, or 0 on overflow */
static size_t bitw_close(bitw_t* w) {
  bitw_add(w, 1, 1);
  if ((w->total_bits >> 3) >= w->cap - 8) return 0;
  if (w->nacc) { w->out[w->nbytes] = (uint8_t)w->acc; return w->nbytes + 1; }
  return w->nbytes;
}

/* ------------------------------------------------------------------------- */
/* FSE pieces used by the huff0 weight header                                  */
/* ------------------------------------------------------------------------- */
unsigned zo_optimal_table_log(unsigned max_log, size_t src_size, unsigned max_sv, unsigned minus) {
  /* all arithmetic in 32-bit unsigned, wrap-around included (FSE_optimalTableLog_internal) */
  uint32_t max_bits_src = hb32((uint32_t)(src_size - 1)) - minus;
  uint32_t t = max_log ? max_log : HUF_LOG_DEFAULT;
  uint32_t a = hb32((uint32_t)src_size) + 1, b = hb32(max_sv) + 2;
  uint32_t min_bits = a < b ? a : b;
  if (max_bits_src < t) t = max_bits_src;
  if (min_bits > t) t = min_bits;
  if (t < FSE_LOG_MIN) t = FSE_LOG_MIN;
  if (t > FSE_LOG_MAX) t = FSE_LOG_MAX;
  return t;
}

/* how often the secondary normalisation ran (tests use it to show that their inputs reach that path; not thread-safe,
   so a lower bound when zo_compress_frame runs on several threads) */
static unsigned long g_m2_calls;
unsigned long zo_debug_m2_calls(void) { return g_m2_calls; }

/* secondary normalisation (FSE_normalizeM2) */
static size_t fse_normalize_m2(short* norm, unsigned tl, const unsigned* count, size_t total,
                               unsigned max_sv, short low_prob) {
  const short UNSET = -2;
  uint32_t s, distributed = 0, to_dist;
  uint32_t low_thr = (uint32_t)(total >> tl);
  uint32_t low_one = (uint32_t)((total * 3) >> (tl + 1));
  for (s = 0; s <= max_sv; s++) {
    if (count[s] == 0) { norm[s] = 0; continue; }
    if (count[s] <= low_thr) { norm[s] = low_prob; distributed++; total -= count[s]; continue; }
    if (count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; continue; }
    norm[s] = UNSET;
  }
  to_dist = (1u << tl) - distributed;
  if (to_dist == 0) return 0;
  if ((total / to_dist) > low_one) {
    low_one = (uint32_t)((total * 3) / (to_dist * 2));
    for (s = 0; s <= max_sv; s++)
      if (norm[s] == UNSET && count[s] <= low_one) { norm[s] = 1; distributed++; total -= count[s]; }
    to_dist = (1u << tl) - distributed;
  }
  if (distributed == max_sv + 1) {
    uint32_t best = 0, best_c = 0;
    for (s = 0; s <= max_sv; s++) if (count[s] > best_c) { best = s; best_c = count[s]; }
    norm[best] += (short)to_dist;
    return 0;
  }
  if (total == 0) {
    for (s = 0; to_dist > 0; s = (s + 1) % (max_sv + 1))
      if (norm[s] > 0) { to_dist--; norm[s]++; }
    return 0;
  }
  {
    uint64_t vlog = 62 - tl, mid = (1ULL << (vlog - 1)) - 1;
    uint64_t rstep = (((1ULL << vlog) * to_dist) + mid) / (uint32_t)total;
    uint64_t run = mid;
    for (s = 0; s <= max_sv; s++) {
      if (norm[s] != UNSET) continue;
      uint64_t end = run + (uint64_t)count[s] * rstep;
      uint32_t w = (uint32_t)(end >> vlog) - (uint32_t)(run >> vlog);
      if (w < 1) return ZO_ERR_GENERIC;
      norm[s] = (short)w; run = end;
    }
  }
  return 0;
}

size_t zo_fse_normalize_count(short* norm, unsigned tl, const unsigned* count, size_t total,
                              unsigned max_sv, int low_prob_i) {
  static const uint32_t rtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
  const short low_prob = (short)low_prob_i;
  if (tl == 0) tl = 11;
  if (tl < FSE_LOG_MIN) return ZO_ERR_GENERIC;
  if (tl > FSE_LOG_MAX) return ZO_ERR_TABLELOG_TOO_LARGE;
  {
    uint32_t a = hb32((uint32_t)total) + 1, b = hb32(max_sv) + 2;
    if (tl < (a < b ? a : b)) return ZO_ERR_GENERIC;
  }
  uint64_t scale = 62 - tl, step = (1ULL << 62) / (uint32_t)total, vstep = 1ULL << (scale - 20);
  int still = 1 << tl;
  unsigned s, largest = 0;
  short largest_p = 0;
  uint32_t low_thr = (uint32_t)(total >> tl);
  for (s = 0; s <= max_sv; s++) {
    if (count[s] == total) return 0; /* rle */
    if (count[s] == 0) { norm[s] = 0; continue; }
    if (count[s] <= low_thr) { norm[s] = low_prob; still--; continue; }
    short p = (short)(((uint64_t)count[s] * step) >> scale);
    if (p < 8) {
      uint64_t beat = vstep * rtb[p];
      p += ((uint64_t)count[s] * step) - ((uint64_t)p << scale) > beat;
    }
    if (p > largest_p) { largest_p = p; largest = s; }
    norm[s] = p; still -= p;
  }
  if (-still >= (norm[largest] >> 1)) {
    size_t e;
    g_m2_calls++;
    e = fse_normalize_m2(norm, tl, count, total, max_sv, low_prob);
    if (zo_huf_is_error(e)) return e;
  } else {
    norm[largest] += (short)still;
  }
  return tl;
}

/* FSE_writeNCount: returns bytes written (dst assumed large enough) */
static size_t fse_write_ncount(uint8_t* out, const short* norm, unsigned max_sv, unsigned tl) {
  uint8_t* const o0 = out;
  const int table_size = 1 << tl;
  int nb_bits = (int)tl + 1, remaining = table_size + 1, threshold = table_size;
  uint32_t bits = 0; int nbit = 0;
  unsigned sym = 0; const unsigned alpha = max_sv + 1;
  int prev0 = 0;
  bits += (tl - FSE_LOG_MIN) << nbit; nbit += 4;
  while (sym < alpha && remaining > 1) {
    if (prev0) {
      unsigned start = sym;
      while (sym < alpha && !norm[sym]) sym++;
      if (sym == alpha) break;
      while (sym >= start + 24) {
        start += 24; bits += 0xFFFFu << nbit;
        out[0] = (uint8_t)bits; out[1] = (uint8_t)(bits >> 8); out += 2; bits >>= 16;
      }
      while (sym >= start + 3) { start += 3; bits += 3u << nbit; nbit += 2; }
      bits += (sym - start) << nbit; nbit += 2;
      if (nbit > 16) { out[0] = (uint8_t)bits; out[1] = (uint8_t)(bits >> 8); out += 2; bits >>= 16; nbit -= 16; }
    }
    {
      int c = norm[sym++];
      const int mx = (2 * threshold - 1) - remaining;
      remaining -= c < 0 ? -c : c;
      c++;
      if (c >= threshold) c += mx;
      bits += (uint32_t)c << nbit; nbit += nb_bits; nbit -= (c < mx);
      prev0 = (c == 1);
      if (remaining < 1) return ZO_ERR_GENERIC;
      while (remaining < threshold) { nb_bits--; threshold >>= 1; }
    }
    if (nbit > 16) { out[0] = (uint8_t)bits; out[1] = (uint8_t)(bits >> 8); out += 2; bits >>= 16; nbit -= 16; }
  }
  if (remaining != 1) return ZO_ERR_GENERIC;
  out[0] = (uint8_t)bits; out[1] = (uint8_t)(bits >> 8);
  out += (nbit + 7) / 8;
  return (size_t)(out - o0);
}

typedef struct { int delta_find_state; uint32_t delta_nb_bits; } fse_tt_t;

/* encode table for an alphabet of <= 13 symbols, table log <= 6 */
static void fse_build_ctable(uint16_t* state_tab, fse_tt_t* tt, const short* norm, unsigned max_sv,
                             unsigned tl) {
  const uint32_t size = 1u << tl, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
  uint32_t cumul[HUF_LOG_MAX + 3];
  uint8_t cell_sym[1u << WEIGHT_FSE_LOG];
  uint32_t high = size - 1, u, s;
  cumul[0] = 0;
  for (u = 1; u <= max_sv + 1; u++) {
    if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; cell_sym[high--] = (uint8_t)(u - 1); }
    else cumul[u] = cumul[u - 1] + (uint32_t)norm[u - 1];
  }
  cumul[max_sv + 1] = size + 1;
  {
    uint32_t pos = 0;
    for (s = 0; s <= max_sv; s++) {
      int i, f = norm[s];
      for (i = 0; i < f; i++) {
        cell_sym[pos] = (uint8_t)s;
        pos = (pos + step) & mask;
        while (pos > high) pos = (pos + step) & mask;
      }
    }
  }
  for (u = 0; u < size; u++) { s = cell_sym[u]; state_tab[cumul[s]++] = (uint16_t)(size + u); }
  {
    int total = 0;
    for (s = 0; s <= max_sv; s++) {
      int f = norm[s];
      if (f == 0) { tt[s].delta_nb_bits = ((tl + 1) << 16) - (1u << tl); tt[s].delta_find_state = 0; }
      else if (f == -1 || f == 1) { tt[s].delta_nb_bits = (tl << 16) - (1u << tl); tt[s].delta_find_state = total - 1; total++; }
      else {
        uint32_t max_bits_out = tl - hb32((uint32_t)f - 1);
        uint32_t min_state_plus = (uint32_t)f << max_bits_out;
        tt[s].delta_nb_bits = (max_bits_out << 16) - min_state_plus;
        tt[s].delta_find_state = total - f; total += f;
      }
    }
  }
}

static uint32_t fse_state_init(const uint16_t* st, const fse_tt_t* tt, unsigned sym) {
  uint32_t nb = (tt[sym].delta_nb_bits + (1u << 15)) >> 16;
  uint32_t v = (nb << 16) - tt[sym].delta_nb_bits;
  return st[(int)(v >> nb) + tt[sym].delta_find_state];
}
static uint32_t fse_encode(bitw_t* w, const uint16_t* st, const fse_tt_t* tt, uint32_t state, unsigned sym) {
  uint32_t nb = (state + tt[sym].delta_nb_bits) >> 16;
  bitw_add(w, state, nb);
  return st[(int)(state >> nb) + tt[sym].delta_find_state];
}

/* two-state backward FSE encode of src[0..n) */
static size_t fse_compress_with(uint8_t* dst, size_t cap, const uint8_t* src, size_t n,
                                const uint16_t* st, const fse_tt_t* tt, unsigned tl) {
  bitw_t w; size_t i = n; uint32_t s1, s2;
  if (n <= 2) return 0;
  if (!bitw_init(&w, dst, cap)) return 0;
  if (n & 1) { s1 = fse_state_init(st, tt, src[--i]); s2 = fse_state_init(st, tt, src[--i]); s1 = fse_encode(&w, st, tt, s1, src[--i]); }
  else       { s2 = fse_state_init(st, tt, src[--i]); s1 = fse_state_init(st, tt, src[--i]); }
  while (i > 0) { s2 = fse_encode(&w, st, tt, s2, src[--i]); s1 = fse_encode(&w, st, tt, s1, src[--i]); }
  bitw_add(&w, s2, tl); bitw_add(&w, s1, tl);
  return bitw_close(&w);
}

/* How the weight coder writes a count that rounds below one table cell.  zstd >= 1.4.7 (this oracle's pin, and what the
 * product's encoder emits) passes useLowProbCount = 0 to FSE_normalizeCount from HUF_compressWeights: such a weight gets a
 * full cell (+1).  The huff0 of the FiniteStateEntropy library the reference's PyPI wheels are built from
 * (/root/reference/setup.py:23-28, .gitmodules:4-6; un-vendored) predates that parameter and always writes the
 * "less than one" marker (-1): the symbol gets the table's top cell.  Both forms decode with every huff0 decoder; only the
 * tree-description bytes differ.  -1 here makes the oracle write what a real wheel writes (tests of the decoders only). */
static int g_weight_low_prob = +1;
void zo_set_weight_low_prob(int v) { g_weight_low_prob = (v < 0) ? -1 : +1; }
int zo_get_weight_low_prob(void) { return g_weight_low_prob; }

/* HUF_compressWeights: 0 = not compressible, 1 = single value, else size / error */
static size_t huf_compress_weights(uint8_t* dst, size_t cap, const uint8_t* w, size_t nw) {
  unsigned count[HUF_LOG_MAX + 1] = {0};
  short norm[HUF_LOG_MAX + 1];
  unsigned max_sv = HUF_LOG_MAX, max_c = 0, tl; size_t i;
  uint16_t st[1u << WEIGHT_FSE_LOG]; fse_tt_t tt[HUF_LOG_MAX + 1];
  uint8_t* op = dst;
  if (nw <= 1) return 0;
  for (i = 0; i < nw; i++) count[w[i]]++;
  while (count[max_sv] == 0) max_sv--;
  for (i = 0; i <= max_sv; i++) if (count[i] > max_c) max_c = count[i];
  if (max_c == nw) return 1;
  if (max_c == 1) return 0;
  tl = zo_optimal_table_log(WEIGHT_FSE_LOG, nw, max_sv, 2);
  { size_t e = zo_fse_normalize_count(norm, tl, count, nw, max_sv, g_weight_low_prob); if (zo_huf_is_error(e)) return e; }
  { size_t h = fse_write_ncount(op, norm, max_sv, tl); if (zo_huf_is_error(h)) return h; op += h; }
  fse_build_ctable(st, tt, norm, max_sv, tl);
  { size_t c = fse_compress_with(op, cap - (size_t)(op - dst), w, nw, st, tt, tl); if (c == 0) return 0; op += c; }
  return (size_t)(op - dst);
}

/* ------------------------------------------------------------------------- */
/* huff0 encoder                                                               */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nb; } hnode_t;

/* order: count descending, equal counts keep ascending symbol order */
static void huf_sort(hnode_t* node, const unsigned* count, unsigned max_sv) {
  uint32_t base[33] = {0}, cur[33]; unsigned n;
  for (n = 0; n <= max_sv; n++) base[hb32(count[n] + 1)]++;
  for (n = 30; n > 0; n--) base[n - 1] += base[n];
  memcpy(cur, base, sizeof(cur));
  for (n = 0; n <= max_sv; n++) {
    uint32_t c = count[n], r = hb32(c + 1) + 1, pos = cur[r]++;
    while (pos > base[r] && c > node[pos - 1].count) { node[pos] = node[pos - 1]; pos--; }
    node[pos].count = c; node[pos].byte = (uint8_t)n;
  }
}

static unsigned huf_limit_height(hnode_t* node, unsigned last, unsigned max_nb) {
  const unsigned largest = node[last].nb;
  if (largest <= max_nb) return largest;
  int cost = 0; const unsigned base_cost = 1u << (largest - max_nb);
  int n = (int)last;
  while (node[n].nb > max_nb) { cost += (int)(base_cost - (1u << (largest - node[n].nb))); node[n].nb = (uint8_t)max_nb; n--; }
  while (node[n].nb == max_nb) n--;
  cost >>= (largest - max_nb);
  {
    const uint32_t NONE = 0xF0F0F0F0u;
    uint32_t rank_last[HUF_LOG_MAX + 2]; unsigned i;
    for (i = 0; i < HUF_LOG_MAX + 2; i++) rank_last[i] = NONE;
    { unsigned cur = max_nb; int pos;
      for (pos = n; pos >= 0; pos--) { if (node[pos].nb >= cur) continue; cur = node[pos].nb; rank_last[max_nb - cur] = (uint32_t)pos; } }
    while (cost > 0) {
      unsigned d = hb32((uint32_t)cost) + 1;
      for (; d > 1; d--) {
        uint32_t hp = rank_last[d], lp = rank_last[d - 1];
        if (hp == NONE) continue;
        if (lp == NONE) break;
        if (node[hp].count <= 2 * node[lp].count) break;
      }
      while (d <= HUF_LOG_MAX && rank_last[d] == NONE) d++;
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
  }
  return max_nb;
}

size_t zo_huf_build_ctable(const unsigned* count, unsigned max_sv, unsigned max_nb_bits,
                           uint8_t* nb_bits, uint16_t* val) {
  enum { START = 256 };
  hnode_t tab0[1 + 2 * 256]; hnode_t* node = tab0 + 1;
  int non_null, low_s, low_n, node_nb = START, root, n;
  if (max_nb_bits == 0) max_nb_bits = HUF_LOG_DEFAULT;
  if (max_sv > HUF_SYM_MAX) return ZO_ERR_GENERIC;
  memset(tab0, 0, sizeof(tab0));
  huf_sort(node, count, max_sv);
  non_null = (int)max_sv;
  while (node[non_null].count == 0) non_null--;
  low_s = non_null; root = node_nb + low_s - 1; low_n = node_nb;
  node[node_nb].count = node[low_s].count + node[low_s - 1].count;
  node[low_s].parent = node[low_s - 1].parent = (uint16_t)node_nb;
  node_nb++; low_s -= 2;
  for (n = node_nb; n <= root; n++) node[n].count = 1u << 30;
  tab0[0].count = 1u << 31; /* node[-1]: barrier below the smallest leaf */
  while (node_nb <= root) {
    int n1 = (node[low_s].count < node[low_n].count) ? low_s-- : low_n++;
    int n2 = (node[low_s].count < node[low_n].count) ? low_s-- : low_n++;
    node[node_nb].count = node[n1].count + node[n2].count;
    node[n1].parent = node[n2].parent = (uint16_t)node_nb;
    node_nb++;
  }
  node[root].nb = 0;
  for (n = root - 1; n >= START; n--) node[n].nb = (uint8_t)(node[node[n].parent].nb + 1);
  for (n = 0; n <= non_null; n++) node[n].nb = (uint8_t)(node[node[n].parent].nb + 1);
  max_nb_bits = huf_limit_height(node, (unsigned)non_null, max_nb_bits);
  {
    uint16_t per_rank[HUF_LOG_MAX + 1] = {0}, val_rank[HUF_LOG_MAX + 1] = {0};
    if (max_nb_bits > HUF_LOG_MAX) return ZO_ERR_GENERIC;
    for (n = 0; n <= non_null; n++) per_rank[node[n].nb]++;
    { uint16_t mn = 0; for (n = (int)max_nb_bits; n > 0; n--) { val_rank[n] = mn; mn = (uint16_t)(mn + per_rank[n]); mn >>= 1; } }
    for (n = 0; n <= (int)max_sv; n++) nb_bits[node[n].byte] = node[n].nb;
    for (n = 0; n <= (int)max_sv; n++) val[n] = val_rank[nb_bits[n]]++;
  }
  return max_nb_bits;
}

size_t zo_huf_write_ctable(void* dst, size_t cap, const uint8_t* nb_bits, unsigned max_sv,
                           unsigned huff_log) {
  uint8_t to_weight[HUF_LOG_MAX + 1]; uint8_t w[HUF_SYM_MAX + 1];
  uint8_t* op = (uint8_t*)dst; unsigned n;
  if (max_sv > HUF_SYM_MAX) return ZO_ERR_GENERIC;
  to_weight[0] = 0;
  for (n = 1; n < huff_log + 1; n++) to_weight[n] = (uint8_t)(huff_log + 1 - n);
  for (n = 0; n < max_sv; n++) w[n] = to_weight[nb_bits[n]];
  {
    size_t h = huf_compress_weights(op + 1, cap - 1, w, max_sv);
    if (zo_huf_is_error(h)) return h;
    if (h > 1 && h < max_sv / 2) { op[0] = (uint8_t)h; return h + 1; }
  }
  if (max_sv > 128) return ZO_ERR_GENERIC;
  if (((max_sv + 1) / 2) + 1 > cap) return ZO_ERR_DST_TOO_SMALL;
  op[0] = (uint8_t)(128 + (max_sv - 1));
  w[max_sv] = 0;
  for (n = 0; n < max_sv; n += 2) op[n / 2 + 1] = (uint8_t)((w[n] << 4) + w[n + 1]);
  return ((max_sv + 1) / 2) + 1;
}

/* one backward bit-stream: codes of src[n-1], src[n-2], ... src[0], end mark, pad */
static size_t huf_encode_stream(uint8_t* dst, size_t cap, const uint8_t* src, size_t n,
                                const uint8_t* nb, const uint16_t* val) {
  bitw_t w; size_t i;
  if (cap < 8) return 0;
  if (!bitw_init(&w, dst, cap)) return 0;
  for (i = n; i-- > 0;) bitw_add(&w, val[src[i]], nb[src[i]]);
  return bitw_close(&w);
}

static size_t huf_encode_4x(uint8_t* dst, size_t cap, const uint8_t* src, size_t n,
                            const uint8_t* nb, const uint16_t* val) {
  const size_t seg = (n + 3) / 4;
  uint8_t* op = dst; uint8_t* const oend = dst + cap; int k;
  if (cap < 6 + 1 + 1 + 1 + 8) return 0;
  if (n < 12) return 0;
  op += 6;
  for (k = 0; k < 4; k++) {
    size_t len = (k < 3) ? seg : n - 3 * seg;
    size_t c = huf_encode_stream(op, (size_t)(oend - op), src + (size_t)k * seg, len, nb, val);
    if (c == 0) return 0;
    if (k < 3) { dst[2 * k] = (uint8_t)c; dst[2 * k + 1] = (uint8_t)(c >> 8); }
    op += c;
  }
  return (size_t)(op - dst);
}

size_t zo_huf_compress(void* dst, size_t cap, const void* src_, size_t n) {
  const uint8_t* src = (const uint8_t*)src_;
  uint8_t* const o0 = (uint8_t*)dst; uint8_t* op = o0; uint8_t* const oend = o0 + cap;
  unsigned count[256] = {0}; unsigned max_sv = 255, largest = 0, huff_log; size_t i;
  uint8_t nb[256]; uint16_t val[256];
  if (n == 0) return 0;
  if (cap == 0) return 0;
  if (n > HUF_BLOCK_MAX) return ZO_ERR_SRCSIZE_WRONG;
  for (i = 0; i < n; i++) count[src[i]]++;
  while (count[max_sv] == 0) max_sv--;
  for (i = 0; i <= max_sv; i++) if (count[i] > largest) largest = count[i];
  if (largest == n) { o0[0] = src[0]; return 1; }
  if (largest <= (n >> 7) + 4) return 0;
  huff_log = zo_optimal_table_log(HUF_LOG_DEFAULT, n, max_sv, 1);
  {
    size_t mb = zo_huf_build_ctable(count, max_sv, huff_log, nb, val);
    if (zo_huf_is_error(mb)) return mb;
    huff_log = (unsigned)mb;
  }
  {
    size_t h = zo_huf_write_ctable(op, cap, nb, max_sv, huff_log);
    if (zo_huf_is_error(h)) return h;
    if (h + 12 >= n) return 0;
    op += h;
  }
  {
    size_t c = huf_encode_4x(op, (size_t)(oend - op), src, n, nb, val);
    if (c == 0) return 0;
    op += c;
  }
  if ((size_t)(op - o0) >= n - 1) return 0;
  return (size_t)(op - o0);
}

/* ------------------------------------------------------------------------- */
/* huff0 decoder                                                               */
/* ------------------------------------------------------------------------- */
/* backward bit reader: bits [0,pos) are unread; zero-fill below bit 0 */
typedef struct { const uint8_t* p; size_t nbytes; int64_t pos; } bitr_t;

static int bitr_init(bitr_t* r, const uint8_t* p, size_t n) {
  if (n < 1) return -1;
  if (p[n - 1] == 0) return -1; /* end mark missing */
  r->p = p; r->nbytes = n; r->pos = (int64_t)(n - 1) * 8 + hb32(p[n - 1]);
  return 0;
}
/* value of the nb bits just below pos (MSB = bit pos-1) */
static uint32_t bitr_peek(const bitr_t* r, unsigned nb) {
  uint32_t v = 0; unsigned k;
  for (k = 0; k < nb; k++) {
    int64_t b = r->pos - 1 - (int64_t)k;
    uint32_t bit = (b >= 0) ? ((r->p[b >> 3] >> (b & 7)) & 1u) : 0u;
    v = (v << 1) | bit;
  }
  return v;
}

static size_t fse_decode_weights(uint8_t* out, size_t max_out, const uint8_t* src, size_t n) {
  /* FSE_readNCount */
  short norm[256]; unsigned nsym = 0, tl; size_t hdr_bytes;
  {
    uint64_t bitpos = 0; const uint64_t nbits_total = (uint64_t)n * 8;
    #define RD(nb_) ({ uint32_t v_ = 0; unsigned k_; for (k_ = 0; k_ < (nb_); k_++) { uint64_t b_ = bitpos + k_; uint32_t bit_ = (b_ < nbits_total) ? ((src[b_ >> 3] >> (b_ & 7)) & 1u) : 0u; v_ |= bit_ << k_; } v_; })
    int remaining, threshold, nb_bits, prev0 = 0;
    if (n < 1) return ZO_ERR_SRCSIZE_WRONG;
    tl = RD(4) + FSE_LOG_MIN; bitpos += 4;
    if (tl > 15) return ZO_ERR_TABLELOG_TOO_LARGE;
    remaining = (1 << tl) + 1; threshold = 1 << tl; nb_bits = (int)tl + 1;
    while (remaining > 1 && nsym <= 255) {
      if (prev0) {
        unsigned n0 = nsym;
        while (RD(16) == 0xFFFF) { n0 += 24; bitpos += 16; if (bitpos > nbits_total + 64) return ZO_ERR_CORRUPTION; }
        while (RD(2) == 3) { n0 += 3; bitpos += 2; if (bitpos > nbits_total + 64) return ZO_ERR_CORRUPTION; }
        n0 += RD(2); bitpos += 2;
        if (n0 > 255) return ZO_ERR_GENERIC;
        while (nsym < n0) norm[nsym++] = 0;
      }
      {
        const int mx = (2 * threshold - 1) - remaining; int c;
        if ((int)RD((unsigned)nb_bits - 1) < mx) { c = (int)RD((unsigned)nb_bits - 1); bitpos += (unsigned)nb_bits - 1; }
        else { c = (int)RD((unsigned)nb_bits); if (c >= threshold) c -= mx; bitpos += (unsigned)nb_bits; }
        c--;
        remaining -= c < 0 ? -c : c;
        norm[nsym++] = (short)c; prev0 = !c;
        while (remaining < threshold) { nb_bits--; threshold >>= 1; }
      }
    }
    #undef RD
    if (remaining != 1) return ZO_ERR_CORRUPTION;
    if (bitpos > nbits_total) return ZO_ERR_CORRUPTION;
    hdr_bytes = (size_t)((bitpos + 7) >> 3);
  }
  if (tl > WEIGHT_FSE_LOG) return ZO_ERR_TABLELOG_TOO_LARGE;
  /* FSE_buildDTable */
  {
    const uint32_t size = 1u << tl, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint8_t cell[1u << WEIGHT_FSE_LOG]; uint16_t next[256];
    uint8_t d_sym[1u << WEIGHT_FSE_LOG], d_nb[1u << WEIGHT_FSE_LOG]; uint16_t d_base[1u << WEIGHT_FSE_LOG];
    uint32_t high = size - 1, pos = 0, u; unsigned s;
    for (s = 0; s < nsym; s++) { if (norm[s] == -1) { cell[high--] = (uint8_t)s; next[s] = 1; } else next[s] = (uint16_t)norm[s]; }
    for (s = 0; s < nsym; s++) { int i; for (i = 0; i < norm[s]; i++) { cell[pos] = (uint8_t)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; } }
    if (pos != 0) return ZO_ERR_GENERIC;
    for (u = 0; u < size; u++) {
      uint32_t ns = next[cell[u]]++; unsigned nb = tl - hb32(ns);
      d_sym[u] = cell[u]; d_nb[u] = (uint8_t)nb; d_base[u] = (uint16_t)((ns << nb) - size);
    }
    /* FSE_decompress_usingDTable, two interleaved states, backward stream */
    {
      bitr_t r; size_t o = 0; uint32_t s1, s2;
      if (bitr_init(&r, src + hdr_bytes, n - hdr_bytes)) return ZO_ERR_CORRUPTION;
      s1 = bitr_peek(&r, tl); r.pos -= tl;
      s2 = bitr_peek(&r, tl); r.pos -= tl;
      if (r.pos < 0) return ZO_ERR_CORRUPTION;
      for (;;) {
        if (o >= max_out) return ZO_ERR_DST_TOO_SMALL;
        out[o++] = d_sym[s1];
        { unsigned nb = d_nb[s1]; uint32_t v = bitr_peek(&r, nb); r.pos -= nb; s1 = d_base[s1] + v; }
        if (r.pos < 0) { if (o >= max_out) return ZO_ERR_DST_TOO_SMALL; out[o++] = d_sym[s2]; break; }
        if (o >= max_out) return ZO_ERR_DST_TOO_SMALL;
        out[o++] = d_sym[s2];
        { unsigned nb = d_nb[s2]; uint32_t v = bitr_peek(&r, nb); r.pos -= nb; s2 = d_base[s2] + v; }
        if (r.pos < 0) { if (o >= max_out) return ZO_ERR_DST_TOO_SMALL; out[o++] = d_sym[s1]; break; }
      }
      return o;
    }
  }
}

size_t zo_huf_read_stats(uint8_t* w, unsigned* n_sym, unsigned* table_log, const void* src_,
                         size_t src_size) {
  const uint8_t* ip = (const uint8_t*)src_; size_t isz, osz; uint32_t total = 0, n;
  unsigned rank1 = 0;
  if (!src_size) return ZO_ERR_SRCSIZE_WRONG;
  isz = ip[0];
  if (isz >= 128) {
    osz = isz - 127; isz = (osz + 1) / 2;
    if (isz + 1 > src_size) return ZO_ERR_SRCSIZE_WRONG;
    if (osz >= 256) return ZO_ERR_CORRUPTION;
    for (n = 0; n < osz; n += 2) { w[n] = ip[1 + n / 2] >> 4; w[n + 1] = ip[1 + n / 2] & 15; }
  } else {
    if (isz + 1 > src_size) return ZO_ERR_SRCSIZE_WRONG;
    osz = fse_decode_weights(w, 255, ip + 1, isz);
    if (zo_huf_is_error(osz)) return osz;
  }
  for (n = 0; n < osz; n++) {
    if (w[n] >= HUF_LOG_MAX) return ZO_ERR_CORRUPTION;
    total += (1u << w[n]) >> 1; rank1 += (w[n] == 1);
  }
  if (total == 0) return ZO_ERR_CORRUPTION;
  {
    unsigned tl = hb32(total) + 1; uint32_t rest, last;
    if (tl > HUF_LOG_MAX) return ZO_ERR_CORRUPTION;
    rest = (1u << tl) - total;
    if ((1u << hb32(rest)) != rest) return ZO_ERR_CORRUPTION;
    last = hb32(rest) + 1;
    w[osz] = (uint8_t)last; rank1 += (last == 1);
    *table_log = tl;
  }
  if (rank1 < 2 || (rank1 & 1)) return ZO_ERR_CORRUPTION;
  *n_sym = (unsigned)(osz + 1);
  return isz + 1;
}

size_t zo_huf_decompress(void* dst_, size_t dst_size, const void* csrc_, size_t csize) {
  uint8_t* dst = (uint8_t*)dst_; const uint8_t* src = (const uint8_t*)csrc_;
  uint8_t w[256]; unsigned nsym, tl; size_t hs;
  uint8_t lut_sym[1u << HUF_LOG_MAX], lut_nb[1u << HUF_LOG_MAX];
  if (dst_size == 0) return ZO_ERR_DST_TOO_SMALL;
  if (csize > dst_size) return ZO_ERR_CORRUPTION;
  if (csize == dst_size) { memcpy(dst, src, dst_size); return dst_size; }
  if (csize == 1) { memset(dst, src[0], dst_size); return dst_size; }
  hs = zo_huf_read_stats(w, &nsym, &tl, src, csize);
  if (zo_huf_is_error(hs)) return hs;
  if (hs >= csize) return ZO_ERR_SRCSIZE_WRONG;
  {
    /* cells in order of ascending weight (longest codes first), ascending symbol inside */
    uint32_t next = 0; unsigned wv, s;
    for (wv = 1; wv <= tl; wv++)
      for (s = 0; s < nsym; s++)
        if (w[s] == wv) { uint32_t len = (1u << wv) >> 1, u; for (u = 0; u < len; u++) { lut_sym[next + u] = (uint8_t)s; lut_nb[next + u] = (uint8_t)(tl + 1 - wv); } next += len; }
    if (next != (1u << tl)) return ZO_ERR_CORRUPTION;
  }
  src += hs; csize -= hs;
  if (csize < 10) return ZO_ERR_CORRUPTION;
  {
    size_t l1 = src[0] | (src[1] << 8), l2 = src[2] | (src[3] << 8), l3 = src[4] | (src[5] << 8);
    size_t l4, seg = (dst_size + 3) / 4, lens[4], outs[4]; const uint8_t* ps[4]; int k;
    if (l1 + l2 + l3 + 6 > csize) return ZO_ERR_CORRUPTION;
    l4 = csize - (l1 + l2 + l3 + 6);
    if (3 * seg > dst_size) return ZO_ERR_CORRUPTION;
    lens[0] = l1; lens[1] = l2; lens[2] = l3; lens[3] = l4;
    outs[0] = outs[1] = outs[2] = seg; outs[3] = dst_size - 3 * seg;
    ps[0] = src + 6; ps[1] = ps[0] + l1; ps[2] = ps[1] + l2; ps[3] = ps[2] + l3;
    for (k = 0; k < 4; k++) {
      bitr_t r; size_t i; uint8_t* o = dst + (size_t)k * seg;
      if (bitr_init(&r, ps[k], lens[k])) return ZO_ERR_CORRUPTION;
      for (i = 0; i < outs[k]; i++) { uint32_t idx = bitr_peek(&r, tl); o[i] = lut_sym[idx]; r.pos -= lut_nb[idx]; }
      if (r.pos != 0) return ZO_ERR_CORRUPTION;
    }
  }
  return dst_size;
}

/* ------------------------------------------------------------------------- */
/* byte-plane transforms                                                       */
/* ------------------------------------------------------------------------- */
static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static void st32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }

void zo_rotate_fwd(uint8_t* buf, size_t len, int num_buf) {
  size_t i, nw = len / 4;
  for (i = 0; i < nw; i++) {
    uint32_t u = ld32(buf + 4 * i);
    if (num_buf == 2) u = ((u << 1) & 0xFF00FF00u) | ((u >> 8) & 0x00800080u) | (u & 0x007F007Fu);
    else              u = ((u << 1) & 0xFF000000u) | ((u >> 8) & 0x00800000u) | (u & 0x007FFFFFu);
    st32(buf + 4 * i, u);
  }
}
void zo_rotate_inv(uint8_t* buf, size_t len, int num_buf) {
  size_t i, nw = len / 4;
  for (i = 0; i < nw; i++) {
    uint32_t u = ld32(buf + 4 * i);
    if (num_buf == 2) u = ((u << 8) & 0x80008000u) | ((u >> 1) & 0x7F807F80u) | (u & 0x007F007Fu);
    else              u = ((u << 8) & 0x80000000u) | ((u >> 1) & 0x7F800000u) | (u & 0x007FFFFFu);
    st32(buf + 4 * i, u);
  }
}
void zo_plane_lens(size_t len, int num_buf, size_t* lens) {
  int p; for (p = 0; p < num_buf; p++) lens[p] = len / (size_t)num_buf + ((size_t)p < len % (size_t)num_buf);
}

/* ------------------------------------------------------------------------- */
/* frame level                                                                 */
/* ------------------------------------------------------------------------- */
size_t zo_compress_bound(size_t n, int num_buf, size_t chunk, size_t hdr_len) {
  size_t k = chunk ? (n + chunk - 1) / chunk : 0;
  return hdr_len + 9 * (size_t)num_buf * k + n;
}

typedef struct {
  const uint8_t* src; size_t n; int P, bits_mode; size_t chunk; double thr;
  size_t K; uint8_t* types; uint32_t* sizes; uint8_t** payload; /* [P*K] */
  size_t next; pthread_mutex_t mu; int err;
} cjob_t;

static void compress_one_chunk(cjob_t* j, size_t c) {
  const size_t off = c * j->chunk;
  const size_t len = (c == j->K - 1) ? j->n - off : j->chunk;
  const int P = j->P; size_t lens[4]; int p; size_t i;
  uint8_t* tmp = (uint8_t*)malloc(len ? len : 1);
  uint8_t* plane = (uint8_t*)malloc(len ? len : 1);
  uint8_t* cbuf = (uint8_t*)malloc(j->chunk ? j->chunk : 1);
  if (!tmp || !plane || !cbuf) { j->err = -2; free(tmp); free(plane); free(cbuf); return; }
  memcpy(tmp, j->src + off, len);
  if (j->bits_mode == 1 && P > 1) zo_rotate_fwd(tmp, len, P);
  zo_plane_lens(len, P, lens);
  for (p = 0; p < P; p++) {
    size_t pl = lens[p], idx = (size_t)p * j->K + c; uint32_t cs;
    for (i = 0; i < pl; i++) plane[i] = tmp[i * (size_t)P + (size_t)p];
    /* dst capacity is the chunk size, as at csrc/zipnn_core.c:366-368; the result is
     * truncated to 32 bits (:280,365) so huff0 error codes land on "store raw". */
    cs = (uint32_t)zo_huf_compress(cbuf, j->chunk, plane, pl);
    if (cs != 0 && (double)cs < (double)pl * j->thr) {
      j->types[idx] = 1; j->sizes[idx] = cs; j->payload[idx] = (uint8_t*)malloc(cs);
      if (!j->payload[idx]) { j->err = -2; break; }
      memcpy(j->payload[idx], cbuf, cs);
    } else {
      j->types[idx] = 0; j->sizes[idx] = (uint32_t)pl; j->payload[idx] = (uint8_t*)malloc(pl ? pl : 1);
      if (!j->payload[idx]) { j->err = -2; break; }
      memcpy(j->payload[idx], plane, pl);
    }
  }
  free(tmp); free(plane); free(cbuf);
}

static void* compress_worker(void* a) {
  cjob_t* j = (cjob_t*)a;
  for (;;) {
    size_t c;
    pthread_mutex_lock(&j->mu); c = j->next++; pthread_mutex_unlock(&j->mu);
    if (c >= j->K) break;
    compress_one_chunk(j, c);
  }
  return NULL;
}

int zo_compress_frame(const uint8_t* hdr, size_t hdr_len, const uint8_t* src, size_t n, int P,
                      int bits_mode, int bytes_mode, size_t chunk, float threshold, int threads,
                      uint8_t* dst, size_t dst_cap, size_t* dst_len) {
  cjob_t j; size_t K, total, off, c; int p, t; int rc = 0;
  if (!(P == 1 || P == 2 || P == 4) || chunk == 0) return -1;
  if ((P == 4 && bytes_mode != 220) || (P != 4 && bytes_mode != 10)) return -1;
  K = (n + chunk - 1) / chunk;
  memset(&j, 0, sizeof(j));
  j.src = src; j.n = n; j.P = P; j.bits_mode = bits_mode; j.chunk = chunk;
  j.thr = (double)threshold; /* "f" in the arg format: float widened to double (zipnn_core.c:407,413) */
  j.K = K;
  j.types = (uint8_t*)calloc((size_t)P * K + 1, 1);
  j.sizes = (uint32_t*)calloc((size_t)P * K + 1, 4);
  j.payload = (uint8_t**)calloc((size_t)P * K + 1, sizeof(uint8_t*));
  pthread_mutex_init(&j.mu, NULL);
  if (!j.types || !j.sizes || !j.payload) { rc = -2; goto done; }
  if (threads <= 1) { for (c = 0; c < K; c++) compress_one_chunk(&j, c); }
  else {
    pthread_t th[64]; if (threads > 64) threads = 64;
    for (t = 0; t < threads; t++) pthread_create(&th[t], NULL, compress_worker, &j);
    for (t = 0; t < threads; t++) pthread_join(th[t], NULL);
  }
  if (j.err) { rc = j.err; goto done; }
  total = hdr_len + 9 * (size_t)P * K;
  for (c = 0; c < (size_t)P * K; c++) total += j.sizes[c];
  if (total > dst_cap) { rc = -3; goto done; }
  memcpy(dst, hdr, hdr_len);
  if (hdr_len >= 32) { uint64_t t64 = total; memcpy(dst + 24, &t64, 8); } /* zipnn_core.c:121 */
  off = hdr_len;
  memcpy(dst + off, j.types, (size_t)P * K); off += (size_t)P * K;
  for (p = 0; p < P; p++) { uint64_t cum = 0; for (c = 0; c < K; c++) { cum += j.sizes[(size_t)p * K + c]; memcpy(dst + off, &cum, 8); off += 8; } }
  for (c = 0; c < (size_t)P * K; c++) { memcpy(dst + off, j.payload[c], j.sizes[c]); off += j.sizes[c]; }
  *dst_len = off;
done:
  if (j.payload) for (c = 0; c < (size_t)P * K; c++) free(j.payload[c]);
  free(j.payload); free(j.types); free(j.sizes); pthread_mutex_destroy(&j.mu);
  return rc;
}

typedef struct {
  const uint8_t* body; int P, bits_mode; size_t chunk, orig, K;
  const uint8_t* types; const uint8_t* cum; const uint8_t* plane_base[4]; size_t plane_total[4];
  uint8_t* dst; size_t next; pthread_mutex_t mu; int err;
} djob_t;

static uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

static void decompress_one_chunk(djob_t* j, size_t c) {
  const int P = j->P; const size_t off = c * j->chunk;
  const size_t len = (c == j->K - 1) ? j->orig - off : j->chunk;
  size_t lens[4], i; int p;
  uint8_t* out = j->dst + off;
  uint8_t* plane = (uint8_t*)malloc(len ? len : 1);
  if (!plane) { j->err = -2; return; }
  zo_plane_lens(len, P, lens);
  for (p = 0; p < P; p++) {
    uint64_t hi = ld64(j->cum + 8 * ((size_t)p * j->K + c));
    uint64_t lo = c ? ld64(j->cum + 8 * ((size_t)p * j->K + c - 1)) : 0;
    const uint8_t* s; size_t clen = (size_t)(hi - lo); uint8_t ty = j->types[(size_t)p * j->K + c];
    if (hi < lo || hi > j->plane_total[p]) { j->err = -4; break; }
    s = j->plane_base[p] + lo;
    if (ty == 0) { if (clen < lens[p]) { j->err = -4; break; } for (i = 0; i < lens[p]; i++) out[i * (size_t)P + (size_t)p] = s[i]; }
    else {
      size_t r = lens[p] ? zo_huf_decompress(plane, lens[p], s, clen) : 0;
      if (zo_huf_is_error(r)) { j->err = -5; break; }
      for (i = 0; i < lens[p]; i++) out[i * (size_t)P + (size_t)p] = plane[i];
    }
  }
  if (!j->err && j->bits_mode == 1 && P > 1) zo_rotate_inv(out, len, P);
  free(plane);
}

static void* decompress_worker(void* a) {
  djob_t* j = (djob_t*)a;
  for (;;) {
    size_t c;
    pthread_mutex_lock(&j->mu); c = j->next++; pthread_mutex_unlock(&j->mu);
    if (c >= j->K) break;
    decompress_one_chunk(j, c);
  }
  return NULL;
}

int zo_decompress_body(const uint8_t* body, size_t body_len, int P, int bits_mode, int bytes_mode,
                       size_t chunk, size_t orig, int threads, uint8_t* dst) {
  djob_t j; size_t K, c, meta; int p, t;
  if (!(P == 1 || P == 2 || P == 4) || chunk == 0) return -1;
  if ((P == 4 && bytes_mode != 220) || (P != 4 && bytes_mode != 10)) return -1;
  K = (orig + chunk - 1) / chunk;
  meta = 9 * (size_t)P * K;
  if (body_len < meta) return -4;
  memset(&j, 0, sizeof(j));
  j.body = body; j.P = P; j.bits_mode = bits_mode; j.chunk = chunk; j.orig = orig; j.K = K; j.dst = dst;
  j.types = body; j.cum = body + (size_t)P * K;
  for (c = 0; c < (size_t)P * K; c++) if (j.types[c] > 1) return -6; /* zipnn_core.c:983-1000 */
  {
    const uint8_t* base = body + meta; size_t remain = body_len - meta;
    for (p = 0; p < P; p++) {
      uint64_t tot = K ? ld64(j.cum + 8 * ((size_t)p * K + K - 1)) : 0;
      if (tot > remain) return -4;
      j.plane_base[p] = base; j.plane_total[p] = (size_t)tot; base += tot; remain -= (size_t)tot;
    }
  }
  pthread_mutex_init(&j.mu, NULL);
  if (threads <= 1) { for (c = 0; c < K && !j.err; c++) decompress_one_chunk(&j, c); }
  else {
    pthread_t th[64]; if (threads > 64) threads = 64;
    for (t = 0; t < threads; t++) pthread_create(&th[t], NULL, decompress_worker, &j);
    for (t = 0; t < threads; t++) pthread_join(th[t], NULL);
  }
  pthread_mutex_destroy(&j.mu);
  return j.err;
}
