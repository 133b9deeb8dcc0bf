/*
 * zn_oracle.h — CPU ORACLE, TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the one ZipNN hot path this repository accelerates
 * (byte-plane split + per-plane huff0 coding over fixed-size chunks, and its
 * inverse).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (zipnn_amd/) never links, imports or
 * executes anything from oracle/.
 *
 * What it restates (reference file:line, all under /root/reference):
 *   - chunking, threshold rule, wire format      csrc/zipnn_core.c:105-244,294-390,401-702
 *   - metadata parse + per-chunk combine         csrc/zipnn_core.c:768-861,881-1164
 *   - 16-bit rotate/split/combine                csrc/data_manipulation_dtype16.c:10-29,64-138,145-216
 *   - 32-bit rotate/split/combine                csrc/data_manipulation_dtype32.c:39-58,78-133,275-294,391-456
 *   - fp8 passthrough                            csrc/data_manipulation_dtype16.c:33-58
 *   - HUF_compress / HUF_decompress / HUF_isError: third-party huff0 (Cyan4973/
 *     FiniteStateEntropy lib/, un-vendored submodule, version unpinned; call sites
 *     csrc/zipnn_core.c:366,807,813).  Restated from its published algorithm as
 *     shipped in zstd 1.4.8 (the libzstd.so.1 of this image, which exports the
 *     same entry points) — see SURVEY.md Appendix B.
 *
 * Parity pinning: this restatement is checked (tests/test_oracle.py) byte-for-byte
 * against (a) libzstd 1.4.8's exported HUF_compress/HUF_decompress and stage
 * functions, (b) oracle/_ref (the reference csrc/ compiled from where it lies,
 * linked to that libzstd), and (c) golden frames produced by the reference's own
 * Python package on top of (b) — tests/golden/.  The reference tree itself holds
 * no golden compressed vectors, so (b)/(c) are the pin.
 */
#ifndef ZN_ORACLE_H
#define ZN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- huff0 (return conventions of HUF_compress / HUF_decompress) ---- */
#define ZO_ERR_GENERIC            ((size_t)-1)
#define ZO_ERR_CORRUPTION         ((size_t)-20)
#define ZO_ERR_TABLELOG_TOO_LARGE ((size_t)-44)
#define ZO_ERR_DST_TOO_SMALL      ((size_t)-70)
#define ZO_ERR_SRCSIZE_WRONG      ((size_t)-72)

unsigned zo_huf_is_error(size_t code);
/* 0 = not compressible, 1 = RLE (dst[0] = the byte), >1 = compressed size, or an error code. */
size_t zo_huf_compress(void* dst, size_t dst_cap, const void* src, size_t n);
/* returns dst_size or an error code. */
size_t zo_huf_decompress(void* dst, size_t dst_size, const void* csrc, size_t csize);

/* ---- huff0 stages, exported for known-answer tests ---- */
unsigned zo_optimal_table_log(unsigned max_log, size_t src_size, unsigned max_sv, unsigned minus);
/* count[0..max_sv] -> nb_bits[0..max_sv], val[0..max_sv]; returns max code length or error. */
size_t zo_huf_build_ctable(const unsigned* count, unsigned max_sv, unsigned max_nb_bits,
                           uint8_t* nb_bits, uint16_t* val);
/* tree description (weights header); returns header size or error. */
size_t zo_huf_write_ctable(void* dst, size_t cap, const uint8_t* nb_bits, unsigned max_sv,
                           unsigned huff_log);
/* header -> weights[0..*n_sym), *table_log; returns header size consumed or error. */
size_t zo_huf_read_stats(uint8_t* weights, unsigned* n_sym, unsigned* table_log, const void* src,
                         size_t src_size);
/* FSE_normalizeCount of zstd 1.4.8 (low_prob = +1 for huff0 weights, -1 otherwise). */
size_t zo_fse_normalize_count(short* norm, unsigned table_log, const unsigned* count,
                              size_t total, unsigned max_sv, int low_prob);

/* Low-probability weight counts in the tree description: +1 (default; zstd 1.4.8, the pin) or -1 (the legacy
 * FiniteStateEntropy huff0 that the reference's PyPI wheels bundle).  Process-wide; set before compressing. */
void zo_set_weight_low_prob(int v);
int zo_get_weight_low_prob(void);
unsigned long zo_debug_m2_calls(void);   /* secondary normalisations so far (test coverage probe) */

/* ---- byte-plane transforms (one chunk) ---- */
/* In-place sign-bit rotate over len/4 words, as the reference does it (a trailing
 * 2-byte element of a 16-bit stream is left un-rotated). num_buf = 2 or 4. */
void zo_rotate_fwd(uint8_t* buf, size_t len, int num_buf);
void zo_rotate_inv(uint8_t* buf, size_t len, int num_buf);
/* plane p receives bytes p, p+P, p+2P, ...; lens[p] = len/P + (p < len%P). */
void zo_plane_lens(size_t len, int num_buf, size_t* lens);

/* ---- frame level (what zipnn_core.zipnn_core / combine_dtype compute) ---- */
size_t zo_compress_bound(size_t n, int num_buf, size_t chunk, size_t hdr_len);
/* dst = hdr ‖ types ‖ cumSizes ‖ payload, with hdr[24:32] patched to the total length.
 * src is NOT modified (the reference rotates it in place — SURVEY.md Appendix D).
 * threads: 0/1 = serial, >1 = that many pthreads over chunks.  Returns 0 or negative. */
int zo_compress_frame(const uint8_t* hdr, size_t hdr_len, const uint8_t* src, size_t n,
                      int num_buf, int bits_mode, int bytes_mode, size_t chunk, float threshold,
                      int threads, uint8_t* dst, size_t dst_cap, size_t* dst_len);
/* body = frame minus (header + ext header).  Returns 0 or negative. */
int zo_decompress_body(const uint8_t* body, size_t body_len, int num_buf, int bits_mode,
                       int bytes_mode, size_t chunk, size_t orig_size, int threads, uint8_t* dst);

#ifdef __cplusplus
}
#endif
#endif
