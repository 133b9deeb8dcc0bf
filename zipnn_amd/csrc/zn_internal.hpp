// zn_internal.hpp — host-side declarations shared by the kernel translation units
// and the C-ABI (zn_api.hip).  Not part of the public interface (include/zipnn_hip.h).
#pragma once

#include "zn_common.hpp"

// What the decoder resolved for one (plane, chunk): where its bytes come from.
struct ZnPlaneDesc {
  uint64_t off;    // RAW: byte offset into the body; HUF: byte offset into the scratch planes; RLE: the byte value
  uint32_t kind;   // ZN_KIND_*
  uint32_t len;    // uncompressed plane length
};

// Slot stride (bytes) of one plane of one chunk in the scratch-plane buffers.
static inline size_t zn_plane_slot(size_t chunk, int P) { return ((chunk + (size_t)P - 1) / (size_t)P + 15) & ~(size_t)15; }

// One tensor ("segment") of a decode launch.  A launch decodes one tensor (the segment travels as a kernel
// argument) or a batch of tensors with the same plane count (a table in device memory; workgroups find their
// tensor by binary search over the three running indices).
struct ZnSeg {
  ZnGeom g;
  const uint8_t* body; uint64_t body_len; uint8_t* dst;
  uint64_t chunk0;   // index of the tensor's first chunk in the launch-wide done[] flags  (merge kernel grid)
  uint64_t desc0;    // index of its first (plane, chunk) in descs[]                          (planes kernel grid)
  uint32_t wg0;      // its first workgroup of the fused kernel
  uint32_t ncg;      // chunks per fused workgroup
  uint32_t tail0;    // a partial last chunk gets P workgroups of the tail kernel / P tail-scratch slots from here …
  uint32_t has_tail; // … if this is non-zero
  const uint8_t* xr; // delta base (orig_size bytes) the decoded bytes are XORed with on the way out, or null
};

// ---- generic decode path (any dtype, any tail) : zn_decode_generic.hip ----
// descs: Σ P·K entries; status: one device word; d_done: Σ K flags written by the fused kernel.
// segs == nullptr: the single tensor `one`.  total_pk / total_k: grid sizes (Σ P·K, Σ K).
void zn_launch_decode_generic(int P, const ZnSeg& one, const ZnSeg* d_segs, uint32_t nseg, uint64_t total_pk, uint64_t total_k,
                              ZnPlaneDesc* d_descs, uint32_t* d_status, const uint8_t* d_done, const uint8_t* d_pdone,
                              const uint8_t* d_tail_scratch, const uint8_t* d_tail_done, hipStream_t stream);

// ---- fused decode path (full chunks; one pass per Huffman plane) : zn_decode_fused.hip ----
uint32_t zn_decode_fused_group(uint64_t K);     // chunks per workgroup for a tensor of K chunks
// d_done: Σ K chunk flags; d_pdone: the same per (plane, chunk) entry (Σ P·K), for the planes kernel.
// ntail: Huffman planes of partial last chunks are decoded by `ntail` extra workgroups at the front of the grid (the
// parallel stream decoder on ragged streams) into padded scratch slots (ZN_TAIL_SLOT bytes per plane);
// tail_done[i] = 1 where that worked — the generic kernels take it from there.
bool zn_launch_decode_fused(int P, const ZnSeg& one, const ZnSeg* d_segs, uint32_t nseg, uint32_t total_wg,
                            uint8_t* d_done, uint8_t* d_pdone, uint32_t* d_status, uint32_t ntail, uint8_t* d_tail_scratch,
                            uint8_t* d_tail_done, bool delta, int wide, bool status_zeroed, ZnPlaneDesc* d_descs_rest, uint32_t* d_tailsync, hipStream_t stream);     // delta: some tensor of the launch has ZnSeg::xr
// d_tailsync (may be null): two zeroed words per tensor with a partial last chunk — with d_descs_rest the launch then finishes those chunks itself (merge workgroups at its end)
// wide = 4 / 2 (waves per stream): the launch's segments have ncg == 1 and zn_k_decode_wide (zn_decode_wide.hpp, small inputs) goes first;
// !status_zeroed: … and zeroes the status words (the call's first launch: the caller then leaves out its memset);
// d_descs_rest (used when the launch has no tail workgroups and no delta base): the fused kernel's `rest` instance decodes what it does not take with the generic
// path's own code (zn_decode_rest.hpp) — returns true then, and the caller leaves out zn_launch_decode_generic
int zn_decode_use_wide(uint64_t total_full_chunks, bool delta, bool weights_like, uint64_t tail_wgs);     // weights_like: every tensor of the call is split with the sign rotate (bf16 / fp32)

// ---- encode ----
struct ZnEncDesc {           // per (plane, chunk): what the emit kernel needs for a plane kept as huff0 / RLE
  uint32_t code[256];        // code value | code length << 16
  uint8_t  hdr[136];         // tree description (RLE: hdr[0] = the byte)
  uint32_t hdr_len;
  uint32_t ssize[4];         // stream sizes in bytes
  uint16_t qcount[4][256];   // symbol counts of each quarter (stats kernel → tables kernel)
};

// One tensor of a compress launch (a single tensor travels as a kernel argument, a batch — tensors of the same
// plane count — as a table in device memory; workgroups find their tensor by binary search over the running
// grid indices).  The per-(plane, chunk) arrays (stored size, type, payload offset, descriptor) of the whole
// launch are concatenated; pc0 is this tensor's base in them, index p·K + c inside.
struct ZnESeg {
  ZnGeom g;
  const uint8_t* src; uint8_t* body;
  float threshold; uint32_t legacy_weights;   // 1: tree descriptions with -1 markers (zn_set_legacy_tree_descriptions)
  uint64_t nfull;      // chunks [0, nfull) go through the fused kernels, [nfull, K) through the generic ones
  uint64_t pc0;        // base of its (plane, chunk) entries
  uint64_t slot0;      // base of its generic-path scratch slots (P·(K-nfull) of them)
  uint64_t total_idx;  // where its body length goes (d_total[total_idx])
  uint64_t T;          // scan: entries per block
  uint32_t chunk0;     // grid key: first fused chunk (stats / emit)
  uint32_t job0;       // grid key: first (plane, fused chunk) job (tables)
  uint32_t tail0;      // grid key: first generic chunk (split)
  uint32_t ptail0;     // grid key: first generic (plane, chunk) (encode planes / gather)
  uint32_t scan0;      // grid key: first scan block
  uint32_t pad2_;
  const uint8_t* xr;   // delta base (g.n bytes): the encoder sees src ^ xr; null = none
};

// Slot stride of the generic path's scratch planes = zn_plane_slot(chunk, P).  All launchers: `one` when
// d_segs == nullptr, else the table; grid totals are sums over the launch's tensors.
bool zn_encode_fused_ok(const ZnGeom& g, const void* d_src, const void* d_xr);    // d_xr: delta base or null
// total_ptails ragged planes (partial last chunks, geometries the fused kernels do not take) ride along as further workgroups of the same
// launches; d_planes / slot: their scratch planes (slot0 of a segment = its first slot)
// d_status_zero (may be null): the call's status word, zeroed by the table kernel; returns whether that kernel was launched
bool zn_launch_encode_fused_stats(int P, const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_chunks, uint32_t total_jobs,
                                  uint32_t total_ptails, uint8_t* d_planes, uint64_t slot,
                                  uint32_t* d_csize, uint8_t* d_type, ZnEncDesc* d_descs, bool delta, uint32_t* d_status_zero, hipStream_t stream);
void zn_launch_encode_fused_emit(int P, const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_chunks, uint32_t total_ptails,
                                 const uint8_t* d_planes, uint64_t slot,
                                 const uint32_t* d_csize, const uint8_t* d_type, const uint64_t* d_offs, const ZnEncDesc* d_descs,
                                 uint32_t* d_status, bool delta, hipStream_t stream);
// per-tensor scan over ALL its chunks: types, cumSizes (into the body), payload offsets, total body length → d_total[total_idx]
// d_spec_status (may be null): the launch's full chunks went through the one-pass encoder — the scan checks its layout speculation (the last plane starts where
// all-raw earlier planes put it) and sets ZN_DEV_MISSPEC there when it does not hold
void zn_launch_scan_sizes(const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_blocks, const uint32_t* d_csize,
                          const uint8_t* d_type, uint64_t* d_offs, uint64_t* d_total, uint32_t* d_spec_status, hipStream_t stream);
// the one-pass encoder over the launch's total_chunks full chunks (zn_k_encode_onepass): d_lb = total_chunks look-back words, d_ticket = a zeroed counter,
// gen = the launch's generation tag (1 .. 2^22 - 1)
void zn_launch_encode_onepass(int P, const ZnESeg& one, const ZnESeg* d_segs, uint32_t nseg, uint32_t total_chunks, uint32_t* d_csize, uint8_t* d_type,
                              uint64_t* d_lb, uint32_t* d_ticket, uint32_t* d_status, uint32_t gen, bool delta, hipStream_t stream);
void zn_scan_geometry(uint64_t PK, uint64_t* T, uint32_t* blocks);     // entries per block / number of blocks for PK entries

// kernel-name log for zn_last_kernels()
void zn_note_kernel(const char* name);

#if defined(__HIPCC__) || defined(ZN_SIMT_EMULATOR)
// which tensor of a batched compress launch does grid index `b` belong to?  (last segment with key ≤ b; wave-uniform)
#define ZN_DEF_EFIND(NAME, FIELD)                                                                                           \
  __device__ __forceinline__ ZnESeg NAME(const ZnESeg& one, const ZnESeg* __restrict__ segs, uint32_t nseg, uint64_t b) { \
    if (segs == nullptr) return one;                                                                                       \
    uint32_t lo = 0, hi = nseg;                                                                                            \
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)segs[mid].FIELD <= b) lo = mid; else hi = mid; } \
    return segs[lo];                                                                                                       \
  }
ZN_DEF_EFIND(zn_efind_chunk, chunk0)
ZN_DEF_EFIND(zn_efind_job, job0)
ZN_DEF_EFIND(zn_efind_tail, tail0)
ZN_DEF_EFIND(zn_efind_ptail, ptail0)
ZN_DEF_EFIND(zn_efind_scan, scan0)
#undef ZN_DEF_EFIND
#endif
