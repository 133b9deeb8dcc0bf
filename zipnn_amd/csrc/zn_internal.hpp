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
};

// ---- generic decode path (any dtype, any tail) : zn_decode_generic.hip ----
// descs: Σ P·K entries; status: one device word; d_done: Σ K flags written by the fused kernel.
// segs == nullptr: the single tensor `one`.  total_pk / total_k: grid sizes (Σ P·K, Σ K).
void zn_launch_decode_generic(int P, const ZnSeg& one, const ZnSeg* d_segs, uint32_t nseg, uint64_t total_pk, uint64_t total_k,
                              ZnPlaneDesc* d_descs, uint32_t* d_status, const uint8_t* d_done, const uint8_t* d_pdone,
                              const uint8_t* d_tail_scratch, const uint8_t* d_tail_done, hipStream_t stream);

// ---- fused decode path (full chunks, ≤1 Huffman plane) : zn_decode_fused.hip ----
uint32_t zn_decode_fused_group(uint64_t K);     // chunks per workgroup for a tensor of K chunks
// d_done: Σ K chunk flags; d_pdone: the same per (plane, chunk) entry (Σ P·K), for the planes kernel.
// ntail: Huffman planes of partial last chunks are decoded by `ntail` extra workgroups at the front of the grid (the
// parallel stream decoder on ragged streams) into padded scratch slots (ZN_TAIL_SLOT bytes per plane);
// tail_done[i] = 1 where that worked — the generic kernels take it from there.
void zn_launch_decode_fused(int P, const ZnSeg& one, const ZnSeg* d_segs, uint32_t nseg, uint32_t total_wg,
                            uint8_t* d_done, uint8_t* d_pdone, uint32_t* d_status, uint32_t ntail, uint8_t* d_tail_scratch,
                            uint8_t* d_tail_done, hipStream_t stream);

// ---- generic encode path : zn_encode_generic.hip ----
// Handles chunks [c0, K).  planes/enc: P*(K-c0) slots each; csize/type/offs: [P*K] (global indexing).
void zn_launch_encode_generic_stats(const ZnGeom& g, uint64_t c0, const uint8_t* d_src, float threshold, uint8_t* d_planes,
                                    uint8_t* d_enc, uint32_t* d_csize, uint8_t* d_type, hipStream_t stream);
// per-plane scan over ALL chunks: types, cumSizes (into the body), payload offsets, total body length
void zn_launch_scan_sizes(const ZnGeom& g, const uint32_t* d_csize, const uint8_t* d_type, uint64_t* d_offs,
                          uint64_t* d_total, uint8_t* d_body, hipStream_t stream);
void zn_launch_encode_generic_gather(const ZnGeom& g, uint64_t c0, const uint8_t* d_planes, const uint8_t* d_enc,
                                     const uint32_t* d_csize, const uint8_t* d_type, const uint64_t* d_offs, uint8_t* d_body,
                                     hipStream_t stream);

// ---- fused encode path (full chunks [0, nfull)) : zn_encode_fused.hip ----
struct ZnEncDesc {           // per (plane, chunk): what the emit kernel needs for a plane kept as huff0 / RLE
  uint32_t code[256];        // code value | code length << 16
  uint8_t  hdr[136];         // tree description (RLE: hdr[0] = the byte)
  uint32_t hdr_len;
  uint32_t ssize[4];         // stream sizes in bytes
  uint16_t qcount[4][256];   // symbol counts of each quarter (stats kernel → tables kernel)
};
bool zn_encode_fused_ok(const ZnGeom& g, const void* d_src);
void zn_launch_encode_fused_stats(const ZnGeom& g, uint64_t nfull, const uint8_t* d_src, float threshold, uint32_t* d_csize,
                                  uint8_t* d_type, ZnEncDesc* d_descs, hipStream_t stream);
void zn_launch_encode_fused_emit(const ZnGeom& g, uint64_t nfull, const uint8_t* d_src, const uint32_t* d_csize, const uint8_t* d_type,
                                 const uint64_t* d_offs, const ZnEncDesc* d_descs, uint8_t* d_body, uint32_t* d_status, hipStream_t stream);

// kernel-name log for zn_last_kernels()
void zn_note_kernel(const char* name);
