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

// ---- generic decode path (any dtype, any tail) : zn_decode_generic.hip ----
// descs: [P*K]; scratch: P*K slots of zn_plane_slot bytes; status: one device word.
// d_done: [K] flags written by the fused kernel (1 = chunk already decoded), or nullptr.
void zn_launch_decode_generic(const ZnGeom& g, const uint8_t* d_body, uint64_t body_len,
                              ZnPlaneDesc* d_descs, uint32_t* d_status, uint8_t* d_dst, const uint8_t* d_done,
                              hipStream_t stream);

// ---- fused decode path (full chunks, ≤1 Huffman plane) : zn_decode_fused.hip ----
void zn_launch_decode_fused(const ZnGeom& g, const uint8_t* d_body, uint64_t body_len, uint8_t* d_dst, uint8_t* d_done,
                            uint32_t* d_status, hipStream_t stream);

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
