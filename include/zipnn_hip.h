/*
 * zipnn_hip.h — C ABI of libzipnn_hip.so, the MI355X (gfx950) implementation of
 * ZipNN's compress/decompress hot path.
 *
 * This is the drop-in boundary.  In the reference the boundary is the CPython
 * extension `zipnn_core` (csrc/zipnn_core_module.c:9-23) with two functions:
 *
 *   zipnn_core.zipnn_core(header, data, numBuf, bits_mode, bytes_mode, is_redata,
 *                         origChunkSize, compThreshold, checkThAfterPercent, threads)
 *                                                  csrc/zipnn_core.c:401-702, called at zipnn/zipnn.py:714-725
 *   zipnn_core.combine_dtype(data_after_header, numBuf, bits_mode, bytes_mode,
 *                            origChunkSize, origSize, threads)
 *                                                  csrc/zipnn_core.c:881-1164, called at zipnn/zipnn.py:1143-1151
 *
 * zn_compress() / zn_decompress() below take exactly those arguments (minus the dead
 * ones: is_redata, checkThAfterPercent, threads — SURVEY.md Appendix D) as plain
 * pointers and sizes, and produce / consume exactly the same bytes (wire format:
 * SURVEY.md Appendix A).  The *_dev variants are the same operations on buffers that
 * already live in HBM (what bench.py times, and what a device-aware safetensors
 * loader calls).  No torch types, no exceptions, no longjmp: every function returns
 * 0 or a negative zn_status and never touches the caller's input buffers (the
 * reference rotates its input in place, csrc/data_manipulation_dtype16.c:68 — we
 * deliberately do not).
 *
 * Parameters shared by all entry points
 *   num_buf     1 (fp8), 2 (bf16/fp16), 4 (fp32)            zipnn/zipnn.py:786-815
 *   bits_mode   1 = sign-bit rotate (bf16/fp32), 0 = none   csrc/data_manipulation_dtype16.c:10-29, dtype32.c:39-58
 *   bytes_mode  10 for num_buf 1/2, 220 for num_buf 4        csrc/data_manipulation_dtype16.c:77, dtype32.c:219-268
 *   chunk       origChunkSize in bytes (power of two; callers pass min(128 KiB, chunk)
 *               for num_buf == 1, zipnn/zipnn.py:721,1148)
 *   threshold   compThreshold: a plane is kept Huffman-coded iff
 *               0 < csize < plane_len * (double)threshold    csrc/zipnn_core.c:371-373
 */
#ifndef ZIPNN_HIP_H
#define ZIPNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum zn_status {
  ZN_OK = 0,
  ZN_E_ARG = -1,      /* bad argument (unsupported num_buf/bytes_mode, chunk == 0, ...) */
  ZN_E_HIP = -2,      /* a HIP runtime call failed; zn_last_hip_error() has the text */
  ZN_E_CAP = -3,      /* destination capacity too small */
  ZN_E_CORRUPT = -4,  /* compressed body is malformed (sizes, huff0 header, streams) */
  ZN_E_TYPE = -5,     /* a chunk-type byte is not 0/1 (reference: MemoryError "Compress Type is not correct", zipnn_core.c:993-996) */
  ZN_E_NODEV = -6,    /* no usable GPU */
  ZN_E_ALLOC = -7,    /* device/host allocation failed */
  ZN_E_TIMEOUT = -8   /* a bounded device-side wait between workgroups of one launch ran out (a preempted or faulted device): the data was not judged */
} zn_status;

/* ABI version of this header (bumped on incompatible change). */
int zn_abi_version(void);
const char* zn_strerror(int status);
/* Text of the last failing HIP call on this thread ("" if none). */
const char* zn_last_hip_error(void);
/* Number of visible HIP devices (0 when none; never fails). */
int zn_device_count(void);

/* numChunks = ceil(n / chunk)                                   csrc/zipnn_core.c:420,893 */
size_t zn_num_chunks(size_t n, size_t chunk);
/* Upper bound of a frame: hdr_len + 9*num_buf*numChunks + n     csrc/zipnn_core.c:105-118 */
size_t zn_compress_bound(size_t n, int num_buf, size_t chunk, size_t hdr_len);

/* ---- host-buffer entry points (replace the two zipnn_core functions) ---- */

/* dst <- hdr ‖ types[P][K] ‖ cumSizes[P][K] (u64) ‖ payload (plane-major); when
 * hdr_len >= 32 the total frame length is written to dst[24:32] (zipnn_core.c:121).
 * `hdr`/`src`/`dst` are host pointers; the call stages through HBM on `device`. */
int zn_compress(const void* hdr, size_t hdr_len, const void* src, size_t n, int num_buf,
                int bits_mode, int bytes_mode, size_t chunk, float threshold, int device,
                void* dst, size_t dst_cap, size_t* dst_len);

/* body = frame minus (32-byte header + shape ext-header); dst receives orig_size bytes. */
int zn_decompress(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode,
                  size_t chunk, size_t orig_size, int device, void* dst);

/* ---- device-resident entry points (inputs and outputs already in HBM) ---- */

/* d_src: n bytes on the current device.  d_body (capacity body_cap >=
 * zn_compress_bound(n, num_buf, chunk, 0)) receives types ‖ cumSizes ‖ payload; its
 * length is returned in *body_len (one 8-byte device→host read).  `stream` is a
 * hipStream_t (NULL = the null stream); work is enqueued on it and the call returns
 * after the length has been read back. */
int zn_compress_dev(const void* d_src, size_t n, int num_buf, int bits_mode, int bytes_mode,
                    size_t chunk, float threshold, void* d_body, size_t body_cap,
                    size_t* body_len, void* stream);

/* d_body: body_len bytes on the current device; d_dst receives orig_size bytes.
 * Asynchronous on `stream` unless `check` is non-zero, in which case the call
 * synchronises and returns ZN_E_CORRUPT / ZN_E_TYPE if a kernel flagged bad input.
 * Streams: calls on one stream are ordered by it; once a device has seen calls on more than one stream, every call records an event that the
 * next one waits for on the device (the library's per-device workspace is shared).  Destroy a stream only after the asynchronous calls issued
 * on it have finished (synchronise it, or ask zn_decode_status, first). */
int zn_decompress_dev(const void* d_body, size_t body_len, int num_buf, int bits_mode,
                      int bytes_mode, size_t chunk, size_t orig_size, void* d_dst, void* stream,
                      int check);

/* Batched compress: `count` tensors (any mix of dtypes) through one launch per stage, one read-back of all body
 * lengths — what a safetensors producer or the streaming writer does per file (reference
 * scripts/zipnn_compress_safetensors.py:75-120, zipnn/zipnn.py:612-635).  Per item the semantics of
 * zn_compress_dev; body_len is filled in for every item. */
typedef struct zn_cbatch_item {
  const void* d_src; size_t n;            /* tensor bytes on the current device */
  int num_buf, bits_mode, bytes_mode;
  size_t chunk; float threshold;
  void* d_body; size_t body_cap;          /* >= zn_compress_bound(n, num_buf, chunk, 0) */
  size_t body_len;                        /* out */
  const void* d_delta;                    /* NULL, or n bytes on the device: the tensor is compressed as src ^ delta */
} zn_cbatch_item;
int zn_compress_batch_dev(zn_cbatch_item* items, size_t count, void* stream);

/* Batched decompress: `count` tensors (any mix of dtypes) decoded by one set of kernel launches, so that
 * many small tensors fill the device as one large one does.  What a safetensors loader does per file:
 * replaces the per-tensor loop around decompress_safetensors_tensor (reference zipnn/zipnn.py:1584-1596,
 * scripts/zipnn_decompress_safetensors.py:75-120).  Same semantics per item as zn_decompress_dev; an error
 * in any item fails the call. */
typedef struct zn_batch_item {
  const void* d_body; size_t body_len;   /* frame body (after the header), on the current device */
  void* d_dst; size_t orig_size;         /* receives orig_size bytes */
  int num_buf, bits_mode, bytes_mode;
  size_t chunk;
  const void* d_delta;                   /* NULL, or orig_size bytes on the device: dst <- decoded ^ delta */
} zn_batch_item;
int zn_decompress_batch_dev(const zn_batch_item* items, size_t count, void* stream, int check);

/* Delta ("byte"/"file" delta_compressed_type of the reference, zipnn/zipnn.py:625-640 and :983-1004: the bytes
 * are XORed with a second buffer of the same length before compression and after decompression).  The XOR is
 * fused into the kernels that read the tensor / write the output — no extra pass over HBM.  d_delta = NULL gives
 * zn_compress_dev / zn_decompress_dev.  Batched: set d_delta in the items above. */
int zn_compress_delta_dev(const void* d_src, const void* d_delta, size_t n, int num_buf, int bits_mode,
                          int bytes_mode, size_t chunk, float threshold, void* d_body, size_t body_cap,
                          size_t* body_len, void* stream);
int zn_decompress_delta_dev(const void* d_body, size_t body_len, const void* d_delta, int num_buf, int bits_mode,
                            int bytes_mode, size_t chunk, size_t orig_size, void* d_dst, void* stream, int check);
/* … and with host buffers (`delta`: n / orig_size host bytes, or NULL): zn_compress / zn_decompress plus one more upload. */
int zn_compress_delta(const void* hdr, size_t hdr_len, const void* src, const void* delta, size_t n, int num_buf,
                      int bits_mode, int bytes_mode, size_t chunk, float threshold, int device,
                      void* dst, size_t dst_cap, size_t* dst_len);
int zn_decompress_delta(const void* body, size_t body_len, const void* delta, int num_buf, int bits_mode,
                        int bytes_mode, size_t chunk, size_t orig_size, int device, void* dst);

/* One call, several GPUs of the node (SURVEY.md §8b proposes exactly this `devices, ndev` form of the two entry points; the
 * reference's counterpart is its `threads` argument — csrc/zipnn_core.c:401 "y*y*iiiinfii", :881 "y*iiinni" — which fans the
 * chunks out over pthreads, zipnn_core.c:294-390 / 768-861).  Device i codes the contiguous chunk range [i K / ndev,
 * (i + 1) K / ndev) on a host thread, stream and workspace of its own; no collective; the host concatenates types and payload
 * per plane and re-bases cumSizes.  The frame / the bytes are identical to the single-device call's.  ndev = 1 IS that call.
 * The same device may be listed more than once (its ranges then run one after the other). */
int zn_compress_multi(const void* hdr, size_t hdr_len, const void* src, size_t n, int num_buf, int bits_mode, int bytes_mode,
                      size_t chunk, float threshold, const int* devices, int ndev, void* dst, size_t dst_cap, size_t* dst_len);
int zn_decompress_multi(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                        size_t orig_size, const int* devices, int ndev, void* dst);

/* The same with the TENSOR resident in HBM, range by range — what a sharded loader / saver holds: device i owns the bytes of the
 * chunk range zn_multi_range(n, chunk, ndev, i) gives, at d_dst[i] / d_src[i] (device memory of devices[i]; NULL for an empty
 * range).  The frame (or its body) stays a host buffer — it comes from / goes to a file.  Decompress: each range's size tables are
 * re-based on the host (9 num_buf bytes per chunk), its payload slices go from `body` straight through that device's pinned pipe
 * into HBM, the decoded bytes never cross PCIe.  Compress: each device codes its range where it lies; only the compressed bodies
 * come back.  One host thread, stream and workspace per listed device; no collective.  Bytes identical to the single-device call. */
int zn_multi_range(size_t n, size_t chunk, int ndev, int i, size_t* off, size_t* len);
int zn_decompress_multi_dev(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                            size_t orig_size, const int* devices, int ndev, void* const* d_dst);
int zn_compress_multi_dev(const void* hdr, size_t hdr_len, const void* const* d_src, size_t n, int num_buf, int bits_mode,
                          int bytes_mode, size_t chunk, float threshold, const int* devices, int ndev, void* dst, size_t dst_cap,
                          size_t* dst_len);

/* The two halves of the same for ONE process per GPU (torch.distributed ranks, each with its own device): a rank decodes the chunk
 * range [chunk_lo, chunk_hi) of a host-resident frame body straight into its own HBM (the sub-body is put together on the device:
 * re-based size tables from the host, payload slices through the pinned pipe; nothing decoded crosses PCIe) … */
int zn_decompress_range_dev(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                            size_t orig_size, size_t chunk_lo, size_t chunk_hi, int device, void* d_dst);
/* … and the bodies the ranks wrote for CONSECUTIVE chunk ranges (rank i: zn_compress_dev of its num_chunks[i] chunks) become the one
 * body a single device would have written: types and payload concatenated per plane, cumSizes re-based (host pointers; one thread
 * per part does the copying).  dst_cap >= the sum of the body lengths. */
int zn_merge_range_bodies(const void* const* bodies, const size_t* body_lens, const size_t* num_chunks, int nparts, int num_buf,
                          void* dst, size_t dst_cap, size_t* dst_len);

/* Plumbing for callers that keep the tensors in HBM but hold pageable host buffers (files, Python bytes): the same
 * pinned, multi-threaded transfer the host-buffer entry points above use internally (zipnn_amd/csrc/zn_host_pipe.hpp),
 * on the current device; returns when the n bytes have arrived.  No reference counterpart — the reference's buffers
 * never leave the host (zipnn/zipnn.py:714-725, 1143-1151 hand host pointers straight to the C core). */
int zn_copy_to_device(void* d_dst, const void* src, size_t n);
int zn_copy_to_host(void* dst, const void* d_src, size_t n);

/* ---------------------------------------------------------------------------------------------------------------------------
 * DEVELOPER / TEST KNOBS — not part of the drop-in boundary.  The setters below are process-wide mutable state: they exist so that
 * the test-suite and the A/B scripts can force every kernel form on every input (tests/, scripts/) and so that a frame can be made
 * byte-identical to a PyPI wheel's; a production caller never needs them — every default is "automatic", and the bytes a call produces
 * or accepts do not depend on them (zn_set_legacy_tree_descriptions excepted).  They are NOT synchronised with calls in flight: set them before the first call of a
 * process (or between calls, from the one thread that drives the library) — never while another host thread is inside an entry point.
 * One host thread per GPU, as the multi-device entry points use the library, is safe with the defaults.
 * ------------------------------------------------------------------------------------------------------------------------- */

/* chunks one workgroup of the fused decoder takes, 1..4; 0 (default) = automatic: the group size whose launch takes the fewest
 * rounds of workgroups x (one parse of the group's tree descriptions + its chunks), a round being as many workgroups as the device
 * holds at once (profiles/r05_decode_group_rule.txt).  Returns 0 or ZN_E_ARG. */
int zn_set_decode_group(int chunks_per_workgroup);
/* what that rule (or the knob) gives a launch of `chunks` chunks on the current device: 1..4 */
int zn_decode_group_for(unsigned long long chunks);

/* the small-input form of the decoder (one workgroup per chunk, four or two waves per huff0 stream) — 0 = never,
 * 1 (default) = automatic: calls whose tensors are all split with the sign rotate (bits_mode 1 with 2 or 4 byte planes: bf16, fp32), have no
 * delta base and few workgroups per compute unit of the device — one per full chunk, plus, for a tensor with a partial last chunk (round 6), its tail
 * and merge workgroups, which ride in the same launch; 2 / 3 = every call without a
 * delta base, 16- / 8-wave form.  The bytes produced are the same in every mode.  Returns 0 or ZN_E_ARG. */
int zn_set_decode_wide(int mode);

/* slices of the three-stage pipeline upload | code | download that large pageable buffers can go through in the host-buffer entry points
 * (zn_compress / zn_decompress).  0 (default) = automatic (compress: 4-8 slices from 192 MiB up; decompress: one shot — measured: no gain
 * there — unless the direct transfers of zn_set_host_direct are on), 1 = never, 2..64 = that many, both directions.  Returns 0 or ZN_E_ARG. */
int zn_set_host_slices(int slices);

/* how the host-buffer entry points (zn_compress / zn_decompress, the *_multi forms, zn_copy_to_host / _to_device) move a caller's pageable buffer:
 * bit 2 (default, = 4): result buffers get madvise(MADV_HUGEPAGE) before they are first written — a page-size hint, nothing else;
 * bit 0 / bit 1 (opt-in; 7 = everything): device-to-host / host-to-device transfers of 128 MiB and more may go DIRECT — the caller's pages pinned piece
 * by piece (hipHostRegister), DMA straight between them and HBM, the two directions of a call pipelined in 4-8 slices; only for calls whose result buffer
 * is already backed by pages (recycled buffers), anything else takes the staged copy through the library's own pinned buffers.  1 GiB bf16 each way:
 * 23 ms direct with recycled buffers against 31-36 ms staged; why it is not the default: profiles/r06_host_path.txt (calls that fault a fresh result
 * buffer in get slower in a long-lived process that has pinned user memory).  Also ZIPNN_AMD_HOST_DIRECT=0..7 in the environment.  Returns 0 or ZN_E_ARG. */
int zn_set_host_direct(int mode);

/* Pinned host memory from the library's arena, for buffers that cross PCIe — above all the RESULT of zn_compress / zn_decompress, which the reference's extension
 * also allocates itself and returns as a memoryview (csrc/zipnn_core.c:596, 1126).  A transfer between such a block and HBM is one DMA (57 GB/s, no staging copy, no
 * page faults, and releasing it costs nothing: a pageable 1 GiB result costs its owner 40-130 ms of munmap): with a FRESH block per call — allocate, call, release —
 * 1 GiB of bf16 takes 28-30 ms to compress and 33-36 ms to decompress, all in, against 59-61 and 75 ms with a fresh pageable result; freed blocks are recycled (up to ZIPNN_AMD_HOST_ARENA_MB = 8192 MiB of them are kept;
 * zn_release_workspace returns them to the driver).  zn_host_alloc: NULL when the driver has no more pinned memory; zn_host_free: 0, or ZN_E_ARG for a pointer that
 * is not a live block.  Thread-safe. */
void* zn_host_alloc(size_t n);
int zn_host_free(void* p);

/* the one-pass encoder (full chunks histogrammed, coded and placed by one workgroup each, the chunk's second read aimed at the Infinity Cache) —
 * 0 = never (the four-kernel encoder only), 1 (default; ZIPNN_AMD_ONEPASS=0/1/2 in the environment changes the default) = automatic: calls whose
 * bf16-like tensors (two planes, sign rotate) bring at least 6144 full chunks — where it has measured faster —, 2 = every call with full chunks.
 * The four-kernel encoder is the fall-back for tensors whose planes other than the last do not all stay raw.  The bytes produced are the same in
 * every mode.  Returns 0 or ZN_E_ARG. */
int zn_set_encode_onepass(int mode);

/* Which huff0 the compressed bytes imitate where the two in circulation differ — the FSE-coded tree description of a plane, when one
 * of its code-weight values is rare enough to round below one FSE cell: 0 (default) = zstd >= 1.4.7 (a full cell, "+1"; what the
 * reference writes when built against a current libzstd, oracle/_ref), 1 = the FiniteStateEntropy library the reference's PyPI wheels
 * bundle (/root/reference/setup.py:23-28; the "less than one" marker, "-1").  Every huff0 decoder — this library's included — reads both;
 * with 1 the frames are byte-identical to a wheel's.  (The one knob that changes bytes; read once per compress call, at its start.)
 * Returns 0 or ZN_E_ARG. */
int zn_set_legacy_tree_descriptions(int on);

/* Frees the per-device workspaces this library caches (scratch planes, size tables). */
int zn_release_workspace(void);

/* The device-side verdict of the CALLING THREAD's last zn_decompress*_dev call on the current device with check = 0 (no read-back,
 * asynchronous): waits for `stream` and returns ZN_OK / ZN_E_TYPE / ZN_E_CORRUPT exactly as that call would have with check = 1.  For
 * callers that overlap host work with the decode (a checkpoint loader builds its tensor views while the kernels run) — the reference's
 * combine_dtype (csrc/zipnn_core.c:881) is synchronous and reports through its return value at once.  Every decode call writes its
 * verdict to a slot of its own (sixteen per device, handed out in turn), so decodes by other threads or on other streams in between
 * do not disturb it; after sixteen further decode calls on the device the slot is reused and the answer is ZN_E_CORRUPT ("cannot
 * vouch for it") rather than a guess.  A thread whose most recent check = 0 call was not on this device (it made none, or its last one
 * went to another device) gets the device's most recent decode: ask on the device you launched on, before you launch on another. */
int zn_decode_status(void* stream);

/* Names of the kernels the last *_dev call launched, ';'-separated (for profiles). */
const char* zn_last_kernels(void);

/* Diagnostic: how many chunks of the last zn_decompress_dev call on the current device were
 * decoded by the fused single-pass kernel (the rest went through the generic two-kernel path).
 * Synchronises the device.  Returns a count >= 0 or a negative zn_status. */
long long zn_last_fused_chunks(void);

/* Diagnostic: how many Huffman planes of partial last chunks the parallel tail kernel decoded in the last
 * zn_decompress(_batch)_dev call (the rest of a partial chunk goes through the serial generic kernel). */
long long zn_last_tail_planes(void);

#ifdef __cplusplus
}
#endif
#endif /* ZIPNN_HIP_H */
