// zn_decode_generic.hip — generic decode path: handles every dtype (1/2/4 planes), partial
// last chunks, RLE planes, and any mix of raw/Huffman planes.  Two kernels:
//
//   zn_k_decode_planes   one wave per (plane, chunk): parse its metadata, classify it
//                        (raw / RLE / huff0) and, for huff0 blocks, build the decode LUT in
//                        LDS and decode the four backward streams straight into the plane's
//                        (strided) byte positions of the output — no scratch memory.
//   zn_k_merge_planes    one workgroup per chunk: fill in the raw / RLE planes from the body,
//                        undo the sign-bit rotate (in place, word by word), XOR with the delta base if there is one.
//
// Replaces: decompression_chunk_worker (reference csrc/zipnn_core.c:768-861), the metadata
// parse of py_combine_dtype (:929-1028), HUF_decompress (call site :807) and
// combine_buffers_dtype16/32 + revert_all_floats_* (data_manipulation_dtype16.c:145-216,
// data_manipulation_dtype32.c:275-294,391-456).
//
// This is the correctness-first path; the bandwidth path for full chunks with one
// Huffman-coded plane is zn_decode_fused.hip.
#include "zn_internal.hpp"
#include "zn_huf_wave.hpp"
#include "zn_decode_common.hpp"

#include "zn_decode_rest.hpp"

// Grid-stride over the launch's (plane, chunk) entries; pdone[b] != 0: the fused kernel already wrote that chunk
// (checked before anything else, so a launch where everything is done costs a few microseconds).
__global__ __launch_bounds__(ZN_WAVE) void zn_k_decode_planes(ZnSeg one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint64_t total,
                                                              ZnPlaneDesc* __restrict__ descs_all, uint32_t* __restrict__ status,
                                                              const uint8_t* __restrict__ pdone, const uint8_t* __restrict__ tail_done,
                                                              const uint32_t* __restrict__ left) {
  if (left && *left == 0u) return;             // the fused kernel took every chunk of the launch
  __shared__ ZnPlanesLds L;
  __shared__ uint32_t n_todo; __shared__ uint16_t todo[ZN_WAVE];
  // this wave's items are b = blockIdx.x + i * gridDim.x; their flags are read 64 at a time (one latency)
  const uint64_t nit = (total > blockIdx.x) ? (total - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0;
  for (uint64_t base = 0; base < nit; base += ZN_WAVE) {
    if (threadIdx.x == 0) n_todo = 0;
    __syncthreads();
    const uint64_t i = base + threadIdx.x, b = blockIdx.x + i * gridDim.x;
    if (i < nit && !(pdone && pdone[b])) todo[atomicAdd(&n_todo, 1u)] = (uint16_t)threadIdx.x;
    __syncthreads();
    const uint32_t n = n_todo;
    for (uint32_t k = 0; k < n; k++) {
      const uint64_t bb = blockIdx.x + (base + todo[k]) * gridDim.x;
      __syncthreads();                         // the tables of the previous item are no longer in use
      zn_decode_plane_item(L, one, segs, nseg, bb, descs_all, status, tail_done, threadIdx.x);
    }
    __syncthreads();
  }
}

template <int P>
__global__ __launch_bounds__(256) void zn_k_merge_planes(ZnSeg one, const ZnSeg* __restrict__ segs, uint32_t nseg, uint64_t total,
                                                         const ZnPlaneDesc* __restrict__ descs_all, const uint8_t* __restrict__ done_all,
                                                         const uint8_t* __restrict__ tails, const uint32_t* __restrict__ left) {
  if (left && *left == 0u) return;             // the fused kernel took every chunk of the launch
  __shared__ uint32_t n_todo; __shared__ uint16_t todo[256];
  // items = (chunk, sixteenth); this workgroup's items are it = blockIdx.x + i * gridDim.x; the chunks' flags are
  // read 256 at a time (one latency)
  const uint64_t items = total * ZN_MERGE_SUB;
  const uint64_t nit = (items > blockIdx.x) ? (items - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0;
  for (uint64_t base = 0; base < nit; base += 256u) {
    if (threadIdx.x == 0) n_todo = 0;
    __syncthreads();
    const uint64_t i = base + threadIdx.x, it = blockIdx.x + i * gridDim.x;
    if (i < nit && !(done_all && done_all[it / ZN_MERGE_SUB])) todo[atomicAdd(&n_todo, 1u)] = (uint16_t)threadIdx.x;   // not written by the fused kernel
    __syncthreads();
    const uint32_t n = n_todo;
    for (uint32_t k = 0; k < n; k++) {
      const uint64_t it2 = blockIdx.x + (base + todo[k]) * gridDim.x;
      zn_merge_chunk_item<P>(one, segs, nseg, it2 / ZN_MERGE_SUB, (uint32_t)(it2 % ZN_MERGE_SUB), descs_all, tails);
    }
    __syncthreads();
  }
}

void zn_launch_decode_generic(int P, const ZnSeg& one, const ZnSeg* d_segs, uint32_t nseg, uint64_t total_pk, uint64_t total_k,
                              ZnPlaneDesc* d_descs, uint32_t* d_status, const uint8_t* d_done, const uint8_t* d_pdone,
                              const uint8_t* d_tail_scratch, const uint8_t* d_tail_done, hipStream_t stream) {
  if (total_k == 0) return;
  // d_done != null: the fused kernel ran before us and counted the chunks it left in d_status[1 + q]
  const uint32_t* left = d_done ? d_status + 1 + (P == 1 ? 0 : P == 2 ? 1 : 2) : nullptr;
  const uint32_t gp = (uint32_t)(total_pk < 2048u ? total_pk : 2048u), gm = (uint32_t)(total_k * ZN_MERGE_SUB < 1024u ? total_k * ZN_MERGE_SUB : 1024u);
  hipLaunchKernelGGL(zn_k_decode_planes, dim3(gp), dim3(ZN_WAVE), 0, stream, one, d_segs, nseg, total_pk, d_descs, d_status, d_pdone, d_tail_done, left);
  zn_note_kernel("zn_k_decode_planes");
  if (P == 1) hipLaunchKernelGGL(zn_k_merge_planes<1>, dim3(gm), dim3(256), 0, stream, one, d_segs, nseg, total_k, d_descs, d_done, d_tail_scratch, left);
  else if (P == 2) hipLaunchKernelGGL(zn_k_merge_planes<2>, dim3(gm), dim3(256), 0, stream, one, d_segs, nseg, total_k, d_descs, d_done, d_tail_scratch, left);
  else hipLaunchKernelGGL(zn_k_merge_planes<4>, dim3(gm), dim3(256), 0, stream, one, d_segs, nseg, total_k, d_descs, d_done, d_tail_scratch, left);
  zn_note_kernel("zn_k_merge_planes");
}
