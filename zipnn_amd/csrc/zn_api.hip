// zn_api.hip — the C ABI of libzipnn_hip.so (include/zipnn_hip.h).
//
// Host-side control only: argument checks, per-device workspace cache, kernel launches,
// the one 8-byte read-back of the compressed length.  All data work is in the kernels.
#include "../../include/zipnn_hip.h"
#include "zn_internal.hpp"
#include "zn_host_pipe.hpp"

#include <mutex>
#include <atomic>
#include <condition_variable>
#include <thread>
#include <chrono>
#include <string>
#include <vector>
#include <string.h>
#include <stdlib.h>

namespace {

thread_local std::string t_hip_err;
thread_local std::string t_kernels;
// which status slot the calling thread's last check = 0 decode used (zn_decode_status): device, slot, the slot's generation at that call
thread_local int t_status_dev = -1;
thread_local uint32_t t_status_slot = 0;
thread_local uint64_t t_status_gen = 0;

#define ZN_HIP(call)                                                            \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) {                                                     \
      t_hip_err = std::string(#call) + ": " + hipGetErrorString(e_);            \
      return ZN_E_HIP;                                                          \
    }                                                                           \
  } while (0)

// Grow-only device buffers cached per device; guarded by one mutex (calls on the same
// device serialise on their workspace — independent GPUs run in independent processes).
struct Workspace {
  size_t last_K = 0;             // chunks of the last decompress call (for zn_last_fused_chunks)
  size_t last_tails = 0;         // tail planes of the last decompress call (for zn_last_tail_planes)
  void* buf[13] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t op_backoff = 0, op_penalty = 0;   // automatic one-pass encoding sits out `op_backoff` calls after a misspeculated one (compress_items)
  uint32_t lb_gen = 0;           // generation tag of the one-pass encoder's look-back words (WS_LB): a launch only believes words of its own generation, so the array is never zeroed between calls
  ZnSeg* h_segs = nullptr; size_t h_segs_cap = 0;   // pinned staging for the segment table of a batched call (capacity in ZnSeg units)
  uint64_t* h_totals = nullptr; size_t h_totals_cap = 0;   // pinned: body lengths of a batched compress
  uint64_t* h_total = nullptr;   // pinned host word for the length read-back
  uint32_t* h_status = nullptr;
  hipEvent_t busy = nullptr;     // orders the workspace between streams (ws_acquire / ws_release)
  hipStream_t only_stream = nullptr; bool have_stream = false, multi = false;   // one stream so far (its handle is only ever COMPARED) / several: an event per call
  // (a stream created at a destroyed one's address compares equal: the header's contract — destroy a stream only after its asynchronous calls have
  //  finished — is what covers that; hipStreamGetId, which could tell the two apart, is a hip_7.1 symbol and the runtime PyTorch 2.10 loads is 7.0)
  // decode status words: ZN_STATUS_SLOTS slots of four words behind the buffer's first 64 bytes, handed out in turn — a decode on another thread or stream
  // does not zero the word a check = 0 caller has yet to read (zn_decode_status)
  uint64_t status_gen = 0, slot_gen[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t last_slot = 0;
  ZnHostPipe pipe;               // pinned bounce buffers + copy stream of the host-buffer entry points
  ZnHostPipe pipe2;              // a second one: the pipelined host path downloads slice i - 1 while it uploads slice i + 1
  hipStream_t cstream = nullptr; // … and codes slice i on a stream of its own
  hipStream_t dstream = nullptr, dstream2 = nullptr; hipEvent_t dfork = nullptr, djoin = nullptr, djoin2 = nullptr;   // a mixed batched decode: the one-plane tensors' launches on one of these streams, the others' on the other
};
#define ZN_STATUS_SLOTS 16u
#define ZN_WORDS_BYTES (64u + ZN_STATUS_SLOTS * 4u * sizeof(uint32_t))
enum { WS_PLANES = 0, WS_ENC, WS_META_A, WS_META_B, WS_META_C, WS_WORDS, WS_DESC, WS_SEGS, WS_HOST_IN, WS_HOST_OUT, WS_TOTALS, WS_HOST_DELTA, WS_LB, WS_COUNT };
static_assert(WS_COUNT == 13, "Workspace::buf size");

// One lock PER DEVICE (the workspace tables of different devices share nothing): threads that drive different GPUs of a
// node from one process — north_star's "independent HIP streams" — do not serialise each other.
std::mutex g_dev_mu[64];
Workspace g_ws[64];

// the host entry points take a device ordinal: switch to it for the call and leave the caller's current device as it was
struct DeviceScope {
  int prev = -1; bool ok = true;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); ok = false; }
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
std::mutex g_host_mu[64];     // serialises the host-buffer entry points of a device (they share two staging buffers)

int ws_reserve(Workspace& w, int slot, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (w.cap[slot] >= bytes) return ZN_OK;
  if (w.buf[slot]) { ZN_HIP(hipFree(w.buf[slot])); w.buf[slot] = nullptr; w.cap[slot] = 0; }
  hipError_t e = hipMalloc(&w.buf[slot], bytes);
  if (e != hipSuccess) { t_hip_err = std::string("hipMalloc: ") + hipGetErrorString(e); (void)hipGetLastError(); return ZN_E_ALLOC; }
  w.cap[slot] = bytes;
  return ZN_OK;
}

int ws_host_words(Workspace& w) {
  if (!w.h_total) ZN_HIP(hipHostMalloc((void**)&w.h_total, 64, hipHostMallocDefault));
  w.h_status = (uint32_t*)(w.h_total + 4);
  return ZN_OK;
}

// The workspace is shared by every call on the device.  Host-side the mutex serialises them; device-side a call on another stream must not
// start before the previous call's kernels are done with the buffers: an event recorded behind every call, waited for by the next.  That
// record costs 3.5-6 us of a 60-120 us decode call (measured), and calls that follow each other on ONE stream are ordered by the stream itself:
// as long as the device's workspace has only ever seen one stream, nothing is recorded (`mark` excepted: a batched call's pinned segment table
// is rewritten from the HOST by the next batched call, which waits for the event).  The first call on a second stream synchronises the device
// once and switches the workspace to event-per-call for good.  (No stream handle is ever used after the call it came with: the caller may
// have destroyed it — hipEventRecord on a destroyed stream crashes, tried.)
int ws_acquire(Workspace& w, hipStream_t stream) {
  if (!w.busy) ZN_HIP(hipEventCreateWithFlags(&w.busy, hipEventDisableTiming));
  if (!w.multi) {
    if (!w.have_stream || stream == w.only_stream) return ZN_OK;
    ZN_HIP(hipDeviceSynchronize());              // a second stream: whatever the first one still holds of the workspace is over after this
    w.multi = true;
    return ZN_OK;
  }
  ZN_HIP(hipStreamWaitEvent(stream, w.busy, 0));
  return ZN_OK;
}
int ws_release(Workspace& w, hipStream_t stream, bool mark = false) {
  if (!w.multi) { w.only_stream = stream; w.have_stream = true; if (!mark) return ZN_OK; }
  ZN_HIP(hipEventRecord(w.busy, stream));
  return ZN_OK;
}

int check_geom(size_t n, int num_buf, int bytes_mode, size_t chunk, ZnGeom* g, int bits_mode) {
  if (!(num_buf == 1 || num_buf == 2 || num_buf == 4)) return ZN_E_ARG;
  if ((num_buf == 4 && bytes_mode != 220) || (num_buf != 4 && bytes_mode != 10)) return ZN_E_ARG;
  if (chunk == 0 || (chunk % (size_t)num_buf) != 0 || chunk > (1ull << 31)) return ZN_E_ARG;
  g->n = n; g->chunk = chunk; g->K = (n + chunk - 1) / chunk; g->P = (uint32_t)num_buf;
  g->rot = (bits_mode == 1 && num_buf > 1) ? 1u : 0u;
  if (g->K * (uint64_t)num_buf > 0x7FFFFFFFull) return ZN_E_ARG;
  return ZN_OK;
}

}  // namespace

void zn_note_kernel(const char* name) { if (!t_kernels.empty()) t_kernels += ";"; t_kernels += name; }

extern "C" {

int zn_abi_version(void) { return 3; }

const char* zn_strerror(int s) {
  switch (s) {
    case ZN_OK: return "ok";
    case ZN_E_ARG: return "bad argument";
    case ZN_E_HIP: return "HIP runtime call failed";
    case ZN_E_CAP: return "destination capacity too small";
    case ZN_E_CORRUPT: return "compressed data is corrupt";
    case ZN_E_TYPE: return "Compress Type is not correct in Decompression function";
    case ZN_E_NODEV: return "no HIP device";
    case ZN_E_ALLOC: return "allocation failed";
    case ZN_E_TIMEOUT: return "a device-side wait between workgroups timed out";
    default: return "unknown status";
  }
}

const char* zn_last_hip_error(void) { return t_hip_err.c_str(); }
const char* zn_last_kernels(void) { return t_kernels.c_str(); }

int zn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

size_t zn_num_chunks(size_t n, size_t chunk) { return chunk ? (n + chunk - 1) / chunk : 0; }

size_t zn_compress_bound(size_t n, int num_buf, size_t chunk, size_t hdr_len) {
  return hdr_len + 9u * (size_t)num_buf * zn_num_chunks(n, chunk) + n;
}

// Which of the two huff0 forms of a tree description the encoder writes (both decode with every huff0): 0 = zstd >= 1.4.7's (the
// default; what the oracle's pin and the reference built against a current libzstd write), 1 = the FiniteStateEntropy library's, which
// the reference's PyPI wheels bundle: weight counts that round below one FSE cell as -1 markers.  Process-wide.
static std::atomic<int> g_legacy_weights{0};
int zn_set_legacy_tree_descriptions(int on) {
  if (on != 0 && on != 1) return ZN_E_ARG;
  g_legacy_weights.store(on, std::memory_order_relaxed);
  return ZN_OK;
}

// The one-pass encoder (zn_k_encode_onepass): 0 = never (the four-kernel encoder only), 1 (default) = automatic — where it has measured faster: calls whose
// two-plane, sign-rotated tensors (bf16) bring at least ZN_ONEPASS_MIN_CHUNKS full chunks (profiles/r05_encoder_onepass.txt: 4 GiB 2.17 vs 2.27 ms,
// 2 GiB 1.145 vs 1.183, 1 GiB 0.633 vs 0.625; fp16 / fp32 / fp8 lose 9-30 % to it: longer table builds inside the workgroup) —, 2 = every call with
// full chunks (tests).  ZIPNN_AMD_ONEPASS=0/1/2 in the environment sets the default of a process.  The four-kernel encoder takes over whenever the
// one-pass kernel's layout speculation fails.  (Developer / test knob.)
#define ZN_ONEPASS_MIN_CHUNKS 6144u
static std::atomic<int> g_encode_onepass{-1};
static int zn_encode_onepass_mode() {
  int v = g_encode_onepass.load(std::memory_order_relaxed);
  if (v < 0) { const char* e = getenv("ZIPNN_AMD_ONEPASS"); v = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; g_encode_onepass.store(v, std::memory_order_relaxed); }
  return v;
}
extern "C" int zn_set_encode_onepass(int mode) {
  if (mode < 0 || mode > 2) return ZN_E_ARG;
  g_encode_onepass.store(mode, std::memory_order_relaxed);
  return ZN_OK;
}

// Compress `count` tensors: one launch per stage and plane count over all of them (a single tensor travels to the
// kernels as an argument, a batch as a segment table), one read-back of all body lengths.
static int compress_items(zn_cbatch_item* items, size_t count, hipStream_t stream) {
  if (count && !items) return ZN_E_ARG;
  std::vector<ZnESeg> segs[3];                   // by plane count: 1, 2, 4
  std::vector<size_t> owner[3];                  // item index of each segment
  uint64_t pc_all = 0, slot_all = 0; size_t slot = 0;
  uint64_t chunks_of[3] = {0, 0, 0}, jobs_of[3] = {0, 0, 0}, tails_of[3] = {0, 0, 0}, ptails_of[3] = {0, 0, 0}, scan_of[3] = {0, 0, 0};
  bool any_fused = false; bool delta_of[3] = {false, false, false};
  for (size_t i = 0; i < count; i++) {
    zn_cbatch_item& it = items[i];
    ZnESeg sg; memset(&sg, 0, sizeof(sg));
    int rc = check_geom(it.n, it.num_buf, it.bytes_mode, it.chunk, &sg.g, it.bits_mode);
    if (rc) return rc;
    if (it.n && (!it.d_src || !it.d_body)) return ZN_E_ARG;
    if (it.body_cap < zn_compress_bound(it.n, it.num_buf, it.chunk, 0)) return ZN_E_CAP;
    it.body_len = 0;
    const int q = sg.g.P == 1 ? 0 : sg.g.P == 2 ? 1 : 2;
    const uint64_t PK = (uint64_t)sg.g.P * sg.g.K;
    sg.src = (const uint8_t*)it.d_src; sg.body = (uint8_t*)it.d_body; sg.threshold = it.threshold;
    sg.legacy_weights = g_legacy_weights.load(std::memory_order_relaxed) ? 1u : 0u;
    sg.xr = it.n ? (const uint8_t*)it.d_delta : nullptr;
    if (sg.xr) delta_of[q] = true;
    // full chunks go through the fused encoder, the partial tail (or everything, for geometries the fused
    // kernels do not take) through the generic one
    sg.nfull = zn_encode_fused_ok(sg.g, it.d_src, sg.xr) ? (uint64_t)(it.n / it.chunk) : 0;
    any_fused = any_fused || sg.nfull;
    const uint64_t KL = sg.g.K - sg.nfull;
    sg.pc0 = pc_all; sg.slot0 = slot_all; sg.total_idx = i;
    uint32_t blocks = 0; zn_scan_geometry(PK, &sg.T, &blocks);
    sg.chunk0 = (uint32_t)chunks_of[q]; sg.job0 = (uint32_t)jobs_of[q]; sg.tail0 = (uint32_t)tails_of[q];
    sg.ptail0 = (uint32_t)ptails_of[q]; sg.scan0 = (uint32_t)scan_of[q];
    chunks_of[q] += sg.nfull; jobs_of[q] += sg.nfull * sg.g.P; tails_of[q] += KL; ptails_of[q] += KL * sg.g.P; scan_of[q] += blocks;
    pc_all += PK; slot_all += KL * sg.g.P;
    const size_t sl = zn_plane_slot(it.chunk, it.num_buf);
    if (sl > slot) slot = sl;                    // one slot stride for the whole launch
    if (jobs_of[q] > 0x7FFFFFFFull || ptails_of[q] > 0x7FFFFFFFull || pc_all > 0x7FFFFFFFFFull) return ZN_E_ARG;
    segs[q].push_back(sg); owner[q].push_back(i);
  }
  if (count == 0) return ZN_OK;
  int dev = 0;
  ZN_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return ZN_E_ARG;
  t_kernels.clear();
  std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
  Workspace& w = g_ws[dev];
  int rc;
  const size_t nseg_all = segs[0].size() + segs[1].size() + segs[2].size();
  const bool table = nseg_all > 1;
  if ((rc = ws_reserve(w, WS_PLANES, slot_all * slot))) return rc;
  if ((rc = ws_reserve(w, WS_META_A, pc_all * sizeof(uint32_t)))) return rc;   // stored sizes
  if ((rc = ws_reserve(w, WS_META_B, pc_all))) return rc;                      // types
  if ((rc = ws_reserve(w, WS_META_C, pc_all * sizeof(uint64_t)))) return rc;   // payload offsets
  if ((rc = ws_reserve(w, WS_DESC, pc_all * sizeof(ZnEncDesc)))) return rc;
  if ((rc = ws_reserve(w, WS_WORDS, ZN_WORDS_BYTES))) return rc;
  // (one slot more than there are tensors: the call's status word rides behind the body lengths, so that both come back in ONE copy)
  if ((rc = ws_reserve(w, WS_TOTALS, (count + 4) * sizeof(uint64_t)))) return rc;      // (… and the one-pass encoder's three ticket counters behind that)
  if ((rc = ws_host_words(w))) return rc;
  if (w.h_totals_cap < count + 1) {
    if (w.h_totals) { ZN_HIP(hipHostFree(w.h_totals)); w.h_totals = nullptr; w.h_totals_cap = 0; }
    ZN_HIP(hipHostMalloc((void**)&w.h_totals, (count + 1) * sizeof(uint64_t), hipHostMallocDefault));
    w.h_totals_cap = count + 1;
  }
  if (table) {
    if ((rc = ws_reserve(w, WS_SEGS, nseg_all * sizeof(ZnESeg)))) return rc;
    const size_t need = nseg_all * sizeof(ZnESeg);
    if (w.h_segs_cap * sizeof(ZnSeg) < need) {
      if (w.h_segs) { ZN_HIP(hipHostFree(w.h_segs)); w.h_segs = nullptr; w.h_segs_cap = 0; }
      ZN_HIP(hipHostMalloc((void**)&w.h_segs, need, hipHostMallocDefault));
      w.h_segs_cap = (need + sizeof(ZnSeg) - 1) / sizeof(ZnSeg);
    }
  }
  // The one-pass encoder takes the full chunks of the call when every tensor's prefix sums fit its look-back words (40 bits)
  const int op_mode = zn_encode_onepass_mode();
  bool op_of[3] = {false, false, false};         // by plane count: this call's full chunks of that plane count go through the one-pass kernel
  for (int q = 0; q < 3; q++) {
    if (op_mode == 0 || chunks_of[q] == 0) continue;
    bool ok = true, rot = true;
    for (const ZnESeg& sg : segs[q]) { if (sg.g.n >= (1ull << 38)) ok = false; if (sg.nfull && !sg.g.rot) rot = false; }
    // automatic mode: plain weights only.  A call with a delta base (XOR of a fine-tuned checkpoint: both planes compress) breaks the layout speculation almost
    // always, and a misspeculation costs the whole call twice — so delta calls never take it, and after a misspeculated call the device sits out
    // `op_backoff` automatic calls (8, doubling up to 1 024 while misspeculations keep coming; a call that speculated right halves the penalty)
    const bool auto_ok = q == 1 && rot && !delta_of[q] && chunks_of[q] >= ZN_ONEPASS_MIN_CHUNKS;
    if (op_mode == 1 && auto_ok && w.op_backoff > 0) { w.op_backoff--; continue; }
    op_of[q] = ok && (op_mode == 2 || auto_ok);
  }
  const bool onepass = op_of[0] || op_of[1] || op_of[2];
  const uint64_t chunks_all = chunks_of[0] + chunks_of[1] + chunks_of[2];
  if (onepass) {
    const size_t had = w.cap[WS_LB];
    if ((rc = ws_reserve(w, WS_LB, chunks_all * sizeof(uint64_t)))) return rc;
    if (w.cap[WS_LB] != had) { ZN_HIP(hipMemset(w.buf[WS_LB], 0, w.cap[WS_LB])); w.lb_gen = 0; }      // (a fresh array: no word of any generation)
  }
  if ((rc = ws_acquire(w, stream))) return rc;
  uint64_t* d_totals = (uint64_t*)w.buf[WS_TOTALS];
  uint32_t* d_status = (uint32_t*)(d_totals + count);
  uint32_t* d_csize = (uint32_t*)w.buf[WS_META_A]; uint8_t* d_type = (uint8_t*)w.buf[WS_META_B]; uint64_t* d_offs = (uint64_t*)w.buf[WS_META_C];
  if (table) {
    ZN_HIP(hipEventSynchronize(w.busy));         // the previous batched call may still be reading the pinned staging
    ZnESeg* hs = (ZnESeg*)w.h_segs; size_t o = 0;
    for (int q = 0; q < 3; q++) for (const ZnESeg& sg : segs[q]) hs[o++] = sg;
    ZN_HIP(hipMemcpyAsync(w.buf[WS_SEGS], hs, nseg_all * sizeof(ZnESeg), hipMemcpyHostToDevice, stream));
  }
  // One pass over the launches; `op`: the full chunks through the one-pass encoder (their ragged planes, if any, through the ragged workgroups of the
  // three fused launches), else everything through the four-kernel encoder.
  auto launch_all = [&](bool op) -> int {
    bool status_zeroed = false;                  // (by the first table kernel of the call; a memset only when none is launched — or in front of the one-pass kernel, which may set bits)
    if (op) {
      ZN_HIP(hipMemsetAsync(d_status, 0, 4 * sizeof(uint64_t), stream));      // the status word and the three ticket counters behind it
      status_zeroed = true;
    }
    size_t seg_base = 0;
    uint64_t lb_base = 0;
    for (int stage = 0; stage < 3; stage++) {    // stats (fused + generic) for every plane count, then the scans, then emit / gather
      seg_base = 0;
      if (stage == 1 && !status_zeroed) { ZN_HIP(hipMemsetAsync(d_status, 0, sizeof(uint32_t), stream)); status_zeroed = true; }
      for (int q = 0; q < 3; q++) {
        if (segs[q].empty()) continue;
        const int P = q == 0 ? 1 : q == 1 ? 2 : 4;
        const ZnESeg* d_segs = table ? (const ZnESeg*)w.buf[WS_SEGS] + seg_base : nullptr;
        const uint32_t nseg = (uint32_t)segs[q].size();
        const ZnESeg& one = segs[q][0];
        const bool opq = op && op_of[q];
        if (stage == 0) {
          if (zn_launch_encode_fused_stats(P, one, d_segs, nseg, opq ? 0u : (uint32_t)chunks_of[q], opq ? 0u : (uint32_t)jobs_of[q], (uint32_t)ptails_of[q], (uint8_t*)w.buf[WS_PLANES], slot,
                                           d_csize, d_type, (ZnEncDesc*)w.buf[WS_DESC], delta_of[q], status_zeroed ? nullptr : d_status, stream)) status_zeroed = true;
          if (opq) {
            w.lb_gen = (w.lb_gen + 1u) & 0x3FFFFFu;
            if (w.lb_gen == 0) { ZN_HIP(hipMemsetAsync(w.buf[WS_LB], 0, w.cap[WS_LB], stream)); w.lb_gen = 1; }       // (the tag has wrapped: forget every older word)
            zn_launch_encode_onepass(P, one, d_segs, nseg, (uint32_t)chunks_of[q], d_csize, d_type, (uint64_t*)w.buf[WS_LB] + lb_base,
                                     (uint32_t*)(d_totals + count + 1 + q), d_status, w.lb_gen, delta_of[q], stream);
            lb_base += chunks_of[q];
          }
        } else if (stage == 1) {
          zn_launch_scan_sizes(one, d_segs, nseg, (uint32_t)scan_of[q], d_csize, d_type, d_offs, d_totals, opq ? d_status : nullptr, stream);
        } else {
          zn_launch_encode_fused_emit(P, one, d_segs, nseg, opq ? 0u : (uint32_t)chunks_of[q], (uint32_t)ptails_of[q], (const uint8_t*)w.buf[WS_PLANES], slot,
                                      d_csize, d_type, d_offs, (const ZnEncDesc*)w.buf[WS_DESC], d_status, delta_of[q], stream);
        }
        seg_base += nseg;
      }
    }
    ZN_HIP(hipGetLastError());
    ZN_HIP(hipMemcpyAsync(w.h_totals, d_totals, (count + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));     // body lengths + the status word
    return ZN_OK;
  };
  if ((rc = launch_all(onepass))) return rc;
  if (onepass) {
    ZN_HIP(hipStreamSynchronize(stream));
    if ((uint32_t)w.h_totals[count] & (ZN_DEV_MISSPEC | ZN_DEV_SYNC_TIMEOUT)) {    // a tensor that does not look like weights: its bytes are where a decoder would not look for them
      const bool timed_out = ((uint32_t)w.h_totals[count] & ZN_DEV_SYNC_TIMEOUT) != 0;      // (… or a look-back that gave up waiting: the four-kernel encoder waits for nobody)
      w.op_penalty = w.op_penalty ? (w.op_penalty < 1024u ? 2u * w.op_penalty : 1024u) : 8u;
      w.op_backoff = w.op_penalty;
      if ((rc = launch_all(false))) return rc;
      zn_note_kernel(timed_out ? "(one-pass look-back timed out: four-kernel encoder)" : "(one-pass speculation failed: four-kernel encoder)");
    } else if (w.op_penalty) w.op_penalty >>= 1;
  }
  if ((rc = ws_release(w, stream, table))) return rc;
  ZN_HIP(hipStreamSynchronize(stream));
  for (size_t i = 0; i < count; i++) items[i].body_len = (size_t)w.h_totals[i];
  if ((uint32_t)w.h_totals[count]) return ZN_E_CORRUPT;   // internal consistency check of the encoder failed
  return ZN_OK;
}

int zn_compress_delta_dev(const void* d_src, const void* d_delta, size_t n, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                          float threshold, void* d_body, size_t body_cap, size_t* body_len, void* stream_) {
  if (!body_len) return ZN_E_ARG;
  zn_cbatch_item it; it.d_delta = d_delta;
  it.d_src = d_src; it.n = n; it.num_buf = num_buf; it.bits_mode = bits_mode; it.bytes_mode = bytes_mode; it.chunk = chunk;
  it.threshold = threshold; it.d_body = d_body; it.body_cap = body_cap; it.body_len = 0;
  int rc;
  try { rc = compress_items(&it, 1, (hipStream_t)stream_); } catch (...) { return ZN_E_ALLOC; }
  *body_len = it.body_len;
  return rc;
}

int zn_compress_dev(const void* d_src, size_t n, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                    float threshold, void* d_body, size_t body_cap, size_t* body_len, void* stream_) {
  return zn_compress_delta_dev(d_src, nullptr, n, num_buf, bits_mode, bytes_mode, chunk, threshold, d_body, body_cap, body_len, stream_);
}

int zn_compress_batch_dev(zn_cbatch_item* items, size_t count, void* stream_) {
  try { return compress_items(items, count, (hipStream_t)stream_); } catch (...) { return ZN_E_ALLOC; }
}

// Decode `count` tensors in one set of launches per plane count.  A single tensor travels to the kernels as an
// argument; a batch as a segment table in device memory.
#define ZN_REST_MAX_CHUNKS 2048u          // (512 MiB of 256 KiB chunks)
#define ZN_REST_TAIL_MAX_CHUNKS 24576u    // (6 GiB: a launch WITH partial chunks stays with the rest instance up to here)
// A batch that mixes one-plane tensors (fp8: dense codes, the decode is LDS-bound at a fifth of the HBM roofline) with two- / four-plane ones (memory-bound) decodes
// the two kinds CONCURRENTLY: the one-plane launches go to a second stream, forked from and joined to the caller's with events, so that the dispatcher fills a CU with
// workgroups of both kernels — one kind waits on LDS look-ups while the other waits on HBM.  ZIPNN_AMD_DECODE_OVERLAP=0 keeps everything on the caller's stream.
static int zn_decode_overlap_on() {
  static std::atomic<int> v{-1};
  int x = v.load(std::memory_order_relaxed);
  if (x < 0) { const char* e = getenv("ZIPNN_AMD_DECODE_OVERLAP"); x = (e && e[0] >= '0' && e[0] <= '9') ? e[0] - '0' : 1; v.store(x, std::memory_order_relaxed); }
  return x;
}
static int decompress_items(const zn_batch_item* items, size_t count, hipStream_t stream, int check) {
  if (count && !items) return ZN_E_ARG;
  std::vector<ZnSeg> segs[3];                    // by plane count: 1, 2, 4
  uint64_t pk_of[3] = {0, 0, 0}, k_of[3] = {0, 0, 0}; uint64_t wg_of[3] = {0, 0, 0}, tail_of[3] = {0, 0, 0};
  bool delta_of[3] = {false, false, false}, rest_ok[3] = {true, true, true};
  uint64_t total_chunks = 0;
  bool any_delta = false;
  uint64_t full_chunks = 0; bool all_rotated = true; uint64_t tail_wgs = 0;
  uint64_t kq[3] = {0, 0, 0};                    // chunks per launch (one launch per plane count)
  for (size_t i = 0; i < count; i++) {
    total_chunks += zn_num_chunks(items[i].orig_size, items[i].chunk); if (items[i].d_delta) any_delta = true;
    // (FULL chunks: a partial last chunk is the tail workgroups' — the workgroup its group index maps to only skips it, and counting it would push a tensor of exactly
    //  n rounds into n + 1: 6 144 chunks + a tail took single-chunk groups, 0.64 ms, where three-chunk groups take 0.59)
    if (items[i].chunk) kq[items[i].num_buf == 1 ? 0 : items[i].num_buf == 2 ? 1 : 2] += items[i].orig_size / items[i].chunk;
    if (items[i].chunk) full_chunks += items[i].orig_size / items[i].chunk;
    // (tensors without the sign rotate — fp16, fp8, integers: their Huffman planes are dense codes, which the wide kernel parses and declines)
    if (!(items[i].bits_mode == 1 && items[i].num_buf > 1)) all_rotated = false;
    // (… a partial last chunk: its tail workgroups ride at the front of the wide launch and its merge workgroups at the end of it, as in a fused launch; the tail
    //  workgroups take workgroup slots of their own — 64 MiB + 250 KB = 256 chunks + 8 took two rounds of the 16-wave form on 256 CUs, 113 µs against the 8-wave form's 73)
    if (items[i].chunk && items[i].orig_size % items[i].chunk) tail_wgs += 4u * (uint64_t)(items[i].num_buf > 0 ? items[i].num_buf : 1) + 32u;      // (… + the tensor's merge workgroups at the end of the grid)
  }
  const int wide = zn_decode_use_wide(full_chunks, any_delta, all_rotated, tail_wgs);       // small calls: a 16-wave workgroup per full chunk (zn_decode_wide.hpp)
  uint32_t ncg_of[3];
  for (int q = 0; q < 3; q++) ncg_of[q] = wide ? 1u : zn_decode_fused_group(kq[q]);
  for (size_t i = 0; i < count; i++) {
    const zn_batch_item& it = items[i];
    ZnSeg sg;
    int rc = check_geom(it.orig_size, it.num_buf, it.bytes_mode, it.chunk, &sg.g, it.bits_mode);
    if (rc) return rc;
    if (it.body_len < 9u * (size_t)sg.g.P * sg.g.K) return ZN_E_CORRUPT;
    if (it.orig_size == 0) continue;
    if (!it.d_body || !it.d_dst) return ZN_E_ARG;
    const int q = sg.g.P == 1 ? 0 : sg.g.P == 2 ? 1 : 2;
    const uint32_t ncg = ncg_of[q];
    sg.body = (const uint8_t*)it.d_body; sg.body_len = it.body_len; sg.dst = (uint8_t*)it.d_dst;
    sg.xr = (const uint8_t*)it.d_delta;
    if (sg.xr) delta_of[q] = true;
    // (a LARGE call in a geometry the fused kernel takes no chunk of — its rule: whole rows per stream, a 16-byte aligned destination — goes to the generic
    //  KERNELS, which spread it over the whole chip with one wave per plane; the fused kernel's rest instance decodes the odd chunk, or the odd small tensor)
    { const uint64_t unit = 64ull * (sg.g.P == 1 ? 16u : 8u);
      if ((it.chunk % (4ull * sg.g.P * unit) != 0 || (((uintptr_t)it.d_dst) & 15u) != 0) && total_chunks > 1024u) rest_ok[q] = false; }
    sg.chunk0 = k_of[q]; sg.desc0 = pk_of[q]; sg.wg0 = (uint32_t)wg_of[q]; sg.ncg = ncg;
    sg.tail0 = (uint32_t)tail_of[q]; sg.has_tail = (it.orig_size % it.chunk) != 0 ? 1u : 0u;   // partial last chunk
    if (sg.has_tail) tail_of[q] += sg.g.P;
    k_of[q] += sg.g.K; pk_of[q] += (uint64_t)sg.g.P * sg.g.K; wg_of[q] += (sg.g.K + ncg - 1u) / ncg;
    if (pk_of[q] > 0x7FFFFFFFull || wg_of[q] > 0x7FFFFFFFull) return ZN_E_ARG;
    segs[q].push_back(sg);
  }
  const uint64_t all_k = k_of[0] + k_of[1] + k_of[2], all_pk = pk_of[0] + pk_of[1] + pk_of[2];
  if (all_k == 0) return ZN_OK;
  int dev = 0;
  ZN_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return ZN_E_ARG;
  t_kernels.clear();
  std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
  Workspace& w = g_ws[dev];
  int rc;
  const size_t nseg_all = segs[0].size() + segs[1].size() + segs[2].size();
  const bool table = nseg_all > 1;
  if ((rc = ws_reserve(w, WS_META_C, all_pk * sizeof(ZnPlaneDesc)))) return rc;
  if ((rc = ws_reserve(w, WS_META_B, all_k))) return rc;                     // per-chunk "done by the fused kernel" flags
  if ((rc = ws_reserve(w, WS_ENC, all_pk))) return rc;                       // … and the same per (plane, chunk)
  const uint64_t all_tail = tail_of[0] + tail_of[1] + tail_of[2];
  if ((rc = ws_reserve(w, WS_PLANES, all_tail * ZN_TAIL_SLOT))) return rc;   // decoded Huffman planes of partial last chunks
  const size_t sync_off = (4u * all_tail + 15u) & ~(size_t)15u;              // … and whether the tail workgroups produced them (four flag bytes per plane: one per huff0 stream); behind that two flag words per tensor with a partial chunk
  if ((rc = ws_reserve(w, WS_META_A, sync_off + 2u * sizeof(uint32_t) * all_tail))) return rc;
  if ((rc = ws_reserve(w, WS_WORDS, ZN_WORDS_BYTES))) return rc;
  if ((rc = ws_host_words(w))) return rc;
  if (table) {
    if ((rc = ws_reserve(w, WS_SEGS, nseg_all * sizeof(ZnSeg)))) return rc;
    if (w.h_segs_cap < nseg_all) {
      if (w.h_segs) { ZN_HIP(hipHostFree(w.h_segs)); w.h_segs = nullptr; w.h_segs_cap = 0; }
      ZN_HIP(hipHostMalloc((void**)&w.h_segs, nseg_all * sizeof(ZnSeg), hipHostMallocDefault));
      w.h_segs_cap = nseg_all;
    }
  }
  if ((rc = ws_acquire(w, stream))) return rc;
  // this call's status slot (status word + the three "left to the generic kernels" counters)
  const uint32_t slot = (uint32_t)(w.status_gen % ZN_STATUS_SLOTS);
  w.slot_gen[slot] = ++w.status_gen; w.last_slot = slot;
  if (!check) { t_status_dev = dev; t_status_slot = slot; t_status_gen = w.status_gen; }
  uint32_t* d_status = (uint32_t*)w.buf[WS_WORDS] + 16 + 4u * slot;
  w.last_K = all_k; w.last_tails = all_tail;
  // status + the three "left to the generic kernels" counters (a wide call: its first kernel zeroes them)
  bool status_zeroed = false;
  if (!wide || all_tail) { ZN_HIP(hipMemsetAsync(d_status, 0, 4 * sizeof(uint32_t), stream)); status_zeroed = true; }
  if (all_tail) ZN_HIP(hipMemsetAsync(w.buf[WS_META_A], 0, sync_off + 2u * sizeof(uint32_t) * all_tail, stream));
  if (table) {
    // the previous batched call may still be reading the pinned staging: wait for it on the host
    ZN_HIP(hipEventSynchronize(w.busy));
    size_t o = 0;
    for (int q = 0; q < 3; q++) for (const ZnSeg& sg : segs[q]) w.h_segs[o++] = sg;
    ZN_HIP(hipMemcpyAsync(w.buf[WS_SEGS], w.h_segs, nseg_all * sizeof(ZnSeg), hipMemcpyHostToDevice, stream));
  }
  size_t seg_base = 0; uint64_t k_base = 0, pk_base = 0, tail_base = 0;
  const hipStream_t stream_main = stream;
  const bool overlap = zn_decode_overlap_on() && k_of[0] >= 512u && k_of[1] + k_of[2] >= 512u;      // (both kinds fill the chip: below that the fork and join cost more than they hide)
  if (overlap) {
    // (TWO streams of our own, created one behind the other — the runtime deals its hardware queues out in turn, so these two do not share one; the caller's stream
    //  may share a queue with either: under torch.distributed, with RCCL's streams in the process, the caller's stream + ONE new stream ran the two kinds one after the
    //  other.  Equal priorities: a higher or a lower one for the fp8 stream measured 4-5 % slower, profiles/r05_llama8b_overlap.txt)
    if (!w.dstream) ZN_HIP(hipStreamCreateWithFlags(&w.dstream, hipStreamNonBlocking));
    if (!w.dstream2) ZN_HIP(hipStreamCreateWithFlags(&w.dstream2, hipStreamNonBlocking));
    if (!w.djoin2) ZN_HIP(hipEventCreateWithFlags(&w.djoin2, hipEventDisableTiming));
    if (!w.dfork) ZN_HIP(hipEventCreateWithFlags(&w.dfork, hipEventDisableTiming));
    if (!w.djoin) ZN_HIP(hipEventCreateWithFlags(&w.djoin, hipEventDisableTiming));
    ZN_HIP(hipEventRecord(w.dfork, stream_main));          // behind the memsets and the segment table
    // (from here to the join nothing returns: a failure is remembered, the side streams are joined — or, if even that fails, drained — and THEN the call fails;
    //  kernels already queued on them must not outlive the call's claim on d_dst and the workspace)
    hipError_t e1 = hipStreamWaitEvent(w.dstream, w.dfork, 0), e2 = hipStreamWaitEvent(w.dstream2, w.dfork, 0);
    if (e1 != hipSuccess || e2 != hipSuccess) { t_hip_err = std::string("hipStreamWaitEvent(fork): ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2); (void)hipGetLastError(); return ZN_E_HIP; }     // (nothing queued on the side streams yet)
    zn_note_kernel("(two streams)");                        // (zn_last_kernels: the launches that follow are split over the two)
  }
  const bool fp8_last = overlap && zn_decode_overlap_on() == 2;
  size_t seg_b[3]; uint64_t k_b[3], pk_b[3], tail_b[3];          // where plane count q starts in the launch-wide arrays
  for (int q = 0; q < 3; q++) { seg_b[q] = seg_base; k_b[q] = k_base; pk_b[q] = pk_base; tail_b[q] = tail_base; seg_base += segs[q].size(); k_base += k_of[q]; pk_base += pk_of[q]; tail_base += tail_of[q]; }
  for (int qi = 0; qi < 3; qi++) {
    const int q = fp8_last ? (qi + 1) % 3 : qi;
    if (segs[q].empty()) continue;
    const int P = q == 0 ? 1 : q == 1 ? 2 : 4;
    stream = !overlap ? stream_main : (q == 0 ? w.dstream : w.dstream2);
    seg_base = seg_b[q]; k_base = k_b[q]; pk_base = pk_b[q]; tail_base = tail_b[q];
    const ZnSeg* d_segs = table ? (const ZnSeg*)w.buf[WS_SEGS] + seg_base : nullptr;
    const uint32_t nseg = (uint32_t)segs[q].size();
    uint8_t* d_done = (uint8_t*)w.buf[WS_META_B] + k_base;
    ZnPlaneDesc* d_descs = (ZnPlaneDesc*)w.buf[WS_META_C] + pk_base;
    uint8_t* d_tails = (uint8_t*)w.buf[WS_PLANES] + tail_base * ZN_TAIL_SLOT;
    uint8_t* d_tail_done = (uint8_t*)w.buf[WS_META_A] + 4u * tail_base;
    uint8_t* d_pdone = (uint8_t*)w.buf[WS_ENC] + pk_base;
    // (behind the wide kernel, in launches without partial chunks, the fused kernel's rest instance also does the generic kernels' job)
    // (round 5, three boxes, interleaved: at 4 GiB the plain instance + the two generic launches behind it decode 1.0-1.4 % FASTER than the rest instance alone —
    //  1.445 / 1.445 / 1.500 against 1.460 / 1.466 / 1.516 ms, profiles/r05_rest_instance_ab.txt —: the generic path's code in its cold paths costs the rest instance
    //  twelve spilled registers.  What it saves is two launches, ≈ 4-8 us: it is the instance of calls of up to ZN_REST_MAX_CHUNKS chunks.)
    // (… unless the launch has partial chunks: their merge workgroups ride in the rest instance — 1.0-1.4 % of a large call — where the plain instance needs the two
    //  generic launches behind it, and the generic merge of ONE partial chunk takes 33 µs: 1 GiB + 200 KB 445 µs that way, 417 this way; 4 GiB + 200 KB 1 573 / 1 555.
    //  From about 6 GiB on the percent outweighs the 33 µs: the Llama-3-8B batch — 16 GB of bf16, its 8 KB norm vectors the partial chunks — 12.12 ms against 11.94)
    if (total_chunks > ZN_REST_MAX_CHUNKS && (tail_of[q] == 0 || k_of[q] > ZN_REST_TAIL_MAX_CHUNKS)) rest_ok[q] = false;
    const bool rest = zn_launch_decode_fused(P, segs[q][0], d_segs, nseg, (uint32_t)wg_of[q], d_done, d_pdone, d_status, (uint32_t)tail_of[q], d_tails,
                                             d_tail_done, delta_of[q], wide, status_zeroed, rest_ok[q] ? d_descs : nullptr,
                                             tail_of[q] ? (uint32_t*)((uint8_t*)w.buf[WS_META_A] + sync_off) + 2u * tail_base : nullptr, stream);
    status_zeroed = true;
    if (!rest) zn_launch_decode_generic(P, segs[q][0], d_segs, nseg, pk_of[q], k_of[q], d_descs, d_status, d_done, d_pdone, d_tails, d_tail_done, stream);
  }
  stream = stream_main;
  if (overlap) {
    hipError_t ej = hipEventRecord(w.djoin, w.dstream);
    if (ej == hipSuccess) ej = hipStreamWaitEvent(stream_main, w.djoin, 0);
    hipError_t ej2 = hipEventRecord(w.djoin2, w.dstream2);
    if (ej2 == hipSuccess) ej2 = hipStreamWaitEvent(stream_main, w.djoin2, 0);
    if (ej != hipSuccess || ej2 != hipSuccess) {           // the join itself failed: drain the side streams on the host, so that nothing of this call is still running when it returns
      (void)hipStreamSynchronize(w.dstream); (void)hipStreamSynchronize(w.dstream2);
      t_hip_err = std::string("two-stream join: ") + hipGetErrorString(ej != hipSuccess ? ej : ej2); (void)hipGetLastError();
      (void)ws_release(w, stream_main, table);
      return ZN_E_HIP;
    }
  }
  { const hipError_t el = hipGetLastError();               // a launch that failed: the workspace is still handed back in stream order (the side streams are joined above)
    if (el != hipSuccess) { t_hip_err = std::string("kernel launch: ") + hipGetErrorString(el); (void)ws_release(w, stream_main, table); return ZN_E_HIP; } }
  if (check) ZN_HIP(hipMemcpyAsync(w.h_status, d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  if ((rc = ws_release(w, stream, table))) return rc;
  if (check) {
    ZN_HIP(hipStreamSynchronize(stream));
    const uint32_t st = *w.h_status;
    if (st & ZN_DEV_BAD_TYPE) return ZN_E_TYPE;
    if (st & ZN_DEV_CORRUPT) return ZN_E_CORRUPT;
    if (st & ZN_DEV_SYNC_TIMEOUT) { t_hip_err = "a workgroup gave up waiting for another one of the same launch (device preempted or faulted?)"; return ZN_E_TIMEOUT; }
  }
  return ZN_OK;
}

int zn_decompress_delta_dev(const void* d_body, size_t body_len, const void* d_delta, int num_buf, int bits_mode, int bytes_mode,
                            size_t chunk, size_t orig_size, void* d_dst, void* stream_, int check) {
  zn_batch_item it; it.d_delta = d_delta;
  it.d_body = d_body; it.body_len = body_len; it.d_dst = d_dst; it.orig_size = orig_size;
  it.num_buf = num_buf; it.bits_mode = bits_mode; it.bytes_mode = bytes_mode; it.chunk = chunk;
  try {
    return decompress_items(&it, 1, (hipStream_t)stream_, check);
  } catch (...) { return ZN_E_ALLOC; }
}

int zn_decompress_dev(const void* d_body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                      size_t orig_size, void* d_dst, void* stream_, int check) {
  return zn_decompress_delta_dev(d_body, body_len, nullptr, num_buf, bits_mode, bytes_mode, chunk, orig_size, d_dst, stream_, check);
}

int zn_decompress_batch_dev(const zn_batch_item* items, size_t count, void* stream_, int check) {
  try {                                          // (the segment lists are std::vectors: nothing may unwind through the C ABI)
    return decompress_items(items, count, (hipStream_t)stream_, check);
  } catch (...) { return ZN_E_ALLOC; }
}

// Large host buffers take a three-stage pipeline over slices of the chunks — upload slice i + 1 | code slice i | download slice i - 1 —
// so that both directions of the PCIe link work at once (defined below, behind the range helpers).  0 = not applicable: the one-shot path.
static int zn_host_slices(size_t n, size_t chunk, bool decompress, bool direct);
static int zn_compress_host_pipelined(const void* hdr, size_t hdr_len, const void* src, size_t n, int num_buf, int bits_mode, int bytes_mode,
                                      size_t chunk, float threshold, int dev, int S, bool direct, void* dst, size_t dst_cap, size_t* dst_len);
static int zn_decompress_host_pipelined(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                                        size_t orig_size, int dev, int S, bool direct, void* dst);

int zn_compress_delta(const void* hdr, size_t hdr_len, const void* src, const void* delta, size_t n, int num_buf, int bits_mode,
                      int bytes_mode, size_t chunk, float threshold, int device, void* dst, size_t dst_cap, size_t* dst_len) {
  if (!dst_len || (hdr_len && !hdr) || (n && !src) || !dst) return ZN_E_ARG;
  if (zn_device_count() <= 0) return ZN_E_NODEV;
  DeviceScope scope(device);              // (restored on every return path)
  if (!scope.ok) { t_hip_err = "hipSetDevice"; return ZN_E_HIP; }
  const size_t bound = zn_compress_bound(n, num_buf, chunk, 0);
  zn_host_thp_hint(dst, dst_cap);         // (a fresh result buffer faults in 2 MiB at a time, and is freed as fast: zn_host_pipe.hpp)
  // The DIRECT transfers (the caller's buffers pinned, DMA straight between them and HBM: zn_host_pipe.hpp) are for calls whose RESULT buffer is already
  // backed by pages — a recycled buffer, or one its owner has written before.  A call that has to fault its result in goes the staged way of rounds 1-5 in
  // BOTH directions: page faults in a process with pinned user memory were measured slow and disruptive to every DMA in flight (profiles/r06_host_path.txt).
  // (a result buffer from zn_host_alloc needs none of this: its downloads are plain DMAs whichever way the call is cut — zn_host_pipe_copy sees to that)
  const bool direct = zn_host_pipe_detail::direct_enabled() && !zn_arena().covers(dst, dst_cap) && zn_host_resident(dst, hdr_len + 9u * (size_t)num_buf * zn_num_chunks(n, chunk) + n / 2);
  if (!delta && (num_buf == 1 || num_buf == 2 || num_buf == 4) && chunk) {
    const int S = zn_host_slices(n, chunk, false, direct);
    if (S >= 2) {
      int dev_ = 0;
      ZN_HIP(hipGetDevice(&dev_));
      if (dev_ < 0 || dev_ >= 64) return ZN_E_ARG;
      try { return zn_compress_host_pipelined(hdr, hdr_len, src, n, num_buf, bits_mode, bytes_mode, chunk, threshold, dev_, S, direct, dst, dst_cap, dst_len); }
      catch (...) { return ZN_E_ALLOC; }
    }
  }
  // device staging for host buffers: cached with the workspace (grow-only), one host-path call at a time per device
  int dev = 0;
  ZN_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return ZN_E_ARG;
  std::lock_guard<std::mutex> hk(g_host_mu[dev]);
  void* d_src = nullptr; void* d_body = nullptr; void* d_delta = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
    Workspace& w = g_ws[dev];
    int rc0;
    if (delta && n) { if ((rc0 = ws_reserve(w, WS_HOST_DELTA, n))) return rc0; d_delta = w.buf[WS_HOST_DELTA]; }
    if ((rc0 = ws_reserve(w, WS_HOST_IN, n ? n : 16))) return rc0;
    if ((rc0 = ws_reserve(w, WS_HOST_OUT, bound ? bound : 16))) return rc0;
    d_src = w.buf[WS_HOST_IN]; d_body = w.buf[WS_HOST_OUT];
  }
  int rc = ZN_OK; size_t body_len = 0;
  do {
    // (pageable host memory through the pinned, multi-threaded pipe of zn_host_pipe.hpp; g_host_mu serialises its use)
    ZnHostPipe& pipe = g_ws[dev].pipe;
    if (n && zn_host_pipe_copy(pipe, d_src, const_cast<void*>(src), n, true, nullptr, !direct) != hipSuccess) { rc = ZN_E_HIP; t_hip_err = "host pipe H2D"; break; }
    if (d_delta && zn_host_pipe_copy(pipe, d_delta, const_cast<void*>(delta), n, true, nullptr, !direct) != hipSuccess) { rc = ZN_E_HIP; t_hip_err = "host pipe H2D"; break; }
    rc = zn_compress_delta_dev(d_src, d_delta, n, num_buf, bits_mode, bytes_mode, chunk, threshold, d_body, bound ? bound : 16, &body_len, nullptr);
    if (rc) break;
    if (hdr_len + body_len > dst_cap) { rc = ZN_E_CAP; break; }
    if (hdr_len) memcpy(dst, hdr, hdr_len);
    if (body_len && zn_host_pipe_copy(pipe, d_body, (uint8_t*)dst + hdr_len, body_len, false, nullptr, !direct) != hipSuccess) { rc = ZN_E_HIP; t_hip_err = "host pipe D2H"; break; }
    *dst_len = hdr_len + body_len;
    if (hdr_len >= 32) { const uint64_t total = *dst_len; memcpy((uint8_t*)dst + 24, &total, 8); }   // zipnn_core.c:121
  } while (0);
  return rc;
}

int zn_compress(const void* hdr, size_t hdr_len, const void* src, size_t n, int num_buf, int bits_mode, int bytes_mode,
                size_t chunk, float threshold, int device, void* dst, size_t dst_cap, size_t* dst_len) {
  return zn_compress_delta(hdr, hdr_len, src, nullptr, n, num_buf, bits_mode, bytes_mode, chunk, threshold, device, dst, dst_cap, dst_len);
}

int zn_decompress_delta(const void* body, size_t body_len, const void* delta, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                        size_t orig_size, int device, void* dst) {
  if ((body_len && !body) || (orig_size && !dst)) return ZN_E_ARG;
  if (zn_device_count() <= 0) return ZN_E_NODEV;
  DeviceScope scope(device);              // (restored on every return path)
  if (!scope.ok) { t_hip_err = "hipSetDevice"; return ZN_E_HIP; }
  int dev = 0;
  ZN_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return ZN_E_ARG;
  zn_host_thp_hint(dst, orig_size);
  // (an arena result stays one shot like a pageable one: measured, the slice pipeline with a STAGED upload beside the arena's download DMAs is slower than their sum —
  //  36-38 ms against 33 for 1 GiB; the copy threads and the two DMA directions meet in the host's memory: profiles/r06_host_path.txt)
  const bool direct = zn_host_pipe_detail::direct_enabled() && !zn_arena().covers(dst, orig_size) && zn_host_resident(dst, orig_size);      // (see zn_compress_delta)
  if (!delta && (num_buf == 1 || num_buf == 2 || num_buf == 4) && chunk) {
    const int S = zn_host_slices(orig_size, chunk, true, direct);
    if (S >= 2) {
      try { return zn_decompress_host_pipelined(body, body_len, num_buf, bits_mode, bytes_mode, chunk, orig_size, dev, S, direct, dst); }
      catch (...) { return ZN_E_ALLOC; }
    }
  }
  std::lock_guard<std::mutex> hk(g_host_mu[dev]);
  void* d_body = nullptr; void* d_dst = nullptr; void* d_delta = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
    Workspace& w = g_ws[dev];
    int rc0;
    if (delta && orig_size) { if ((rc0 = ws_reserve(w, WS_HOST_DELTA, orig_size))) return rc0; d_delta = w.buf[WS_HOST_DELTA]; }
    if ((rc0 = ws_reserve(w, WS_HOST_IN, body_len ? body_len : 16))) return rc0;
    if ((rc0 = ws_reserve(w, WS_HOST_OUT, orig_size ? orig_size : 16))) return rc0;
    d_body = w.buf[WS_HOST_IN]; d_dst = w.buf[WS_HOST_OUT];
  }
  int rc = ZN_OK;
  do {
    ZnHostPipe& pipe = g_ws[dev].pipe;
    if (body_len && zn_host_pipe_copy(pipe, d_body, const_cast<void*>(body), body_len, true, nullptr, !direct) != hipSuccess) { rc = ZN_E_HIP; t_hip_err = "host pipe H2D"; break; }
    if (d_delta && zn_host_pipe_copy(pipe, d_delta, const_cast<void*>(delta), orig_size, true, nullptr, !direct) != hipSuccess) { rc = ZN_E_HIP; t_hip_err = "host pipe H2D"; break; }
    rc = zn_decompress_delta_dev(d_body, body_len, d_delta, num_buf, bits_mode, bytes_mode, chunk, orig_size, d_dst, nullptr, 1);
    if (rc) break;
    if (orig_size && zn_host_pipe_copy(pipe, d_dst, dst, orig_size, false, nullptr, !direct) != hipSuccess) { rc = ZN_E_HIP; t_hip_err = "host pipe D2H"; break; }
  } while (0);
  return rc;
}

int zn_decompress(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                  size_t orig_size, int device, void* dst) {
  return zn_decompress_delta(body, body_len, nullptr, num_buf, bits_mode, bytes_mode, chunk, orig_size, device, dst);
}

// ---- one call, several GPUs (`const int* devices, int ndev`; north_star: "chunks partition embarrassingly
// across the GPUs of one node on independent HIP streams, no collectives") ------------------------------------------------
// Chunks are independent, so device i codes the contiguous chunk range [i K / G, (i + 1) K / G) as a tensor of its own, on a
// host thread of its own (per-device locks, workspaces and streams: zn_api.hip above), and the host does the plane-major
// bookkeeping: types and payload of the ranges concatenated per plane, cumSizes re-based by what the earlier ranges put into
// the plane.  The frame is byte-identical to the one a single device writes (zipnn_amd/sharding.py states the same
// arithmetic in numpy for the one-process-per-GPU path).  The reference's counterpart is its pthread fan-out over chunks
// (csrc/zipnn_core.c:294-301, 509-523).
extern "C++" {
namespace {
// host staging of the multi-device COMPRESS calls (a range's body before it is placed in the frame): one grow-only buffer per
// range slot, kept between calls (a fresh 0.5 GiB vector per range cost more in zero-fill and first-touch page faults than the
// coding itself); g_multi_mu guards it.  Decompress stages nothing on the host: the ranges' payload slices go from the caller's
// body straight into HBM.
std::mutex g_multi_mu;
struct ZnStage { uint8_t* p = nullptr; size_t cap = 0; };
ZnStage g_multi_stage[64];
uint8_t* zn_stage(int slot, size_t need) {
  ZnStage& s = g_multi_stage[slot];
  if (s.cap < need) { free(s.p); s.p = (uint8_t*)malloc(need + (need >> 3) + 4096); s.cap = s.p ? need + (need >> 3) + 4096 : 0; }
  return s.p;
}
struct ZnRange { size_t lo, hi; };                 // chunk range of one device
ZnRange zn_range_of(size_t K, int g, int G) { return ZnRange{(size_t)g * K / (size_t)G, (size_t)(g + 1) * K / (size_t)G}; }
uint64_t zn_rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// the worker threads of one call: joined on EVERY way out of the scope (a std::thread destroyed while joinable terminates the
// process), and a thread that cannot be started — EAGAIN under a thread limit — fails the call instead of throwing through it
struct ZnWorkers {
  std::vector<std::thread> th;
  bool failed = false;
  template <class F> void start(F&& f) { if (failed) return; try { th.emplace_back(std::forward<F>(f)); } catch (...) { failed = true; } }
  void join() { for (auto& t : th) if (t.joinable()) t.join(); }
  ~ZnWorkers() { join(); }
};

// The size tables of a frame body, checked once: plane bases, and cumSizes monotone and inside the payload (every per-range
// sub-body below is cut out of them).
struct ZnBodyView { const uint8_t* types; const uint8_t* cums; const uint8_t* pay; size_t P, K, pay_len; std::vector<uint64_t> base; };
int zn_body_view(const void* body, size_t body_len, int num_buf, size_t chunk, size_t orig_size, ZnBodyView* v) {
  const size_t P = (size_t)num_buf, K = (orig_size + chunk - 1) / chunk;
  if (K > body_len / (9 * P)) return ZN_E_CORRUPT;
  const uint8_t* b = (const uint8_t*)body;
  v->P = P; v->K = K; v->types = b; v->cums = b + P * K; v->pay = v->cums + 8 * P * K; v->pay_len = body_len - 9 * P * K;
  v->base.assign(P, 0);
  uint64_t acc = 0;
  for (size_t p = 0; p < P; p++) {
    v->base[p] = acc;
    uint64_t prev = 0;
    for (size_t i = 0; i < K; i++) { const uint64_t c = zn_rd64(v->cums + 8 * (p * K + i)); if (c < prev || c > v->pay_len - acc) return ZN_E_CORRUPT; prev = c; }
    acc += prev;
  }
  return ZN_OK;
}
uint64_t zn_cum_before(const ZnBodyView& v, size_t p, size_t c) { return c ? zn_rd64(v.cums + 8 * (p * v.K + c - 1)) : 0; }
size_t zn_sub_need(const ZnBodyView& v, const ZnRange& r) {
  size_t need = 9 * v.P * (r.hi - r.lo);
  for (size_t p = 0; p < v.P; p++) need += (size_t)(zn_cum_before(v, p, r.hi) - zn_cum_before(v, p, r.lo));
  return need;
}

// Decode the chunk range `r` of a host-resident frame body on `device`.  The range's sub-body is put together IN HBM: its
// size tables (9 P k bytes, re-based) are built on the host, its P payload slices go from the caller's body through the
// device's pinned pipe to their place behind them — no host copy of the payload.  The decoded bytes land in `d_dst` (device
// memory of `device`, when given) or are copied to `h_dst` (host).
int zn_decode_range(const ZnBodyView& v, const ZnRange& r, int num_buf, int bits_mode, int bytes_mode, size_t chunk, size_t orig_size,
                    int device, void* h_dst, void* d_dst) {
  const size_t P = v.P, K = v.K, k = r.hi - r.lo;
  if (!k) return ZN_OK;
  const size_t off = r.lo * chunk, len = (r.hi * chunk < orig_size ? r.hi * chunk : orig_size) - off;
  DeviceScope scope(device);              // (restored on every return path)
  if (!scope.ok) { t_hip_err = "hipSetDevice"; return ZN_E_HIP; }
  int dev = 0;
  ZN_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return ZN_E_ARG;
  std::lock_guard<std::mutex> hk(g_host_mu[dev]);
  const size_t need = zn_sub_need(v, r);
  uint8_t* d_in = nullptr; void* d_out = d_dst;
  {
    std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
    Workspace& w = g_ws[dev];
    int rc0;
    if ((rc0 = ws_reserve(w, WS_HOST_IN, need + 16))) return rc0;
    d_in = (uint8_t*)w.buf[WS_HOST_IN];
    if (!d_out) { if ((rc0 = ws_reserve(w, WS_HOST_OUT, len ? len : 16))) return rc0; d_out = w.buf[WS_HOST_OUT]; }
  }
  std::vector<uint8_t> meta(9 * P * k);
  std::vector<uint64_t> s0(P);
  for (size_t p = 0; p < P; p++) {
    s0[p] = zn_cum_before(v, p, r.lo);
    memcpy(meta.data() + p * k, v.types + p * K + r.lo, k);
    for (size_t i = 0; i < k; i++) { const uint64_t c = zn_rd64(v.cums + 8 * (p * K + r.lo + i)) - s0[p]; memcpy(meta.data() + P * k + 8 * (p * k + i), &c, 8); }
  }
  ZnHostPipe& pipe = g_ws[dev].pipe;
  if (zn_host_pipe_copy(pipe, d_in, meta.data(), meta.size(), true) != hipSuccess) { t_hip_err = "host pipe H2D"; (void)hipGetLastError(); return ZN_E_HIP; }
  size_t at = meta.size();
  for (size_t p = 0; p < P; p++) {
    const size_t m = (size_t)(zn_cum_before(v, p, r.hi) - s0[p]);
    if (m && zn_host_pipe_copy(pipe, d_in + at, const_cast<uint8_t*>(v.pay + v.base[p] + s0[p]), m, true) != hipSuccess) { t_hip_err = "host pipe H2D"; (void)hipGetLastError(); return ZN_E_HIP; }
    at += m;
  }
  int rc = zn_decompress_dev(d_in, need, num_buf, bits_mode, bytes_mode, chunk, len, d_out, nullptr, 1);
  if (rc) return rc;
  if (h_dst && len && zn_host_pipe_copy(pipe, d_out, (uint8_t*)h_dst + off, len, false) != hipSuccess) { t_hip_err = "host pipe D2H"; (void)hipGetLastError(); return ZN_E_HIP; }
  return ZN_OK;
}

// Bodies of CONSECUTIVE chunk ranges (part i: ks[i] chunks, plen[i] bytes = types ‖ cumSizes ‖ payload of that range alone) ->
// hdr ‖ one body: types and payload concatenated per plane, cumSizes re-based by what the earlier ranges put into the plane;
// one thread per range does the copying.
int zn_assemble_ranges(const void* hdr, size_t hdr_len, const std::vector<const uint8_t*>& part, const std::vector<size_t>& plen,
                       const std::vector<size_t>& ks, size_t P, void* dst, size_t dst_cap, size_t* dst_len) {
  const size_t G = part.size();
  size_t K = 0, pay_total = 0;
  for (size_t g = 0; g < G; g++) {
    const size_t k = ks[g];
    if (!k) continue;
    if (!part[g] || plen[g] < 9 * P * k) return ZN_E_ARG;
    K += k; pay_total += plen[g] - 9 * P * k;
  }
  const size_t total = hdr_len + 9 * P * K + pay_total;
  if (total > dst_cap) return ZN_E_CAP;
  uint8_t* o = (uint8_t*)dst;
  if (hdr_len) memcpy(o, hdr, hdr_len);
  uint8_t* types = o + hdr_len; uint8_t* cums = types + P * K; uint8_t* pay = cums + 8 * P * K;
  // where every (plane, range) piece goes and what re-bases its cumSizes
  std::vector<size_t> dst_at(G * P, 0), c0(G, 0); std::vector<uint64_t> rebase(G * P, 0);
  { size_t c = 0; for (size_t g = 0; g < G; g++) { c0[g] = c; c += ks[g]; } }
  // every part by itself: its cumSizes non-decreasing inside each plane and its plane totals adding up to ITS OWN payload length — a
  // global sum alone lets a part that is short by what another one is long through, and the copies below then read past its end
  for (size_t g = 0; g < G; g++) {
    const size_t k = ks[g];
    if (!k) continue;
    const uint8_t* c = part[g] + P * k;
    const size_t own = plen[g] - 9 * P * k;
    size_t sum = 0;
    for (size_t p = 0; p < P; p++) {
      uint64_t prev = 0;
      for (size_t i = 0; i < k; i++) { const uint64_t v = zn_rd64(c + 8 * (p * k + i)); if (v < prev) return ZN_E_CORRUPT; prev = v; }
      if (prev > own - sum) return ZN_E_CORRUPT;
      sum += (size_t)prev;
    }
    if (sum != own) return ZN_E_CORRUPT;
  }
  size_t pay_at = 0;
  for (size_t p = 0; p < P; p++) {
    uint64_t run = 0;                              // bytes the earlier ranges put into plane p
    for (size_t g = 0; g < G; g++) {
      const size_t k = ks[g];
      if (!k) continue;
      const size_t tot = (size_t)zn_rd64(part[g] + P * k + 8 * (p * k + k - 1));
      dst_at[g * P + p] = pay_at; rebase[g * P + p] = run;
      pay_at += tot; run += tot;
    }
  }
  if (pay_at != pay_total) return ZN_E_CORRUPT;   // a part whose cumSizes disagree with its length
  {
    ZnWorkers cp;
    for (size_t g = 0; g < G; g++) {
      const size_t k = ks[g];
      if (!k) continue;
      cp.start([&, g, k]() {
        const uint8_t* b = part[g];
        size_t base = 0;                             // where plane p starts in this range's payload
        for (size_t p = 0; p < P; p++) {
          memcpy(types + p * K + c0[g], b + p * k, k);
          const uint8_t* c = b + P * k + 8 * p * k;
          const uint64_t run = rebase[g * P + p];
          for (size_t i = 0; i < k; i++) { const uint64_t v = zn_rd64(c + 8 * i) + run; memcpy(cums + 8 * (p * K + c0[g] + i), &v, 8); }
          const size_t tot = (size_t)zn_rd64(c + 8 * (k - 1));
          memcpy(pay + dst_at[g * P + p], b + 9 * P * k + base, tot);
          base += tot;
        }
      });
    }
    cp.join();
    if (cp.failed) return ZN_E_ALLOC;
  }
  *dst_len = total;
  if (hdr_len >= 32) { const uint64_t t64 = total; memcpy(o + 24, &t64, 8); }   // zipnn_core.c:121
  return ZN_OK;
}

// Compress with the chunk ranges on several devices; `part_src(g)` says where range g's bytes are: a host pointer (h != null:
// staged through HBM by zn_compress) or a device pointer on devices[g].  The ranges' bodies come back to host staging and are
// placed in the frame by one thread per range.
struct ZnPartSrc { const void* h; const void* d; };
template <class SRC>
int zn_compress_ranges(const void* hdr, size_t hdr_len, size_t n, int num_buf, int bits_mode, int bytes_mode, size_t chunk, float threshold,
                       const int* devices, int ndev, void* dst, size_t dst_cap, size_t* dst_len, SRC&& part_src) {
  const size_t P = (size_t)num_buf, K = (n + chunk - 1) / chunk;
  std::lock_guard<std::mutex> mk(g_multi_mu);
  std::vector<uint8_t*> part((size_t)ndev, nullptr);
  std::vector<size_t> plen((size_t)ndev, 0);
  for (int g = 0; g < ndev; g++) {                 // (staging reserved here: the worker threads only fill it)
    const ZnRange r = zn_range_of(K, g, ndev);
    if (r.hi <= r.lo) continue;
    const size_t off = r.lo * chunk, len = (r.hi * chunk < n ? r.hi * chunk : n) - off;
    if (!(part[(size_t)g] = zn_stage(g, zn_compress_bound(len, num_buf, chunk, 0) + 16))) return ZN_E_ALLOC;
  }
  std::vector<int> rcs((size_t)ndev, ZN_OK);
  {
    ZnWorkers wk;
    for (int g = 0; g < ndev; g++) {
      const ZnRange r = zn_range_of(K, g, ndev);
      if (r.hi <= r.lo) continue;
      wk.start([&, g, r]() {
        try {
          const size_t off = r.lo * chunk, len = (r.hi * chunk < n ? r.hi * chunk : n) - off;
          const size_t cap = zn_compress_bound(len, num_buf, chunk, 0) + 16;
          const ZnPartSrc ps = part_src(g, off);
          if (ps.h || !len) {
            rcs[(size_t)g] = zn_compress(nullptr, 0, ps.h, len, num_buf, bits_mode, bytes_mode, chunk, threshold, devices[g], part[(size_t)g], cap, &plen[(size_t)g]);
            return;
          }
          // the range is in HBM already: code it there, bring only the body back
          DeviceScope scope(devices[g]);
          if (!scope.ok) { rcs[(size_t)g] = ZN_E_HIP; return; }
          int dev = 0;
          if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); rcs[(size_t)g] = ZN_E_HIP; return; }
          std::lock_guard<std::mutex> hk(g_host_mu[dev]);
          void* d_body = nullptr;
          { std::lock_guard<std::mutex> lk(g_dev_mu[dev]); const int rc0 = ws_reserve(g_ws[dev], WS_HOST_OUT, cap); if (rc0) { rcs[(size_t)g] = rc0; return; } d_body = g_ws[dev].buf[WS_HOST_OUT]; }
          size_t blen = 0;
          int rc = zn_compress_dev(ps.d, len, num_buf, bits_mode, bytes_mode, chunk, threshold, d_body, cap, &blen, nullptr);
          if (!rc && blen && zn_host_pipe_copy(g_ws[dev].pipe, d_body, part[(size_t)g], blen, false) != hipSuccess) { (void)hipGetLastError(); rc = ZN_E_HIP; }
          plen[(size_t)g] = blen; rcs[(size_t)g] = rc;
        } catch (...) { rcs[(size_t)g] = ZN_E_ALLOC; }
      });
    }
    wk.join();
    if (wk.failed) return ZN_E_ALLOC;
  }
  for (int g = 0; g < ndev; g++) if (rcs[(size_t)g]) return rcs[(size_t)g];
  std::vector<const uint8_t*> cpart((size_t)ndev, nullptr); std::vector<size_t> ks((size_t)ndev, 0);
  for (int g = 0; g < ndev; g++) { const ZnRange r = zn_range_of(K, g, ndev); cpart[(size_t)g] = part[(size_t)g]; ks[(size_t)g] = r.hi - r.lo; }
  return zn_assemble_ranges(hdr, hdr_len, cpart, plen, ks, P, dst, dst_cap, dst_len);
}
}  // namespace
}  // extern "C++"

// ---- the host-buffer entry points on ONE device, pipelined over slices of the chunks -------------------------------------------
// zn_compress / zn_decompress hand over pageable host buffers: one shot = upload everything, code, download everything, each PCIe
// direction idle while the other works (1 GiB bf16: ≈ 30 GB/s each way).  Here the chunks are cut into S contiguous slices and
// three host threads run upload(i + 1) ‖ code(i) ‖ download(i - 1): two pinned pipes, the kernels on a stream of their own.
// Decompress: slice i's sub-body is put together in HBM as in zn_decode_range.  Compress: a slice's plane-0 payload goes to its
// place in the frame as soon as the slice is coded (its offset needs only the slices before it); the later planes follow when
// the last slice is done (plane p starts behind ALL of plane p - 1).  Same bytes as the one-shot path, slice by slice.
namespace {
std::atomic<int> g_host_slices{0};               // zn_set_host_slices: 0 automatic, 1 never, 2..64 that many
struct ZnGate {                                   // "stage X has finished n items" between two pipeline threads
  std::mutex m; std::condition_variable cv; size_t done = 0; std::atomic<bool> fail{false};     // (fail is also read without the mutex)
  void publish(size_t n) { { std::lock_guard<std::mutex> lk(m); done = n; } cv.notify_all(); }
  void abort() { { std::lock_guard<std::mutex> lk(m); fail = true; } cv.notify_all(); }
  bool wait_for(size_t n) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return done >= n || fail; }); return !fail; }
};
// Declared BEHIND the ZnWorkers of a pipelined call: when the coordinating thread leaves the scope by an exception (std::bad_alloc from a
// vector), this aborts the gates before ~ZnWorkers joins the threads — which would otherwise wait on a gate for ever.
struct ZnGateGuard {
  ZnGate* a; ZnGate* b; bool armed = true;
  ZnGateGuard(ZnGate* a_, ZnGate* b_) : a(a_), b(b_) {}
  void disarm() { armed = false; }
  ~ZnGateGuard() { if (armed) { a->abort(); b->abort(); } }
};
int zn_pipeline_streams(Workspace& w) {
  if (!w.cstream) ZN_HIP(hipStreamCreateWithFlags(&w.cstream, hipStreamNonBlocking));
  return ZN_OK;
}
}  // namespace

// Measured (1 GiB bf16, MI355X box, profiles/r03_host_buffer_path.txt): the transfers are bound by the host-side staging copies
// (pageable <-> pinned: 48-54 GB/s per direction with 8 threads, more threads do not help), and the two directions disturb each
// other when they run at once: compress 36.9 -> 32.1 ms pipelined (4 slices), decompress 35.2 -> 35.5-38.7 ms.  So the automatic
// choice pipelines compress only; zn_set_host_slices forces either.
static int zn_host_slices(size_t n, size_t chunk, bool decompress, bool direct) {
  const size_t K = chunk ? (n + chunk - 1) / chunk : 0;
  const int forced = g_host_slices.load(std::memory_order_relaxed);
  size_t S;
  if (forced == 1) return 0;
  if (forced >= 2) S = (size_t)forced;
  // (round 6: with the caller's buffers pinned ahead of the transfers — ZnHostMap — the two directions no longer share the host's copy threads, and the
  //  decompress pipeline pays as well: 1 GiB 33 -> 23 ms, profiles/r06_host_path.txt; without the direct path it stays what rounds 3-5 measured: compress only)
  else if (decompress && !direct) return 0;
  else { if (n < ((size_t)192 << 20)) return 0; S = n / ((size_t)(direct ? 128 : 256) << 20); if (S < 4) S = 4; if (S > 8) S = 8; }   // four to eight slices
  if (S > K) S = K;
  return S >= 2 ? (int)S : 0;
}
void* zn_host_alloc(size_t n) { try { return zn_arena().alloc(n); } catch (...) { return nullptr; } }
int zn_host_free(void* p) { if (!p) return ZN_OK; try { return zn_arena().release(p) ? ZN_OK : ZN_E_ARG; } catch (...) { return ZN_E_ALLOC; } }
int zn_set_host_direct(int mode) {
  if (mode < 0 || mode > 7) return ZN_E_ARG;
  zn_host_pipe_detail::direct_mode_ref().store(mode, std::memory_order_relaxed);
  return ZN_OK;
}
int zn_set_host_slices(int slices) {
  if (slices < 0 || slices > 64) return ZN_E_ARG;
  g_host_slices.store(slices, std::memory_order_relaxed);
  return ZN_OK;
}

static int zn_decompress_host_pipelined(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk,
                                        size_t orig_size, int dev, int S, bool direct, void* dst) {
  ZnBodyView v;
  int rc = zn_body_view(body, body_len, num_buf, chunk, orig_size, &v);
  if (rc) return rc;
  const size_t P = v.P, K = v.K;
  std::lock_guard<std::mutex> hk(g_host_mu[dev]);
  std::vector<ZnRange> rg((size_t)S); std::vector<size_t> need((size_t)S), in_off((size_t)S);
  size_t in_total = 0;
  for (int i = 0; i < S; i++) { rg[(size_t)i] = zn_range_of(K, i, S); need[(size_t)i] = zn_sub_need(v, rg[(size_t)i]); in_off[(size_t)i] = in_total; in_total += (need[(size_t)i] + 16 + 255) & ~(size_t)255; }
  uint8_t* d_in = nullptr; uint8_t* d_out = nullptr; hipStream_t cs = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
    Workspace& w = g_ws[dev];
    if ((rc = ws_reserve(w, WS_HOST_IN, in_total + 16))) return rc;
    if ((rc = ws_reserve(w, WS_HOST_OUT, orig_size))) return rc;
    if ((rc = zn_pipeline_streams(w))) return rc;
    d_in = (uint8_t*)w.buf[WS_HOST_IN]; d_out = (uint8_t*)w.buf[WS_HOST_OUT]; cs = w.cstream;
  }
  ZnGate up, dec;
  int rc_up = ZN_OK, rc_down = ZN_OK, rc_dec = ZN_OK;
  // the caller's two buffers made DMA-able ahead of the transfers (zn_host_pipe.hpp: ZnHostMap): the payload for reading, the result — hinted to
  // huge pages, touched, pinned, piece by piece on helper threads — for writing.  Declared ahead of the workers: unpinned after they are joined.
  ZnHostMap src_map, dst_map;
  // (user memory is only pinned when the caller asked for it — zn_set_host_direct — and never a block of the library's own arena: that is pinned already)
  const bool pin_user = direct;
  if (pin_user && !zn_arena().covers(v.pay, v.pay_len)) src_map.start(const_cast<uint8_t*>(v.pay), v.pay_len, false, ~(size_t)0, dev); else src_map.gave_up = true;
  if (pin_user && !zn_arena().covers(dst, orig_size)) dst_map.start(dst, orig_size, true, ~(size_t)0, dev); else dst_map.gave_up = true;
  try {
    ZnWorkers wk;
    ZnGateGuard guard(&up, &dec);
    wk.start([&]() {                                // upload: re-based size tables + the P payload slices of every slice
      try {
        if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); rc_up = ZN_E_HIP; up.abort(); return; }
        ZnHostPipe& pipe = g_ws[dev].pipe;
        for (int i = 0; i < S; i++) {
          const ZnRange r = rg[(size_t)i]; const size_t k = r.hi - r.lo;
          std::vector<uint8_t> meta(9 * P * k);
          std::vector<uint64_t> s0(P);
          for (size_t p = 0; p < P; p++) {
            s0[p] = zn_cum_before(v, p, r.lo);
            memcpy(meta.data() + p * k, v.types + p * K + r.lo, k);
            for (size_t j = 0; j < k; j++) { const uint64_t c = zn_rd64(v.cums + 8 * (p * K + r.lo + j)) - s0[p]; memcpy(meta.data() + P * k + 8 * (p * k + j), &c, 8); }
          }
          uint8_t* d = d_in + in_off[(size_t)i];
          bool ok = zn_host_pipe_copy(pipe, d, meta.data(), meta.size(), true) == hipSuccess;      // (small: never pinned)
          size_t at = meta.size();
          for (size_t p = 0; ok && p < P; p++) {
            const size_t m = (size_t)(zn_cum_before(v, p, r.hi) - s0[p]);
            if (m) ok = zn_host_copy_mapped(src_map, pipe, d + at, const_cast<uint8_t*>(v.pay + v.base[p] + s0[p]), m, true) == hipSuccess;
            at += m;
          }
          if (!ok || dec.fail) { if (!ok) { (void)hipGetLastError(); rc_up = ZN_E_HIP; } up.abort(); return; }
          up.publish((size_t)i + 1);
        }
      } catch (...) { rc_up = ZN_E_ALLOC; up.abort(); }
    });
    wk.start([&]() {                                // download: slice i's bytes as soon as they are decoded
      try {
        if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); rc_down = ZN_E_HIP; return; }
        ZnHostPipe& pipe = g_ws[dev].pipe2;
        for (int i = 0; i < S; i++) {
          if (!dec.wait_for((size_t)i + 1)) return;
          const ZnRange r = rg[(size_t)i];
          const size_t off = r.lo * chunk, len = (r.hi * chunk < orig_size ? r.hi * chunk : orig_size) - off;
          if (len && zn_host_copy_mapped(dst_map, pipe, d_out + off, (uint8_t*)dst + off, len, false) != hipSuccess) { (void)hipGetLastError(); rc_down = ZN_E_HIP; return; }
        }
      } catch (...) { rc_down = ZN_E_ALLOC; }
    });
    if (wk.failed) { up.abort(); dec.abort(); wk.join(); return ZN_E_ALLOC; }
    for (int i = 0; i < S; i++) {                   // decode, on this thread
      if (!up.wait_for((size_t)i + 1)) { rc_dec = rc_up ? rc_up : ZN_E_HIP; break; }
      const ZnRange r = rg[(size_t)i];
      const size_t off = r.lo * chunk, len = (r.hi * chunk < orig_size ? r.hi * chunk : orig_size) - off;
      rc_dec = zn_decompress_dev(d_in + in_off[(size_t)i], need[(size_t)i], num_buf, bits_mode, bytes_mode, chunk, len, d_out + off, cs, 1);
      if (rc_dec) break;
      dec.publish((size_t)i + 1);
    }
    if (rc_dec) { dec.abort(); up.abort(); }
    guard.disarm();
    wk.join();
  } catch (...) { return ZN_E_ALLOC; }
  return rc_dec ? rc_dec : rc_up ? rc_up : rc_down;
}

static int zn_compress_host_pipelined(const void* hdr, size_t hdr_len, const void* src, size_t n, int num_buf, int bits_mode, int bytes_mode,
                                      size_t chunk, float threshold, int dev, int S, bool direct, void* dst, size_t dst_cap, size_t* dst_len) {
  const size_t P = (size_t)num_buf, K = (n + chunk - 1) / chunk;
  if (hdr_len + 9 * P * K > dst_cap) return ZN_E_CAP;
  std::lock_guard<std::mutex> hk(g_host_mu[dev]);
  std::vector<ZnRange> rg((size_t)S); std::vector<size_t> boff((size_t)S), bcap((size_t)S), blen((size_t)S, 0);
  size_t b_total = 0;
  for (int i = 0; i < S; i++) {
    rg[(size_t)i] = zn_range_of(K, i, S);
    const size_t off = rg[(size_t)i].lo * chunk, len = (rg[(size_t)i].hi * chunk < n ? rg[(size_t)i].hi * chunk : n) - off;
    bcap[(size_t)i] = zn_compress_bound(len, num_buf, chunk, 0) + 16; boff[(size_t)i] = b_total; b_total += (bcap[(size_t)i] + 255) & ~(size_t)255;
  }
  uint8_t* d_src = nullptr; uint8_t* d_body = nullptr; hipStream_t cs = nullptr;
  int rc;
  {
    std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
    Workspace& w = g_ws[dev];
    if ((rc = ws_reserve(w, WS_HOST_IN, n))) return rc;
    if ((rc = ws_reserve(w, WS_HOST_OUT, b_total + 16))) return rc;
    if ((rc = zn_pipeline_streams(w))) return rc;
    d_src = (uint8_t*)w.buf[WS_HOST_IN]; d_body = (uint8_t*)w.buf[WS_HOST_OUT]; cs = w.cstream;
  }
  uint8_t* o = (uint8_t*)dst;
  uint8_t* types = o + hdr_len; uint8_t* cums = types + P * K; uint8_t* pay = cums + 8 * P * K;
  const size_t pay_cap = dst_cap - hdr_len - 9 * P * K;
  struct Job { const uint8_t* d; uint8_t* h; size_t len; };
  std::vector<Job> jobs; jobs.reserve((size_t)S + P);                 // (appended by this thread only, read by the downloader up to `queued.done`)
  ZnGate up, queued; bool all_queued = false;
  int rc_up = ZN_OK, rc_down = ZN_OK, rc_enc = ZN_OK;
  std::vector<std::vector<uint8_t>> metas((size_t)S);
  ZnHostMap src_map, dst_map;                       // (as in the decompress pipeline; the result is prepared as far as a weights-like tensor will need it, further on demand)
  const bool pin_user = direct;
  if (pin_user && !zn_arena().covers(src, n)) src_map.start(const_cast<void*>(src), n, false, ~(size_t)0, dev); else src_map.gave_up = true;
  if (pin_user && !zn_arena().covers(dst, dst_cap)) dst_map.start(dst, dst_cap, true, hdr_len + 9 * P * K + n / 2 + n / 4, dev); else dst_map.gave_up = true;
  try {
    ZnWorkers wk;
    ZnGateGuard guard(&up, &queued);
    wk.start([&]() {                                // upload the tensor slice by slice
      try {
        if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); rc_up = ZN_E_HIP; up.abort(); return; }
        for (int i = 0; i < S; i++) {
          const size_t off = rg[(size_t)i].lo * chunk, len = (rg[(size_t)i].hi * chunk < n ? rg[(size_t)i].hi * chunk : n) - off;
          if (len && zn_host_copy_mapped(src_map, g_ws[dev].pipe, d_src + off, const_cast<uint8_t*>((const uint8_t*)src + off), len, true) != hipSuccess) { (void)hipGetLastError(); rc_up = ZN_E_HIP; up.abort(); return; }
          if (queued.fail) { up.abort(); return; }
          up.publish((size_t)i + 1);
        }
      } catch (...) { rc_up = ZN_E_ALLOC; up.abort(); }
    });
    wk.start([&]() {                                // download the payload pieces in the order they are queued
      try {
        if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); rc_down = ZN_E_HIP; return; }
        size_t next = 0;
        for (;;) {
          size_t have; bool fin;
          { std::unique_lock<std::mutex> lk(queued.m); queued.cv.wait(lk, [&] { return queued.done > next || queued.fail || all_queued; }); have = queued.done; fin = all_queued; if (queued.fail) return; }
          for (; next < have; next++) {
            const Job j = jobs[next];
            if (j.len && zn_host_copy_mapped(dst_map, g_ws[dev].pipe2, const_cast<uint8_t*>(j.d), j.h, j.len, false) != hipSuccess) { (void)hipGetLastError(); rc_down = ZN_E_HIP; return; }
          }
          if (fin && next >= have) { std::lock_guard<std::mutex> lk(queued.m); if (queued.done == next) return; }
        }
      } catch (...) { rc_down = ZN_E_ALLOC; }
    });
    if (wk.failed) { up.abort(); queued.abort(); wk.join(); return ZN_E_ALLOC; }
    std::vector<uint64_t> tot((size_t)S * P, 0);
    size_t pay0_at = 0;
    for (int i = 0; i < S && !rc_enc; i++) {        // code slice i; its plane 0 can leave at once
      if (!up.wait_for((size_t)i + 1)) { rc_enc = rc_up ? rc_up : ZN_E_HIP; break; }
      const ZnRange r = rg[(size_t)i]; const size_t k = r.hi - r.lo;
      const size_t off = r.lo * chunk, len = (r.hi * chunk < n ? r.hi * chunk : n) - off;
      uint8_t* b = d_body + boff[(size_t)i];
      rc_enc = zn_compress_dev(d_src + off, len, num_buf, bits_mode, bytes_mode, chunk, threshold, b, bcap[(size_t)i], &blen[(size_t)i], cs);
      if (rc_enc) break;
      metas[(size_t)i].resize(9 * P * k);
      if (hipMemcpyAsync(metas[(size_t)i].data(), b, 9 * P * k, hipMemcpyDeviceToHost, cs) != hipSuccess || hipStreamSynchronize(cs) != hipSuccess) { (void)hipGetLastError(); rc_enc = ZN_E_HIP; break; }
      for (size_t p = 0; p < P; p++) tot[(size_t)i * P + p] = zn_rd64(metas[(size_t)i].data() + P * k + 8 * (p * k + k - 1));
      const size_t t0 = (size_t)tot[(size_t)i * P];
      if (pay0_at + t0 > pay_cap) { rc_enc = ZN_E_CAP; break; }
      jobs.push_back(Job{b + 9 * P * k, pay + pay0_at, t0});
      pay0_at += t0;
      queued.publish(jobs.size());
    }
    size_t total = 0;
    if (!rc_enc) {                                  // every slice is coded: the later planes have their places now
      std::vector<size_t> base(P, 0); size_t acc = 0;
      for (size_t p = 0; p < P; p++) { base[p] = acc; for (int i = 0; i < S; i++) acc += (size_t)tot[(size_t)i * P + p]; }
      if (acc > pay_cap) rc_enc = ZN_E_CAP;
      else {
        // planes 1.. are contiguous in the frame across the slices: their pieces are gathered on the device (into the input staging,
        // which nothing reads any more) and leave as ONE transfer per plane instead of S small ones
        size_t g_at = 0;
        for (size_t p = 1; p < P && !rc_enc; p++) {
          const size_t g0 = g_at;
          for (int i = 0; i < S; i++) {
            const size_t k = rg[(size_t)i].hi - rg[(size_t)i].lo;
            size_t before = 0; for (size_t q = 0; q < p; q++) before += (size_t)tot[(size_t)i * P + q];
            const size_t m = (size_t)tot[(size_t)i * P + p];
            if (m && hipMemcpyAsync(d_src + g_at, d_body + boff[(size_t)i] + 9 * P * k + before, m, hipMemcpyDeviceToDevice, cs) != hipSuccess) { (void)hipGetLastError(); rc_enc = ZN_E_HIP; break; }
            g_at += m;
          }
          if (!rc_enc) jobs.push_back(Job{d_src + g0, pay + base[p], g_at - g0});
        }
        if (!rc_enc && hipStreamSynchronize(cs) != hipSuccess) { (void)hipGetLastError(); rc_enc = ZN_E_HIP; }
        total = hdr_len + 9 * P * K + acc;
      }
    }
    if (rc_enc) { queued.abort(); up.abort(); }
    else { { std::lock_guard<std::mutex> lk(queued.m); queued.done = jobs.size(); all_queued = true; } queued.cv.notify_all(); }
    if (!rc_enc) {                                  // the size tables, re-based, while the last pieces are on their way
      if (hdr_len) memcpy(o, hdr, hdr_len);
      std::vector<uint64_t> run(P, 0);
      for (int i = 0; i < S; i++) {
        const ZnRange r = rg[(size_t)i]; const size_t k = r.hi - r.lo;
        const uint8_t* m = metas[(size_t)i].data();
        for (size_t p = 0; p < P; p++) {
          memcpy(types + p * K + r.lo, m + p * k, k);
          for (size_t j = 0; j < k; j++) { const uint64_t c = zn_rd64(m + P * k + 8 * (p * k + j)) + run[p]; memcpy(cums + 8 * (p * K + r.lo + j), &c, 8); }
          run[p] += tot[(size_t)i * P + p];
        }
      }
      if (hdr_len >= 32) { const uint64_t t64 = total; memcpy(o + 24, &t64, 8); }   // zipnn_core.c:121
    }
    guard.disarm();
    wk.join();
    if (!rc_enc && !rc_down) *dst_len = total;
  } catch (...) { return ZN_E_ALLOC; }
  return rc_enc ? rc_enc : rc_up ? rc_up : rc_down;
}

int zn_compress_multi(const void* hdr, size_t hdr_len, const void* src, size_t n, int num_buf, int bits_mode, int bytes_mode,
                      size_t chunk, float threshold, const int* devices, int ndev, void* dst, size_t dst_cap, size_t* dst_len) {
  if (!dst_len || (hdr_len && !hdr) || (n && !src) || !dst || !devices || ndev <= 0 || ndev > 64 || !chunk) return ZN_E_ARG;
  if (num_buf != 1 && num_buf != 2 && num_buf != 4) return ZN_E_ARG;
  if (ndev == 1) return zn_compress(hdr, hdr_len, src, n, num_buf, bits_mode, bytes_mode, chunk, threshold, devices[0], dst, dst_cap, dst_len);
  try {
    return zn_compress_ranges(hdr, hdr_len, n, num_buf, bits_mode, bytes_mode, chunk, threshold, devices, ndev, dst, dst_cap, dst_len,
                              [&](int, size_t off) { return ZnPartSrc{(const uint8_t*)src + off, nullptr}; });
  } catch (...) { return ZN_E_ALLOC; }
}

int zn_compress_multi_dev(const void* hdr, size_t hdr_len, const void* const* d_src, size_t n, int num_buf, int bits_mode, int bytes_mode,
                          size_t chunk, float threshold, const int* devices, int ndev, void* dst, size_t dst_cap, size_t* dst_len) {
  if (!dst_len || (hdr_len && !hdr) || (n && !d_src) || !dst || !devices || ndev <= 0 || ndev > 64 || !chunk) return ZN_E_ARG;
  if (num_buf != 1 && num_buf != 2 && num_buf != 4) return ZN_E_ARG;
  if (zn_device_count() <= 0) return ZN_E_NODEV;
  const size_t K = (n + chunk - 1) / chunk;
  for (int g = 0; g < ndev; g++) { const ZnRange r = zn_range_of(K, g, ndev); if (r.hi > r.lo && !d_src[g]) return ZN_E_ARG; }
  try {
    return zn_compress_ranges(hdr, hdr_len, n, num_buf, bits_mode, bytes_mode, chunk, threshold, devices, ndev, dst, dst_cap, dst_len,
                              [&](int g, size_t) { return ZnPartSrc{nullptr, d_src[g]}; });
  } catch (...) { return ZN_E_ALLOC; }
}

int zn_multi_range(size_t n, size_t chunk, int ndev, int i, size_t* off, size_t* len) {
  if (!chunk || ndev <= 0 || i < 0 || i >= ndev || !off || !len) return ZN_E_ARG;
  const size_t K = (n + chunk - 1) / chunk;
  const ZnRange r = zn_range_of(K, i, ndev);
  *off = r.lo * chunk; *len = r.hi > r.lo ? (r.hi * chunk < n ? r.hi * chunk : n) - r.lo * chunk : 0;
  return ZN_OK;
}

int zn_merge_range_bodies(const void* const* bodies, const size_t* body_lens, const size_t* num_chunks, int nparts, int num_buf,
                          void* dst, size_t dst_cap, size_t* dst_len) {
  if (nparts < 0 || (nparts && (!bodies || !body_lens || !num_chunks)) || !dst || !dst_len) return ZN_E_ARG;
  if (num_buf != 1 && num_buf != 2 && num_buf != 4) return ZN_E_ARG;
  try {
    std::vector<const uint8_t*> part((size_t)nparts); std::vector<size_t> plen((size_t)nparts), ks((size_t)nparts);
    for (int i = 0; i < nparts; i++) { part[(size_t)i] = (const uint8_t*)bodies[i]; plen[(size_t)i] = body_lens[i]; ks[(size_t)i] = num_chunks[i]; }
    return zn_assemble_ranges(nullptr, 0, part, plen, ks, (size_t)num_buf, dst, dst_cap, dst_len);
  } catch (...) { return ZN_E_ALLOC; }
}

int zn_decompress_range_dev(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk, size_t orig_size,
                            size_t chunk_lo, size_t chunk_hi, int device, void* d_dst) {
  if ((body_len && !body) || !chunk) return ZN_E_ARG;
  if (num_buf != 1 && num_buf != 2 && num_buf != 4) return ZN_E_ARG;
  const size_t K = (orig_size + chunk - 1) / chunk;
  if (chunk_lo > chunk_hi || chunk_hi > K) return ZN_E_ARG;
  if (chunk_lo == chunk_hi) return ZN_OK;
  if (!d_dst) return ZN_E_ARG;
  if (zn_device_count() <= 0) return ZN_E_NODEV;
  try {
    ZnBodyView v;
    const int rc = zn_body_view(body, body_len, num_buf, chunk, orig_size, &v);
    if (rc) return rc;
    return zn_decode_range(v, ZnRange{chunk_lo, chunk_hi}, num_buf, bits_mode, bytes_mode, chunk, orig_size, device, nullptr, d_dst);
  } catch (...) { return ZN_E_ALLOC; }
}

static int zn_decompress_ranges(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk, size_t orig_size,
                                const int* devices, int ndev, void* h_dst, void* const* d_dst) {
  ZnBodyView v;
  int rc = zn_body_view(body, body_len, num_buf, chunk, orig_size, &v);
  if (rc) return rc;
  std::vector<int> rcs((size_t)ndev, ZN_OK);
  {
    ZnWorkers wk;
    for (int g = 0; g < ndev; g++) {
      const ZnRange r = zn_range_of(v.K, g, ndev);
      if (r.hi <= r.lo) continue;
      wk.start([&, g, r]() {
        try { rcs[(size_t)g] = zn_decode_range(v, r, num_buf, bits_mode, bytes_mode, chunk, orig_size, devices[g], h_dst, d_dst ? d_dst[g] : nullptr); }
        catch (...) { rcs[(size_t)g] = ZN_E_ALLOC; }
      });
    }
    wk.join();
    if (wk.failed) return ZN_E_ALLOC;
  }
  for (int g = 0; g < ndev; g++) if (rcs[(size_t)g]) return rcs[(size_t)g];
  return ZN_OK;
}

int zn_decompress_multi(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk, size_t orig_size,
                        const int* devices, int ndev, void* dst) {
  if ((body_len && !body) || (orig_size && !dst) || !devices || ndev <= 0 || ndev > 64 || !chunk) return ZN_E_ARG;
  if (num_buf != 1 && num_buf != 2 && num_buf != 4) return ZN_E_ARG;
  if (ndev == 1 || orig_size == 0) return zn_decompress(body, body_len, num_buf, bits_mode, bytes_mode, chunk, orig_size, devices[0], dst);
  if (zn_device_count() <= 0) return ZN_E_NODEV;
  try { return zn_decompress_ranges(body, body_len, num_buf, bits_mode, bytes_mode, chunk, orig_size, devices, ndev, dst, nullptr); }
  catch (...) { return ZN_E_ALLOC; }
}

int zn_decompress_multi_dev(const void* body, size_t body_len, int num_buf, int bits_mode, int bytes_mode, size_t chunk, size_t orig_size,
                            const int* devices, int ndev, void* const* d_dst) {
  if ((body_len && !body) || !devices || ndev <= 0 || ndev > 64 || !chunk || (orig_size && !d_dst)) return ZN_E_ARG;
  if (num_buf != 1 && num_buf != 2 && num_buf != 4) return ZN_E_ARG;
  if (orig_size == 0) return ZN_OK;
  if (zn_device_count() <= 0) return ZN_E_NODEV;
  const size_t K = (orig_size + chunk - 1) / chunk;
  for (int g = 0; g < ndev; g++) { const ZnRange r = zn_range_of(K, g, ndev); if (r.hi > r.lo && !d_dst[g]) return ZN_E_ARG; }
  try { return zn_decompress_ranges(body, body_len, num_buf, bits_mode, bytes_mode, chunk, orig_size, devices, ndev, nullptr, d_dst); }
  catch (...) { return ZN_E_ALLOC; }
}

// The device-side status of the last decode call on the current device that was made with check = 0: waits for `stream`, reads the
// status word that call left in the workspace.  A loader launches its batched decode without the read-back, builds its tensor views
// while the kernels run, and asks here once at the end.
int zn_decode_status(void* stream_) {
  try {
    int dev = 0;
    ZN_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return ZN_E_ARG;
    std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
    Workspace& w = g_ws[dev];
    if (!w.buf[WS_WORDS] || w.cap[WS_WORDS] < ZN_WORDS_BYTES || w.status_gen == 0) {
      // no decode has run on this device — or its status words are gone (zn_release_workspace).  A thread that still holds the token of an unverified
      // check = 0 decode here gets "cannot vouch for it", not "ok" (ADVICE r5): the verdict went with the buffer
      if (t_status_dev == dev && t_status_gen != 0) { t_status_gen = 0; t_status_dev = -1; return ZN_E_CORRUPT; }
      return ZN_OK;
    }
    int rc;
    if ((rc = ws_host_words(w))) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    // the calling thread's last check = 0 decode on this device — or, for a thread that made none (a loader that launches on one thread and asks on another),
    // the device's last decode.  Sixteen decodes later the slot belongs to another call: its answer is gone, and saying "ok" would be a guess
    uint32_t slot = w.last_slot;
    if (t_status_dev == dev) {
      if (w.slot_gen[t_status_slot] != t_status_gen) return ZN_E_CORRUPT;
      slot = t_status_slot;
    }
    if (w.multi && w.busy) ZN_HIP(hipStreamWaitEvent(stream, w.busy, 0));      // (several streams on this device: behind the last call's kernels, whichever stream they ran on — ADVICE r4)
    ZN_HIP(hipMemcpyAsync(w.h_status, (uint32_t*)w.buf[WS_WORDS] + 16 + 4u * slot, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    ZN_HIP(hipStreamSynchronize(stream));
    const uint32_t st = *w.h_status;
    if (st & ZN_DEV_BAD_TYPE) return ZN_E_TYPE;
    if (st & ZN_DEV_CORRUPT) return ZN_E_CORRUPT;
    if (st & ZN_DEV_SYNC_TIMEOUT) return ZN_E_TIMEOUT;
    return ZN_OK;
  } catch (...) { return ZN_E_ALLOC; }
}

long long zn_last_fused_chunks(void) {
  try {
    int dev = 0;
    ZN_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return ZN_E_ARG;
    std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
    Workspace& w = g_ws[dev];
    if (!w.last_K || !w.buf[WS_META_B]) return 0;
    std::string flags(w.last_K, '\0');
    ZN_HIP(hipDeviceSynchronize());
    ZN_HIP(hipMemcpy(&flags[0], w.buf[WS_META_B], w.last_K, hipMemcpyDeviceToHost));
    long long n = 0;
    for (char f : flags) n += (f != 0);
    return n;
  } catch (...) { return ZN_E_ALLOC; }
}

long long zn_last_tail_planes(void) {
  try {
    int dev = 0;
    ZN_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return ZN_E_ARG;
    std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
    Workspace& w = g_ws[dev];
    if (!w.last_tails || !w.buf[WS_META_A]) return 0;
    std::string flags(4u * w.last_tails, '\0');     // (four flag bytes per plane: one per huff0 stream)
    ZN_HIP(hipDeviceSynchronize());
    ZN_HIP(hipMemcpy(&flags[0], w.buf[WS_META_A], 4u * w.last_tails, hipMemcpyDeviceToHost));
    long long n = 0;
    for (size_t i = 0; i < w.last_tails; i++) n += (flags[4 * i] && flags[4 * i + 1] && flags[4 * i + 2] && flags[4 * i + 3]);
    return n;
  } catch (...) { return ZN_E_ALLOC; }
}

static int zn_copy_host(void* d, void* h, size_t n, bool to_device) {
  if (n == 0) return ZN_OK;
  if (!d || !h) return ZN_E_ARG;
  if (zn_device_count() <= 0) return ZN_E_NODEV;
  int dev = 0;
  ZN_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return ZN_E_ARG;
  std::lock_guard<std::mutex> hk(g_host_mu[dev]);        // one user of the device's bounce buffers at a time
  try {
    if (zn_host_pipe_copy(g_ws[dev].pipe, d, h, n, to_device) != hipSuccess) { t_hip_err = to_device ? "host pipe H2D" : "host pipe D2H"; (void)hipGetLastError(); return ZN_E_HIP; }
  } catch (...) { return ZN_E_ALLOC; }
  return ZN_OK;
}
int zn_copy_to_device(void* d_dst, const void* src, size_t n) { return zn_copy_host(d_dst, const_cast<void*>(src), n, true); }
int zn_copy_to_host(void* dst, const void* d_src, size_t n) { return zn_copy_host(const_cast<void*>(d_src), dst, n, false); }

#if defined(ZN_SIMT_EMULATOR)
// test hook of the emulated build (tests/test_kernels_simt.py): holds one device's workspace lock for `ms` milliseconds,
// so that a test can show that calls on ANOTHER device do not wait for it and calls on the same one do
int zn_debug_hold_device_lock(int dev, int ms) {
  if (dev < 0 || dev >= 64) return ZN_E_ARG;
  std::lock_guard<std::mutex> lk(g_dev_mu[dev]);
  std::this_thread::sleep_for(std::chrono::milliseconds(ms));
  return ZN_OK;
}
#endif

int zn_release_workspace(void) {
  zn_arena().trim();
  { std::lock_guard<std::mutex> mk(g_multi_mu); for (int i = 0; i < 64; i++) { free(g_multi_stage[i].p); g_multi_stage[i].p = nullptr; g_multi_stage[i].cap = 0; } }
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
  for (int d = 0; d < 64; d++) {
    // same order as the host-buffer entry points: the device's host lock (its bounce buffers may be in use), then the table
    std::lock_guard<std::mutex> hk(g_host_mu[d]);
    std::lock_guard<std::mutex> lk(g_dev_mu[d]);
    Workspace& w = g_ws[d];
    bool any = w.h_total != nullptr || w.busy != nullptr || w.h_segs != nullptr || w.h_totals != nullptr || w.pipe.pin[0] != nullptr || w.pipe2.pin[0] != nullptr || w.cstream != nullptr || w.dstream != nullptr;
    for (int i = 0; i < WS_COUNT; i++) any = any || w.buf[i];
    if (!any) continue;
    if (hipSetDevice(d) != hipSuccess) { (void)hipGetLastError(); continue; }
    for (int i = 0; i < WS_COUNT; i++) if (w.buf[i]) { (void)hipFree(w.buf[i]); w.buf[i] = nullptr; w.cap[i] = 0; }
    if (w.h_segs) { (void)hipHostFree(w.h_segs); w.h_segs = nullptr; w.h_segs_cap = 0; }
    if (w.h_totals) { (void)hipHostFree(w.h_totals); w.h_totals = nullptr; w.h_totals_cap = 0; }
    if (w.h_total) { (void)hipHostFree(w.h_total); w.h_total = nullptr; w.h_status = nullptr; }
    zn_host_pipe_release(w.pipe);
    zn_host_pipe_release(w.pipe2);
    if (w.cstream) { (void)hipStreamSynchronize(w.cstream); (void)hipStreamDestroy(w.cstream); w.cstream = nullptr; }
    if (w.dstream) { (void)hipStreamSynchronize(w.dstream); (void)hipStreamDestroy(w.dstream); w.dstream = nullptr; }
    if (w.dstream2) { (void)hipStreamSynchronize(w.dstream2); (void)hipStreamDestroy(w.dstream2); w.dstream2 = nullptr; }
    if (w.djoin2) { (void)hipEventDestroy(w.djoin2); w.djoin2 = nullptr; }
    if (w.dfork) { (void)hipEventDestroy(w.dfork); w.dfork = nullptr; }
    if (w.djoin) { (void)hipEventDestroy(w.djoin); w.djoin = nullptr; }
    if (w.busy) { (void)hipDeviceSynchronize(); (void)hipEventDestroy(w.busy); w.busy = nullptr; }
    w.have_stream = false; w.multi = false;
    for (uint32_t i = 0; i < ZN_STATUS_SLOTS; i++) w.slot_gen[i] = 0;      // (the status slots went with the buffers: a thread's old token matches nothing;
    w.last_slot = 0; w.last_K = 0; w.last_tails = 0;                       //  status_gen keeps counting, so that no later call can be handed an old token's generation)
    w.op_backoff = 0; w.op_penalty = 0;
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  return ZN_OK;
}

}  // extern "C"
