// pass_probe.hip — developer probe: the decode chain of zn_decode_chain.hpp in isolation (ISA inspection; hipcc -S).
#include <hip/hip_runtime.h>
#include "../../zipnn_amd/csrc/zn_decode_chain.hpp"

template <int TF, int TB, int U>
__global__ __launch_bounds__(256, 4) void k_probe(const uint2* glut, const uint32_t* gin, uint32_t* out, uint32_t TL, int reps) {
  __shared__ uint2 lut[2048];
  __shared__ uint32_t in[4][260];
  __shared__ uint32_t ring[4][1024];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t i = tid; i < 2048; i += 256) lut[i] = glut[i];
  for (uint32_t i = tid; i < 4 * 260; i += 256) ((uint32_t*)in)[i] = gin[i];
  for (uint32_t i = tid; i < 4096; i += 256) ((uint32_t*)ring)[i] = 0;
  __syncthreads();
  uint32_t total = 0;
  for (int r = 0; r < reps; r++) {
    const int32_t hi_k = 32 * (257 - (int32_t)lane * 4), stop = hi_k - 128;
    ZnRec rec; uint32_t acc = 0; int nfull = 0, nbnd = 0;
    const int32_t s = hi_k - (int32_t)((lane * 7u + r) % 10u);
    const bool took = zn_pass1<TF, TB, U, false>(lut, in[wave], 0, TL, s, stop, true, rec, acc, nfull, nbnd);
    if (!took) break;
    uint32_t N;
    const uint32_t n = (acc >> 8) & 0xFFu;
    const uint32_t o = zn_wave_excl_scan(n, lane, &N);
    zn_pass2<TF, TB>(ring[wave], o, rec, nfull, nbnd, [](auto) {});
    total += N + (acc & 0xFFu);
  }
  out[blockIdx.x * 256 + tid] = total + ring[wave][lane];
}
template __global__ void k_probe<17, 3, 8>(const uint2*, const uint32_t*, uint32_t*, uint32_t, int);
