// wave_probe.hip — developer probe: one instance of zn_fused_wave (the per-wave body of the fused decode kernel) in a
// minimal kernel, to read its register use / spills off the compiler's remarks in seconds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -Rpass-analysis=kernel-resource-usage scripts/ubench/wave_probe.hip
#include "../../zipnn_amd/csrc/zn_decode_fused.hip"

#ifndef PROBE_P
#define PROBE_P 2
#endif
#ifndef PROBE_H
#define PROBE_H 1
#endif
#ifndef PROBE_DC
#define PROBE_DC 4
#endif
__global__ __launch_bounds__(256, ZN_F_WAVES_PER_SIMD) void k_wave_probe(ZnGeom g, const uint8_t* body, uint64_t body_len, uint8_t* dst, const ZnFusedPlane* gpl,
                                                                       const uint8_t* stream, uint32_t slen, uint32_t TL, uint32_t* okout) {
  __shared__ ZnFusedLds L;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  for (uint32_t i = tid; i < 2048; i += 256) L.lut[i] = ((const uint2*)body)[i];
  __syncthreads();
  ZnFusedPlane pl[PROBE_P]; const uint8_t* rawq[PROBE_P];
  for (int p = 0; p < PROBE_P; p++) { pl[p] = gpl[p]; rawq[p] = body + pl[p].off + wave * 16384u; }
  const bool ok = zn_fused_wave<PROBE_P, PROBE_H, PROBE_DC, false>(g, body, body + body_len, dst + wave * 65536u, nullptr, pl, rawq, L.lut, L.ring[wave], L.in[wave], lane,
                                                                  g.chunk / PROBE_P / 4u, TL, PROBE_DC, stream + wave * slen, slen, false);
  if (lane == 0) okout[blockIdx.x * 4 + wave] = ok;
}
