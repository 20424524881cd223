/* wave.h — the CDNA4 execution vocabulary used by the codec kernels.
 *
 * One wavefront (64 lanes) encodes one (stream, frame).  Workgroups are exactly one wave
 * (__launch_bounds__(64)), so a workgroup barrier is free: the compiler drops s_barrier and keeps only
 * the LDS ordering (s_waitcnt lgkmcnt(0)).  Lane-parallel phases alternate with lane-0 serial phases
 * (range coder, bit allocation); cross-lane traffic goes through DPP/ds_bpermute shuffles, never
 * through global memory.  All working arrays live in LDS and are addressed through address_space(3)
 * pointers so every access is a ds_* instruction, not a flat_* one. */
#ifndef OPUS_AMD_WAVE_H
#define OPUS_AMD_WAVE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WV_DEV  __device__ __forceinline__
#define WV_DEVN __device__ __noinline__
#define WV_LDS  __attribute__((address_space(3)))
#define WV_TABLE __device__ const
#define WV_WIDTH 64

WV_DEV int wv_lane() { return (int)threadIdx.x; }
/* orders LDS traffic between lanes of the wave (block == wave) */
WV_DEV void wv_sync() { __syncthreads(); }

WV_DEV int32_t wv_shfl(int32_t v, int src) { return __shfl(v, src, 64); }
WV_DEV int32_t wv_bcast(int32_t v, int src) { return __builtin_amdgcn_readlane(v, src); }
WV_DEV int32_t wv_sum(int32_t v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
WV_DEV uint32_t wv_sumu(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64); return v; }
WV_DEV int64_t wv_sum64(int64_t v)
{
   for (int o = 32; o > 0; o >>= 1) {
      int lo = __shfl_xor((int)(uint32_t)v, o, 64), hi = __shfl_xor((int)(v >> 32), o, 64);
      v += (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
   }
   return v;
}
WV_DEV int32_t wv_max(int32_t v) { for (int o = 32; o > 0; o >>= 1) { int32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; } return v; }
WV_DEV int32_t wv_min(int32_t v) { for (int o = 32; o > 0; o >>= 1) { int32_t t = __shfl_xor(v, o, 64); v = t < v ? t : v; } return v; }
WV_DEV uint32_t wv_or(uint32_t v) { for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64); return v; }
WV_DEV uint64_t wv_ballot(int pred) { return __ballot(pred); }
/* inclusive prefix sum over lanes */
WV_DEV int32_t wv_scan_incl(int32_t v)
{
   int l = wv_lane();
   for (int o = 1; o < 64; o <<= 1) { int32_t t = __shfl_up(v, o, 64); if (l >= o) v += t; }
   return v;
}
/* PVQ greedy-search arg-max: maximise num/den (den > 0, exact 16x16 cross products), lowest index wins ties.
 * Every lane receives the winning (num, den, idx). */
WV_DEV void wv_argmax_ratio(int32_t &num, int32_t &den, int32_t &idx)
{
   for (int o = 32; o > 0; o >>= 1) {
      int32_t n2 = __shfl_xor(num, o, 64), d2 = __shfl_xor(den, o, 64), i2 = __shfl_xor(idx, o, 64);
      int32_t lhs = (int32_t)(int16_t)den * (int32_t)(int16_t)n2, rhs = (int32_t)(int16_t)d2 * (int32_t)(int16_t)num;
      bool take = lhs > rhs || (lhs == rhs && i2 < idx);
      if (take) { num = n2; den = d2; idx = i2; }
   }
}
#endif
