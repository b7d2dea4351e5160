// Cross-lane / block reduction helpers for 256-bit field elements (wave64, gfx950).
#pragma once
#include "field.cuh"

namespace spk {

typedef FqP S;  // scalar field of the bench engine

__device__ __forceinline__ fe_t shfl_xor_fe(const fe_t& a, int mask) {
  fe_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = __shfl_xor(a.v[i], mask, 64);
  return r;
}
__device__ __forceinline__ fe_t wave_sum(fe_t a) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) a = fe_add<S>(a, shfl_xor_fe(a, m));
  return a;
}
// Sum NACC accumulators over a 256-thread block; result valid in thread 0. smem: NACC * 4 elements.
template <int NACC>
__device__ __forceinline__ void block_sum(fe_t (&acc)[NACC], fe_t* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) smem[k * 4 + wave] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
      fe_t s = smem[k * 4];
      for (int w = 1; w < nwaves; ++w) s = fe_add<S>(s, smem[k * 4 + w]);
      acc[k] = s;
    }
  }
}

}  // namespace spk
