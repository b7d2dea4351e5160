// Cross-lane / block reduction helpers for 256-bit field elements (wave64, gfx950).
#pragma once
#include "field.hpp"

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

// ---- lazy (unreduced) accumulation for large reductions ---------------------------------------------------------------
// A sum of up to 2^32 canonical elements fits 9 x u32. Tree levels are then 9 add-with-carry instead of a modular add
// (8 addc + 8 subb + 8 select), and the single reduction mod p happens once per group in the second-stage kernel.
struct lazy9_t {
  uint32_t v[9];
};
__device__ __forceinline__ lazy9_t lazy_from(const fe_t& a) {
  lazy9_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = a.v[i];
  r.v[8] = 0;
  return r;
}
__device__ __forceinline__ lazy9_t lazy_add(const lazy9_t& a, const lazy9_t& b) {
  lazy9_t r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = sp_addc(a.v[i], b.v[i], c);
  return r;
}
__device__ __forceinline__ lazy9_t lazy_wave_sum(lazy9_t a) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    lazy9_t o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.v[i] = __shfl_xor(a.v[i], m, 64);
    a = lazy_add(a, o);
  }
  return a;
}
// value mod p, canonical: lo + hi * (2^256 mod p), folded twice, then conditional subtractions
__device__ __forceinline__ fe_t lazy_reduce(const lazy9_t& a) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = a.v[i];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const uint32_t hi = t[8];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint64_t x = (uint64_t)hi * S::R1(i) + t[i] + carry;
      t[i] = (uint32_t)x;
      carry = x >> 32;
    }
    t[8] = (uint32_t)carry;
  }
  uint32_t top = t[8];  // <= 1 after two folds (hi <= 2^32 first, then hi <= ~2^32/2^224... tiny)
#pragma unroll
  for (int k = 0; k < 3; ++k) fe_cond_sub_p_top<S>(t, top);
  fe_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  return r;
}

}  // namespace spk
