// HIP kernels (gfx950) of the NeutronNova NIFS data path: SURVEY.md 8(a) rows a21 (NeutronNovaNIFS::prove rounds) and a13 (small-value path).
// Reference semantics: src/neutronnova_zk.rs:98-432 (prove_helper*), :739-775 (fold_ab_pair!), :779-1165 (round structure),
// src/big_num/small_value.rs:31-222. All arithmetic is exact modular integer arithmetic, so the summation order is free and results are
// bit-identical to the reference's values whatever the reduction tree.
//
// Layout: the layers of one matrix are ONE contiguous array [layer][k], k < total = left * right (k = i * left + j, E[k] = e_left[j] * f[i],
// src/neutronnova_zk.rs:113-118). A block owns 256 consecutive k of one (pair, i): with left a multiple of 256 (every real size: left =
// 2^ceil(ell/2) >= 256 from 2^15 constraints up) the block sum is multiplied by f[i] and by the pair's rho weight ONCE (FACTORED); tiny test
// sizes take the per-element form.
#pragma once
#include "device_utils.hpp"

namespace spk {

struct NifsGeom {
  unsigned long long total;  // elements per layer
  unsigned left_log2;
  const fe_t* e_left;  // left entries
  const fe_t* f;       // right entries
};

__device__ __forceinline__ fe_t nifs_e(const NifsGeom& g, unsigned long long k, bool factored) {
  const fe_t el = g.e_left[k & ((1ull << g.left_log2) - 1)];
  return factored ? el : fe_mul<S>(el, g.f[k >> g.left_log2]);
}

// Round 0 (:779-851): quad = sum_pairs w[p] * sum_k E[k] (A_odd - A_even)(B_odd - B_even); e0 is identically zero (compute_e0 = false, :117).
template <bool FACTORED>
__global__ void __launch_bounds__(256) k_nifs_round0(const fe_t* __restrict__ A, const fe_t* __restrict__ B, NifsGeom g, const fe_t* __restrict__ w,
                                                     fe_t* __restrict__ partials) {
  __shared__ fe_t smem[4];
  const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long p = blockIdx.y;
  fe_t acc[1] = {fe_zero()};
  if (k < g.total) {
    const fe_t a0 = A[(2 * p) * g.total + k], a1 = A[(2 * p + 1) * g.total + k];
    const fe_t b0 = B[(2 * p) * g.total + k], b1 = B[(2 * p + 1) * g.total + k];
    acc[0] = fe_mul<S>(nifs_e(g, k, FACTORED), fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0)));
  }
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) {
    fe_t s = acc[0];
    if (FACTORED) s = fe_mul<S>(s, g.f[((unsigned long long)blockIdx.x * 256) >> g.left_log2]);
    partials[p * gridDim.x + blockIdx.x] = fe_mul<S>(s, w[p]);
  }
}

// Rounds t >= 1 (:855-1097), merged exactly as the reference merges them: fold layers (4j, 4j+1) and (4j+2, 4j+3) with the previous
// challenge, store them compacted at (2j, 2j+1) of the destination buffer (compact_folded_layers), and evaluate the pair on the values just
// produced: e0_ab = sum E[k] lo_a lo_b (prove_helper_ab_only, :186-246; the C part is subtracted by the caller from c_vals), quad as above.
// Traffic: 8 elements read + 4 written per k and prove pair = 384 B; nothing is re-read.
template <bool FACTORED>
__global__ void __launch_bounds__(256) k_nifs_fold_prove(const fe_t* __restrict__ A, const fe_t* __restrict__ B, fe_t* __restrict__ A_out,
                                                         fe_t* __restrict__ B_out, NifsGeom g, fe_t r, const fe_t* __restrict__ w,
                                                         fe_t* __restrict__ partials) {
  __shared__ fe_t smem[8];
  const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long j = blockIdx.y;
  fe_t acc[2] = {fe_zero(), fe_zero()};
  if (k < g.total) {
    const fe_t* a = A + (4 * j) * g.total + k;
    const fe_t* b = B + (4 * j) * g.total + k;
    const fe_t a0 = a[0], a1 = a[g.total], a2 = a[2 * g.total], a3 = a[3 * g.total];
    const fe_t b0 = b[0], b1 = b[g.total], b2 = b[2 * g.total], b3 = b[3 * g.total];
    const fe_t la = fe_add<S>(a0, fe_mul<S>(r, fe_sub<S>(a1, a0))), ha = fe_add<S>(a2, fe_mul<S>(r, fe_sub<S>(a3, a2)));
    const fe_t lb = fe_add<S>(b0, fe_mul<S>(r, fe_sub<S>(b1, b0))), hb = fe_add<S>(b2, fe_mul<S>(r, fe_sub<S>(b3, b2)));
    A_out[(2 * j) * g.total + k] = la;
    A_out[(2 * j + 1) * g.total + k] = ha;
    B_out[(2 * j) * g.total + k] = lb;
    B_out[(2 * j + 1) * g.total + k] = hb;
    const fe_t e = nifs_e(g, k, FACTORED);
    acc[0] = fe_mul<S>(e, fe_mul<S>(la, lb));
    acc[1] = fe_mul<S>(e, fe_mul<S>(fe_sub<S>(ha, la), fe_sub<S>(hb, lb)));
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    fe_t s0 = acc[0], s1 = acc[1];
    if (FACTORED) {
      const fe_t fi = g.f[((unsigned long long)blockIdx.x * 256) >> g.left_log2];
      s0 = fe_mul<S>(s0, fi);
      s1 = fe_mul<S>(s1, fi);
    }
    fe_t* dst = partials + 2 * (j * gridDim.x + blockIdx.x);
    dst[0] = fe_mul<S>(s0, w[j]);
    dst[1] = fe_mul<S>(s1, w[j]);
  }
}

// Evaluate pairs (2j, 2j+1) of already-folded layers (first round after a shard hand-off): same sums as k_nifs_fold_prove without the fold.
template <bool FACTORED>
__global__ void __launch_bounds__(256) k_nifs_prove_pairs(const fe_t* __restrict__ A, const fe_t* __restrict__ B, NifsGeom g, const fe_t* __restrict__ w,
                                                          fe_t* __restrict__ partials) {
  __shared__ fe_t smem[8];
  const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long j = blockIdx.y;
  fe_t acc[2] = {fe_zero(), fe_zero()};
  if (k < g.total) {
    const fe_t la = A[(2 * j) * g.total + k], ha = A[(2 * j + 1) * g.total + k];
    const fe_t lb = B[(2 * j) * g.total + k], hb = B[(2 * j + 1) * g.total + k];
    const fe_t e = nifs_e(g, k, FACTORED);
    acc[0] = fe_mul<S>(e, fe_mul<S>(la, lb));
    acc[1] = fe_mul<S>(e, fe_mul<S>(fe_sub<S>(ha, la), fe_sub<S>(hb, lb)));
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    fe_t s0 = acc[0], s1 = acc[1];
    if (FACTORED) {
      const fe_t fi = g.f[((unsigned long long)blockIdx.x * 256) >> g.left_log2];
      s0 = fe_mul<S>(s0, fi);
      s1 = fe_mul<S>(s1, fi);
    }
    fe_t* dst = partials + 2 * (j * gridDim.x + blockIdx.x);
    dst[0] = fe_mul<S>(s0, w[j]);
    dst[1] = fe_mul<S>(s1, w[j]);
  }
}

// Plain fold of layer pairs (fold_ab_pair!, :739-760): out[i] = in[2i] + r (in[2i+1] - in[2i]); blockIdx.z selects the matrix.
__global__ void __launch_bounds__(256) k_nifs_fold(const fe_t* __restrict__ A, const fe_t* __restrict__ B, fe_t* __restrict__ A_out, fe_t* __restrict__ B_out,
                                                   unsigned long long total, fe_t r) {
  const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= total) return;
  const fe_t* in = (blockIdx.z ? B : A) + (2ull * blockIdx.y) * total + k;
  fe_t* out = (blockIdx.z ? B_out : A_out) + (unsigned long long)blockIdx.y * total + k;
  const fe_t lo = in[0], hi = in[total];
  *out = fe_add<S>(lo, fe_mul<S>(r, fe_sub<S>(hi, lo)));
}

// c_vals[b] = sum_k E[k] Cz_b[k] (:652-703): lets every later round skip the C layers.
template <bool FACTORED>
__global__ void __launch_bounds__(256) k_nifs_cvals(const fe_t* __restrict__ C, NifsGeom g, fe_t* __restrict__ partials) {
  __shared__ fe_t smem[4];
  const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long b = blockIdx.y;
  fe_t acc[1] = {fe_zero()};
  if (k < g.total) {
    const fe_t c = C[b * g.total + k];
    if (!fe_is_zero(c)) acc[0] = fe_mul<S>(nifs_e(g, k, FACTORED), c);
  }
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) {
    fe_t s = acc[0];
    if (FACTORED) s = fe_mul<S>(s, g.f[((unsigned long long)blockIdx.x * 256) >> g.left_log2]);
    partials[b * gridDim.x + blockIdx.x] = s;
  }
}

// Sum NACC-interleaved partials: row = blockIdx.y holds n entries of NACC elements; block x sums entries [x*chunk, (x+1)*chunk) into
// out[(row * gridDim.x + x) * NACC ...].
template <int NACC>
__global__ void __launch_bounds__(256) k_nifs_sum(const fe_t* __restrict__ in, unsigned long long n, unsigned long long chunk, fe_t* __restrict__ out) {
  __shared__ fe_t smem[4 * NACC];
  const fe_t* row = in + (unsigned long long)blockIdx.y * n * NACC;
  const unsigned long long lo = (unsigned long long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  fe_t acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) acc[q] = fe_zero();
  for (unsigned long long i = lo + threadIdx.x; i < hi; i += 256) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = fe_add<S>(acc[q], row[i * NACC + q]);
  }
  block_sum<NACC>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) out[((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x) * NACC + q] = acc[q];
  }
}

}  // namespace spk

// ======== small-value path (a13) ==========================================================================================================
namespace spk {

constexpr unsigned long long SMALL_VALUE_MAX = (1ull << 62) - 1;  // small_value.rs:31

// to_small_vec_or_zero (small_value.rs:41-86): i64 image of each element, 0 + flag when neither v nor p - v is <= 2^62 - 1.
// `flags` is OR-ed into (callers accumulate the union over several tables, :1548-1572).
// blockIdx.y = layer: layer b reads in[b * n + i], writes out[b * n + i], and ORs into the SAME flags[i] (one launch for all instances).
__global__ void __launch_bounds__(256) k_to_small(const fe_t* __restrict__ in, unsigned long long n, long long* __restrict__ out, unsigned char* __restrict__ flags) {
  const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  in += (unsigned long long)blockIdx.y * n;
  out += (unsigned long long)blockIdx.y * n;
  const fe_t c = fe_to_canonical<S>(in[i]);
  const unsigned long long lo = (unsigned long long)c.v[0] | ((unsigned long long)c.v[1] << 32);
  long long r = 0;
  bool large = false;
  if ((c.v[2] | c.v[3] | c.v[4] | c.v[5] | c.v[6] | c.v[7]) == 0 && lo <= SMALL_VALUE_MAX) {
    r = (long long)lo;
  } else {
    uint32_t d[8], bw = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = sp_subb(S::P(q), c.v[q], bw);
    const unsigned long long dl = (unsigned long long)d[0] | ((unsigned long long)d[1] << 32);
    if ((d[2] | d[3] | d[4] | d[5] | d[6] | d[7]) == 0 && dl > 0 && dl <= SMALL_VALUE_MAX)
      r = -(long long)dl;
    else
      large = true;
  }
  out[i] = r;
  if (large) flags[i] = 1;
}
// zero the i64 mirrors of every layer at the globally large positions (:1575-1586); layers = blockIdx.y
__global__ void __launch_bounds__(256) k_small_mask(long long* __restrict__ v, unsigned long long total, const unsigned char* __restrict__ flags) {
  const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (k < total && flags[k]) v[(unsigned long long)blockIdx.y * total + k] = 0;
}

// Unreduced accumulator of (Montgomery limbs) x (128-bit magnitude) products: 13 x u32 = 416 bits, room for 2^32 terms. A negative term is
// entered as (p - e) x |q| (SmallAccumulator keeps two buckets instead, small_value.rs:99-166; the value mod p is the same).
struct lazy13_t {
  uint32_t v[13];
};
__device__ __forceinline__ lazy13_t lazy13_zero() {
  lazy13_t r;
#pragma unroll
  for (int i = 0; i < 13; ++i) r.v[i] = 0;
  return r;
}
__device__ __forceinline__ lazy13_t lazy13_add(const lazy13_t& a, const lazy13_t& b) {
  lazy13_t r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 13; ++i) r.v[i] = sp_addc(a.v[i], b.v[i], c);
  return r;
}
__device__ __forceinline__ lazy13_t mul_small(const fe_t& e, unsigned long long lo, unsigned long long hi) {
  const uint32_t q[4] = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
  lazy13_t r = lazy13_zero();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned long long carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned long long x = (unsigned long long)e.v[i] * q[j] + r.v[i + j] + carry;
      r.v[i + j] = (uint32_t)x;
      carry = x >> 32;
    }
    r.v[8 + j] = (uint32_t)carry;
  }
  return r;
}
// N mod p read as Montgomery limbs: N = lo + hi * 2^256, hi * 2^256 = montmul(hi, R^2)
__device__ __forceinline__ fe_t lazy13_reduce(const lazy13_t& a) {
  uint32_t lo[8];
  fe_t hi = fe_zero();
#pragma unroll
  for (int i = 0; i < 8; ++i) lo[i] = a.v[i];
#pragma unroll
  for (int i = 0; i < 5; ++i) hi.v[i] = a.v[8 + i];
  fe_t r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) r2.v[i] = S::R2(i);
  const fe_t l = fe_cond_sub_p<S>(lo, 0);
  return fe_add<S>(l, fe_mul<S>(hi, r2));
}
// block sum of one lazy13 accumulator; result valid in thread 0. smem: 4 entries.
__device__ __forceinline__ lazy13_t lazy13_block_sum(lazy13_t a, lazy13_t* smem) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    lazy13_t o;
#pragma unroll
    for (int i = 0; i < 13; ++i) o.v[i] = __shfl_xor(a.v[i], m, 64);
    a = lazy13_add(a, o);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) smem[wave] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) a = lazy13_add(a, smem[w]);
  }
  return a;
}
__device__ __forceinline__ unsigned long long abs64(long long v) { return v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v; }

// Round 0 from the i64 mirrors (prove_helper_small, :255-320): 32 B read per k and pair instead of 128 B, no modular product in the loop.
// A block owns 256 * ppt consecutive k (ppt <= left / 256 when FACTORED, so they share x_out): the 13-word cross-lane reduction, which costs more
// than the 32-mad product itself, is paid once per ppt terms.
template <bool FACTORED>
__global__ void __launch_bounds__(256) k_nifs_round0_small(const long long* __restrict__ A, const long long* __restrict__ B, NifsGeom g,
                                                           const fe_t* __restrict__ w, fe_t* __restrict__ partials, int ppt) {
  __shared__ lazy13_t smem[4];
  const unsigned long long base = (unsigned long long)blockIdx.x * 256 * ppt;
  const unsigned long long p = blockIdx.y;
  lazy13_t acc = lazy13_zero();
#pragma unroll 1
  for (int q = 0; q < ppt; ++q) {
    const unsigned long long k = base + (unsigned long long)q * 256 + threadIdx.x;
    if (k < g.total) {
      const long long da = A[(2 * p + 1) * g.total + k] - A[(2 * p) * g.total + k];
      const long long db = B[(2 * p + 1) * g.total + k] - B[(2 * p) * g.total + k];
      if (da != 0 && db != 0) {
        const unsigned long long ua = abs64(da), ub = abs64(db);
        fe_t e = nifs_e(g, k, FACTORED);
        if ((da < 0) != (db < 0)) e = fe_neg<S>(e);
        acc = lazy13_add(acc, mul_small(e, ua * ub, __umul64hi(ua, ub)));
      }
    }
  }
  acc = lazy13_block_sum(acc, smem);
  if (threadIdx.x == 0) {
    fe_t s = lazy13_reduce(acc);
    if (FACTORED) s = fe_mul<S>(s, g.f[base >> g.left_log2]);
    partials[p * gridDim.x + blockIdx.x] = fe_mul<S>(s, w[p]);
  }
}
// field-arithmetic correction of round 0 at the large positions (:297-313); grid (ceil(nlarge/256), pairs)
__global__ void __launch_bounds__(256) k_nifs_round0_large(const fe_t* __restrict__ A, const fe_t* __restrict__ B, NifsGeom g, const unsigned* __restrict__ large,
                                                           unsigned nlarge, const fe_t* __restrict__ w, fe_t* __restrict__ partials) {
  __shared__ fe_t smem[4];
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long p = blockIdx.y;
  fe_t acc[1] = {fe_zero()};
  if (idx < nlarge) {
    const unsigned long long k = large[idx];
    const fe_t a0 = A[(2 * p) * g.total + k], a1 = A[(2 * p + 1) * g.total + k];
    const fe_t b0 = B[(2 * p) * g.total + k], b1 = B[(2 * p + 1) * g.total + k];
    acc[0] = fe_mul<S>(nifs_e(g, k, false), fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0)));
  }
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) partials[p * gridDim.x + blockIdx.x] = fe_mul<S>(acc[0], w[p]);
}
// c_vals from the i64 mirror of C (:652-676) — the large positions are added by k_nifs_cvals_large
template <bool FACTORED>
__global__ void __launch_bounds__(256) k_nifs_cvals_small(const long long* __restrict__ C, NifsGeom g, fe_t* __restrict__ partials, int ppt) {
  __shared__ lazy13_t smem[4];
  const unsigned long long base = (unsigned long long)blockIdx.x * 256 * ppt;
  const unsigned long long b = blockIdx.y;
  lazy13_t acc = lazy13_zero();
#pragma unroll 1
  for (int q = 0; q < ppt; ++q) {
    const unsigned long long k = base + (unsigned long long)q * 256 + threadIdx.x;
    if (k < g.total) {
      const long long c = C[b * g.total + k];
      if (c != 0) {
        fe_t e = nifs_e(g, k, FACTORED);
        if (c < 0) e = fe_neg<S>(e);
        acc = lazy13_add(acc, mul_small(e, abs64(c), 0));
      }
    }
  }
  acc = lazy13_block_sum(acc, smem);
  if (threadIdx.x == 0) {
    fe_t s = lazy13_reduce(acc);
    if (FACTORED) s = fe_mul<S>(s, g.f[base >> g.left_log2]);
    partials[b * gridDim.x + blockIdx.x] = s;
  }
}
__global__ void __launch_bounds__(256) k_nifs_cvals_large(const fe_t* __restrict__ C, NifsGeom g, const unsigned* __restrict__ large, unsigned nlarge,
                                                          fe_t* __restrict__ partials) {
  __shared__ fe_t smem[4];
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long b = blockIdx.y;
  fe_t acc[1] = {fe_zero()};
  if (idx < nlarge) {
    const unsigned long long k = large[idx];
    acc[0] = fe_mul<S>(nifs_e(g, k, false), C[b * g.total + k]);
  }
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) partials[b * gridDim.x + blockIdx.x] = acc[0];
}

}  // namespace spk
