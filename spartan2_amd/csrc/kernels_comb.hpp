// Fixed-base comb tables of the commitment key and the row-commitment kernel over them (gfx950). Its own translation unit (capi_comb.hip): these are
// THROUGHPUT kernels and want the default scheduler (141 VGPRs, 3 waves per SIMD); capi_group.hip is built for ILP at low occupancy (-amdgpu-sched-strategy=
// max-ilp: the same kernel takes 313 registers there and runs one wave per SIMD).
#pragma once
#include "curve.hpp"

namespace spk {

typedef FqP SF;
__device__ __forceinline__ unsigned comb_row_len(size_t row, size_t cols, size_t n) {
  const size_t lo = row * cols;
  return (unsigned)((lo + cols <= n) ? cols : n - lo);
}

// ---- fixed-base comb tables for the commitment key ("many rows, one key": PCS::commit of non-small witnesses, BASELINE config 4) ---------------
// The reference keeps per-base FixedBaseMul tables only for keys of <= 64 bases (hyrax_pc.rs:81-96, msm.rs:653-773) because a CPU cannot afford
// them for 2048; with 288 GB of HBM the same idea scales: table[w][j][d-1] = d * 2^(C w) * ck[j] for signed C-bit digits d in 1..2^(C-1)
// (C = 12: 22 windows x 2048 bases x 2048 entries x 64 B = 5.9 GB). A row commitment is then sum_j sum_w +-table[w][j][|digit_w(s_j)|-1]: no sort,
// no buckets, no window sums, no doublings, and every lane does the same number of additions (the bucket form loses ~40 % to the spread of the
// bucket sizes inside a wave). Same group element as the Pippenger sum, hence the same affine bytes.
// build, stage 1: one lane per (window, base): P = 2^(C w) ck[j] by doublings, then its multiples 1..E as Jacobian points
// (the chain of E additions is cut into `segs` pieces so that a build fills the chip: piece g starts at (g E / segs + 1) P by double-and-add)
__global__ void __launch_bounds__(64) k_comb_multiples(const aff_t* __restrict__ bases, unsigned ncols, int c, int w0, int nw, unsigned E, unsigned segs,
                                                      jac_t* __restrict__ out /* [nw][ncols][E] */) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)nw * ncols * segs) return;
  const unsigned seg = (unsigned)(gid % segs), j = (unsigned)((gid / segs) % ncols), wl = (unsigned)(gid / segs / ncols);
  jac_t p = jac_from_affine(bases[j]);
  for (int k = 0; k < c * (w0 + (int)wl); ++k) p = jac_dbl(p);
  const aff_t pa = jac_to_affine(p);
  const unsigned per = E / segs, first = seg * per + 1;  // this piece holds the multiples first .. first + per - 1
  jac_t acc = jac_identity();
  for (int b = 31; b >= 0; --b) {
    acc = jac_dbl(acc);
    if ((first >> b) & 1u) acc = jac_add_mixed(acc, pa);
  }
  jac_t* dst = out + ((size_t)wl * ncols + j) * E + (first - 1);
  dst[0] = acc;
  for (unsigned d = 1; d < per; ++d) {
    acc = jac_add_mixed(acc, pa);
    dst[d] = acc;
  }
}
// build, stage 2: Jacobian -> affine, 8 points per lane sharing one inversion (Montgomery's trick)
__global__ void __launch_bounds__(256) k_comb_normalize(const jac_t* __restrict__ in, size_t n, aff_t* __restrict__ out) {
  const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (base >= n) return;
  const int cnt = n - base < 8 ? (int)(n - base) : 8;
  fe_t pre[8];
  fe_t run = fe_one<B>();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    pre[i] = run;
    if (i < cnt) {
      const fe_t z = in[base + i].z;
      if (!fe_is_zero(z)) run = fe_mul<B>(run, z);
    }
  }
  fe_t inv = fe_inv<B>(run);
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    if (i < cnt) {
      const jac_t p = in[base + i];
      aff_t a;
      if (fe_is_zero(p.z)) {
        a.x = fe_zero();
        a.y = fe_zero();
      } else {
        const fe_t zi = fe_mul<B>(inv, pre[i]);
        inv = fe_mul<B>(inv, p.z);
        const fe_t zi2 = fe_sqr<B>(zi);
        a.x = fe_mul<B>(p.x, zi2);
        a.y = fe_mul<B>(fe_mul<B>(p.y, zi2), zi);
      }
      out[base + i] = a;
    }
  }
}
// use: one 256-thread block per selected row; thread t takes the bases t, t + 256, ...; per scalar the signed C-bit digits go through LDS (so the
// window loop stays rolled: one table entry in flight ahead of the addition that consumes the previous one); 8-level tree over the block at the end.
template <int C, int MINW>
__global__ void __launch_bounds__(256, MINW) k_comb_rows(const fe_t* __restrict__ canon, const unsigned* __restrict__ rows, size_t cols, size_t n,
                                                   const aff_t* __restrict__ table, int windows, jac_t* __restrict__ out) {
  constexpr int MAXW = (257 + C - 1) / C;
  constexpr unsigned E = 1u << (C - 1);
  __shared__ short dg[MAXW][256];
  __shared__ xyzz_t red[256];
  const size_t row = rows[blockIdx.x];
  const unsigned len = comb_row_len(row, cols, n);
  xyzz_t acc = xyzz_identity();  // XYZZ: 10 products per mixed addition instead of 11, 14 instead of 16 in the tree
  for (unsigned j = threadIdx.x; j < len; j += 256) {
    const fe_t sc = canon[row * cols + j];
    int carry = 0;
#pragma unroll
    for (int w = 0; w < MAXW; ++w) {
      const int pos = C * w, idx = pos >> 5, sh = pos & 31;
      unsigned raw = 0;
      if (idx < 8) {
        raw = sc.v[idx] >> sh;
        if (sh + C > 32 && idx + 1 < 8) raw |= sc.v[idx + 1] << (32 - sh);
        raw &= (1u << C) - 1u;
      }
      int r = (int)raw + carry;
      if (r > (int)E) {
        r -= (int)(1u << C);
        carry = 1;
      } else {
        carry = 0;
      }
      dg[w][threadIdx.x] = (short)r;
    }
    // (own column of dg only: no barrier needed)
    const aff_t* tj = table + (size_t)j * E;
    aff_t qn;
    int dn = dg[0][threadIdx.x];
    if (dn) qn = tj[(dn < 0 ? -dn : dn) - 1];
#pragma unroll 1
    for (int w = 0; w < windows; ++w) {
      const int d = dn;
      aff_t q = qn;
      if (w + 1 < windows) {
        dn = dg[w + 1][threadIdx.x];
        if (dn) qn = table[((size_t)(w + 1) * cols + j) * E + (dn < 0 ? -dn : dn) - 1];
      }
      if (d) {
        if (d < 0) q.y = fe_neg<B>(q.y);
        acc = xyzz_add_mixed(acc, q);
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] = xyzz_add(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = xyzz_to_jac(red[0]);
}


}  // namespace spk
