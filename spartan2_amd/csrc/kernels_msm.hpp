// HIP kernels for the group side (Hyrax commitments) on gfx950:
//   k_to_canonical        to_repr() of a scalar array (Montgomery -> canonical), also flags "all small / all bits"
//   k_msm_sort            K11 stage 1: signed 8-bit digits (msm.rs:110-145) + LDS counting sort by bucket, one block per window
//   k_msm_bucket_sum      K11 stage 2: bucket accumulation, 8 lanes per bucket, mixed adds + shuffle tree
//   k_msm_window_reduce   K11 stage 3: sum_k k*B_k per window by suffix scan + tree in LDS (summation by parts, msm.rs:169-174)
//   k_msm_binary_rows     K10 msm_binary per Hyrax row (msm.rs:418-451 via hyrax_pc.rs:230-300)
//   k_fixed_base_rows     K12 FixedBaseMul::mul per scalar (msm.rs:691-725): 32 table lookups + wave tree
//   k_fixed_base_table    FixedBaseMul::precompute (msm.rs:653-689)
//   k_rowmat_vec          K9  bind_with_delayed (hyrax_pc.rs:38-54)
// The window Horner (256 doublings) and the final normalisation are short sequential tails done by the host side
// of the library (capi_group.hip) — one CPU core is ~20x faster than one GPU lane at a dependent chain.
// Results are group elements; parity is on canonical affine coordinates, so summation order is free.
#pragma once
#include "curve.hpp"
#include "device_utils.hpp"

// SP_STAMP(i): wall-clock stamps (100 MHz) of block 0 / thread 0 for the latency kernels - compiled in only by tools/fb_stamps.hip
#ifdef SP_KERNEL_STAMPS
__device__ unsigned long long sp_stamps[192];
#define SP_STAMP(i)                                                     \
  do {                                                                  \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                          \
      sp_stamps[i] = wall_clock64();                                    \
      sp_stamps[32 + (i)] = clock64();                                  \
    }                                                                   \
  } while (0)
__device__ unsigned sp_stage_ctr[4];
__device__ unsigned sp_hwid[4];
__device__ unsigned sp_predelay;  // 10 ns ticks every wave spins for before it starts (is the slow phase tied to time since launch or to the tree level?)
#ifdef SP_KERNEL_STAGE_STAMPS  // (each stage stamp costs a global counter round trip: only for looking inside a level, not for timing one)
#define SP_STAGE_STAMP()                                                          \
  do {                                                                            \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) sp_stamps[64 + 32 * (threadIdx.x >> 6) + (sp_stage_ctr[threadIdx.x >> 6]++ & 31)] = wall_clock64(); \
  } while (0)
#else
#define SP_STAGE_STAMP()
#endif
#else
#define SP_STAMP(i)
#define SP_STAGE_STAMP()
#endif

namespace spk {

typedef FqP SF;  // scalar field

__device__ __forceinline__ jac_t shfl_down_jac(const jac_t& a, int delta) {
  jac_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.x.v[i] = __shfl_down(a.x.v[i], delta, 64);
    r.y.v[i] = __shfl_down(a.y.v[i], delta, 64);
    r.z.v[i] = __shfl_down(a.z.v[i], delta, 64);
  }
  return r;
}

__device__ __forceinline__ xyzz_t shfl_down_xyzz(const xyzz_t& a, int delta) {
  xyzz_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.x.v[i] = __shfl_down(a.x.v[i], delta, 64);
    r.y.v[i] = __shfl_down(a.y.v[i], delta, 64);
    r.zz.v[i] = __shfl_down(a.zz.v[i], delta, 64);
    r.zzz.v[i] = __shfl_down(a.zzz.v[i], delta, 64);
  }
  return r;
}

// canonical (non-Montgomery) limbs of each scalar
__global__ void __launch_bounds__(256) k_to_canonical(const fe_t* __restrict__ in, size_t n, fe_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = fe_to_canonical<SF>(in[i]);
}
// eq(r, i) = left[i >> lo_bits] * right[i & mask] (canonical limbs for MSM scalars, Montgomery form for field work): an eq table of up to 10 variables formed from its
// two half tables (<= 32 entries each, passed by value — no upload, no synchronisation)
struct EqTensorArgs {
  fe_t left[32], right[32];
  int lo_bits;
  unsigned n;
};
template <bool CANONICAL>
__global__ void __launch_bounds__(256) k_eq_tensor(EqTensorArgs a, fe_t* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const fe_t w = fe_mul<SF>(a.left[i >> a.lo_bits], a.right[i & ((1u << a.lo_bits) - 1)]);
  out[i] = CANONICAL ? fe_to_canonical<SF>(w) : w;
}
// Sign folding for the digit path: s -> (min(s, n - s), sign) with n the group order, so every folded scalar is < 2^255:
// the top byte is < 128, the carry window (msm.rs:137-145) stays almost empty and buckets are balanced. s*P == (n-s)*(-P).
// The sign lives in bit 31 of limb 7 of the output.
__global__ void __launch_bounds__(256) k_fold_sign(const fe_t* __restrict__ canon, size_t n, fe_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const fe_t c = canon[i];
    fe_t d;
    uint32_t bw = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d.v[k] = sp_subb(SF::P(k), c.v[k], bw);  // n - c (c canonical, so no borrow)
    uint32_t lt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) (void)sp_subb(d.v[k], c.v[k], lt);  // lt == 1 iff d < c
    fe_t r = lt ? d : c;
    if (fe_is_zero(c)) r = c;  // n - 0 == n is not a scalar
    r.v[7] |= (lt && !fe_is_zero(c)) ? 0x80000000u : 0u;
    out[i] = r;
  }
}

// per-row classification for PCS::commit (hyrax_pc.rs:243-292): bit0 = row has a non-zero, bit1 = some value > 1,
// bit2 = some value >= 2^64. rows of `cols` canonical scalars, last row may be short.
__global__ void __launch_bounds__(256) k_classify_rows(const fe_t* __restrict__ canon, size_t n, size_t cols, unsigned* __restrict__ flags) {
  const size_t row = blockIdx.x;
  const size_t lo = row * cols, hi = (lo + cols < n) ? lo + cols : n;
  unsigned f = 0;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const fe_t c = canon[i];
    unsigned upper = c.v[2] | c.v[3] | c.v[4] | c.v[5] | c.v[6] | c.v[7];
    if (upper) f |= 7u;
    else if (c.v[1] || c.v[0] > 1) f |= 3u;
    else if (c.v[0]) f |= 1u;
  }
  if (f) atomicOr(&flags[row], f);
}

// ---- K11 signed-digit Pippenger, window c = 8 ---------------------------------------------------------------------
constexpr int MSM_C = 8;
constexpr int MSM_BUCKETS = 1 << (MSM_C - 1);  // 128 signed buckets per window
constexpr int MSM_MAX_WINDOWS = 33;            // 32 byte windows + the carry window (msm.rs:137-145)

// signed digit of window w: d = ((s + sum_j 2^(8j+7)) >> 8w & 255) - 128 restated without big-integer adds:
// raw byte + carry-in, carry-in(w) = 1 iff the lower windows recoded with a carry — computed by scanning bytes.
__device__ __forceinline__ int signed_digit(const fe_t& c, int w) {
  // carry into window w: propagate from window 0 (cheap: <= 32 steps, scalars are read once per block pass)
  int carry = 0;
  int d = 0;
  for (int k = 0; k <= w; ++k) {
    int raw = (k < 32) ? (int)((c.v[k >> 2] >> (8 * (k & 3))) & (k == 31 ? 0x7f : 0xff)) : 0;  // bit 255 is the fold sign
    raw += carry;
    if (raw >= 128) {
      d = raw - 256;
      carry = 1;
    } else {
      d = raw;
      carry = 0;
    }
  }
  return d;
}

// All signed digits of a scalar in one pass (the carry is a sequential chain over the windows): digits[w * n + j], bit 7 of the SIGN byte array is not
// needed — the fold sign lives in the scalar. Fully unrolled, so the limbs stay in registers (the per-window form above indexes them dynamically).
__global__ void __launch_bounds__(256) k_msm_digits(const fe_t* __restrict__ canon, unsigned n, int windows, signed char* __restrict__ digits) {
  const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const fe_t c = canon[j];
  int carry = 0;
#pragma unroll
  for (int k = 0; k < MSM_MAX_WINDOWS; ++k) {
    if (k < windows) {
      int raw = (k < 32) ? (int)((c.v[k >> 2] >> (8 * (k & 3))) & (k == 31 ? 0x7f : 0xff)) : 0;
      raw += carry;
      int d;
      if (raw >= 128) {
        d = raw - 256;
        carry = 1;
      } else {
        d = raw;
        carry = 0;
      }
      digits[(size_t)k * n + j] = (signed char)d;
    }
  }
}
// k_msm_sort on precomputed digits
__global__ void __launch_bounds__(256) k_msm_sort_digits(const signed char* __restrict__ digits, const fe_t* __restrict__ canon, unsigned n,
                                                         unsigned* __restrict__ order, unsigned* __restrict__ start) {
  __shared__ unsigned hist[MSM_BUCKETS + 1];
  __shared__ unsigned cursor[MSM_BUCKETS];
  const int w = blockIdx.x;
  const signed char* dg = digits + (size_t)w * n;
  for (int k = threadIdx.x; k <= MSM_BUCKETS; k += blockDim.x) hist[k] = 0;
  __syncthreads();
  for (unsigned j = threadIdx.x; j < n; j += blockDim.x) {
    const int d = dg[j];
    if (d) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // exclusive prefix over the 128 counts by one wave (two per lane + a shuffle scan)
    const unsigned a = hist[2 * threadIdx.x], b = hist[2 * threadIdx.x + 1];
    unsigned incl = a + b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned o = __shfl_up(incl, off, 64);
      if ((int)threadIdx.x >= off) incl += o;
    }
    const unsigned excl = incl - (a + b);
    hist[2 * threadIdx.x] = excl;
    hist[2 * threadIdx.x + 1] = excl + a;
    cursor[2 * threadIdx.x] = excl;
    cursor[2 * threadIdx.x + 1] = excl + a;
    if (threadIdx.x == 63) hist[MSM_BUCKETS] = incl;
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= MSM_BUCKETS; k += blockDim.x) start[(size_t)w * (MSM_BUCKETS + 1) + k] = hist[k];
  for (unsigned j = threadIdx.x; j < n; j += blockDim.x) {
    const int d = dg[j];
    if (d) {
      unsigned pos = atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
      const bool neg = (d < 0) != ((canon[j].v[7] >> 31) != 0);
      order[(size_t)w * n + pos] = j | (neg ? 0x80000000u : 0u);
    }
  }
}

// One block per window. Outputs, per window w: order[w*n + pos] = base index | (negate << 31), grouped by bucket;
// start[w*(BUCKETS+1) + k] = first position of bucket k+1's list (k = |digit| - 1).
__global__ void __launch_bounds__(256) k_msm_sort(const fe_t* __restrict__ canon, unsigned n, unsigned* __restrict__ order,
                                                  unsigned* __restrict__ start) {
  __shared__ unsigned hist[MSM_BUCKETS + 1];
  __shared__ unsigned cursor[MSM_BUCKETS];
  const int w = blockIdx.x;
  for (int k = threadIdx.x; k <= MSM_BUCKETS; k += blockDim.x) hist[k] = 0;
  __syncthreads();
  for (unsigned j = threadIdx.x; j < n; j += blockDim.x) {
    int d = signed_digit(canon[j], w);
    if (d) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int k = 0; k < MSM_BUCKETS; ++k) {
      unsigned cnt = hist[k];
      hist[k] = run;
      cursor[k] = run;
      run += cnt;
    }
    hist[MSM_BUCKETS] = run;
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= MSM_BUCKETS; k += blockDim.x) start[(size_t)w * (MSM_BUCKETS + 1) + k] = hist[k];
  for (unsigned j = threadIdx.x; j < n; j += blockDim.x) {
    const fe_t c = canon[j];
    int d = signed_digit(c, w);
    if (d) {
      unsigned pos = atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
      const bool neg = (d < 0) != ((c.v[7] >> 31) != 0);
      order[(size_t)w * n + pos] = j | (neg ? 0x80000000u : 0u);
    }
  }
}

// 8 lanes per bucket; grid covers windows * 128 buckets * 8 lanes.
#ifndef MSM_LPB
#define MSM_LPB 8
#endif
constexpr int MSM_LANES_PER_BUCKET = MSM_LPB;
__global__ void __launch_bounds__(256) k_msm_bucket_sum(const aff_t* __restrict__ bases, unsigned n, const unsigned* __restrict__ order,
                                                        const unsigned* __restrict__ start, int windows, jac_t* __restrict__ buckets) {
  // blockIdx.y = row of a shared-weights batch (msm.rs:228-356): same digits, different bases; 0 for a plain MSM
  bases += (size_t)blockIdx.y * n;
  buckets += (size_t)blockIdx.y * windows * MSM_BUCKETS;
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned bucket = gid / MSM_LANES_PER_BUCKET, sub = gid % MSM_LANES_PER_BUCKET;
  const unsigned total = (unsigned)windows * MSM_BUCKETS;
  xyzz_t acc = xyzz_identity();
  if (bucket < total) {
    const unsigned w = bucket / MSM_BUCKETS, k = bucket % MSM_BUCKETS;
    const unsigned lo = start[(size_t)w * (MSM_BUCKETS + 1) + k], hi = start[(size_t)w * (MSM_BUCKETS + 1) + k + 1];
    for (unsigned p = lo + sub; p < hi; p += MSM_LANES_PER_BUCKET) {
      const unsigned e = order[(size_t)w * n + p];
      aff_t q = bases[e & 0x7fffffffu];
      if (e & 0x80000000u) q = aff_neg(q);
      acc = xyzz_add_mixed(acc, q);
    }
  }
#pragma unroll
  for (int d = MSM_LANES_PER_BUCKET / 2; d >= 1; d >>= 1) {
    xyzz_t o = shfl_down_xyzz(acc, d);
    if (sub < (unsigned)d) acc = xyzz_add(acc, o);
  }
  if (bucket < total && sub == 0) buckets[bucket] = xyzz_to_jac(acc);
}

// ---- block-cooperative addition in XYZZ coordinates -------------------------------------------------------------------------------------------
// The MSM tail is a chain of DEPENDENT additions on few points; a lone wave issues the base-field products of one addition back to back (~1 us
// each: the SIMD is saturated by one wave's quarter-rate 64-bit multiply-adds), ~15 us per Jacobian addition. Here FOUR wave groups ("roles", each
// on its own SIMD) share every addition: each role computes one product per dependency level for all ITEMS point pairs and the levels meet in LDS.
// Coordinates are (X, Y, ZZ, ZZZ) with x = X / ZZ, y = Y / ZZZ (add-2008-s): 14 products in FOUR levels of <= 4 - the Jacobian add-2007-bl this
// replaces needs 16 in five - so a chain of dependent additions is a fifth shorter. Points enter (Jacobian buckets, affine table entries) and leave
// (Jacobian sums for the host's Horner / normalisation) through two products each; the group element, hence every byte downstream, is the same.
template <int ITEMS>
struct CoopAdd {
  fe_t t[9][ITEMS];   // U1 -> Q | U2 -> P -> Y3a | S1 | S2 -> R | PP -> Y3b | RR | ZZ1 ZZ2 -> ZZ3 | ZZZ1 ZZZ2 -> ZZZ3 | PPP
  xyzz_t fix[ITEMS];  // results of the special cases (identity operand, P = +-Q), computed while the inputs are still intact
  int flag[ITEMS];
};
// All ITEMS * 4 threads of the block call this (it synchronises). role = threadIdx / ITEMS, i = threadIdx % ITEMS; P[i] += Q index given by the
// caller as pointers into LDS; `active` = this item takes part. The sum is written to dst[i] (may alias P: results are stored after the last level).
// CODE SIZE is what this routine is written around. Every wave of a latency kernel runs each instruction once per tree level, and a product is ~3 KB
// of straight-line code: with one inlined product per (stage, role) - 14 of them, plus the doubling case - a level was ~100 KB against a 64 KB
// instruction cache, and the same level took 5 us with its code cached and 10-12 us without (tools/fb_stamps.hip). Here every stage has ONE product that
// all four roles execute on operands they pick by address, so a level is ~15 KB and stays cached from the second level on. (A real call per product is
// no way out: 18 us per level with the call ABI's moves and scratch set-up.)
template <int ITEMS>
__device__ __forceinline__ void xyzz_add_block4(CoopAdd<ITEMS>& L, const xyzz_t* P, const xyzz_t* Q, xyzz_t* dst, int role, int i, bool active) {
  // EXEC stays FULL through the products: items that do not take part compute on whatever their slots hold and only the flag / result stores are
  // predicated. Measured (tools/fb_stamps.hip, profiles/r03_sparse_exec.txt): the same level of the same tree takes 5.2 us with every lane computing
  // and, in two launches out of three, 10-13 us once only 8 or 4 lanes of each wave are enabled - the long dependent v_mad_u64_u32 / v_addc chains
  // run 2.5-3x slower under a sparse EXEC mask on this part.
  // (A wave none of whose items takes part skips the products altogether: it would only compete with the wave it shares its SIMD with.)
  fe_t(*t)[ITEMS] = L.t;
  const bool run = __ballot(active) != 0;  // wave-uniform
  // stage 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
  if (run) {
    if (active && role == 0) {
      int f = 0;
      if (xyzz_is_identity(*P)) {
        L.fix[i] = *Q;
        f = 1;
      } else if (xyzz_is_identity(*Q)) {
        L.fix[i] = *P;
        f = 1;
      }
      L.flag[i] = f;
    }
    const xyzz_t* a = (role & 1) ? Q : P;
    const xyzz_t* b = (role & 1) ? P : Q;
    const fe_t* xp = (role & 2) ? &a->y : &a->x;
    const fe_t* yp = (role & 2) ? &b->zzz : &b->zz;
    t[role][i] = fe_mul_rowwise<B>(*xp, *yp);
  }
  SP_STAGE_STAMP();
  __syncthreads();
  // stage 2: PP = (U2 - U1)^2, P kept | RR = (S2 - S1)^2, R kept | ZZ1 ZZ2 | ZZZ1 ZZZ2
  if (run) {
    fe_t x, y;
    if (role < 2) {
      x = fe_sub<B>(t[2 * role + 1][i], t[2 * role][i]);
      t[2 * role + 1][i] = x;
      y = x;
    } else {
      x = *((role & 1) ? &P->zzz : &P->zz);
      y = *((role & 1) ? &Q->zzz : &Q->zz);
    }
    t[4 + role][i] = fe_mul_rowwise<B>(x, y);
  }
  SP_STAGE_STAMP();
  __syncthreads();
  // stage 3: PPP = P PP (+ the P = +-Q case) | Q = U1 PP | ZZ3 = ZZ1 ZZ2 PP
  if (run && role < 3) {
    if (role == 0 && active && __builtin_expect(fe_is_zero(t[1][i]) && !L.flag[i], 0)) {
      L.fix[i] = fe_is_zero(t[3][i]) ? xyzz_dbl(*P) : xyzz_identity();
      L.flag[i] = 1;
    }
    const int xi = role == 0 ? 1 : role == 1 ? 0 : 6, oi = role == 0 ? 8 : role == 1 ? 0 : 6;
    t[oi][i] = fe_mul_rowwise<B>(t[xi][i], t[4][i]);
  }
  SP_STAGE_STAMP();
  __syncthreads();
  // stage 4: R (Q - X3) | S1 PPP | ZZZ3 = ZZZ1 ZZZ2 PPP
  if (run && role < 3) {
    const int xi = role == 0 ? 3 : role == 1 ? 2 : 7, oi = role == 0 ? 1 : role == 1 ? 4 : 7;
    fe_t y = t[8][i];
    if (role == 0) {
      const fe_t q = t[0][i];
      const fe_t x3 = fe_sub<B>(fe_sub<B>(t[5][i], y), fe_dbl<B>(q));
      y = fe_sub<B>(q, x3);
    }
    t[oi][i] = fe_mul_rowwise<B>(t[xi][i], y);
  }
  SP_STAGE_STAMP();
  __syncthreads();
  if (active && role == 0) {
    xyzz_t r;
    if (L.flag[i]) {
      r = L.fix[i];
    } else {
      r.x = fe_sub<B>(fe_sub<B>(t[5][i], t[8][i]), fe_dbl<B>(t[0][i]));
      r.y = fe_sub<B>(t[1][i], t[4][i]);
      r.zz = t[6][i];
      r.zzz = t[7][i];
    }
    dst[i] = r;
  }
  __syncthreads();
}

// per window: W = sum_k k B_k by suffix scan + tree as below, every addition shared by four wave groups (512 threads per window)
__global__ void __launch_bounds__(4 * MSM_BUCKETS) k_msm_window_reduce_coop(const jac_t* __restrict__ buckets, jac_t* __restrict__ window_sums) {
  __shared__ CoopAdd<MSM_BUCKETS> L;
  __shared__ xyzz_t s[MSM_BUCKETS];
  const int w = blockIdx.x + blockIdx.y * gridDim.x;
  // wave -> (role, item block): wave % 4 is the SIMD a wave lands on, so the four roles of an item block sit on four different SIMDs and a level
  // with <= 64 active items costs ONE product time; the second item block rotates its roles by two so the short levels (3 and 2 products) balance.
  const int wave = threadIdx.x >> 6, blk = wave >> 2;
  const int role = (wave + 2 * blk) & 3, k = blk * 64 + (threadIdx.x & 63);
  if (role == 0) s[k] = xyzz_from_jac(buckets[(size_t)w * MSM_BUCKETS + k]);
  __syncthreads();
  for (int off = 1; off < MSM_BUCKETS; off <<= 1) {  // suffix scan: s_k += s_{k+off}; in place is safe, results are stored after the last level
    const bool active = k + off < MSM_BUCKETS;
    xyzz_add_block4<MSM_BUCKETS>(L, &s[k], &s[active ? k + off : k], s, role, k, active);
  }
  for (int off = MSM_BUCKETS / 2; off >= 1; off >>= 1) {
    const bool active = k < off;
    xyzz_add_block4<MSM_BUCKETS>(L, &s[k], &s[active ? k + off : k], s, role, k, active);
  }
  if (threadIdx.x == 0) window_sums[w] = xyzz_to_jac(s[0]);
}

// per window: W = sum_{k=1..128} k * B_k = sum_k S_k with S_k = sum_{j >= k} B_j (suffix scan, then tree)
__global__ void __launch_bounds__(MSM_BUCKETS) k_msm_window_reduce(const jac_t* __restrict__ buckets, jac_t* __restrict__ window_sums) {
  __shared__ xyzz_t s[MSM_BUCKETS];
  const int w = blockIdx.x + blockIdx.y * gridDim.x, k = threadIdx.x;  // blockIdx.y = row of a shared-weights batch
  s[k] = xyzz_from_jac(buckets[(size_t)w * MSM_BUCKETS + k]);
  __syncthreads();
  for (int off = 1; off < MSM_BUCKETS; off <<= 1) {  // Hillis-Steele suffix scan
    xyzz_t o = (k + off < MSM_BUCKETS) ? s[k + off] : xyzz_identity();
    __syncthreads();
    if (k + off < MSM_BUCKETS) s[k] = xyzz_add(s[k], o);
    __syncthreads();
  }
  for (int off = MSM_BUCKETS / 2; off >= 1; off >>= 1) {
    if (k < off) s[k] = xyzz_add(s[k], s[k + off]);
    __syncthreads();
  }
  if (k == 0) window_sums[w] = xyzz_to_jac(s[0]);
}

// ---- batched row MSMs: many rows, ONE base vector, different scalars (PCS::commit on non-small witnesses, hyrax_pc.rs:230-300; the 2048 x 2048
// full-scalar row MSMs of BASELINE config 4). Throughput regime — every (row, window, bucket) gets its own lane and plain sequential mixed
// additions (no shuffle trees: the machine is full); sort and window reduction are the single-MSM kernels with a row dimension.
// rows[y] = index of the y-th selected row; its scalars are canon[rows[y] * cols .. + len(y)).
__device__ __forceinline__ unsigned batched_row_len(size_t row, size_t cols, size_t n) {
  const size_t lo = row * cols;
  return (unsigned)((lo + cols <= n) ? cols : n - lo);
}
__global__ void __launch_bounds__(256) k_fold_sign_rows(const fe_t* __restrict__ canon, const unsigned* __restrict__ rows, size_t cols, size_t n,
                                                        fe_t* __restrict__ out /* [nrows][cols] */) {
  const size_t row = rows[blockIdx.y];
  const unsigned len = batched_row_len(row, cols, n);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < len; i += gridDim.x * blockDim.x) {
    const fe_t c = canon[row * cols + i];
    fe_t d;
    uint32_t bw = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) d.v[k] = sp_subb(SF::P(k), c.v[k], bw);
    uint32_t lt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) (void)sp_subb(d.v[k], c.v[k], lt);
    fe_t o = lt ? d : c;
    if (lt) o.v[7] |= 0x80000000u;
    out[(size_t)blockIdx.y * cols + i] = o;
  }
}
// grid (windows, nrows): LDS counting sort of one (row, window), as k_msm_sort
__global__ void __launch_bounds__(256) k_msm_sort_rows(const fe_t* __restrict__ scal /* [nrows][cols] or canon when rows != null */, const unsigned* __restrict__ rows,
                                                       size_t cols, size_t n, unsigned* __restrict__ order, unsigned* __restrict__ start) {
  __shared__ unsigned hist[MSM_BUCKETS + 1];
  __shared__ unsigned cursor[MSM_BUCKETS];
  const int w = blockIdx.x, windows = gridDim.x;
  const size_t y = blockIdx.y;
  const size_t row = rows[y];
  const unsigned len = batched_row_len(row, cols, n);
  const fe_t* src = scal + y * cols;
  order += (y * windows + w) * cols;
  start += (y * windows + w) * (MSM_BUCKETS + 1);
  for (int k = threadIdx.x; k <= MSM_BUCKETS; k += blockDim.x) hist[k] = 0;
  __syncthreads();
  for (unsigned j = threadIdx.x; j < len; j += blockDim.x) {
    int d = signed_digit(src[j], w);
    if (d) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int k = 0; k < MSM_BUCKETS; ++k) {
      unsigned cnt = hist[k];
      hist[k] = run;
      cursor[k] = run;
      run += cnt;
    }
    hist[MSM_BUCKETS] = run;
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= MSM_BUCKETS; k += blockDim.x) start[k] = hist[k];
  for (unsigned j = threadIdx.x; j < len; j += blockDim.x) {
    const fe_t c = src[j];
    int d = signed_digit(c, w);
    if (d) {
      unsigned pos = atomicAdd(&cursor[(d < 0 ? -d : d) - 1], 1u);
      const bool neg = (d < 0) != ((c.v[7] >> 31) != 0);
      order[pos] = j | (neg ? 0x80000000u : 0u);
    }
  }
}
// one lane per (row, window, bucket)
__global__ void __launch_bounds__(256) k_msm_bucket_sum_rows(const aff_t* __restrict__ bases, size_t cols, const unsigned* __restrict__ order,
                                                             const unsigned* __restrict__ start, size_t total_buckets, jac_t* __restrict__ buckets) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= total_buckets) return;
  const size_t rw = b / MSM_BUCKETS, k = b % MSM_BUCKETS;  // rw = row-slot * windows + window
  const unsigned lo = start[rw * (MSM_BUCKETS + 1) + k], hi = start[rw * (MSM_BUCKETS + 1) + k + 1];
  xyzz_t acc = xyzz_identity();
  for (unsigned p = lo; p < hi; ++p) {
    const unsigned e = order[rw * cols + p];
    aff_t q = bases[e & 0x7fffffffu];
    if (e & 0x80000000u) q = aff_neg(q);
    acc = xyzz_add_mixed(acc, q);
  }
  buckets[b] = xyzz_to_jac(acc);
}

// Shared-weights batch (msm.rs:228-356) in the throughput regime - many rows, few points per bucket (the commitment fold of hundreds of instances:
// 512 rows x 256 weights at BASELINE config 5): one lane per (row, window, bucket) walks its bucket's entries sequentially; the digit order is
// shared by every row, the bases are the row's own. (The 8-lanes-per-bucket form with its shuffle tree pays three dependent additions per bucket
// even when the bucket holds two points: 11.8 ms against 1.4 ms here.)
__global__ void __launch_bounds__(256) k_msm_bucket_sum_shared(const aff_t* __restrict__ bases /* [rows][n] */, unsigned n, const unsigned* __restrict__ order,
                                                               const unsigned* __restrict__ start, int windows, size_t total_buckets, jac_t* __restrict__ buckets) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total_buckets) return;
  // neighbouring lanes take the SAME (window, bucket) of neighbouring rows (round 6): the bucket's entry list is shared by every row, so a wave runs one
  // trip count - with the lanes of a wave on 64 buckets of one row it ran the longest of 64 lists (~2 entries on average, 6-7 the longest) - and a wave
  // whose bucket is empty leaves at once. The result keeps its (row, window, bucket) place.
  const size_t per_row = (size_t)windows * MSM_BUCKETS, rows = total_buckets / per_row;
  const size_t row = gid % rows, wk = gid / rows;
  const size_t b = row * per_row + wk;
  const size_t w = wk / MSM_BUCKETS, k = wk % MSM_BUCKETS;
  const unsigned lo = start[w * (MSM_BUCKETS + 1) + k], hi = start[w * (MSM_BUCKETS + 1) + k + 1];
  const aff_t* rb = bases + row * n;
  xyzz_t acc = xyzz_identity();
  for (unsigned p = lo; p < hi; ++p) {
    const unsigned e = order[w * n + p];
    aff_t q = rb[e & 0x7fffffffu];
    if (e & 0x80000000u) q = aff_neg(q);
    acc = xyzz_add_mixed(acc, q);
  }
  buckets[b] = xyzz_to_jac(acc);
}

// Window sums for the batched path, work-efficient form: 8 adjacent lanes per (row, window); lane s runs the classical running sum over its 16
// buckets (acc_s = sum_i (i+1) B_{16s+i}, run_s = sum_i B_{16s+i}: 32 additions), then W = sum_s acc_s + 16 * sum_s s * run_s.
// ~280 additions per window instead of the 128 x 14 of the scan form (which is the right shape only when a single MSM must finish fast).
__global__ void __launch_bounds__(256) k_msm_window_reduce_seg(const jac_t* __restrict__ buckets, size_t nwin, jac_t* __restrict__ window_sums) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t win = gid >> 3;
  const int seg = (int)(gid & 7);
  xyzz_t run = xyzz_identity(), acc = xyzz_identity();
  if (win < nwin) {
    const jac_t* b = buckets + win * MSM_BUCKETS + seg * 16;
    for (int i = 15; i >= 0; --i) {
      run = xyzz_add(run, xyzz_from_jac(b[i]));
      acc = xyzz_add(acc, run);
    }
  }
  // X = sum_s acc_s (3-level tree over the 8 lanes); Y = sum_s s * run_s by a running sum in lane 0 of the group
  xyzz_t x = acc;
#pragma unroll
  for (int d = 4; d >= 1; d >>= 1) {
    xyzz_t o = shfl_down_xyzz(x, d);
    if (seg < d) x = xyzz_add(x, o);
  }
  xyzz_t rr = xyzz_identity(), y = xyzz_identity();
  for (int s = 7; s >= 1; --s) {
    xyzz_t rs = shfl_down_xyzz(run, s);  // lane 0 of the group receives run_s
    if (seg == 0) {
      rr = xyzz_add(rr, rs);
      y = xyzz_add(y, rr);
    }
  }
  if (win < nwin && seg == 0) {
    jac_t yj = xyzz_to_jac(y);
    for (int k = 0; k < 4; ++k) yj = jac_dbl(yj);
    window_sums[win] = jac_add(xyzz_to_jac(x), yj);
  }
}

// Window Horner on the device for batches (one lane per row): acc = 2^8 acc + W_w, high to low (msm.rs:150-175). A single MSM
// leaves this 256-doubling chain to the host; with hundreds of rows the lanes run it in parallel.
__global__ void __launch_bounds__(64) k_msm_horner_rows(const jac_t* __restrict__ window_sums, int windows, size_t rows, jac_t* __restrict__ out) {
  const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  jac_t acc = jac_identity();
  for (int w = windows - 1; w >= 0; --w) {
    for (int k = 0; k < MSM_C; ++k) acc = jac_dbl(acc);
    acc = jac_add(acc, window_sums[row * windows + w]);
  }
  out[row] = acc;
}

// ---- vartime_scalar_mul (src/provider/msm.rs:779-867): width-5 wNAF, one lane per point, the digits of the ONE scalar by value ---------------
struct WnafArgs {
  signed char d[260];
  int len;
};
__global__ void __launch_bounds__(64) k_wnaf_rows(const aff_t* __restrict__ pts, size_t n, WnafArgs w, jac_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aff_t p = pts[i];
  // odd multiples P, 3P, ..., 31P (msm.rs:788-794), kept affine-free: Jacobian table in registers/scratch
  jac_t tab[16];
  tab[0] = jac_from_affine(p);
  const jac_t dbl = jac_dbl(tab[0]);
  for (int k = 1; k < 16; ++k) tab[k] = jac_add(tab[k - 1], dbl);
  jac_t acc = jac_identity();
  bool started = false;
  for (int k = w.len - 1; k >= 0; --k) {
    if (started) acc = jac_dbl(acc);
    const int d = w.d[k];
    if (d > 0) {
      started = true;
      acc = jac_add(acc, tab[(d - 1) / 2]);
    } else if (d < 0) {
      started = true;
      jac_t q = tab[(-d - 1) / 2];
      q.y = fe_neg<B>(q.y);
      acc = jac_add(acc, q);
    }
  }
  out[i] = acc;
}

// ---- K14: R1CSWitness::fold_multiple (src/r1cs/mod.rs:570-660): out[j] = sum_i w[i] * Ws[i][j] ---------------------------------
// The small-value fast path of the reference (skip zeros, add w_i for ones, :615-631) is kept: SHA witnesses are bits.
__global__ void __launch_bounds__(256) k_fold_tables(const fe_t* const* __restrict__ tables, const fe_t* __restrict__ weights, size_t n, size_t len,
                                                     fe_t* __restrict__ out) {
  const fe_t one = fe_one<SF>();
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < len; j += (size_t)gridDim.x * blockDim.x) {
    fe_t acc = fe_zero();
    for (size_t i = 0; i < n; ++i) {
      const fe_t x = tables[i][j];
      if (fe_is_zero(x)) continue;
      acc = fe_add<SF>(acc, fe_eq(x, one) ? weights[i] : fe_mul<SF>(weights[i], x));
    }
    out[j] = acc;
  }
}

// ---- K10: binary rows ------------------------------------------------------------------------------------------------
// One block per Hyrax row: sum of bases[j] where the canonical scalar of column j equals 1.
__global__ void __launch_bounds__(256) k_msm_binary_rows(const fe_t* __restrict__ canon, size_t n, size_t cols, const aff_t* __restrict__ bases,
                                                         const unsigned* __restrict__ row_flags, jac_t* __restrict__ out) {
  __shared__ xyzz_t s[256];
  const size_t row = blockIdx.x;
  const size_t lo = row * cols, hi = (lo + cols < n) ? lo + cols : n;
  xyzz_t acc = xyzz_identity();
  if (row_flags[row] == 1u) {  // only rows that really are 0/1 valued; others are handled by the digit path
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
      if (canon[i].v[0] & 1u) acc = xyzz_add_mixed(acc, bases[i - lo]);
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) s[threadIdx.x] = xyzz_add(s[threadIdx.x], s[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[row] = xyzz_to_jac(s[0]);
}

// (K12, FixedBaseMul::precompute for one and for many bases, Curve::batch_normalize: kernels_bulk.hpp / capi_bulk.hip - throughput kernels, compiled without
// this translation unit's max-ilp scheduling, under which they spill)
// One 32-lane half-wave per scalar: lane j adds table[j][byte j], then a 5-level tree.
// `tables` holds ntables consecutive 32*255-entry tables; scalar idx uses table idx % ntables.
__global__ void __launch_bounds__(256) k_fixed_base_rows(const fe_t* __restrict__ scalars, size_t n, const aff_t* __restrict__ tables, size_t ntables,
                                                         jac_t* __restrict__ out) {
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t idx = gid >> 5;
  const int j = (int)(gid & 31);
  xyzz_t acc = xyzz_identity();
  if (idx < n) {
    const fe_t c = fe_to_canonical<SF>(scalars[idx]);
    const unsigned digit = (c.v[j >> 2] >> (8 * (j & 3))) & 0xffu;
    if (digit) acc = xyzz_from_affine(tables[(idx % ntables) * (32 * 255) + (size_t)j * 255 + digit - 1]);
  }
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {
    xyzz_t o = shfl_down_xyzz(acc, d);
    if (j < d) acc = xyzz_add(acc, o);
  }
  if (idx < n && j == 0) out[idx] = xyzz_to_jac(acc);
}

// Row sums for the narrow-key commit (hyrax_pc.rs:221-260: one commitment = the sum of `per` table walks): one 64-lane wave per row adds the row's
// `per` <= 65 points with a shuffle tree (6 levels instead of `per` sequential host additions per row).
__global__ void __launch_bounds__(256) k_sum_rows_of_points(const jac_t* __restrict__ pts, size_t rows, unsigned per, jac_t* __restrict__ out) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned lane = threadIdx.x & 63;
  xyzz_t acc = xyzz_identity();
  if (row < rows)
    for (unsigned k = lane; k < per; k += 64) acc = xyzz_add(acc, xyzz_from_jac(pts[row * per + k]));  // per <= 65: at most two per lane
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    xyzz_t o = shfl_down_xyzz(acc, d);
    if (lane < (unsigned)d) acc = xyzz_add(acc, o);
  }
  if (row < rows && lane == 0) out[row] = xyzz_to_jac(acc);
}

// The same with the block-cooperative addition: four scalars per 512-thread block, their 4 x 32 table entries are the 128 items; the five tree
// levels cost one cooperative addition each instead of ~15 us (the chain of dependent additions is all there is: 84 scalars do not fill the chip either way).
__global__ void __launch_bounds__(4 * 128) k_fixed_base_rows_coop(const fe_t* __restrict__ scalars, size_t n, const aff_t* __restrict__ tables, size_t ntables,
                                                                  jac_t* __restrict__ out) {
  __shared__ CoopAdd<128> L;
  __shared__ xyzz_t s[128];
  const int wave = threadIdx.x >> 6, blk = wave >> 2;
  const int role = (wave + 2 * blk) & 3, k = blk * 64 + (threadIdx.x & 63);  // roles of an item block on four different SIMDs (see k_msm_window_reduce_coop)
  const size_t idx = (size_t)blockIdx.x * 4 + (k >> 5);
  const int j = k & 31;
  if (role == 0) {
    xyzz_t acc = xyzz_identity();
    if (idx < n) {
      const fe_t c = fe_to_canonical<SF>(scalars[idx]);
      const unsigned digit = (c.v[j >> 2] >> (8 * (j & 3))) & 0xffu;
      if (digit) acc = xyzz_from_affine(tables[(idx % ntables) * (32 * 255) + (size_t)j * 255 + digit - 1]);
    }
    s[k] = acc;
  }
  __syncthreads();
  for (int off = 16; off >= 1; off >>= 1) {  // in place is safe: results are stored after the last level of the addition
    const bool active = j < off;
    xyzz_add_block4<128>(L, &s[k], &s[active ? k + off : k], s, role, k, active);
  }
  if (role == 0 && j == 0 && idx < n) out[idx] = xyzz_to_jac(s[k]);
}

// k_fixed_base_rows_coop with its inputs and outputs in mapped host memory (<= 128 scalars: a round commitment of the ZK verifier circuit, the blinds of
// the zero rows): the scalars are read through the bus, scalar i's result goes into slot i (the point, then the tag: sequence, sequence + plain sum,
// sequence * K + position-weighted sum, sequence) which the host polls - no copy launch on either side and no
// stream synchronise around a 50 us kernel.
// ITEMS = 64: two scalars per 256-thread block, ONE wave per role, so each of the four products of a level has a SIMD to itself. The products are
// issue-bound (v_mad_u64_u32 at quarter rate: ~136 of them per Montgomery product whatever the number of active lanes), so the 128-item form - two
// waves per SIMD, both issuing full-length products even when a tree level has two active lanes left - takes twice as long per level.
// WBITS = 16: tables of k_fixed_base_tables16 - 16 entries per scalar, four tree levels.
// XYZZ_OUT: the result slot carries (X, Y, ZZ, ZZZ) as they are (32 words) instead of the Jacobian form (24) - the caller that adds the points on the
// host anyway (a commitment = the sum of its scalars' walks) saves the two conversion products per scalar here and two products per addition there.
// Slots are 48 words apart with the tag at words 32..35 in either form.
constexpr int FB_SLOT_WORDS = 48, FB_SLOT_TAG = 32;
template <int ITEMS, int WBITS, bool XYZZ_OUT>
__global__ void __launch_bounds__(4 * ITEMS) k_fixed_base_rows_coop_mapped(const fe_t* __restrict__ scalars, size_t n, const aff_t* __restrict__ tables, size_t ntables,
                                                                           unsigned* __restrict__ slots, unsigned seq) {
  constexpr int PER = 256 / WBITS;  // table entries (= tree leaves) per scalar
  constexpr size_t WIN = ((size_t)1 << WBITS) - 1;
  __shared__ CoopAdd<ITEMS> L;
  __shared__ xyzz_t s[ITEMS];
  const int wave = threadIdx.x >> 6, blk = wave >> 2;
  const int role = (wave + 2 * blk) & 3, k = blk * 64 + (threadIdx.x & 63);
  const size_t idx = (size_t)blockIdx.x * (ITEMS / PER) + (k / PER);
  const int j = k % PER;
  SP_STAMP(0);
#ifdef SP_KERNEL_STAMPS
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) sp_hwid[threadIdx.x >> 6] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
  {
    const unsigned long long w0 = wall_clock64();
    while (wall_clock64() - w0 < sp_predelay) __builtin_amdgcn_s_sleep(8);
  }
  SP_STAMP(0);
#endif
  if (role == 0) {
    xyzz_t acc = xyzz_identity();
    if (idx < n) {
      const fe_t c = fe_to_canonical<SF>(scalars[idx]);
      SP_STAMP(1);
      const unsigned digit = WBITS == 8 ? (c.v[j >> 2] >> (8 * (j & 3))) & 0xffu : (c.v[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      if (digit) acc = xyzz_from_affine(tables[(idx % ntables) * (PER * WIN) + (size_t)j * WIN + digit - 1]);
    }
    s[k] = acc;
  }
  __syncthreads();
  SP_STAMP(2);
  int lvl = 0;
  for (int off = PER / 2; off >= 1; off >>= 1) {
    const bool active = j < off;
    xyzz_add_block4<ITEMS>(L, &s[k], &s[active ? k + off : k], s, role, k, active);
    ++lvl;
    SP_STAMP(2 + lvl);
  }
  (void)lvl;
  if (role == 0 && j == 0 && idx < n) {
    constexpr int D = XYZZ_OUT ? 32 : 24;
    unsigned rw[32];
    if (XYZZ_OUT) {
      const xyzz_t r = s[k];
      const unsigned* p = reinterpret_cast<const unsigned*>(&r);
#pragma unroll
      for (int w = 0; w < 32; ++w) rw[w] = p[w];
    } else {
      const jac_t r = xyzz_to_jac(s[k]);
      const unsigned* p = reinterpret_cast<const unsigned*>(&r);
#pragma unroll
      for (int w = 0; w < 24; ++w) rw[w] = p[w];
    }
    unsigned* slot = slots + idx * FB_SLOT_WORDS;
    unsigned a = seq, b = seq * 0x9E3779B1u;
#pragma unroll
    for (int w = 0; w < D; ++w) {
      slot[w] = rw[w];
      a += rw[w];
      b += (unsigned)(w + 1) * rw[w];
    }
    const unsigned long long lo = ((unsigned long long)a << 32) | seq, hi = ((unsigned long long)seq << 32) | b;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + FB_SLOT_TAG + 2), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + FB_SLOT_TAG), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    SP_STAMP(20);
  }
}

// FixedBaseMul::multi_mul (msm.rs:727-773) over per-base tables, sum_i s_i P_i, in ONE launch with no copy on either side and no host-side tail — the
// latency form behind sp_fbtables_multi_mul. Its call site is comm_LZ of the Hyrax opening (hyrax_pc.rs:387-478): the commitment of L^T W equals
// sum_i L_i * comm_W[i], the rows of comm_W are known when the witness is prepared, so their window tables are built there and the 512-point MSM
// (digits, sort, bucket sums, a 14-level window reduction, 256 doublings on the host) that used to follow the last row challenge becomes 14 levels of
// additions. The scalars are read straight from mapped host memory; block b owns scalars 4b .. 4b+3 (128 table entries, seven cooperative levels down
// to one point); the last block to take a ticket adds the block sums (<= 512 of them) the same way and publishes the Jacobian sum into a
// self-validating slot in mapped host memory that the host polls: words 0..23 the point, 24 = sequence, 25 = sequence + plain sum, 26 = sequence * K +
// position-weighted sum, 27 = sequence again (the order in which the stores land does not matter; the host accepts the slot only when all four agree).
constexpr unsigned MULTI_MUL_SLOT_K = 0x9E3779B1u;
constexpr int MULTI_MUL_MAX_BLOCKS = 1024;  // 4096 scalars (the 2048 bases of a key + h: every MSM over the key as one table walk)
// `has_last`: scalar n - 1 is the kernel argument `last` instead of scalars[n - 1] (the blind of h, known on the host, behind n - 1 scalars that a
// kernel earlier on the stream left in device memory: comm_LZ = <LZ, ck> + r_LZ h of hyrax_pc.rs:454-455 without a host round trip for LZ).
// `has_last` == 2 (EXPAND): the scalars are eq(r_1 .. r_k, .) one level short - `scalars` holds P = eq(r_1 .. r_(k-1), .) (ceil((n - 1) / 2) entries, the
// new variable is the index LSB, eq.rs:66-76) followed by S0, S1, and `last` is r_k: scalar idx < n - 1 is P[idx / 2] r_k or P[idx / 2] (1 - r_k), scalar
// n - 1 is S0 + r_k (S1 - S0). One product per scalar here instead of 2^(k-1) on the host between the challenge and the launch (comm_LZ's walk, the chain
// the end of a prove waits for, starts ~8 us earlier).
// `gblocks` > 0 (GROUPS): every `gblocks` consecutive blocks form a group with a ticket and a result slot of its own (slot + 32 g words); the host adds
// the groups' sums. A cooperative addition is ~4.8 us of dependent products on the device and ~0.5 us on the host, so the top levels of the tree are
// the host's: with 8 groups over 107 blocks the join is 4 levels instead of 7 (-14 us) for 7 host additions (+3.5 us).
__global__ void __launch_bounds__(4 * 128) k_multi_mul_coop(const fe_t* __restrict__ scalars, size_t n, const aff_t* __restrict__ tables, xyzz_t* __restrict__ partial,
                                                            unsigned* __restrict__ ticket, unsigned* __restrict__ slot, unsigned seq, fe_t last, int has_last,
                                                            unsigned gblocks) {
  __shared__ CoopAdd<128> L;
  __shared__ xyzz_t s[128], s2[128];
  __shared__ unsigned s_last;
  const int wave = threadIdx.x >> 6, blk = wave >> 2;
  const int role = (wave + 2 * blk) & 3, k = blk * 64 + (threadIdx.x & 63);  // roles of an item block on four different SIMDs (see k_msm_window_reduce_coop)
  const size_t idx = (size_t)blockIdx.x * 4 + (k >> 5);
  const int j = k & 31;
  if (role == 0) {
    xyzz_t acc = xyzz_identity();
    if (idx < n) {
      fe_t sc;
      if (has_last == 2) {
        const size_t np = n / 2;  // = ceil((n - 1) / 2)
        if (idx == n - 1) {
          const fe_t s0 = scalars[np], s1 = scalars[np + 1];
          sc = fe_add<SF>(s0, fe_mul<SF>(last, fe_sub<SF>(s1, s0)));
        } else {
          const fe_t base = scalars[idx >> 1], hi = fe_mul<SF>(base, last);
          sc = (idx & 1) ? hi : fe_sub<SF>(base, hi);
        }
      } else {
        sc = (has_last && idx == n - 1) ? last : scalars[idx];
      }
      const fe_t c = fe_to_canonical<SF>(sc);
      const unsigned digit = (c.v[j >> 2] >> (8 * (j & 3))) & 0xffu;
      if (digit) acc = xyzz_from_affine(tables[idx * (32 * 255) + (size_t)j * 255 + digit - 1]);
    }
    s[k] = acc;
  }
  __syncthreads();
  for (int off = 64; off >= 1; off >>= 1) {  // in place is safe: results are stored after the last level of the addition
    const bool active = k < off;
    xyzz_add_block4<128>(L, &s[k], &s[active ? k + off : k], s, role, k, active);
  }
  // the group this block joins: all blocks (gblocks == 0) or blocks [g0, g0 + nbg)
  const unsigned grp = gblocks ? blockIdx.x / gblocks : 0u, g0 = grp * gblocks;
  const unsigned nbg = gblocks ? (gridDim.x - g0 < gblocks ? gridDim.x - g0 : gblocks) : gridDim.x;
  if (gblocks) {
    ticket += grp;
    slot += 32 * grp;
    partial += g0;
  }
  if (nbg > 1) {
    if (threadIdx.x == 0) {
      partial[gblocks ? blockIdx.x - g0 : blockIdx.x] = s[0];
      __threadfence();  // the block sum is visible device-wide before the ticket is taken
      s_last = atomicAdd(ticket, 1u) == nbg - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (role == 0) {
      xyzz_t acc = xyzz_identity();
      if (k < (int)nbg) {  // other blocks' sums: loads that cannot be served from a stale line of this XCD's caches
        const unsigned* pw = reinterpret_cast<const unsigned*>(&partial[k]);
        unsigned* aw = reinterpret_cast<unsigned*>(&acc);
#pragma unroll
        for (int w = 0; w < (int)(sizeof(xyzz_t) / 4); ++w) aw[w] = __hip_atomic_load(pw + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s[k] = acc;
    }
    __syncthreads();
    // more than 128 block sums: item k first takes in sums k + 128, k + 256, ... (one cooperative addition each), then the tree
    for (int base = 128; base < (int)nbg; base += 128) {
      const bool active = base + k < (int)nbg;
      if (role == 0 && active) {
        xyzz_t acc;
        const unsigned* pw = reinterpret_cast<const unsigned*>(&partial[base + k]);
        unsigned* aw = reinterpret_cast<unsigned*>(&acc);
#pragma unroll
        for (int w = 0; w < (int)(sizeof(xyzz_t) / 4); ++w) aw[w] = __hip_atomic_load(pw + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s2[k] = acc;
      }
      __syncthreads();
      xyzz_add_block4<128>(L, &s[k], &s2[k], s, role, k, active);
    }
    int top = 1;
    const int live = (int)nbg < 128 ? (int)nbg : 128;
    while (2 * top < live) top <<= 1;
    for (int off = top; off >= 1; off >>= 1) {
      const bool active = k < off;
      xyzz_add_block4<128>(L, &s[k], &s[active ? k + off : k], s, role, k, active);
    }
    if (threadIdx.x == 0) atomicExch(ticket, 0u);  // ready for the next launch (stream order makes it visible)
  }
  if (threadIdx.x == 0) {
    const jac_t r = xyzz_to_jac(s[0]);
    const unsigned* rw = reinterpret_cast<const unsigned*>(&r);
    unsigned a = seq, b = seq * MULTI_MUL_SLOT_K;
#pragma unroll
    for (int w = 0; w < 24; ++w) {
      slot[w] = rw[w];
      a += rw[w];
      b += (unsigned)(w + 1) * rw[w];
    }
    const unsigned long long lo = ((unsigned long long)a << 32) | seq, hi = ((unsigned long long)seq << 32) | b;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + 26), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + 24), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The same sum for MANY scalars (the 2048 bases of a key + h: delta and comm_LZ of the opening, 65 568 table entries): 1024 (scalar, digit) items per
// 512-thread block instead of 128, so a walk over the whole key is 65 blocks — two of them run side by side on a quarter of the chip — and the
// last-block join adds 65 sums instead of 513. Per block: every thread takes TWO entries (one mixed addition on its own lane: all eight waves busy),
// lanes 32..63 of each wave hand their sum to lanes 0..31 through shuffles (one full addition per lane), and the 256 sums left go down the
// block-cooperative tree of k_multi_mul_coop (eight levels). Same ticket / slot protocol, same result.
constexpr int MULTI_MUL_WIDE_ITEMS = 1024;
__global__ void __launch_bounds__(4 * 128) k_multi_mul_wide(const fe_t* __restrict__ scalars, size_t n, const aff_t* __restrict__ tables, xyzz_t* __restrict__ partial,
                                                            unsigned* __restrict__ ticket, unsigned* __restrict__ slot, unsigned seq, fe_t last, int has_last, int raw64) {
  // raw64: `scalars` points at 64-byte uniform blocks, one per scalar, reduced here as from_uniform (src/provider/traits.rs:275-280) - the IPA's mask
  // vector straight from the randomness stream, so that delta's walk can start before the host has drawn a single element
  __shared__ CoopAdd<128> L;
  __shared__ xyzz_t s[256];
  __shared__ unsigned s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, blk = wave >> 2;
  {
    const size_t e0 = (size_t)blockIdx.x * MULTI_MUL_WIDE_ITEMS + 2 * (size_t)threadIdx.x, idx = e0 >> 5;
    const int j = (int)(e0 & 31);
    xyzz_t acc = xyzz_identity();
    if (idx < n) {
      fe_t sc;
      if (has_last && idx == n - 1) sc = last;
      else if (raw64) sc = fe_from_uniform<SF>(reinterpret_cast<const uint8_t*>(scalars) + 64 * idx);
      else sc = scalars[idx];
      const fe_t c = fe_to_canonical<SF>(sc);
      const unsigned d0 = (c.v[j >> 2] >> (8 * (j & 3))) & 0xffu, d1 = (c.v[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xffu;
      const aff_t* tab = tables + idx * (32 * 255);
      if (d0) acc = xyzz_from_affine(tab[(size_t)j * 255 + d0 - 1]);
      if (d1) acc = xyzz_add_mixed(acc, tab[(size_t)(j + 1) * 255 + d1 - 1]);
    }
    const xyzz_t other = shfl_down_xyzz(acc, 32);
    if (lane < 32) s[wave * 32 + lane] = xyzz_add(acc, other);
  }
  __syncthreads();
  const int role = (wave + 2 * blk) & 3, k = blk * 64 + lane;  // roles of an item block on four different SIMDs (see k_msm_window_reduce_coop)
  xyzz_add_block4<128>(L, &s[k], &s[k + 128], s, role, k, true);
  for (int off = 64; off >= 1; off >>= 1) {
    const bool active = k < off;
    xyzz_add_block4<128>(L, &s[k], &s[active ? k + off : k], s, role, k, active);
  }
  if (gridDim.x > 1) {
    if (threadIdx.x == 0) {
      partial[blockIdx.x] = s[0];
      __threadfence();  // the block sum is visible device-wide before the ticket is taken
      s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (role == 0) {
      xyzz_t acc = xyzz_identity();
      if (k < (int)gridDim.x) {  // other blocks' sums: loads that cannot be served from a stale line of this XCD's caches
        const unsigned* pw = reinterpret_cast<const unsigned*>(&partial[k]);
        unsigned* aw = reinterpret_cast<unsigned*>(&acc);
#pragma unroll
        for (int w = 0; w < (int)(sizeof(xyzz_t) / 4); ++w) aw[w] = __hip_atomic_load(pw + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s[k] = acc;
    }
    __syncthreads();
    int top = 1;
    const int live = (int)gridDim.x < 128 ? (int)gridDim.x : 128;  // (the launcher keeps the grid within 128 blocks = 4096 scalars)
    while (2 * top < live) top <<= 1;
    for (int off = top; off >= 1; off >>= 1) {
      const bool active = k < off;
      xyzz_add_block4<128>(L, &s[k], &s[active ? k + off : k], s, role, k, active);
    }
    if (threadIdx.x == 0) atomicExch(ticket, 0u);  // ready for the next launch (stream order makes it visible)
  }
  if (threadIdx.x == 0) {
    const jac_t r = xyzz_to_jac(s[0]);
    const unsigned* rw = reinterpret_cast<const unsigned*>(&r);
    unsigned a = seq, b = seq * MULTI_MUL_SLOT_K;
#pragma unroll
    for (int w = 0; w < 24; ++w) {
      slot[w] = rw[w];
      a += rw[w];
      b += (unsigned)(w + 1) * rw[w];
    }
    const unsigned long long lo = ((unsigned long long)a << 32) | seq, hi = ((unsigned long long)seq << 32) | b;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + 26), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot + 24), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// (K9, bind_with_delayed: kernels_bulk.hpp)

}  // namespace spk
