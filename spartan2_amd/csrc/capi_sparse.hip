// libspartan_hip.so — R1CS sparse kernels and entry points of include/spartan_hip.h.
//   k_spmv3     K5/K6  PrecomputedSparseMatrix::multiply_vec (src/r1cs/sparse.rs:194-233) for A, B, C in one launch, and
//                      multiply_vec_incremental_into (src/r1cs/mod.rs:1170-1211) = cached + filtered rows (sparse.rs:305-380)
//   k_polyabc*  K7     bind_and_prepare_poly_ABC / accumulate_rows (src/r1cs/mod.rs:1235-1398) as a column-major GATHER
//                      (no 256-bit atomics): short columns one lane each, long columns (the constant-1 column of the
//                      booleanity rows) one block each.
// Entries keep the reference's classes: +-1 and |k| in 2..7 as an int8 code (add / sub / double-add chains, sparse.rs:137-155),
// everything else as a full field coefficient.
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <vector>

#include "core.hpp"
#include "device_utils.hpp"

using sp::fail;
typedef FqP S;

namespace spk {

struct SplitDev {          // major-ordered sparse structure, entries split by coefficient class
  const unsigned* sptr;    // [n_major + 1] small-class offsets
  const unsigned* sidx;    // minor index of each small-class entry
  const signed char* scode;  // +-1 .. +-7
  const unsigned* gptr;    // [n_major + 1] general-class offsets
  const unsigned* gidx;
  const fe_t* gval;
};

__device__ __forceinline__ fe_t small_mul(int code, const fe_t& x) {  // sparse.rs:137-155
  const int a = code < 0 ? -code : code;
  fe_t r;
  switch (a) {
    case 1: r = x; break;
    case 2: r = fe_dbl<S>(x); break;
    case 3: r = fe_add<S>(fe_dbl<S>(x), x); break;
    case 4: r = fe_dbl<S>(fe_dbl<S>(x)); break;
    case 5: r = fe_add<S>(fe_dbl<S>(fe_dbl<S>(x)), x); break;
    case 6: { fe_t d = fe_dbl<S>(x); r = fe_add<S>(fe_dbl<S>(d), d); break; }
    default: { fe_t d = fe_dbl<S>(x); r = fe_add<S>(fe_add<S>(fe_dbl<S>(d), d), x); break; }
  }
  return r;
}
__device__ __forceinline__ fe_t acc_small(const fe_t& acc, int code, const fe_t& x) {
  if (code == 1) return fe_add<S>(acc, x);
  if (code == -1) return fe_sub<S>(acc, x);
  fe_t m = small_mul(code, x);
  return code < 0 ? fe_sub<S>(acc, m) : fe_add<S>(acc, m);
}
// sum over the entries of one major index, strided (first, step) so a block can share a long list
__device__ __forceinline__ fe_t gather_major(const SplitDev& m, size_t major, const fe_t* __restrict__ x, unsigned first, unsigned step) {
  fe_t acc = fe_zero();
  for (unsigned k = m.sptr[major] + first, e = m.sptr[major + 1]; k < e; k += step) acc = acc_small(acc, m.scode[k], x[m.sidx[k]]);
  for (unsigned k = m.gptr[major] + first, e = m.gptr[major + 1]; k < e; k += step) acc = fe_add<S>(acc, fe_mul<S>(m.gval[k], x[m.gidx[k]]));
  return acc;
}

// The same sum with FOUR entries in flight: index / code loads first, then the four gathers, then the four accumulations. One lane walking a
// column otherwise pays two dependent memory latencies (index, then x[index]) per entry, and the column kernels are bound by exactly that chain
// (k_polyabc_short: 82 us for 5 M entries at config 2, nowhere near a bandwidth limit).
__device__ __forceinline__ fe_t gather_major_x4(const SplitDev& m, size_t major, const fe_t* __restrict__ x) {
  fe_t acc = fe_zero();
  unsigned k = m.sptr[major];
  const unsigned e = m.sptr[major + 1];
  for (; k + 4 <= e; k += 4) {
    const unsigned i0 = m.sidx[k], i1 = m.sidx[k + 1], i2 = m.sidx[k + 2], i3 = m.sidx[k + 3];
    const int c0 = m.scode[k], c1 = m.scode[k + 1], c2 = m.scode[k + 2], c3 = m.scode[k + 3];
    const fe_t x0 = x[i0], x1 = x[i1], x2 = x[i2], x3 = x[i3];
    acc = acc_small(acc, c0, x0);
    acc = acc_small(acc, c1, x1);
    acc = acc_small(acc, c2, x2);
    acc = acc_small(acc, c3, x3);
  }
  if (k + 2 <= e) {
    const unsigned i0 = m.sidx[k], i1 = m.sidx[k + 1];
    const int c0 = m.scode[k], c1 = m.scode[k + 1];
    const fe_t x0 = x[i0], x1 = x[i1];
    acc = acc_small(acc, c0, x0);
    acc = acc_small(acc, c1, x1);
    k += 2;
  }
  if (k < e) acc = acc_small(acc, m.scode[k], x[m.sidx[k]]);
  for (unsigned g = m.gptr[major], ge = m.gptr[major + 1]; g < ge; ++g) acc = fe_add<S>(acc, fe_mul<S>(m.gval[g], x[m.gidx[g]]));
  return acc;
}

struct Spmv3Args {
  SplitDev m[3];
  const fe_t* base[3];  // cached products to add (nullptr for a plain multiply_vec)
  fe_t* out[3];
};
__global__ void __launch_bounds__(256) k_spmv3(Spmv3Args a, const fe_t* __restrict__ z, size_t nrows) {
  const int which = blockIdx.y;
  const SplitDev m = a.m[which];
  const fe_t* base = a.base[which];
  fe_t* out = a.out[which];
  for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += (size_t)gridDim.x * blockDim.x) {
    fe_t acc = gather_major(m, row, z, 0, 1);
    if (base) acc = fe_add<S>(acc, base[row]);
    out[row] = acc;
  }
}

// The tau-independent halves of the outer sum-check's FIRST evaluation (evaluation_points_cubic_with_three_inputs, src/sumcheck.rs:1041-1105), formed
// right behind the matrix-vector product while both run in the shadow of commit_zeros: P0[i] = A0 B0 - C0, P1[i] = (A1 - A0)(B1 - B0) for the pair
// (i, i + N/2) the top variable joins (src/polys/multilinear.rs:101). The first evaluation on the critical path then reads 2 x 16 MiB instead of
// 84 MiB. (A fused form — one thread computing both rows of all three products — was measured: its six row walks per thread cost more than this
// second streaming pass, 116 us against 50 + 25 us.)
__global__ void __launch_bounds__(256) k_round0_products(const fe_t* __restrict__ A, const fe_t* __restrict__ B, const fe_t* __restrict__ C, size_t half,
                                                         fe_t* __restrict__ p0, fe_t* __restrict__ p1) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
    const fe_t a0 = A[i], a1 = A[i + half], b0 = B[i], b1 = B[i + half], c0 = C[i];
    p0[i] = fe_sub<S>(fe_mul<S>(a0, b0), c0);
    p1[i] = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  }
}

struct PolyAbcArgs {
  SplitDev m[3];  // column-major A, B, C
  fe_t r, r2;
};
constexpr unsigned LONG_COLUMN = 512;
__device__ __forceinline__ unsigned col_len(const PolyAbcArgs& a, size_t col) {
  unsigned n = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) n += (a.m[i].sptr[col + 1] - a.m[i].sptr[col]) + (a.m[i].gptr[col + 1] - a.m[i].gptr[col]);
  return n;
}
// `order` lists the short columns by decreasing entry count, so the 64 columns of a wave have (nearly) the same length: a wave costs its longest
// column, and SHA circuits mix 1-entry columns with columns of dozens of entries.
__global__ void __launch_bounds__(256) k_polyabc_short(PolyAbcArgs a, const fe_t* __restrict__ rx, const unsigned* __restrict__ order, size_t n_short,
                                                       fe_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_short; i += (size_t)gridDim.x * blockDim.x) {
    const size_t col = order[i];
    fe_t sa = gather_major_x4(a.m[0], col, rx), sb = gather_major_x4(a.m[1], col, rx), sc = gather_major_x4(a.m[2], col, rx);
    // two thirds of a SHA circuit's columns have no entry in B: with the columns grouped by their entry counts that is uniform across a wave, and the
    // skipped product is a third of this kernel's multiplications (the per-column epilogue outweighs the 1.05 M general-coefficient products)
    if (!fe_is_zero(sb)) sa = fe_add<S>(sa, fe_mul<S>(a.r, sb));
    if (!fe_is_zero(sc)) sa = fe_add<S>(sa, fe_mul<S>(a.r2, sc));
    out[col] = sa;
  }
}
// Long columns (the constant-1 column has ~one entry per booleanity row): NB blocks share a column, each striding
// over its entry lists; a second one-block pass per column adds the NB partial triples and applies (1, r, r^2).
constexpr unsigned LONG_NB_MAX = 128;
__device__ __forceinline__ unsigned long_nb(unsigned len) {
  unsigned nb = (len + 2047) / 2048;
  return nb < 1 ? 1 : (nb > LONG_NB_MAX ? LONG_NB_MAX : nb);
}
__global__ void __launch_bounds__(256) k_polyabc_long(PolyAbcArgs a, const fe_t* __restrict__ rx, const unsigned* __restrict__ long_cols,
                                                      fe_t* __restrict__ partials) {
  __shared__ fe_t smem[3 * 4];
  const size_t col = long_cols[blockIdx.y];
  const unsigned nb = long_nb(col_len(a, col));
  if (blockIdx.x >= nb) return;
  fe_t acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = gather_major(a.m[i], col, rx, blockIdx.x * blockDim.x + threadIdx.x, nb * blockDim.x);
  block_sum<3>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) partials[((size_t)blockIdx.y * LONG_NB_MAX + blockIdx.x) * 3 + i] = acc[i];
  }
}
// The three kernels above in ONE launch: blocks [0, LONG_NB_MAX * n_long) are the long columns' (they are dispatched first and run under the short
// columns' blocks instead of 17-24 us behind them on the path of the inner sum-check's first round), the rest walk the short columns. The last block
// of a long column to arrive (one counter per column, <= 128 arrivals, partial triples stored write-through and read back past this XCD's L2) adds the
// column's partials and applies (1, r, r^2) - k_polyabc_long_final's work, also under the short columns.
__global__ void __launch_bounds__(256) k_polyabc_short_and_long(PolyAbcArgs a, const fe_t* __restrict__ rx, const unsigned* __restrict__ order, size_t n_short,
                                                                fe_t* __restrict__ out, const unsigned* __restrict__ long_cols, unsigned n_long,
                                                                fe_t* __restrict__ partials, unsigned* __restrict__ tickets, unsigned short_blocks,
                                                                size_t zero_from, size_t zero_n) {
  __shared__ fe_t smem[3 * 4];
  __shared__ unsigned s_last;
  const unsigned long_blocks = LONG_NB_MAX * n_long;
  if (blockIdx.x >= long_blocks + short_blocks) {
    // the zero tail of the output (out_len > num_cols: 32 MB at config 2, a 7 us fill launch in front of this kernel until round 3): the last blocks
    // of the grid stream it out under the column walks
    uint4* z = reinterpret_cast<uint4*>(out + zero_from);
    const size_t n16 = zero_n * 2, nb = gridDim.x - long_blocks - short_blocks;
    for (size_t i = (size_t)(blockIdx.x - long_blocks - short_blocks) * blockDim.x + threadIdx.x; i < n16; i += nb * blockDim.x) z[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  if (blockIdx.x < long_blocks) {
    const unsigned by = blockIdx.x / LONG_NB_MAX, bx = blockIdx.x % LONG_NB_MAX;
    const size_t col = long_cols[by];
    const unsigned nb = long_nb(col_len(a, col));
    if (bx >= nb) return;
    fe_t acc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) acc[i] = gather_major(a.m[i], col, rx, bx * blockDim.x + threadIdx.x, nb * blockDim.x);
    block_sum<3>(acc, smem);
    fe_t* trip = partials + ((size_t)by * LONG_NB_MAX) * 3;
    if (threadIdx.x == 0) {
      unsigned* dst = reinterpret_cast<unsigned*>(trip + (size_t)bx * 3);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int w = 0; w < 8; ++w) __hip_atomic_store(dst + 8 * i + w, acc[i].v[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // release / acquire at agent scope on the ticket (a few hundred long-column blocks at most, all of them under the short columns' walk: the
      // L2 write-back a release costs does not matter here as it did in a thousand-block streaming launch)
      s_last = __hip_atomic_fetch_add(tickets + by, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nb - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      acc[i] = fe_zero();
      if (threadIdx.x < nb) {
        const unsigned* src = reinterpret_cast<const unsigned*>(trip + (size_t)threadIdx.x * 3 + i);
#pragma unroll
        for (int w = 0; w < 8; ++w) acc[i].v[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();  // smem reuse
    block_sum<3>(acc, smem);
    if (threadIdx.x == 0) {
      __hip_atomic_store(tickets + by, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
      out[col] = fe_add<S>(fe_add<S>(acc[0], fe_mul<S>(a.r, acc[1])), fe_mul<S>(a.r2, acc[2]));
    }
    return;
  }
  const size_t nblk = short_blocks;
  for (size_t i = (size_t)(blockIdx.x - long_blocks) * blockDim.x + threadIdx.x; i < n_short; i += nblk * blockDim.x) {
    const size_t col = order[i];
    fe_t sa = gather_major_x4(a.m[0], col, rx), sb = gather_major_x4(a.m[1], col, rx), sc = gather_major_x4(a.m[2], col, rx);
    if (!fe_is_zero(sb)) sa = fe_add<S>(sa, fe_mul<S>(a.r, sb));
    if (!fe_is_zero(sc)) sa = fe_add<S>(sa, fe_mul<S>(a.r2, sc));
    out[col] = sa;
  }
}
__global__ void __launch_bounds__(256) k_polyabc_long_final(PolyAbcArgs a, const unsigned* __restrict__ long_cols, const fe_t* __restrict__ partials,
                                                            fe_t* __restrict__ out) {
  __shared__ fe_t smem[3 * 4];
  const size_t col = long_cols[blockIdx.x];
  const unsigned nb = long_nb(col_len(a, col));
  fe_t acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = threadIdx.x < nb ? partials[((size_t)blockIdx.x * LONG_NB_MAX + threadIdx.x) * 3 + i] : fe_zero();
  block_sum<3>(acc, smem);
  if (threadIdx.x == 0) out[col] = fe_add<S>(fe_add<S>(acc[0], fe_mul<S>(a.r, acc[1])), fe_mul<S>(a.r2, acc[2]));
}

// ---- poly_ABC split at a challenge boundary (round 3) -----------------------------------------------------------------------------------------
// poly_ABC[col] = sum_row M[row, col] eq(r_x, row) and eq(r_x, row) = eq_hi[row >> n_lo] * eq_lo[row & mask]: the outer sum-check draws r_x top variable
// first, so eq_hi is known n_lo rounds before the sum-check ends. k_polyabc_weights then turns every matrix entry into w = coeff * eq_hi[row >> n_lo]
// (a gather from a table of <= 2^12 entries: cache-resident; 32 bytes written per entry, in the column-major entry order) under the remaining,
// latency-bound rounds, and after the last challenge k_polyabc_weighted_* only has to stream the weights and multiply by eq_lo (<= 2^10 entries):
// no 32 MB evals_rx table, no random gathers from it on the critical path.
struct EqSmallArgs {
  fe_t v[12];
  int m;
  fe_t* out;  // pyramid: the table of all m variables at out + 2^m - 1
};
__global__ void __launch_bounds__(1024) k_eq_small(EqSmallArgs a) {
  if (threadIdx.x == 0) a.out[0] = fe_one<S>();
  __syncthreads();
  for (int k = 0; k < a.m; ++k) {
    const fe_t r = a.v[a.m - 1 - k];
    const size_t size = (size_t)1 << k;
    const fe_t* prev = a.out + (size - 1);
    fe_t* next = a.out + (2 * size - 1);
    for (size_t i = threadIdx.x; i < size; i += blockDim.x) {
      const fe_t e = prev[i], y = fe_mul<S>(e, r);
      next[size + i] = y;
      next[i] = fe_sub<S>(e, y);
    }
    __syncthreads();
  }
}
// Sliced-ELL copy of the SHORT columns for the split form: the columns of a wave (64 consecutive entries of `order`, i.e. columns of nearly equal
// length) store their j-th entries side by side — slot = off[wave] + 64 j + lane — so that every load of the final pass is coalesced: 256 bytes of packed
// (row | matrix << 28 | valid << 31) words and 2 KiB of weights per wave and step. `src` tells the weights pass where an entry's coefficient lives
// (k | general << 29 | matrix << 30 into the column-major arrays).
struct EllDev {
  const uint4* meta;  // per wave TWO words: (first slot, small steps of A, general steps of A, small steps of B), (general steps of B, small of C, general of C, -):
                      // a wave's columns are sorted to have the same entry counts per matrix and class, so the six segments need next to no padding, the
                      // final pass has no per-entry matrix selector and the one-pass form multiplies only in its general segments
  const unsigned* row;
  const unsigned* src;
  size_t slots;
};
struct EllCoeff {
  const signed char* scode[3];
  const fe_t* gval[3];
};
__global__ void __launch_bounds__(256) k_polyabc_ell_weights(EllDev e, EllCoeff co, const fe_t* __restrict__ eq_hi, int n_lo, fe_t* __restrict__ w) {
  for (size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x; slot < e.slots; slot += (size_t)gridDim.x * blockDim.x) {
    const unsigned packed = e.row[slot];
    if (!(packed >> 31)) continue;
    const unsigned src = e.src[slot], k = src & 0x1fffffffu;
    const int m = (int)(src >> 30);
    const fe_t h = eq_hi[(packed & 0x0fffffffu) >> n_lo];
    fe_t v;
    if (src & 0x20000000u) {
      v = fe_mul<S>(co.gval[m][k], h);
    } else {
      const int code = co.scode[m][k];
      v = small_mul(code, h);
      if (code < 0) v = fe_neg<S>(v);
    }
    w[slot] = v;
  }
}
__global__ void __launch_bounds__(256) k_polyabc_ell_final(EllDev e, const fe_t* __restrict__ w, const fe_t* __restrict__ eq_lo, unsigned mask, const unsigned* __restrict__ order,
                                                           size_t n_short, fe_t r, fe_t r2, fe_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t wave = i >> 6;
  const unsigned lane = threadIdx.x & 63;
  if (wave * 64 >= n_short) return;
  const uint4 m0 = e.meta[2 * wave], m1 = e.meta[2 * wave + 1];
  const uint4 mt = make_uint4(m0.x, m0.y + m0.z, m0.w + m1.x, m1.y + m1.z);  // per-matrix totals
  // sum of weight * eq_lo over `steps` slots starting at `base` (this lane's column of the segment), four loads in flight
  auto segment = [&](size_t base, unsigned steps) {
    fe_t acc = fe_zero(), acc2 = fe_zero();
    unsigned j = 0;
    for (; j + 2 <= steps; j += 2) {
      const size_t s0 = base + 64 * (size_t)j, s1 = s0 + 64;
      const unsigned p0 = e.row[s0], p1 = e.row[s1];
      const fe_t w0 = w[s0], w1 = w[s1];
      if (p0 >> 31) acc = fe_add<S>(acc, fe_mul<S>(w0, eq_lo[p0 & mask]));
      if (p1 >> 31) acc2 = fe_add<S>(acc2, fe_mul<S>(w1, eq_lo[p1 & mask]));
    }
    if (j < steps) {
      const size_t s0 = base + 64 * (size_t)j;
      const unsigned p0 = e.row[s0];
      if (p0 >> 31) acc = fe_add<S>(acc, fe_mul<S>(w[s0], eq_lo[p0 & mask]));
    }
    return fe_add<S>(acc, acc2);
  };
  const size_t base = (size_t)mt.x + lane;
  fe_t sa = segment(base, mt.y);
  if (mt.z) sa = fe_add<S>(sa, fe_mul<S>(r, segment(base + 64 * (size_t)mt.y, mt.z)));
  if (mt.w) sa = fe_add<S>(sa, fe_mul<S>(r2, segment(base + 64 * ((size_t)mt.y + mt.z), mt.w)));
  if (i < n_short) out[order[i]] = sa;
}
// ONE-PASS poly_ABC over the same sliced-ELL copy: per wave one meta load, then coalesced (row, class) loads, then the gathers from evals_rx — three
// dependent memory rounds for a short column where the column-major walk of k_polyabc_short needs nine (pointers, indices, gathers, for each of the
// three matrices in turn). `cls`: 0 = padding, 1..14 = small codes (-7..-1, 1..7 -> 1..14), 15.. = index into the table of the shape's distinct
// general coefficients (powers of two of the additions' rows: 214 values at config 2), which stays in the L1s.
__device__ __forceinline__ int cls_code(unsigned cls) { return cls <= 7 ? (int)cls - 8 : (int)cls - 7; }  // 1..7 -> -7..-1, 8..14 -> 1..7
__global__ void __launch_bounds__(256) k_polyabc_ell_onepass(EllDev e, const unsigned char* __restrict__ cls, const fe_t* __restrict__ gtab, const fe_t* __restrict__ rx,
                                                             const unsigned* __restrict__ order, size_t n_short, fe_t r, fe_t r2, fe_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t wave = i >> 6;
  const unsigned lane = threadIdx.x & 63;
  if (wave * 64 >= n_short) return;
  const uint4 m0 = e.meta[2 * wave], m1 = e.meta[2 * wave + 1];
  // small-class segment: gathers and add / sub / double-add chains, four entries in flight
  auto seg_small = [&](size_t base, unsigned steps) {
    fe_t acc = fe_zero();
    unsigned j = 0;
    for (; j + 4 <= steps; j += 4) {
      const size_t s0 = base + 64 * (size_t)j;
      const unsigned p0 = e.row[s0], p1 = e.row[s0 + 64], p2 = e.row[s0 + 128], p3 = e.row[s0 + 192];
      const unsigned c0 = cls[s0], c1 = cls[s0 + 64], c2 = cls[s0 + 128], c3 = cls[s0 + 192];
      const fe_t x0 = rx[p0 & 0x0fffffffu], x1 = rx[p1 & 0x0fffffffu], x2 = rx[p2 & 0x0fffffffu], x3 = rx[p3 & 0x0fffffffu];
      if (c0) acc = acc_small(acc, cls_code(c0), x0);
      if (c1) acc = acc_small(acc, cls_code(c1), x1);
      if (c2) acc = acc_small(acc, cls_code(c2), x2);
      if (c3) acc = acc_small(acc, cls_code(c3), x3);
    }
    for (; j < steps; ++j) {
      const size_t s0 = base + 64 * (size_t)j;
      const unsigned c0 = cls[s0];
      const fe_t x0 = rx[e.row[s0] & 0x0fffffffu];
      if (c0) acc = acc_small(acc, cls_code(c0), x0);
    }
    return acc;
  };
  // general-class segment: one product per entry, two in flight
  auto seg_general = [&](size_t base, unsigned steps) {
    fe_t acc = fe_zero();
    unsigned j = 0;
    for (; j + 2 <= steps; j += 2) {
      const size_t s0 = base + 64 * (size_t)j;
      const unsigned p0 = e.row[s0], p1 = e.row[s0 + 64];
      const unsigned c0 = cls[s0], c1 = cls[s0 + 64];
      const fe_t x0 = rx[p0 & 0x0fffffffu], x1 = rx[p1 & 0x0fffffffu];
      const fe_t g0 = gtab[c0 >= 15 ? c0 - 15 : 0], g1 = gtab[c1 >= 15 ? c1 - 15 : 0];
      if (c0) acc = fe_add<S>(acc, fe_mul<S>(g0, x0));
      if (c1) acc = fe_add<S>(acc, fe_mul<S>(g1, x1));
    }
    if (j < steps) {
      const size_t s0 = base + 64 * (size_t)j;
      const unsigned c0 = cls[s0];
      if (c0) acc = fe_add<S>(acc, fe_mul<S>(gtab[c0 - 15], rx[e.row[s0] & 0x0fffffffu]));
    }
    return acc;
  };
  size_t base = (size_t)m0.x + lane;
  fe_t sa = seg_small(base, m0.y);
  base += 64 * (size_t)m0.y;
  sa = fe_add<S>(sa, seg_general(base, m0.z));
  base += 64 * (size_t)m0.z;
  fe_t sb = seg_small(base, m0.w);
  base += 64 * (size_t)m0.w;
  sb = fe_add<S>(sb, seg_general(base, m1.x));
  base += 64 * (size_t)m1.x;
  fe_t sc = seg_small(base, m1.y);
  base += 64 * (size_t)m1.y;
  sc = fe_add<S>(sc, seg_general(base, m1.z));
  if (m0.w + m1.x) sa = fe_add<S>(sa, fe_mul<S>(r, sb));  // (wave-uniform: no B entries in two thirds of a SHA circuit's columns)
  if (m1.y + m1.z) sa = fe_add<S>(sa, fe_mul<S>(r2, sc));
  if (i < n_short) out[order[i]] = sa;
}
// long columns of the split form: no stored weights, the eq factor of an entry is the product of the two small tables
__device__ __forceinline__ fe_t gather_twotable(const SplitDev& m, size_t major, const fe_t* __restrict__ eq_hi, const fe_t* __restrict__ eq_lo, int n_lo, unsigned mask,
                                                unsigned first, unsigned step) {
  fe_t acc = fe_zero();
  for (unsigned k = m.sptr[major] + first, e = m.sptr[major + 1]; k < e; k += step) {
    const unsigned idx = m.sidx[k];
    acc = acc_small(acc, m.scode[k], fe_mul<S>(eq_hi[idx >> n_lo], eq_lo[idx & mask]));
  }
  for (unsigned k = m.gptr[major] + first, e = m.gptr[major + 1]; k < e; k += step) {
    const unsigned idx = m.gidx[k];
    acc = fe_add<S>(acc, fe_mul<S>(m.gval[k], fe_mul<S>(eq_hi[idx >> n_lo], eq_lo[idx & mask])));
  }
  return acc;
}
__global__ void __launch_bounds__(256) k_polyabc_long_twotable(PolyAbcArgs a, const fe_t* __restrict__ eq_hi, const fe_t* __restrict__ eq_lo, int n_lo, unsigned mask,
                                                               const unsigned* __restrict__ long_cols, fe_t* __restrict__ partials) {
  __shared__ fe_t smem[3 * 4];
  const size_t col = long_cols[blockIdx.y];
  const unsigned nb = long_nb(col_len(a, col));
  if (blockIdx.x >= nb) return;
  fe_t acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = gather_twotable(a.m[i], col, eq_hi, eq_lo, n_lo, mask, blockIdx.x * blockDim.x + threadIdx.x, nb * blockDim.x);
  block_sum<3>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) partials[((size_t)blockIdx.y * LONG_NB_MAX + blockIdx.x) * 3 + i] = acc[i];
  }
}

}  // namespace spk

// ---- host side: classification and upload ---------------------------------------------------------------------------
namespace {

struct SplitHost {
  std::vector<unsigned> sptr, sidx, gptr, gidx;
  std::vector<signed char> scode;
  std::vector<fe_t> gval;
};
struct SplitOnDevice {
  unsigned *sptr = nullptr, *sidx = nullptr, *gptr = nullptr, *gidx = nullptr;
  signed char* scode = nullptr;
  fe_t* gval = nullptr;
  spk::SplitDev view() const { return spk::SplitDev{sptr, sidx, scode, gptr, gidx, gval}; }
  void release() {
    hipFree(sptr);
    hipFree(sidx);
    hipFree(gptr);
    hipFree(gidx);
    hipFree(scode);
    hipFree(gval);
  }
};

struct Classifier {  // from_sparse (sparse.rs:49-134)
  fe_t pos[8], neg[8];
  Classifier() {
    for (int k = 1; k <= 7; ++k) {
      pos[k] = fe_from_u64<S>(k);
      neg[k] = fe_neg<S>(pos[k]);
    }
  }
  int code(const fe_t& v) const {
    for (int k = 1; k <= 7; ++k) {
      if (fe_eq(v, pos[k])) return k;
      if (fe_eq(v, neg[k])) return -k;
    }
    return 0;
  }
};

template <class T>
int upload(T** dst, const std::vector<T>& src) {
  size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
  hipError_t e = hipMalloc((void**)dst, bytes);
  if (e != hipSuccess) return fail(SP_ERR_NO_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
  if (!src.empty()) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail(SP_ERR_NO_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
  return SP_OK;
}
int upload_split(const SplitHost& h, SplitOnDevice* d) {
  int rc;
  if ((rc = upload(&d->sptr, h.sptr))) return rc;
  if ((rc = upload(&d->sidx, h.sidx))) return rc;
  if ((rc = upload(&d->scode, h.scode))) return rc;
  if ((rc = upload(&d->gptr, h.gptr))) return rc;
  if ((rc = upload(&d->gidx, h.gidx))) return rc;
  return upload(&d->gval, h.gval);
}

struct Entry {
  unsigned major, minor;
  int code;
  fe_t val;
};
// build a major-ordered split structure from (major, minor) entries already grouped by major
SplitHost build_split(size_t n_major, const std::vector<Entry>& es) {
  SplitHost h;
  h.sptr.assign(n_major + 1, 0);
  h.gptr.assign(n_major + 1, 0);
  for (const Entry& e : es) (e.code ? h.sptr : h.gptr)[e.major + 1]++;
  for (size_t i = 0; i < n_major; ++i) {
    h.sptr[i + 1] += h.sptr[i];
    h.gptr[i + 1] += h.gptr[i];
  }
  h.sidx.resize(h.sptr[n_major]);
  h.scode.resize(h.sptr[n_major]);
  h.gidx.resize(h.gptr[n_major]);
  h.gval.resize(h.gptr[n_major]);
  std::vector<unsigned> sc(h.sptr.begin(), h.sptr.end() - 1), gc(h.gptr.begin(), h.gptr.end() - 1);
  for (const Entry& e : es) {
    if (e.code) {
      unsigned p = sc[e.major]++;
      h.sidx[p] = e.minor;
      h.scode[p] = (signed char)e.code;
    } else {
      unsigned p = gc[e.major]++;
      h.gidx[p] = e.minor;
      h.gval[p] = e.val;
    }
  }
  return h;
}

}  // namespace

struct sp_shape {
  sp_dims dims;
  size_t num_vars = 0, num_cols = 0;
  SplitOnDevice row[3];       // full CSR (multiply_vec)
  SplitOnDevice filtered[3];  // FilteredSpmv rows: col >= num_shared + num_precommitted, row < num_cons_unpadded
  SplitOnDevice col[3];       // column-major, rows < num_cons_unpadded (accumulate_rows)
  unsigned* d_long_cols = nullptr;
  unsigned* d_short_order = nullptr;  // short columns by decreasing entry count (k_polyabc_short)
  size_t n_short = 0;
  fe_t* d_long_partials = nullptr;
  unsigned* d_long_tickets = nullptr;  // one arrival counter per long column (k_polyabc_short_and_long), zero between launches
  size_t n_long_cols = 0;
  uint64_t nnz[3] = {0, 0, 0}, nnz_filtered[3] = {0, 0, 0};
  // sliced-ELL copy of the short columns (kernels above: EllDev)
  uint4* d_ell_meta = nullptr;
  unsigned *d_ell_row = nullptr, *d_ell_src = nullptr;
  unsigned char* d_ell_cls = nullptr;  // one-pass form (k_polyabc_ell_onepass): class of every slot; nullptr when the shape has more than 241 distinct general coefficients
  fe_t* d_ell_gtab = nullptr;
  size_t ell_slots = 0;
};

extern "C" {

int sp_shape_from_csr(sp_ctx* c, const sp_csr* A, const sp_csr* Bm, const sp_csr* C, const sp_dims* dims, sp_shape** out) {
  (void)c;
  sp_shape* s = new sp_shape();
  s->dims = *dims;
  s->num_vars = dims->num_shared + dims->num_precommitted + dims->num_rest;
  s->num_cols = s->num_vars + 1 + dims->num_public + dims->num_challenges;
  const size_t nrows = dims->num_cons, col_min = dims->num_shared + dims->num_precommitted, nr_used = dims->num_cons_unpadded;
  const sp_csr* M[3] = {A, Bm, C};
  Classifier cls;
  std::vector<unsigned> col_count(s->num_cols, 0);
  std::vector<SplitHost> col_host(3);
  for (int m = 0; m < 3; ++m) {
    const size_t nnz = M[m]->indptr[nrows];
    s->nnz[m] = nnz;
    std::vector<Entry> all, filt, bycol;
    all.reserve(nnz);
    for (size_t r = 0; r < nrows; ++r) {
      for (uint64_t k = M[m]->indptr[r]; k < M[m]->indptr[r + 1]; ++k) {
        Entry e;
        e.major = (unsigned)r;
        e.minor = M[m]->indices[k];
        if (e.minor >= s->num_cols) {
          delete s;
          return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_shape_from_csr: column index out of range");
        }
        memcpy(&e.val, M[m]->data + 4 * k, 32);
        e.code = cls.code(e.val);
        all.push_back(e);
        if (r < nr_used && e.minor >= col_min) filt.push_back(e);
      }
    }
    s->nnz_filtered[m] = filt.size();
    // column-major copy of the rows accumulate_rows visits
    std::vector<unsigned> cptr(s->num_cols + 1, 0);
    for (const Entry& e : all)
      if (e.major < nr_used) cptr[e.minor + 1]++;
    for (size_t i = 0; i < s->num_cols; ++i) cptr[i + 1] += cptr[i];
    bycol.resize(cptr[s->num_cols]);
    {
      std::vector<unsigned> cur(cptr.begin(), cptr.end() - 1);
      for (const Entry& e : all)
        if (e.major < nr_used) {
          Entry t = e;
          t.major = e.minor;
          t.minor = e.major;
          bycol[cur[e.minor]++] = t;
        }
    }
    for (size_t i = 0; i < s->num_cols; ++i) col_count[i] += cptr[i + 1] - cptr[i];
    int rc;
    col_host[m] = build_split(s->num_cols, bycol);
    if ((rc = upload_split(build_split(nrows, all), &s->row[m])) || (rc = upload_split(build_split(nrows, filt), &s->filtered[m])) ||
        (rc = upload_split(col_host[m], &s->col[m]))) {
      delete s;
      return rc;
    }
  }
  std::vector<unsigned> long_cols;
  for (size_t i = 0; i < s->num_cols; ++i)
    if (col_count[i] >= spk::LONG_COLUMN) long_cols.push_back((unsigned)i);
  s->n_long_cols = long_cols.size();
  int rc = upload(&s->d_long_cols, long_cols);
  if (rc) return rc;
  {
    std::vector<unsigned> order;
    order.reserve(s->num_cols);
    for (size_t i = 0; i < s->num_cols; ++i)
      if (col_count[i] < spk::LONG_COLUMN) order.push_back((unsigned)i);
    // Sorted by decreasing entry count WITHIN windows of consecutive columns (SPARTAN_POLYABC_WINDOW, 0 = one global sort as in round 2): the lanes of
    // a wave still walk columns of (nearly) equal length, but a block's 256 columns now come from one neighbourhood of the matrix, so its pointer
    // loads are nearly coalesced and its gathers from evals_rx fall into a narrow row range (circuit variables are used near where they are allocated)
    static const size_t window = [] {
      const char* e = getenv("SPARTAN_POLYABC_WINDOW");
      return e ? (size_t)atol(e) : (size_t)0;  // measured (tools/r03_polyabc_window.sh): windows of 1 K - 128 K columns with heaviest-first chunks 115 - 120 us, one global sort 116 - 122 us: locality is not what bounds the kernel
    }();
    // (within one total the columns are grouped by their (A, B, C) entry counts: the lanes of a wave then run the same trip counts in each of the
    // three matrices — no divergence in the one-pass kernel, no padding in the sliced-ELL copy below)
    auto cnt6 = [&](unsigned col, unsigned out6[6]) {  // small / general entry counts of A, B, C
      for (int m = 0; m < 3; ++m) {
        out6[2 * m] = col_host[m].sptr[col + 1] - col_host[m].sptr[col];
        out6[2 * m + 1] = col_host[m].gptr[col + 1] - col_host[m].gptr[col];
      }
    };
    auto by_len = [&](unsigned x, unsigned y) {
      if (col_count[x] != col_count[y]) return col_count[x] > col_count[y];
      unsigned a[6], b[6];
      cnt6(x, a);
      cnt6(y, b);
      for (int k = 0; k < 5; ++k)
        if (a[k] != b[k]) return a[k] > b[k];
      return false;
    };
    if (window == 0) {
      std::stable_sort(order.begin(), order.end(), by_len);
    } else {
      for (size_t lo = 0; lo < order.size(); lo += window) {
        const size_t hi = lo + window < order.size() ? lo + window : order.size();
        std::stable_sort(order.begin() + lo, order.begin() + hi, [&](unsigned x, unsigned y) { return col_count[x] > col_count[y]; });
      }
      // ... and the 256-column chunks (= blocks of the launch) by decreasing cost, heaviest first, so that the grid does not end on long columns
      const size_t nchunks = (order.size() + 255) / 256;
      std::vector<unsigned> chunk(nchunks), cost(nchunks, 0);
      for (size_t c = 0; c < nchunks; ++c) {
        chunk[c] = (unsigned)c;
        for (size_t i = 256 * c; i < order.size() && i < 256 * (c + 1); i += 64) cost[c] += col_count[order[i]];  // the first (longest) column of each wave
      }
      std::stable_sort(chunk.begin(), chunk.end(), [&](unsigned x, unsigned y) { return cost[x] > cost[y]; });
      std::vector<unsigned> re;
      re.reserve(order.size());
      for (unsigned c : chunk)
        for (size_t i = 256 * (size_t)c; i < order.size() && i < 256 * ((size_t)c + 1); ++i) re.push_back(order[i]);
      order.swap(re);
    }
    s->n_short = order.size();
    if ((rc = upload(&s->d_short_order, order))) return rc;
    // sliced-ELL copy of the short columns in this order (the split poly_ABC): per wave three segments (A, B, C), each as long as the wave's longest
    // column in that matrix
    const size_t waves = (order.size() + 63) / 64;
    std::vector<uint4> meta(2 * (waves ? waves : 1));
    size_t slots = 0;
    for (size_t w = 0; w < waves; ++w) {
      unsigned L[6] = {0, 0, 0, 0, 0, 0};
      for (size_t i = 64 * w; i < order.size() && i < 64 * (w + 1); ++i) {
        unsigned c6[6];
        cnt6(order[i], c6);
        for (int k = 0; k < 6; ++k) L[k] = std::max(L[k], c6[k]);
      }
      meta[2 * w] = make_uint4((unsigned)slots, L[0], L[1], L[2]);
      meta[2 * w + 1] = make_uint4(L[3], L[4], L[5], 0u);
      slots += 64 * ((size_t)L[0] + L[1] + L[2] + L[3] + L[4] + L[5]);
    }
    if (slots >= ((size_t)1 << 31) || nrows > ((size_t)1 << 28)) {
      s->ell_slots = 0;  // (out of the packed format's range: the split form is then not offered for this shape)
    } else {
      std::vector<unsigned> erow(slots ? slots : 1, 0u), esrc(slots ? slots : 1, 0u);
      for (size_t i = 0; i < order.size(); ++i) {
        const size_t w = i / 64, lane = i % 64, col = order[i];
        size_t seg = meta[2 * w].x;
        const unsigned L[6] = {meta[2 * w].y, meta[2 * w].z, meta[2 * w].w, meta[2 * w + 1].x, meta[2 * w + 1].y, meta[2 * w + 1].z};
        for (int m = 0; m < 3; ++m) {
          const SplitHost& h = col_host[m];
          unsigned j = 0;
          for (unsigned k = h.sptr[col]; k < h.sptr[col + 1]; ++k, ++j) {
            const size_t slot = seg + 64 * (size_t)j + lane;
            erow[slot] = h.sidx[k] | ((unsigned)m << 28) | 0x80000000u;
            esrc[slot] = k | ((unsigned)m << 30);
          }
          seg += 64 * (size_t)L[2 * m];
          j = 0;
          for (unsigned k = h.gptr[col]; k < h.gptr[col + 1]; ++k, ++j) {
            const size_t slot = seg + 64 * (size_t)j + lane;
            erow[slot] = h.gidx[k] | ((unsigned)m << 28) | 0x80000000u;
            esrc[slot] = k | 0x20000000u | ((unsigned)m << 30);
          }
          seg += 64 * (size_t)L[2 * m + 1];
        }
      }
      s->ell_slots = slots;
      if ((rc = upload(&s->d_ell_meta, meta)) || (rc = upload(&s->d_ell_row, erow)) || (rc = upload(&s->d_ell_src, esrc))) return rc;
      // classes for the one-pass form: small codes, or an index into the table of distinct general coefficients
      std::vector<unsigned char> ecls(slots ? slots : 1, 0);
      std::vector<fe_t> gtab;
      std::map<std::array<uint32_t, 8>, unsigned> gidx_of;
      bool fits = true;
      for (size_t slot = 0; slot < slots && fits; ++slot) {
        if (!(erow[slot] >> 31)) continue;
        const unsigned src = esrc[slot], k = src & 0x1fffffffu;
        const SplitHost& h = col_host[src >> 30];
        if (src & 0x20000000u) {
          std::array<uint32_t, 8> key;
          memcpy(key.data(), &h.gval[k], 32);
          auto it = gidx_of.find(key);
          if (it == gidx_of.end()) {
            if (gtab.size() >= 241) {
              fits = false;
              break;
            }
            it = gidx_of.emplace(key, (unsigned)gtab.size()).first;
            gtab.push_back(h.gval[k]);
          }
          ecls[slot] = (unsigned char)(15 + it->second);
        } else {
          const int code = h.scode[k];
          ecls[slot] = (unsigned char)(code < 0 ? code + 8 : code + 7);
        }
      }
      if (fits) {
        if (gtab.empty()) gtab.push_back(fe_zero());
        if ((rc = upload(&s->d_ell_cls, ecls)) || (rc = upload(&s->d_ell_gtab, gtab))) return rc;
      }
    }
  }
  SP_HIP(hipMalloc((void**)&s->d_long_partials, (long_cols.size() + 1) * spk::LONG_NB_MAX * 3 * sizeof(fe_t)));
  SP_HIP(hipMalloc((void**)&s->d_long_tickets, (long_cols.size() + 1) * sizeof(unsigned)));
  SP_HIP(hipMemset(s->d_long_tickets, 0, (long_cols.size() + 1) * sizeof(unsigned)));
  SP_HIP(hipDeviceSynchronize());
  *out = s;
  return SP_OK;
}
void sp_shape_free(sp_shape* s) {
  if (!s) return;
  for (int m = 0; m < 3; ++m) {
    s->row[m].release();
    s->filtered[m].release();
    s->col[m].release();
  }
  hipFree(s->d_long_cols);
  hipFree(s->d_short_order);
  hipFree(s->d_long_partials);
  hipFree(s->d_long_tickets);
  if (s->d_ell_meta) hipFree(s->d_ell_meta);
  if (s->d_ell_row) hipFree(s->d_ell_row);
  if (s->d_ell_src) hipFree(s->d_ell_src);
  if (s->d_ell_cls) hipFree(s->d_ell_cls);
  if (s->d_ell_gtab) hipFree(s->d_ell_gtab);
  delete s;
}

int sp_shape_info(const sp_shape* s, uint64_t out[8]) {
  if (!s || !out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_shape_info: null argument");
  for (int m = 0; m < 3; ++m) {
    out[m] = s->nnz[m];
    out[3 + m] = s->nnz_filtered[m];
  }
  out[6] = s->n_long_cols;
  out[7] = s->n_short;
  return SP_OK;
}

static int spmv3(sp_ctx* c, const sp_shape* s, const SplitOnDevice* mats, const uint64_t* nnz, const sp_table* z, const sp_table* const* base,
                 sp_table** outs, const char* what) {
  const size_t nrows = s->dims.num_cons;
  if (z->len != s->num_cols) return fail(SP_ERR_INVALID_WITNESS_LENGTH, "multiply_vec: z has the wrong length");
  spk::Spmv3Args a;
  for (int m = 0; m < 3; ++m) {
    if (outs[m]->cap < nrows) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec: output table too short");
    if (base && base[m]->cap < nrows) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec_incremental: cached table too short");
    a.m[m] = mats[m].view();
    a.base[m] = base ? base[m]->d : nullptr;
    a.out[m] = outs[m]->d;
    outs[m]->len = nrows;
    outs[m]->lo_eff = outs[m]->hi_eff = (size_t)-1;
  }
  size_t blocks = (nrows + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  // sum_nnz (4 + 32) [+32 per general coefficient, ignored] + 3*32*N outputs (+ cached reads when incremental)
  uint64_t bytes = 36ull * (nnz[0] + nnz[1] + nnz[2]) + 96ull * nrows * (base ? 2 : 1);
  c->timed(what, bytes, [&] { hipLaunchKernelGGL(spk::k_spmv3, dim3((unsigned)blocks, 3), dim3(256), 0, c->stream, a, z->d, nrows); });
  return SP_OK;
}

int sp_multiply_vec(sp_ctx* c, const sp_shape* s, const sp_table* z, sp_table* az, sp_table* bz, sp_table* cz) {
  sp_table* outs[3] = {az, bz, cz};
  return spmv3(c, s, s->row, s->nnz, z, nullptr, outs, "spmv");
}
int sp_multiply_vec_batched(sp_ctx* c, const sp_shape* s, const sp_table* const* zs, size_t count, sp_table* const* az, sp_table* const* bz, sp_table* const* cz) {
  if (count && (!zs || !az || !bz || !cz)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec_batched: null argument");
  // one launch per vector: the matrices (a few MB of indices) stay in the L2s between launches, which is what the reference's single pass over the
  // matrix for all vectors buys on a CPU
  for (size_t k = 0; k < count; ++k) {
    sp_table* outs[3] = {az[k], bz[k], cz[k]};
    int rc = spmv3(c, s, s->row, s->nnz, zs[k], nullptr, outs, "spmv");
    if (rc) return rc;
  }
  return SP_OK;
}
int sp_multiply_vec_incremental(sp_ctx* c, const sp_shape* s, const sp_table* z, const sp_table* caz, const sp_table* cbz, const sp_table* ccz,
                                sp_table* az, sp_table* bz, sp_table* cz) {
  sp_table* outs[3] = {az, bz, cz};
  const sp_table* base[3] = {caz, cbz, ccz};
  return spmv3(c, s, s->filtered, s->nnz_filtered, z, base, outs, "spmv_incremental");
}

int sp_multiply_vec_incremental_round0(sp_ctx* c, const sp_shape* s, const sp_table* z, const sp_table* caz, const sp_table* cbz, const sp_table* ccz, sp_table* az,
                                       sp_table* bz, sp_table* cz, sp_table* p0, sp_table* p1) {
  const size_t nrows = s->dims.num_cons, half = nrows / 2;
  if (nrows < 2) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec_incremental_round0: needs at least two rows");
  if (p0->cap < half || p1->cap < half) return fail(SP_ERR_INVALID_INPUT_LENGTH, "multiply_vec_incremental_round0: product tables too short");
  int rc = sp_multiply_vec_incremental(c, s, z, caz, cbz, ccz, az, bz, cz);
  if (rc) return rc;
  p0->len = p1->len = half;
  p0->lo_eff = p0->hi_eff = p1->lo_eff = p1->hi_eff = (size_t)-1;
  size_t blocks = (half + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  c->timed("round0_products", 224ull * half, [&] { hipLaunchKernelGGL(spk::k_round0_products, dim3((unsigned)blocks), dim3(256), 0, c->stream, az->d, bz->d, cz->d, half, p0->d, p1->d); });
  return SP_OK;
}

static spk::EllDev ell_view(const sp_shape* s);
int sp_poly_abc(sp_ctx* c, const sp_shape* s, const sp_table* rx, const uint64_t r_[4], size_t out_len, sp_table* out) {
  if (rx->len != s->dims.num_cons) return fail(SP_ERR_INVALID_INPUT_LENGTH, "poly_ABC: rx must have num_cons elements");
  if (out_len < s->num_cols || out->cap < out_len) return fail(SP_ERR_INVALID_INPUT_LENGTH, "poly_ABC: output too short");
  spk::PolyAbcArgs a;
  for (int m = 0; m < 3; ++m) a.m[m] = s->col[m].view();
  memcpy(&a.r, r_, 32);
  a.r2 = fe_mul<S>(a.r, a.r);
  const char* merged_env = getenv("SPARTAN_POLYABC_MERGED");  // "0": fill, short columns, long columns and their final sums as four launches (rounds 1-3)
  const char* ell_env = getenv("SPARTAN_POLYABC_ELL");
  const bool ell_onepass = ell_env && ell_env[0] == '1';
  const bool merged = !ell_onepass && s->n_long_cols && s->n_long_cols <= 64 && !(merged_env && merged_env[0] == '0');
  if (out_len > s->num_cols && !merged) SP_HIP(hipMemsetAsync(out->d + s->num_cols, 0, (out_len - s->num_cols) * sizeof(fe_t), c->stream));
  size_t blocks = (s->n_short + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  uint64_t bytes = 36ull * (s->nnz[0] + s->nnz[1] + s->nnz[2]) + 32ull * out_len;
  // SPARTAN_POLYABC_ELL=1: the short columns from the sliced-ELL copy (k_polyabc_ell_onepass) instead of the column-major walk. Measured equal at config 2
  // (97.6 - 100.8 us against 101.4 - 102.0 us): with coalesced index loads and three dependent memory rounds instead of nine the kernel still takes
  // ~100 us, like the split form's final pass without any large gather - the time is the serial latency of a wave's steps at 4 waves per SIMD. Opt-in.
  c->timed("poly_abc", bytes, [&] {
    if (ell_onepass && s->d_ell_cls && s->ell_slots && s->n_short)
      hipLaunchKernelGGL(spk::k_polyabc_ell_onepass, dim3((unsigned)((s->n_short + 255) / 256)), dim3(256), 0, c->stream, ell_view(s), s->d_ell_cls, s->d_ell_gtab, rx->d,
                         s->d_short_order, s->n_short, a.r, a.r2, out->d);
    else if (merged) {
      const size_t zero_n = out_len - s->num_cols;
      size_t zblocks = (2 * zero_n + 256 * 16 - 1) / (256 * 16);  // 16 stores of 16 bytes per thread
      if (zblocks > 2048) zblocks = 2048;
      hipLaunchKernelGGL(spk::k_polyabc_short_and_long, dim3((unsigned)(blocks + spk::LONG_NB_MAX * s->n_long_cols + zblocks)), dim3(256), 0, c->stream, a, rx->d, s->d_short_order,
                         s->n_short, out->d, s->d_long_cols, (unsigned)s->n_long_cols, s->d_long_partials, s->d_long_tickets, (unsigned)blocks, (size_t)s->num_cols, zero_n);
      return;
    } else
      hipLaunchKernelGGL(spk::k_polyabc_short, dim3((unsigned)blocks), dim3(256), 0, c->stream, a, rx->d, s->d_short_order, s->n_short, out->d);
    if (s->n_long_cols) {
      hipLaunchKernelGGL(spk::k_polyabc_long, dim3(spk::LONG_NB_MAX, (unsigned)s->n_long_cols), dim3(256), 0, c->stream, a, rx->d, s->d_long_cols,
                         s->d_long_partials);
      hipLaunchKernelGGL(spk::k_polyabc_long_final, dim3((unsigned)s->n_long_cols), dim3(256), 0, c->stream, a, s->d_long_cols, s->d_long_partials,
                         out->d);
    }
  });
  return SP_OK;
}

// ---- bind_and_prepare_poly_ABC split at a challenge boundary (see k_polyabc_ell_weights) --------------------------------------------------------
struct sp_polyabc_ws {
  const sp_shape* s = nullptr;
  fe_t* w = nullptr;                        // one weight per ELL slot
  fe_t *eq_hi = nullptr, *eq_lo = nullptr;  // pyramids
  hipEvent_t ready = nullptr;
  size_t n_hi = 0;
  bool begun = false;
};
int sp_poly_abc_ws_create(sp_ctx* c, const sp_shape* s, sp_polyabc_ws** out) {
  if (!c || !s || !out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_poly_abc_ws_create: null argument");
  if (s->n_short && !s->ell_slots) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_poly_abc_ws_create: the shape is outside the packed entry format (2^28 rows, 2^31 slots)");
  sp_polyabc_ws* w = new sp_polyabc_ws();
  w->s = s;
  auto bail = [&](const char* what, hipError_t e) {
    sp_poly_abc_ws_free(w);
    return fail(SP_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
  };
  hipError_t e;
  if ((e = hipMalloc((void**)&w->w, (s->ell_slots + 1) * sizeof(fe_t))) != hipSuccess) return bail("weights", e);
  if ((e = hipMalloc((void**)&w->eq_hi, ((size_t)2 << 12) * sizeof(fe_t))) != hipSuccess) return bail("eq_hi", e);
  if ((e = hipMalloc((void**)&w->eq_lo, ((size_t)2 << 12) * sizeof(fe_t))) != hipSuccess) return bail("eq_lo", e);
  if ((e = hipEventCreateWithFlags(&w->ready, hipEventDisableTiming)) != hipSuccess) return bail("event", e);
  *out = w;
  return SP_OK;
}
void sp_poly_abc_ws_free(sp_polyabc_ws* w) {
  if (!w) return;
  if (w->w) hipFree(w->w);
  if (w->eq_hi) hipFree(w->eq_hi);
  if (w->eq_lo) hipFree(w->eq_lo);
  if (w->ready) hipEventDestroy(w->ready);
  delete w;
}
static spk::EllDev ell_view(const sp_shape* s) { return spk::EllDev{s->d_ell_meta, s->d_ell_row, s->d_ell_src, s->ell_slots}; }
// r_hi = the first n_hi challenges of the outer sum-check (top variables of the row index). Issued on the auxiliary stream; returns at once.
int sp_poly_abc_begin(sp_ctx* c, sp_polyabc_ws* w, const uint64_t* r_hi, size_t n_hi) {
  if (!c || !w || (!r_hi && n_hi)) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_poly_abc_begin: null argument");
  const sp_shape* s = w->s;
  size_t ell = 0;
  while (((size_t)1 << ell) < s->dims.num_cons) ++ell;
  if (n_hi > 12 || n_hi > ell || ell - n_hi > 12) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_poly_abc_begin: both halves of the row index must have at most 12 bits");
  spk::EqSmallArgs ea;
  for (size_t i = 0; i < n_hi; ++i) memcpy(&ea.v[i], r_hi + 4 * i, 32);
  ea.m = (int)n_hi;
  ea.out = w->eq_hi;
  hipLaunchKernelGGL(spk::k_eq_small, dim3(1), dim3(1024), 0, c->stream2, ea);
  if (s->ell_slots) {
    spk::EllCoeff co;
    for (int m = 0; m < 3; ++m) {
      co.scode[m] = s->col[m].scode;
      co.gval[m] = s->col[m].gval;
    }
    size_t blocks = (s->ell_slots + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    c->timed_on(c->stream2, "poly_abc_weights", 40ull * s->ell_slots, [&] {
      hipLaunchKernelGGL(spk::k_polyabc_ell_weights, dim3((unsigned)blocks), dim3(256), 0, c->stream2, ell_view(s), co, w->eq_hi + (((size_t)1 << n_hi) - 1), (int)(ell - n_hi), w->w);
    });
  }
  SP_HIP(hipEventRecord(w->ready, c->stream2));
  w->n_hi = n_hi;
  w->begun = true;
  return SP_OK;
}
// r_lo = the remaining challenges; r = the joint challenge of src/spartan.rs:311. On the main stream, behind the weights.
int sp_poly_abc_finish(sp_ctx* c, sp_polyabc_ws* w, const uint64_t* r_lo, size_t n_lo, const uint64_t r_[4], size_t out_len, sp_table* out) {
  if (!c || !w || (!r_lo && n_lo) || !r_ || !out) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_poly_abc_finish: null argument");
  if (!w->begun) return fail(SP_ERR_INTERNAL, "sp_poly_abc_finish: sp_poly_abc_begin has not run");
  w->begun = false;
  const sp_shape* s = w->s;
  size_t ell = 0;
  while (((size_t)1 << ell) < s->dims.num_cons) ++ell;
  if (w->n_hi + n_lo != ell || n_lo > 12) return fail(SP_ERR_INVALID_INPUT_LENGTH, "sp_poly_abc_finish: the two challenge lists must cover the row variables");
  if (out_len < s->num_cols || out->cap < out_len) return fail(SP_ERR_INVALID_INPUT_LENGTH, "poly_ABC: output too short");
  spk::EqSmallArgs ea;
  for (size_t i = 0; i < n_lo; ++i) memcpy(&ea.v[i], r_lo + 4 * i, 32);
  ea.m = (int)n_lo;
  ea.out = w->eq_lo;
  hipLaunchKernelGGL(spk::k_eq_small, dim3(1), dim3(1024), 0, c->stream, ea);
  SP_HIP(hipStreamWaitEvent(c->stream, w->ready, 0));
  spk::PolyAbcArgs a;
  for (int m = 0; m < 3; ++m) a.m[m] = s->col[m].view();
  memcpy(&a.r, r_, 32);
  a.r2 = fe_mul<S>(a.r, a.r);
  if (out_len > s->num_cols) SP_HIP(hipMemsetAsync(out->d + s->num_cols, 0, (out_len - s->num_cols) * sizeof(fe_t), c->stream));
  const fe_t* eq_lo = w->eq_lo + (((size_t)1 << n_lo) - 1);
  const fe_t* eq_hi = w->eq_hi + (((size_t)1 << w->n_hi) - 1);
  const unsigned mask = (unsigned)(((size_t)1 << n_lo) - 1);
  const uint64_t bytes = 36ull * (s->nnz[0] + s->nnz[1] + s->nnz[2]) + 32ull * out_len;
  c->timed("poly_abc_final", bytes, [&] {
    if (s->n_short)
      hipLaunchKernelGGL(spk::k_polyabc_ell_final, dim3((unsigned)((s->n_short + 255) / 256)), dim3(256), 0, c->stream, ell_view(s), w->w, eq_lo, mask, s->d_short_order, s->n_short,
                         a.r, a.r2, out->d);
    if (s->n_long_cols) {
      hipLaunchKernelGGL(spk::k_polyabc_long_twotable, dim3(spk::LONG_NB_MAX, (unsigned)s->n_long_cols), dim3(256), 0, c->stream, a, eq_hi, eq_lo, (int)n_lo, mask, s->d_long_cols,
                         s->d_long_partials);
      hipLaunchKernelGGL(spk::k_polyabc_long_final, dim3((unsigned)s->n_long_cols), dim3(256), 0, c->stream, a, s->d_long_cols, s->d_long_partials, out->d);
    }
  });
  return SP_OK;
}

}  // extern "C"
