// Throughput-shaped group kernels: FixedBaseMul::precompute for one and for many bases (msm.rs:653-689), Curve::batch_normalize (traits.rs:194-198) and
// bind_with_delayed (hyrax_pc.rs:38-54). They live in a translation unit of their own (capi_bulk.hip) because capi_group.hip is compiled with
// -amdgpu-sched-strategy=max-ilp for its latency chains (the cooperative point additions), under which these kernels spill: k_rowmat_vec_tall 26 VGPRs /
// 108 B of scratch a lane (21 MiB of scratch writes for a 32 MiB read, profiles/r05_pmc_traffic.json), k_fixed_base_table[s] 16, k_fb_fill 282 registers
// (one wave a SIMD). With the default scheduler: 154 / 128 / 133 registers, no spills (tools/spill_report.py, tests/test_spills_cpu.py).
#pragma once
#include "curve.hpp"
#include "device_utils.hpp"

namespace spk {
typedef FqP SF;  // scalar field

// ---- K12: fixed-base multiples of h --------------------------------------------------------------------------------------
// table[j*255 + d-1] = d * 2^(8j) * h (affine). Stage 1: thread j builds its window's 255 Jacobian multiples.
__global__ void k_fixed_base_table(aff_t h, jac_t* __restrict__ table_jac) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 32) return;
  jac_t base = jac_from_affine(h);
  for (int k = 0; k < 8 * j; ++k) base = jac_dbl(base);
  jac_t acc = base;
  table_jac[(size_t)j * 255] = acc;
  for (int d = 1; d < 255; ++d) {
    acc = jac_add(acc, base);
    table_jac[(size_t)j * 255 + d] = acc;
  }
}
// the same for many bases in one launch: block b builds the 32 window rows of bases[b] (FixedBaseMul::precompute for every row commitment of a
// prepared witness: sp_fbtables_create)
__global__ void k_fixed_base_tables(const aff_t* __restrict__ bases, size_t n, jac_t* __restrict__ table_jac) {
  const size_t b = blockIdx.x;
  const int j = threadIdx.x;
  if (b >= n || j >= 32) return;
  jac_t base = jac_from_affine(bases[b]);
  for (int k = 0; k < 8 * j; ++k) base = jac_dbl(base);
  jac_t* row = table_jac + (b * 32 + (size_t)j) * 255;
  jac_t acc = base;
  row[0] = acc;
  for (int d = 1; d < 255; ++d) {
    acc = jac_add(acc, base);
    row[d] = acc;
  }
}
// FixedBaseMul tables with 16-bit windows for the latency paths (a round commitment of the ZK verifier circuit: 16 table entries per scalar and a
// four-level tree instead of 32 and five): table[(b * 16 + w) * 65535 + d - 1] = d * 2^(16 w) * bases[b]. One 256-thread block per (base, window):
// thread 0 walks the 255 giant steps k * 256 * G, then thread t fills multiples 256 t + 1 .. 256 t + 255 from its giant step (~510 dependent
// additions in all; the 65535 multiples of a window would take a single thread 0.8 s).
__global__ void __launch_bounds__(256) k_fixed_base_tables16(const aff_t* __restrict__ bases, size_t n, jac_t* __restrict__ table_jac) {
  const size_t b = blockIdx.x / 16;
  const int w = blockIdx.x % 16;
  if (b >= n) return;
  jac_t* row = table_jac + ((size_t)b * 16 + (size_t)w) * 65535;
  __shared__ jac_t G_sh;
  if (threadIdx.x == 0) {
    jac_t G = jac_from_affine(bases[b]);
    for (int k = 0; k < 16 * w; ++k) G = jac_dbl(G);
    G_sh = G;
    jac_t step = G;
    for (int k = 0; k < 8; ++k) step = jac_dbl(step);  // 256 G
    jac_t acc = step;
    row[256 - 1] = acc;  // multiple 256
    for (int t = 2; t < 256; ++t) {
      acc = jac_add(acc, step);
      row[(size_t)256 * t - 1] = acc;  // multiple 256 t
    }
  }
  __syncthreads();  // (global writes of thread 0 are read back below by the other threads of this block)
  __threadfence_block();
  const jac_t G = G_sh;
  const int t = threadIdx.x;
  jac_t acc = t == 0 ? jac_identity() : row[(size_t)256 * t - 1];
  for (int d = 1; d < 256; ++d) {
    acc = t == 0 && d == 1 ? G : jac_add(acc, G);
    row[(size_t)256 * t + d - 1] = acc;  // multiple 256 t + d
  }
}
__global__ void __launch_bounds__(256) k_jac_to_affine(const jac_t* __restrict__ in, size_t n, aff_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = jac_to_affine(in[i]);
}
// Curve::batch_normalize (src/provider/traits.rs:194-198) for table-sized inputs: Montgomery's trick per THREAD - thread t of T owns the entries
// t, t + T, t + 2T, ... (<= K of them; lanes touch consecutive records in every step), keeps the running products of their Z in `pre` (one element per
// entry), inverts the last one (the ~350 products of Fermat's exponentiation, once per K entries instead of once per entry as k_jac_to_affine) and
// walks back: 1 / Z_k = inv * pre[k - 1], inv *= Z_k. An identity (Z == 0) counts as Z = 1 in the products and is written as (0, 0).
__global__ void __launch_bounds__(64) k_jac_to_affine_batch(const jac_t* __restrict__ in, size_t n, unsigned K, aff_t* __restrict__ out, fe_t* __restrict__ pre) {
  const size_t T = (n + K - 1) / K, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  fe_t acc = fe_one<B>();
  unsigned cnt = 0;
  for (size_t i = t; i < n && cnt < K; i += T, ++cnt) {
    const fe_t z = in[i].z;
    if (!fe_is_zero(z)) acc = fe_mul<B>(acc, z);
    pre[i] = acc;
  }
  fe_t inv = fe_inv_fermat<B>(acc);
  for (unsigned k = cnt; k-- > 0;) {
    const size_t i = t + (size_t)k * T;
    const jac_t p = in[i];
    aff_t r;
    if (fe_is_zero(p.z)) {
      r.x = fe_zero();
      r.y = fe_zero();
    } else {
      const fe_t zi = k ? fe_mul<B>(inv, pre[i - T]) : inv;
      inv = fe_mul<B>(inv, p.z);
      const fe_t zi2 = fe_sqr<B>(zi);
      r.x = fe_mul<B>(p.x, zi2);
      r.y = fe_mul<B>(fe_mul<B>(p.y, zi2), zi);
    }
    out[i] = r;
  }
}

// ---- FixedBaseMul::precompute (msm.rs:653-689) for MANY bases in three launches (sp_fbtables_create, sp_ck_create) ------------------------------------
// k_fixed_base_tables above walks 255 dependent additions behind up to 248 doublings per (base, window) thread on half-empty waves, and k_jac_to_affine
// pays one inversion per entry: 4.6 + 10.2 ms for the 513 row tables of a 2^20-variable witness. Here:
//   1. k_fb_ladder: thread b doubles P_b 255 times (the one chain that cannot be shortened: 2^248 P_b is 248 doublings away) and keeps, per window j,
//      2^(8j) P_b = B_j and 32 B_j, 64 B_j, 128 B_j:  ladder[(4 j + kind) * n + b]
//   2. k_jac_to_affine_batch on the 128 n ladder points
//   3. k_fb_fill: thread (q, j, b) builds entries 32 q + 1 .. 32 q + 32 of window j of base b - starts at 32 q B_j (<= 3 mixed additions of ladder points),
//      then 32 mixed additions of B_j - and normalises its own 32 entries with one inversion (Jacobian forms and running Z products in scratch laid out
//      [entry][chain], so that lanes touch consecutive records). 8 n * 32 threads instead of 32 n: two waves on every SIMD at n = 513.
// Results are the canonical affine multiples either way (tests/test_gpu_group.py::test_fbtables_every_entry).
__global__ void __launch_bounds__(64) k_fb_ladder(const aff_t* __restrict__ bases, size_t n, jac_t* __restrict__ ladder) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  jac_t p = jac_from_affine(bases[b]);
  for (int k = 0; k < 256; ++k) {
    const int r = k & 7;
    if (r == 0 || r >= 5) ladder[(size_t)(4 * (k >> 3) + (r == 0 ? 0 : r - 4)) * n + b] = p;
    if (k < 255) p = jac_dbl(p);
  }
}
__global__ void __launch_bounds__(64) k_fb_fill(const aff_t* __restrict__ ladder, size_t n, jac_t* __restrict__ J, fe_t* __restrict__ pre, aff_t* __restrict__ tables) {
  const size_t C = n * 32, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= 8 * C) return;
  const unsigned q = (unsigned)(tid / C);
  const size_t chain = tid % C, j = chain / n, b = chain % n;
  const aff_t Bj = ladder[(4 * j) * n + b];
  jac_t acc = jac_identity();
  if (q & 1u) acc = jac_add_mixed(acc, ladder[(4 * j + 1) * n + b]);
  if (q & 2u) acc = jac_add_mixed(acc, ladder[(4 * j + 2) * n + b]);
  if (q & 4u) acc = jac_add_mixed(acc, ladder[(4 * j + 3) * n + b]);
  const unsigned cnt = q == 7u ? 31u : 32u;
  fe_t prod = fe_one<B>();
  for (unsigned i = 0; i < cnt; ++i) {
    acc = jac_add_mixed(acc, Bj);
    const size_t at = (size_t)(32u * q + i) * C + chain;
    J[at] = acc;
    if (!fe_is_zero(acc.z)) prod = fe_mul<B>(prod, acc.z);
    pre[at] = prod;
  }
  fe_t inv = fe_inv_fermat<B>(prod);
  aff_t* row = tables + (b * 32 + j) * 255;
  for (unsigned i = cnt; i-- > 0;) {
    const size_t at = (size_t)(32u * q + i) * C + chain;
    const jac_t p = J[at];
    aff_t r;
    if (fe_is_zero(p.z)) {
      r.x = fe_zero();
      r.y = fe_zero();
    } else {
      const fe_t zi = i ? fe_mul<B>(inv, pre[at - C]) : inv;
      inv = fe_mul<B>(inv, p.z);
      const fe_t zi2 = fe_sqr<B>(zi);
      r.x = fe_mul<B>(p.x, zi2);
      r.y = fe_mul<B>(fe_mul<B>(p.y, zi2), zi);
    }
    row[32u * q + i] = r;
  }
}

// ---- K9: LZ[i] = sum_j L[j] * poly[j*cols + i] ---------------------------------------------------------------------------
// grid = (cols/64, row_splits): each block handles 64 columns x a slice of rows with 256 threads = 4 row-lanes per column.
__global__ void __launch_bounds__(256) k_rowmat_vec(const fe_t* __restrict__ poly, size_t rows, size_t cols, const fe_t* __restrict__ L,
                                                    fe_t* __restrict__ partial /* [row_splits][cols] */) {
  __shared__ fe_t s[256];
  const size_t col = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;  // 0..3
  const size_t splits = gridDim.y, per = (rows + splits - 1) / splits;
  const size_t r0 = blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  fe_t acc = fe_zero();
  if (col < cols)
    for (size_t r = r0 + rl; r < r1; r += 4) acc = fe_add<SF>(acc, fe_mul<SF>(L[r], poly[r * cols + col]));
  s[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && col < cols) {
    fe_t t = fe_add<SF>(fe_add<SF>(s[threadIdx.x], s[threadIdx.x + 64]), fe_add<SF>(s[threadIdx.x + 128], s[threadIdx.x + 192]));
    partial[(size_t)blockIdx.y * cols + col] = t;
  }
}
// Streaming form for tall matrices (rows >= 128: 512 x 2048 at BASELINE config 2, a pure 32 MiB read — hyrax_pc.rs:38-54): ONE launch, no partials.
// A block of 512 threads owns RMV_COLS = 8 adjacent columns: lane = (row-lane 0..7, column 0..7), so a wave reads eight 256-byte row segments per pass
// and the 8 waves cover 64 rows; L sits in LDS (a row's weight is read right before its product instead of being held), every lane keeps a modular sum
// of its rows' products, the eight row-lanes of a wave and then the eight waves are combined as lazy 9-word sums (shuffles, LDS) with one reduction per
// column. cols / 8 blocks (256 at config 2: one per CU). 512 threads and not 1024: at 128 VGPRs the four loads in flight + the product's temporaries
// spilled 31 registers (104 B of scratch per lane = 26 MB of writes for a 33 MB read: PMC WRITE_SIZE of profiles/r04_kernel_stats.md).
constexpr int RMV_COLS = 8;
constexpr int RMV_THREADS = 512;
constexpr int RMV_ROWS_PER_PASS = RMV_THREADS / RMV_COLS;  // 64
constexpr int RMV_L_MAX = 1024;                            // rows whose weights fit the block's LDS copy (32 KiB); taller matrices read L from memory
__global__ void __launch_bounds__(RMV_THREADS) k_rowmat_vec_tall(const fe_t* __restrict__ poly, size_t rows, size_t cols, const fe_t* __restrict__ L, fe_t* __restrict__ out) {
  __shared__ lazy9_t sm[RMV_THREADS / 64][RMV_COLS];
  extern __shared__ fe_t sL[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & (RMV_COLS - 1), rl = lane >> 3;
  const size_t col = (size_t)blockIdx.x * RMV_COLS + c;
  const bool l_in_lds = rows <= (size_t)RMV_L_MAX;
  if (l_in_lds) {
    for (size_t r = threadIdx.x; r < rows; r += RMV_THREADS) sL[r] = L[r];
    __syncthreads();
  }
  const fe_t* Lp = l_in_lds ? sL : L;
  fe_t acc = fe_zero();
  if (col < cols) {
    constexpr size_t P = RMV_ROWS_PER_PASS;
    size_t r = (size_t)wave * 8 + rl;
    for (; r + 3 * P < rows; r += 4 * P) {  // four loads in flight per lane
      const fe_t a0 = poly[r * cols + col], a1 = poly[(r + P) * cols + col], a2 = poly[(r + 2 * P) * cols + col], a3 = poly[(r + 3 * P) * cols + col];
      acc = fe_add<SF>(acc, fe_mul<SF>(Lp[r], a0));
      acc = fe_add<SF>(acc, fe_mul<SF>(Lp[r + P], a1));
      acc = fe_add<SF>(acc, fe_mul<SF>(Lp[r + 2 * P], a2));
      acc = fe_add<SF>(acc, fe_mul<SF>(Lp[r + 3 * P], a3));
    }
    for (; r < rows; r += P) acc = fe_add<SF>(acc, fe_mul<SF>(Lp[r], poly[r * cols + col]));
  }
  lazy9_t t = lazy_from(acc);
#pragma unroll
  for (int m = 32; m >= 8; m >>= 1) {
    lazy9_t o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.v[i] = __shfl_xor(t.v[i], m, 64);
    t = lazy_add(t, o);
  }
  if (rl == 0) sm[wave][c] = t;
  __syncthreads();
  if (threadIdx.x < RMV_COLS && col < cols) {
    lazy9_t s = sm[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < RMV_THREADS / 64; ++w) s = lazy_add(s, sm[w][threadIdx.x]);
    out[col] = lazy_reduce(s);
  }
}
__global__ void __launch_bounds__(256) k_sum_columns(const fe_t* __restrict__ partial, size_t splits, size_t cols, fe_t* __restrict__ out) {
  const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  fe_t acc = partial[col];
  for (size_t s = 1; s < splits; ++s) acc = fe_add<SF>(acc, partial[s * cols + col]);
  out[col] = acc;
}

}  // namespace spk
