// Experiment harness for the fused bind+eval kernel (K1+K2): times variants on 3 tables of length L and prints algorithmic
// GB/s (48*L bytes per table per launch). Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ispartan2_amd/csrc -Iinclude tools/fused_bench.hip -o tools/fused_bench
#include <cstdio>
#include <vector>

#include "kernels_poly.hpp"

using namespace spk;

__device__ __forceinline__ fe_t ld_nt(const fe_t* p) {
  fe_t r;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u* q = reinterpret_cast<const v4u*>(p);
  v4u a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
  r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}

// VAR 1: nontemporal loads; VAR 2: bind only (no eval) ; VAR 3: eval arithmetic but no block_sum
template <int VAR, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_var(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r,
                                                const fe_t* __restrict__ eq_in, const fe_t* __restrict__ eq_out, int s, fe_t* __restrict__ partials) {
  __shared__ fe_t smem[2 * 4];
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  fe_t acc[2] = {fe_zero(), fe_zero()};
  if (id < q) {
    fe_t la0, la1, la2, la3, lb0, lb1, lb2, lb3, lc0, lc1, lc2, lc3;
    if (VAR == 1) {
      la0 = ld_nt(A + id); la1 = ld_nt(A + id + q); la2 = ld_nt(A + id + 2 * q); la3 = ld_nt(A + id + 3 * q);
      lb0 = ld_nt(B + id); lb1 = ld_nt(B + id + q); lb2 = ld_nt(B + id + 2 * q); lb3 = ld_nt(B + id + 3 * q);
      lc0 = ld_nt(C + id); lc1 = ld_nt(C + id + q); lc2 = ld_nt(C + id + 2 * q); lc3 = ld_nt(C + id + 3 * q);
    } else {
      la0 = A[id]; la1 = A[id + q]; la2 = A[id + 2 * q]; la3 = A[id + 3 * q];
      lb0 = B[id]; lb1 = B[id + q]; lb2 = B[id + 2 * q]; lb3 = B[id + 3 * q];
      lc0 = C[id]; lc1 = C[id + q]; lc2 = C[id + 2 * q]; lc3 = C[id + 3 * q];
    }
    const fe_t a0 = bind1(la0, la2, r), a1 = bind1(la1, la3, r);
    const fe_t b0 = bind1(lb0, lb2, r), b1 = bind1(lb1, lb3, r);
    const fe_t c0 = bind1(lc0, lc2, r), c1 = bind1(lc1, lc3, r);
    A[id] = a0; A[id + q] = a1; B[id] = b0; B[id + q] = b1; C[id] = c0; C[id + q] = c1;
    if (VAR != 2) {
      fe_t w = eq_in[id & mask];
      const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
      const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
      acc[0] = fe_mul<S>(w, t0e);
      acc[1] = fe_mul<S>(w, tie);
    }
  }
  if (VAR == 2) return;
  if (VAR == 3) {
    if (id < q && (acc[0].v[0] ^ acc[1].v[0]) == 0x12345u) partials[0] = acc[0];
    return;
  }
  if (BLOCK == 256) {
    block_sum<2>(acc, smem);
  } else {
    acc[0] = wave_sum(acc[0]);
    acc[1] = wave_sum(acc[1]);
  }
  if (threadIdx.x == 0) {
    const fe_t eo = eq_out[((size_t)blockIdx.x * blockDim.x) >> s];
    partials[(size_t)blockIdx.x * 2] = fe_mul<S>(acc[0], eo);
    partials[(size_t)blockIdx.x * 2 + 1] = fe_mul<S>(acc[1], eo);
  }
}

// VAR 5/6: the production streaming kernel body, but each block walks ITER chunks (grid = q / 256 / ITER): after the first chunk the waves of a
// SIMD are no longer in lock step, so loads of one overlap arithmetic of another. PREFETCH issues the next chunk's loads before the arithmetic.
struct Chunk {
  fe_t a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
};
__device__ __forceinline__ Chunk load_chunk(const fe_t* A, const fe_t* B, const fe_t* C, size_t id, size_t q) {
  Chunk k;
  k.a0 = A[id]; k.a1 = A[id + q]; k.a2 = A[id + 2 * q]; k.a3 = A[id + 3 * q];
  k.b0 = B[id]; k.b1 = B[id + q]; k.b2 = B[id + 2 * q]; k.b3 = B[id + 3 * q];
  k.c0 = C[id]; k.c1 = C[id + q]; k.c2 = C[id + 2 * q]; k.c3 = C[id + 3 * q];
  return k;
}
template <int ITER, bool PREFETCH>
__global__ void __launch_bounds__(256) k_iter(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in,
                                              int s, lazy9_t* __restrict__ partials) {
  const size_t mask = ((size_t)1 << s) - 1;
  lazy9_t l0 = lazy_from(fe_zero()), l1 = lazy_from(fe_zero());
  size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  Chunk cur = load_chunk(A, B, C, id, q);
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    Chunk nxt;
    if (PREFETCH && it + 1 < ITER) nxt = load_chunk(A, B, C, id + step, q);
    const fe_t a0 = bind1(cur.a0, cur.a2, r), a1 = bind1(cur.a1, cur.a3, r);
    const fe_t b0 = bind1(cur.b0, cur.b2, r), b1 = bind1(cur.b1, cur.b3, r);
    const fe_t c0 = bind1(cur.c0, cur.c2, r), c1 = bind1(cur.c1, cur.c3, r);
    A[id] = a0; A[id + q] = a1; B[id] = b0; B[id + q] = b1; C[id] = c0; C[id + q] = c1;
    const fe_t w = eq_in[id & mask];
    const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
    const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
    l0 = lazy_add(l0, lazy_from(fe_mul<S>(w, t0e)));
    l1 = lazy_add(l1, lazy_from(fe_mul<S>(w, tie)));
    id += step;
    if (it + 1 < ITER) cur = PREFETCH ? nxt : load_chunk(A, B, C, id, q);
  }
  stream_block_partials(lazy_wave_sum(l0), lazy_wave_sum(l1), partials);
}

// VAR 7: the production streaming body with non-temporal STORES (the bound values are not re-read before the next launch), block size BLOCK
__device__ __forceinline__ void st_nt(fe_t* p, const fe_t& v) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u a = {v.v[0], v.v[1], v.v[2], v.v[3]}, b = {v.v[4], v.v[5], v.v[6], v.v[7]};
  v4u* q = reinterpret_cast<v4u*>(p);
  __builtin_nontemporal_store(a, q);
  __builtin_nontemporal_store(b, q + 1);
}
template <bool NT>
__global__ void __launch_bounds__(256) k_nt(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in, int s,
                                            lazy9_t* __restrict__ partials) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const Chunk k = load_chunk(A, B, C, id, q);
  const fe_t a0 = bind1(k.a0, k.a2, r), a1 = bind1(k.a1, k.a3, r);
  const fe_t b0 = bind1(k.b0, k.b2, r), b1 = bind1(k.b1, k.b3, r);
  const fe_t c0 = bind1(k.c0, k.c2, r), c1 = bind1(k.c1, k.c3, r);
  if (NT) {
    st_nt(A + id, a0); st_nt(A + id + q, a1); st_nt(B + id, b0); st_nt(B + id + q, b1); st_nt(C + id, c0); st_nt(C + id + q, c1);
  } else {
    A[id] = a0; A[id + q] = a1; B[id] = b0; B[id + q] = b1; C[id] = c0; C[id + q] = c1;
  }
  const fe_t w = eq_in[id & mask];
  const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}

// SPLIT: a pair's work over two lanes 32 apart - the low lane binds (a0, b0, c0) and forms w (a0 b0 - c0), the high lane binds (a1, b1, c1), takes
// a0 and b0 over the crossbar and forms w (a1 - a0)(b1 - b0): five dependent products per lane instead of ten, twice the waves, half the registers
// held across the loads. One five-level lazy sum inside each half-wave replaces the two six-level ones.
__global__ void __launch_bounds__(256) k_split(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in, int s,
                                               lazy9_t* __restrict__ partials) {
  __shared__ lazy9_t sm[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, role = lane >> 5;
  const size_t pair = (size_t)blockIdx.x * 128 + wave * 32 + (lane & 31);
  const size_t id = pair + (role ? q : 0);
  const size_t mask = ((size_t)1 << s) - 1;
  const fe_t la = A[id], ha = A[id + 2 * q], lb = B[id], hb = B[id + 2 * q], lc = C[id], hc = C[id + 2 * q];
  const fe_t w = eq_in[pair & mask];
  const fe_t a = bind1(la, ha, r), b = bind1(lb, hb, r), c = bind1(lc, hc, r);
  A[id] = a;
  B[id] = b;
  C[id] = c;
  fe_t oa, ob;  // the other half-wave's a and b
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    oa.v[i] = __shfl_xor(a.v[i], 32, 64);
    ob.v[i] = __shfl_xor(b.v[i], 32, 64);
  }
  const fe_t x = role ? fe_sub<S>(a, oa) : a, y = role ? fe_sub<S>(b, ob) : b;
  fe_t v = fe_mul<S>(x, y);
  if (!role) v = fe_sub<S>(v, c);
  lazy9_t t = lazy_from(fe_mul<S>(w, v));
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) {
    lazy9_t o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.v[i] = __shfl_xor(t.v[i], m, 64);
    t = lazy_add(t, o);
  }
  if ((lane & 31) == 0) sm[wave][role] = t;
  __syncthreads();
  if (threadIdx.x < 2) partials[(size_t)blockIdx.x * 2 + threadIdx.x] = lazy_add(lazy_add(sm[0][threadIdx.x], sm[1][threadIdx.x]), lazy_add(sm[2][threadIdx.x], sm[3][threadIdx.x]));
}

// FAT: one 1024-thread block per CU at 2^20 (256 blocks), grid-stride over ITER chunks: the "persistent" form. Wave sums meet in LDS; one lazy
// partial pair per block, reduced and weighted with eq_out IN the block (what a host-summed slot would carry): 256 partial pairs instead of 1024.
template <int ITER>
__global__ void __launch_bounds__(1024) k_fat(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in,
                                              const fe_t* __restrict__ eq_out, int s, fe_t* __restrict__ out) {
  __shared__ lazy9_t sm[16][2];
  const size_t mask = ((size_t)1 << s) - 1;
  lazy9_t l0 = lazy_from(fe_zero()), l1 = lazy_from(fe_zero());
  size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
#pragma unroll
  for (int it = 0; it < ITER; ++it, id += step) {
    const Chunk cur = load_chunk(A, B, C, id, q);
    const fe_t a0 = bind1(cur.a0, cur.a2, r), a1 = bind1(cur.a1, cur.a3, r);
    const fe_t b0 = bind1(cur.b0, cur.b2, r), b1 = bind1(cur.b1, cur.b3, r);
    const fe_t c0 = bind1(cur.c0, cur.c2, r), c1 = bind1(cur.c1, cur.c3, r);
    A[id] = a0; A[id + q] = a1; B[id] = b0; B[id + q] = b1; C[id] = c0; C[id + q] = c1;
    const fe_t w = eq_in[id & mask];
    const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
    const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
    l0 = lazy_add(l0, lazy_from(fe_mul<S>(w, t0e)));
    l1 = lazy_add(l1, lazy_from(fe_mul<S>(w, tie)));
  }
  l0 = lazy_wave_sum(l0);
  l1 = lazy_wave_sum(l1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sm[wave][0] = l0;
    sm[wave][1] = l1;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    lazy9_t t = sm[0][threadIdx.x];
    for (int k = 1; k < 16; ++k) t = lazy_add(t, sm[k][threadIdx.x]);
    out[(size_t)blockIdx.x * 2 + threadIdx.x] = fe_mul<S>(lazy_reduce(t), eq_out[((size_t)blockIdx.x * blockDim.x) >> s]);
  }
}

// W256: the production 256-thread streaming body, but every block reduces its lazy pair and applies eq_out itself (fe partials out): the second stage
// becomes a plain modular sum (k_sum_partials) instead of k_sum_partials_lazy
__global__ void __launch_bounds__(256) k_w256(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in,
                                              const fe_t* __restrict__ eq_out, int s, fe_t* __restrict__ out) {
  __shared__ lazy9_t sm[4][2];
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const Chunk k = load_chunk(A, B, C, id, q);
  const fe_t a0 = bind1(k.a0, k.a2, r), a1 = bind1(k.a1, k.a3, r);
  const fe_t b0 = bind1(k.b0, k.b2, r), b1 = bind1(k.b1, k.b3, r);
  const fe_t c0 = bind1(k.c0, k.c2, r), c1 = bind1(k.c1, k.c3, r);
  A[id] = a0; A[id + q] = a1; B[id] = b0; B[id + q] = b1; C[id] = c0; C[id + q] = c1;
  const fe_t w = eq_in[id & mask];
  const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  const lazy9_t l0 = lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), l1 = lazy_wave_sum(lazy_from(fe_mul<S>(w, tie)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sm[wave][0] = l0;
    sm[wave][1] = l1;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const lazy9_t t = lazy_add(lazy_add(sm[0][threadIdx.x], sm[1][threadIdx.x]), lazy_add(sm[2][threadIdx.x], sm[3][threadIdx.x]));
    out[(size_t)blockIdx.x * 2 + threadIdx.x] = fe_mul<S>(lazy_reduce(t), eq_out[((size_t)blockIdx.x * blockDim.x) >> s]);
  }
}

template <class L>
static float time_us(L&& f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

// UP: every element load (and the weight) issued before the first product, held there by a scheduling barrier: the compiler otherwise sinks the loads
// to their uses and the waves of the single generation of a 2^20 launch walk through seven load -> wait -> compute phases in lock step (the memory
// pipe idles while they all compute). W = waves per SIMD the register budget is sized for (2: 256 VGPRs, 3: 168, 4: 128).
template <int W>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
k_up(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in, int s, lazy9_t* __restrict__ partials) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const fe_t la0 = A[id], la2 = A[id + 2 * q], la1 = A[id + q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb2 = B[id + 2 * q], lb1 = B[id + q], lb3 = B[id + 3 * q];
  const fe_t lc0 = C[id], lc2 = C[id + 2 * q], lc1 = C[id + q], lc3 = C[id + 3 * q];
  const fe_t w = eq_in[id & mask];
  __builtin_amdgcn_sched_barrier(0);
  const fe_t a0 = bind1(la0, la2, r);
  A[id] = a0;
  const fe_t a1 = bind1(la1, la3, r);
  A[id + q] = a1;
  const fe_t b0 = bind1(lb0, lb2, r);
  B[id] = b0;
  const fe_t b1 = bind1(lb1, lb3, r);
  B[id + q] = b1;
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  const fe_t ab = fe_mul<S>(a0, b0);
  const fe_t c0 = bind1(lc0, lc2, r);
  C[id] = c0;
  const fe_t c1 = bind1(lc1, lc3, r);
  C[id + q] = c1;
  const fe_t t0e = fe_sub<S>(ab, c0);
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}
// UP2: two batches - A and B first (16 loads), C's four behind the A / B binds
template <int W>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
k_up2(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in, int s, lazy9_t* __restrict__ partials) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const fe_t la0 = A[id], la2 = A[id + 2 * q], la1 = A[id + q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb2 = B[id + 2 * q], lb1 = B[id + q], lb3 = B[id + 3 * q];
  __builtin_amdgcn_sched_barrier(0);
  const fe_t a0 = bind1(la0, la2, r);
  A[id] = a0;
  const fe_t lc0 = C[id], lc2 = C[id + 2 * q], lc1 = C[id + q], lc3 = C[id + 3 * q];
  const fe_t w = eq_in[id & mask];
  __builtin_amdgcn_sched_barrier(0);
  const fe_t a1 = bind1(la1, la3, r);
  A[id + q] = a1;
  const fe_t b0 = bind1(lb0, lb2, r);
  B[id] = b0;
  const fe_t b1 = bind1(lb1, lb3, r);
  B[id + q] = b1;
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  const fe_t ab = fe_mul<S>(a0, b0);
  const fe_t c0 = bind1(lc0, lc2, r);
  C[id] = c0;
  const fe_t c1 = bind1(lc1, lc3, r);
  C[id + q] = c1;
  const fe_t t0e = fe_sub<S>(ab, c0);
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}

// wave sum with DPP moves for the four in-row levels (one VALU instruction each, no LDS crossbar) and shuffles for the two across rows
template <int CTRL>
__device__ __forceinline__ lazy9_t lazy_dpp_step(const lazy9_t& a) {
  lazy9_t o;
#pragma unroll
  for (int i = 0; i < 9; ++i) o.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v[i], CTRL, 0xf, 0xf, false);
  return lazy_add(a, o);
}
__device__ __forceinline__ lazy9_t lazy_wave_sum_dpp(lazy9_t a) {
  a = lazy_dpp_step<0xB1>(a);   // quad_perm [1,0,3,2]
  a = lazy_dpp_step<0x4E>(a);   // quad_perm [2,3,0,1]
  a = lazy_dpp_step<0x124>(a);  // row_ror:4
  a = lazy_dpp_step<0x128>(a);  // row_ror:8
#pragma unroll
  for (int m = 16; m <= 32; m <<= 1) {
    lazy9_t o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.v[i] = __shfl_xor(a.v[i], m, 64);
    a = lazy_add(a, o);
  }
  return a;
}
// W1: the weight's load first (the compiler otherwise issues it behind the stores, where its wait also waits for every store's acknowledgement);
// DPP: the wave sums with DPP moves
template <bool W1, bool DPP>
__global__ void __launch_bounds__(256) k_tail(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in, int s,
                                              lazy9_t* __restrict__ partials) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  fe_t w;
  if (W1) {
    w = eq_in[id & mask];
    __builtin_amdgcn_sched_barrier(0);
  }
  const fe_t la0 = A[id], la1 = A[id + q], la2 = A[id + 2 * q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb1 = B[id + q], lb2 = B[id + 2 * q], lb3 = B[id + 3 * q];
  const fe_t lc0 = C[id], lc1 = C[id + q], lc2 = C[id + 2 * q], lc3 = C[id + 3 * q];
  const fe_t a0 = bind1(la0, la2, r), a1 = bind1(la1, la3, r);
  const fe_t b0 = bind1(lb0, lb2, r), b1 = bind1(lb1, lb3, r);
  const fe_t c0 = bind1(lc0, lc2, r), c1 = bind1(lc1, lc3, r);
  A[id] = a0; A[id + q] = a1; B[id] = b0; B[id + q] = b1; C[id] = c0; C[id + q] = c1;
  if (!W1) w = eq_in[id & mask];
  const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  if (DPP) stream_block_partials(lazy_wave_sum_dpp(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum_dpp(lazy_from(fe_mul<S>(w, tie))), partials);
  else stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}

// STAGGER: the waves of a 2^20 launch are one generation that starts together and walks through the same load / compute phases together; a start
// delay that differs between the blocks sharing a SIMD (SH selects which bits of the block index pick the delay, T its unit in 64-cycle steps)
// lets one group's loads run under another's products
template <int SH, int T>
__global__ void __launch_bounds__(256) k_stagger(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in, int s,
                                                 lazy9_t* __restrict__ partials) {
  const int g = (blockIdx.x >> SH) & 3;
  if (g == 1) __builtin_amdgcn_s_sleep(T);
  else if (g == 2) __builtin_amdgcn_s_sleep(2 * T);
  else if (g == 3) __builtin_amdgcn_s_sleep(3 * T);
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const fe_t la0 = A[id], la1 = A[id + q], la2 = A[id + 2 * q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb1 = B[id + q], lb2 = B[id + 2 * q], lb3 = B[id + 3 * q];
  const fe_t lc0 = C[id], lc1 = C[id + q], lc2 = C[id + 2 * q], lc3 = C[id + 3 * q];
  const fe_t a0 = bind1(la0, la2, r), a1 = bind1(la1, la3, r);
  const fe_t b0 = bind1(lb0, lb2, r), b1 = bind1(lb1, lb3, r);
  const fe_t c0 = bind1(lc0, lc2, r), c1 = bind1(lc1, lc3, r);
  A[id] = a0; A[id + q] = a1; B[id] = b0; B[id + q] = b1; C[id] = c0; C[id + q] = c1;
  const fe_t w = eq_in[id & mask];
  const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}

int main() {
  for (int logL : {20, 22}) {
    const size_t L = (size_t)1 << logL, q = L / 4;
    fe_t *A, *B, *C, *eq, *eo, *part;
    hipMalloc(&A, L * 32); hipMalloc(&B, L * 32); hipMalloc(&C, L * 32);
    hipMalloc(&eq, 1024 * 32); hipMalloc(&eo, ((q >> 10) + 1) * 32); hipMalloc(&part, (q / 64 + 16) * 96);
    hipMemset(A, 0x11, L * 32); hipMemset(B, 0x22, L * 32); hipMemset(C, 0x33, L * 32);
    hipMemset(eq, 0x05, 1024 * 32); hipMemset(eo, 0x07, ((q >> 10) + 1) * 32);
    fe_t r; for (int i = 0; i < 8; ++i) r.v[i] = 0x01234567u * (i + 1);
    const double bytes = 48.0 * L * 3;
    auto report = [&](const char* name, float us) { printf("L=2^%d %-34s %8.1f us  %7.0f GB/s (%.1f%% of 8 TB/s)\n", logL, name, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0); };
    const MailRef nomail{nullptr, nullptr, 0u};
    lazy9_t* lp = reinterpret_cast<lazy9_t*>(part);
    report("library k_bind_eval_cubic<1>", time_us([&] { hipLaunchKernelGGL((k_bind_eval_cubic<1>), dim3((q + 255) / 256), dim3(256), 0, 0, A, B, C, q, r, eq, eo, 10, part, part, 1u, nomail); }, 10));
    report("library k_bind_eval_cubic_stream<1>", time_us([&] { hipLaunchKernelGGL((k_bind_eval_cubic_stream<1, false>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp, nomail); }, 20));
    report("all loads up front, 2 waves/SIMD budget", time_us([&] { hipLaunchKernelGGL((k_up<2>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("all loads up front, 3 waves/SIMD budget", time_us([&] { hipLaunchKernelGGL((k_up<3>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("all loads up front, 4 waves/SIMD budget", time_us([&] { hipLaunchKernelGGL((k_up<4>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("A,B then C, 3 waves/SIMD budget", time_us([&] { hipLaunchKernelGGL((k_up2<3>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("A,B then C, 4 waves/SIMD budget", time_us([&] { hipLaunchKernelGGL((k_up2<4>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("weight loaded first", time_us([&] { hipLaunchKernelGGL((k_tail<true, false>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("DPP wave sums", time_us([&] { hipLaunchKernelGGL((k_tail<false, true>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("weight first + DPP wave sums", time_us([&] { hipLaunchKernelGGL((k_tail<true, true>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stagger blk>>0, 10", time_us([&] { hipLaunchKernelGGL((k_stagger<0, 10>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stagger blk>>3, 10", time_us([&] { hipLaunchKernelGGL((k_stagger<3, 10>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stagger blk>>3, 25", time_us([&] { hipLaunchKernelGGL((k_stagger<3, 25>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stagger blk>>3, 40", time_us([&] { hipLaunchKernelGGL((k_stagger<3, 40>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stagger blk>>8, 25", time_us([&] { hipLaunchKernelGGL((k_stagger<8, 25>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stagger blk>>5, 25", time_us([&] { hipLaunchKernelGGL((k_stagger<5, 25>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("pair split over two half-waves", time_us([&] { hipLaunchKernelGGL(k_split, dim3(q / 128), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stream body, plain stores", time_us([&] { hipLaunchKernelGGL((k_nt<false>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stream body, non-temporal stores", time_us([&] { hipLaunchKernelGGL((k_nt<true>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stream body, 2 chunks per block", time_us([&] { hipLaunchKernelGGL((k_iter<2, false>), dim3(q / 256 / 2), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stream body, 4 chunks per block", time_us([&] { hipLaunchKernelGGL((k_iter<4, false>), dim3(q / 256 / 4), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stream body, 2 chunks, prefetch", time_us([&] { hipLaunchKernelGGL((k_iter<2, true>), dim3(q / 256 / 2), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("stream body, 4 chunks, prefetch", time_us([&] { hipLaunchKernelGGL((k_iter<4, true>), dim3(q / 256 / 4), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20));
    report("1024-thread blocks, in-block weight", time_us([&] { hipLaunchKernelGGL((k_fat<1>), dim3(q / 1024), dim3(1024), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 20));
    report("the same, 2 chunks per block", time_us([&] { hipLaunchKernelGGL((k_fat<2>), dim3(q / 2048), dim3(1024), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 20));
    report("the same, 4 chunks per block", time_us([&] { hipLaunchKernelGGL((k_fat<4>), dim3(q / 4096), dim3(1024), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 20));
    // the second stages as the library launches them since round 5: single-wave blocks, one host result slot each (mapped pinned memory)
    fe_t* slots_h = nullptr;
    fe_t* slots_d = nullptr;
    hipHostMalloc((void**)&slots_h, (SLOT_BASE_ELEM + 4 * HOST_SUM_MAX_BLOCKS + 16) * sizeof(fe_t), hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&slots_d, slots_h, 0);
    const size_t ngroups = (q / 256) >> 2, b2 = (ngroups + 63) / 64 > HOST_SUM_MAX_BLOCKS ? HOST_SUM_MAX_BLOCKS : (ngroups + 63) / 64;
    report("second stage alone (k_sum_partials_lazy<2>)", time_us([&] { hipLaunchKernelGGL((k_sum_partials_lazy<2>), dim3((unsigned)b2), dim3(64), 0, 0, reinterpret_cast<const uint32_t*>(lp), q / 256, 2, eo, slots_d, 7u); }, 20));
    report("256-thread blocks, in-block weight", time_us([&] { hipLaunchKernelGGL(k_w256, dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 20));
    report("second stage of the mid rounds (k_sum_partials), 512 x 2", time_us([&] { hipLaunchKernelGGL(k_sum_partials, dim3(8), dim3(64), 0, 0, part, (size_t)512, 2, slots_d, 7u); }, 20));
    report("second stage of the mid rounds (k_sum_partials), 128 x 2", time_us([&] { hipLaunchKernelGGL(k_sum_partials, dim3(2), dim3(64), 0, 0, part, (size_t)128, 2, slots_d, 7u); }, 20));
    hipHostFree(slots_h);
    report("var0 block256", time_us([&] { hipLaunchKernelGGL((k_var<0, 256>), dim3((q + 255) / 256), dim3(256), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 10));
    report("var1 nontemporal loads", time_us([&] { hipLaunchKernelGGL((k_var<1, 256>), dim3((q + 255) / 256), dim3(256), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 10));
    report("var0 block64 (wave-only reduce)", time_us([&] { hipLaunchKernelGGL((k_var<0, 64>), dim3((q + 63) / 64), dim3(64), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 10));
    report("var2 bind only", time_us([&] { hipLaunchKernelGGL((k_var<2, 256>), dim3((q + 255) / 256), dim3(256), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 10));
    report("var3 bind+eval, no block reduce", time_us([&] { hipLaunchKernelGGL((k_var<3, 256>), dim3((q + 255) / 256), dim3(256), 0, 0, A, B, C, q, r, eq, eo, 10, part); }, 10));
    hipFree(A); hipFree(B); hipFree(C); hipFree(eq); hipFree(eo); hipFree(part);
  }
  return 0;
}
