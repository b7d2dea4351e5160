// Experiment harness for the evals_rx outer product (k_eq_outer_lastk: 2^20 outputs = 32 MiB written, one product per output): variants of the block
// shape and of where the 2^K last-variable weights come from. Prints the best of 20 launches (HIP events) and whether the output equals the library's.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ispartan2_amd/csrc -Iinclude tools/eq_bench.hip -o tools/eq_bench
#include <cstdio>
#include <cstring>
#include <vector>

#include "kernels_poly.hpp"

using namespace spk;

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

// floor: the same stores with no arithmetic
template <int BLOCK, int HPB>
__global__ void __launch_bounds__(BLOCK) k_store_only(const fe_t* __restrict__ t_hi, size_t n_hi, fe_t* __restrict__ out) {
  const unsigned t = threadIdx.x + (blockIdx.x % (1024 / BLOCK)) * BLOCK;
  const size_t hb = blockIdx.x / (1024 / BLOCK);
  fe_t v = t_hi[hb & 1023];
  v.v[0] += t;
#pragma unroll
  for (int h = 0; h < HPB; ++h) out[((hb * HPB + h) << 10) + t] = v;
}

// BLOCK threads = a slice of the 1024 low indices; HPB high entries per block; the 2^K weights formed in the block (WB = false) or passed by value
// (WB = true: the host forms them, 14 products)
struct EqW16 {
  fe_t w[16];
};
struct EqR4 {  // the last K coordinates themselves (the library's form until round 5: a block's first 2^K threads formed the weights)
  fe_t r[4];
};
template <int BLOCK, int HPB, bool WB>
__global__ void __launch_bounds__(BLOCK) k_var(const fe_t* __restrict__ t_hi, const fe_t* __restrict__ t_lo, int K, size_t n_hi, EqR4 rk, EqW16 wv,
                                              fe_t* __restrict__ out) {
  __shared__ fe_t wsh[16];
  constexpr unsigned SLICES = 1024 / BLOCK;
  const unsigned t = threadIdx.x + (blockIdx.x % SLICES) * BLOCK;
  const size_t hb = blockIdx.x / SLICES;
  fe_t th[HPB];
#pragma unroll
  for (int h = 0; h < HPB; ++h) th[h] = t_hi[hb * HPB + h];  // issued before the weights: uniform loads
  const fe_t tl = t_lo[t >> K];
  fe_t wk;
  if (WB) {
    wk = wv.w[t & ((1u << K) - 1)];
  } else {
    if (threadIdx.x < (1u << K)) {
      const fe_t one = fe_one<S>();
      fe_t w = one;
      for (int i = 0; i < K; ++i) {
        const fe_t f = ((threadIdx.x >> (K - 1 - i)) & 1u) ? rk.r[i] : fe_sub<S>(one, rk.r[i]);
        w = i == 0 ? f : fe_mul<S>(w, f);
      }
      wsh[threadIdx.x] = w;
    }
    __syncthreads();
    wk = wsh[t & ((1u << K) - 1)];
  }
  const fe_t lo = fe_mul<S>(tl, wk);
#pragma unroll
  for (int h = 0; h < HPB; ++h) out[((hb * HPB + h) << 10) + t] = fe_mul<S>(th[h], lo);
}

// ---- access-pattern floors: a lane's 32-byte element as two 16-byte accesses at a 32-byte lane stride (AoS: what every table kernel does) against
// the same bytes moved with lane-contiguous 16-byte accesses (a wave's instruction covers 1 KiB contiguous)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_store_strided(fe_t* __restrict__ out, unsigned seed) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  v4u* o = reinterpret_cast<v4u*>(out + id);
  const v4u v = {seed, (unsigned)id, seed, seed};
  o[0] = v;
  o[1] = v;
}
__global__ void __launch_bounds__(256) k_store_coalesced(fe_t* __restrict__ out, unsigned seed) {
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63;
  v4u* o = reinterpret_cast<v4u*>(out + wave * 64);  // the wave's 2 KiB
  const v4u v = {seed, lane, seed, seed};
  o[lane] = v;
  o[64 + lane] = v;
}
__global__ void __launch_bounds__(256) k_load_strided(const fe_t* __restrict__ in, unsigned* __restrict__ sink) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  const v4u* p = reinterpret_cast<const v4u*>(in + id);
  const v4u a = p[0], b = p[1];
  if ((a.x ^ b.y) == 0x13579bdfu) sink[0] = a.z;
}
__global__ void __launch_bounds__(256) k_load_coalesced(const fe_t* __restrict__ in, unsigned* __restrict__ sink) {
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63;
  const v4u* p = reinterpret_cast<const v4u*>(in + wave * 64);
  const v4u a = p[lane], b = p[64 + lane];
  if ((a.x ^ b.y) == 0x13579bdfu) sink[0] = a.z;
}
// the bind's traffic without its arithmetic: out[id] = in[id] ^ in[id + half] over one table (read two, write one), strided and coalesced
__global__ void __launch_bounds__(256) k_copy_strided(const fe_t* __restrict__ in, size_t half, fe_t* __restrict__ out) {
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  const v4u* p = reinterpret_cast<const v4u*>(in + id);
  const v4u* q = reinterpret_cast<const v4u*>(in + id + half);
  const v4u a = p[0], b = p[1], c = q[0], d = q[1];
  v4u* o = reinterpret_cast<v4u*>(out + id);
  o[0] = a ^ c;
  o[1] = b ^ d;
}
__global__ void __launch_bounds__(256) k_copy_coalesced(const fe_t* __restrict__ in, size_t half, fe_t* __restrict__ out) {
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63;
  const v4u* p = reinterpret_cast<const v4u*>(in + wave * 64);
  const v4u* q = reinterpret_cast<const v4u*>(in + half + wave * 64);
  const v4u a = p[lane], b = p[64 + lane], c = q[lane], d = q[64 + lane];
  v4u* o = reinterpret_cast<v4u*>(out + wave * 64);
  o[lane] = a ^ c;
  o[64 + lane] = b ^ d;
}
// a wave's 64 consecutive 32-byte elements stored with lane-contiguous 16-byte stores: transposed through the wave's own 2 KiB of LDS (no block
// barrier: LDS operations of a wave execute in order)
__device__ __forceinline__ void wave_store_coalesced(fe_t* __restrict__ wave_out, const fe_t& v, v4u* __restrict__ lds) {
  const unsigned lane = threadIdx.x & 63;
  const v4u lo = {v.v[0], v.v[1], v.v[2], v.v[3]}, hi = {v.v[4], v.v[5], v.v[6], v.v[7]};
  lds[2 * lane] = lo;
  lds[2 * lane + 1] = hi;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const v4u a = lds[lane], b = lds[64 + lane];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  v4u* o = reinterpret_cast<v4u*>(wave_out);
  o[lane] = a;
  o[64 + lane] = b;
}
template <int BLOCK, int HPB>
__global__ void __launch_bounds__(BLOCK) k_var_c(const fe_t* __restrict__ t_hi, const fe_t* __restrict__ t_lo, int K, size_t n_hi, EqW16 wv, fe_t* __restrict__ out) {
  __shared__ v4u lds[BLOCK / 64][128];
  constexpr unsigned SLICES = 1024 / BLOCK;
  const unsigned t = threadIdx.x + (blockIdx.x % SLICES) * BLOCK;
  const size_t hb = blockIdx.x / SLICES;
  fe_t th[HPB];
#pragma unroll
  for (int h = 0; h < HPB; ++h) th[h] = t_hi[hb * HPB + h];
  const fe_t tl = t_lo[t >> K];
  const fe_t lo = fe_mul<S>(tl, wv.w[t & ((1u << K) - 1)]);
  const unsigned t0 = t & ~63u;
#pragma unroll
  for (int h = 0; h < HPB; ++h) wave_store_coalesced(out + ((hb * HPB + h) << 10) + t0, fe_mul<S>(th[h], lo), lds[threadIdx.x >> 6]);
}
__global__ void __launch_bounds__(256) k_cubic_c(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in,
                                                 int s, lazy9_t* __restrict__ partials) {
  __shared__ v4u lds[4][128];
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const fe_t la0 = A[id], la1 = A[id + q], la2 = A[id + 2 * q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb1 = B[id + q], lb2 = B[id + 2 * q], lb3 = B[id + 3 * q];
  const fe_t lc0 = C[id], lc1 = C[id + q], lc2 = C[id + 2 * q], lc3 = C[id + 3 * q];
  const fe_t a0 = bind1(la0, la2, r), a1 = bind1(la1, la3, r);
  const fe_t b0 = bind1(lb0, lb2, r), b1 = bind1(lb1, lb3, r);
  const fe_t c0 = bind1(lc0, lc2, r), c1 = bind1(lc1, lc3, r);
  const size_t w0 = id & ~(size_t)63;
  v4u* l = lds[threadIdx.x >> 6];
  wave_store_coalesced(A + w0, a0, l);
  wave_store_coalesced(A + w0 + q, a1, l);
  wave_store_coalesced(B + w0, b0, l);
  wave_store_coalesced(B + w0 + q, b1, l);
  wave_store_coalesced(C + w0, c0, l);
  wave_store_coalesced(C + w0 + q, c1, l);
  const fe_t w = eq_in[id & mask];
  const fe_t t0e = fe_sub<S>(fe_mul<S>(a0, b0), c0);
  const fe_t tie = fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0));
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}
// N elements of a wave stored with lane-contiguous 16-byte stores in ONE LDS round trip (N x 2 KiB of LDS per wave): all writes, one wait, all reads, one
// wait, all global stores
template <int N>
__device__ __forceinline__ void wave_store_coalesced_n(fe_t* const (&wave_out)[N], const fe_t (&v)[N], v4u* __restrict__ lds) {
  const unsigned lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const v4u lo = {v[k].v[0], v[k].v[1], v[k].v[2], v[k].v[3]}, hi = {v[k].v[4], v[k].v[5], v[k].v[6], v[k].v[7]};
    lds[128 * k + 2 * lane] = lo;
    lds[128 * k + 2 * lane + 1] = hi;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  v4u a[N], b[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    a[k] = lds[128 * k + lane];
    b[k] = lds[128 * k + 64 + lane];
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    v4u* o = reinterpret_cast<v4u*>(wave_out[k]);
    o[lane] = a[k];
    o[64 + lane] = b[k];
  }
}
__global__ void __launch_bounds__(256) k_quad_sparse_c(fe_t* __restrict__ A, fe_t* __restrict__ B, size_t q, fe_t r, size_t hiA, size_t hiB, lazy9_t* __restrict__ partials) {
  __shared__ v4u lds[4][4 * 128];
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const fe_t la0 = A[id], la1 = A[id + q], lb0 = B[id], lb1 = B[id + q];
  const fe_t one_minus_r = fe_sub<S>(fe_one<S>(), r);
  fe_t o[4];
  o[0] = id < hiA ? bind1(la0, A[id + 2 * q], r) : fe_mul<S>(la0, one_minus_r);
  o[2] = id < hiB ? bind1(lb0, B[id + 2 * q], r) : fe_mul<S>(lb0, one_minus_r);
  o[1] = fe_mul<S>(la1, one_minus_r);
  o[3] = fe_mul<S>(lb1, one_minus_r);
  const size_t w0 = id & ~(size_t)63;
  fe_t* const outs[4] = {A + w0, A + w0 + q, B + w0, B + w0 + q};
  wave_store_coalesced_n<4>(outs, o, lds[threadIdx.x >> 6]);
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(o[0], o[2]))), lazy_wave_sum(lazy_from(fe_mul<S>(fe_sub<S>(o[1], o[0]), fe_sub<S>(o[3], o[2])))), partials);
}
__global__ void __launch_bounds__(256) k_quad_c(fe_t* __restrict__ A, fe_t* __restrict__ B, size_t q, fe_t r, lazy9_t* __restrict__ partials) {
  __shared__ v4u lds[4][4 * 128];
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const fe_t la0 = A[id], la1 = A[id + q], la2 = A[id + 2 * q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb1 = B[id + q], lb2 = B[id + 2 * q], lb3 = B[id + 3 * q];
  fe_t o[4];
  o[0] = bind1(la0, la2, r);
  o[1] = bind1(la1, la3, r);
  o[2] = bind1(lb0, lb2, r);
  o[3] = bind1(lb1, lb3, r);
  const size_t w0 = id & ~(size_t)63;
  fe_t* const outs[4] = {A + w0, A + w0 + q, B + w0, B + w0 + q};
  wave_store_coalesced_n<4>(outs, o, lds[threadIdx.x >> 6]);
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(o[0], o[2]))), lazy_wave_sum(lazy_from(fe_mul<S>(fe_sub<S>(o[1], o[0]), fe_sub<S>(o[3], o[2])))), partials);
}
__global__ void __launch_bounds__(256) k_cubic_c6(fe_t* __restrict__ A, fe_t* __restrict__ B, fe_t* __restrict__ C, size_t q, fe_t r, const fe_t* __restrict__ eq_in,
                                                  int s, lazy9_t* __restrict__ partials) {
  __shared__ v4u lds[4][6 * 128];
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  const fe_t la0 = A[id], la1 = A[id + q], la2 = A[id + 2 * q], la3 = A[id + 3 * q];
  const fe_t lb0 = B[id], lb1 = B[id + q], lb2 = B[id + 2 * q], lb3 = B[id + 3 * q];
  const fe_t lc0 = C[id], lc1 = C[id + q], lc2 = C[id + 2 * q], lc3 = C[id + 3 * q];
  fe_t o[6];
  o[0] = bind1(la0, la2, r);
  o[1] = bind1(la1, la3, r);
  o[2] = bind1(lb0, lb2, r);
  o[3] = bind1(lb1, lb3, r);
  o[4] = bind1(lc0, lc2, r);
  o[5] = bind1(lc1, lc3, r);
  const size_t w0 = id & ~(size_t)63;
  fe_t* const outs[6] = {A + w0, A + w0 + q, B + w0, B + w0 + q, C + w0, C + w0 + q};
  wave_store_coalesced_n<6>(outs, o, lds[threadIdx.x >> 6]);
  const fe_t w = eq_in[id & mask];
  const fe_t t0e = fe_sub<S>(fe_mul<S>(o[0], o[2]), o[4]);
  const fe_t tie = fe_mul<S>(fe_sub<S>(o[1], o[0]), fe_sub<S>(o[3], o[2]));
  stream_block_partials(lazy_wave_sum(lazy_from(fe_mul<S>(w, t0e))), lazy_wave_sum(lazy_from(fe_mul<S>(w, tie))), partials);
}
// U units of 256 ids per block, every unit's eight loads issued before the first product: unit u's products run while unit u + 1's loads are still in
// flight and its stores drain under unit u + 1's products (one generation of fewer, fatter waves)
template <int U, bool FORCE = false>
__global__ void __launch_bounds__(256) k_quad_units(fe_t* __restrict__ A, fe_t* __restrict__ B, size_t q, fe_t r, lazy9_t* __restrict__ partials) {
  const size_t id0 = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  fe_t la[U][4], lb[U][4];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t id = id0 + 256 * (size_t)u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      la[u][k] = A[id + k * q];
      lb[u][k] = B[id + k * q];
    }
  }
  if (FORCE) __builtin_amdgcn_sched_barrier(0);  // the scheduler sinks the later units' loads below the first products otherwise
  lazy9_t l0 = lazy_from(fe_zero()), l1 = lazy_from(fe_zero());
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t id = id0 + 256 * (size_t)u;
    const fe_t a0 = bind1(la[u][0], la[u][2], r), a1 = bind1(la[u][1], la[u][3], r);
    const fe_t b0 = bind1(lb[u][0], lb[u][2], r), b1 = bind1(lb[u][1], lb[u][3], r);
    A[id] = a0;
    A[id + q] = a1;
    B[id] = b0;
    B[id + q] = b1;
    l0 = lazy_add(l0, lazy_from(fe_mul<S>(a0, b0)));
    l1 = lazy_add(l1, lazy_from(fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0))));
  }
  stream_block_partials(lazy_wave_sum(l0), lazy_wave_sum(l1), partials);
}
// The same two units with the loads as inline asm (the compiler cannot sink them) and hand-placed waits: all 32 loads of a lane issued first, unit 0's
// products under unit 1's loads, unit 0's stores under unit 1's products. 128 VGPRs of loaded data: 2 waves per SIMD, 512 blocks = one generation.
__device__ __forceinline__ v4u gload16(const void* p) {
  v4u r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ v4u gload16_hi(const void* p) {
  v4u r;
  asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ fe_t fe_of(const v4u& lo, const v4u& hi) {
  fe_t f;
  f.v[0] = lo.x; f.v[1] = lo.y; f.v[2] = lo.z; f.v[3] = lo.w;
  f.v[4] = hi.x; f.v[5] = hi.y; f.v[6] = hi.z; f.v[7] = hi.w;
  return f;
}
__global__ void __launch_bounds__(256) k_quad_pipe(fe_t* __restrict__ A, fe_t* __restrict__ B, size_t q, fe_t r, lazy9_t* __restrict__ partials) {
  const size_t id0 = (size_t)blockIdx.x * 512 + threadIdx.x;
  v4u lo[2][8], hi[2][8];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const size_t id = id0 + 256 * (size_t)u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lo[u][k] = gload16(A + id + k * q);
      hi[u][k] = gload16_hi(A + id + k * q);
      lo[u][4 + k] = gload16(B + id + k * q);
      hi[u][4 + k] = gload16_hi(B + id + k * q);
    }
  }
  lazy9_t l0 = lazy_from(fe_zero()), l1 = lazy_from(fe_zero());
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    // unit 0: the 16 loads of unit 1 may still be out; unit 1: everything (its loads and unit 0's stores, which count in vmcnt as well)
    if (u == 0) {
      asm volatile("s_waitcnt vmcnt(16)" : "+v"(lo[0][0]), "+v"(lo[0][1]), "+v"(lo[0][2]), "+v"(lo[0][3]), "+v"(lo[0][4]), "+v"(lo[0][5]), "+v"(lo[0][6]), "+v"(lo[0][7]));
      asm volatile("" : "+v"(hi[0][0]), "+v"(hi[0][1]), "+v"(hi[0][2]), "+v"(hi[0][3]), "+v"(hi[0][4]), "+v"(hi[0][5]), "+v"(hi[0][6]), "+v"(hi[0][7]));
    } else {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(lo[1][0]), "+v"(lo[1][1]), "+v"(lo[1][2]), "+v"(lo[1][3]), "+v"(lo[1][4]), "+v"(lo[1][5]), "+v"(lo[1][6]), "+v"(lo[1][7]));
      asm volatile("" : "+v"(hi[1][0]), "+v"(hi[1][1]), "+v"(hi[1][2]), "+v"(hi[1][3]), "+v"(hi[1][4]), "+v"(hi[1][5]), "+v"(hi[1][6]), "+v"(hi[1][7]));
    }
    const size_t id = id0 + 256 * (size_t)u;
    const fe_t a0 = bind1(fe_of(lo[u][0], hi[u][0]), fe_of(lo[u][2], hi[u][2]), r), a1 = bind1(fe_of(lo[u][1], hi[u][1]), fe_of(lo[u][3], hi[u][3]), r);
    const fe_t b0 = bind1(fe_of(lo[u][4], hi[u][4]), fe_of(lo[u][6], hi[u][6]), r), b1 = bind1(fe_of(lo[u][5], hi[u][5]), fe_of(lo[u][7], hi[u][7]), r);
    A[id] = a0;
    A[id + q] = a1;
    B[id] = b0;
    B[id + q] = b1;
    l0 = lazy_add(l0, lazy_from(fe_mul<S>(a0, b0)));
    l1 = lazy_add(l1, lazy_from(fe_mul<S>(fe_sub<S>(a1, a0), fe_sub<S>(b1, b0))));
  }
  stream_block_partials(lazy_wave_sum(l0), lazy_wave_sum(l1), partials);
}
static void quads() {
  // the inner sum-check's two streaming binds at config 2: tables of 2^21 (first bind, high halves zero beyond hi) and 2^20
  const size_t L = (size_t)1 << 21;
  fe_t *A, *B, *part;
  hipMalloc(&A, L * 32); hipMalloc(&B, L * 32); hipMalloc(&part, (L / 4 / 64 + 16) * 96);
  hipMemset(A, 0x11, L * 32); hipMemset(B, 0x22, L * 32);
  fe_t r; for (int i = 0; i < 8; ++i) r.v[i] = 0x01234567u * (i + 1);
  lazy9_t* lp = reinterpret_cast<lazy9_t*>(part);
  const MailRef nomail{nullptr, nullptr, 0u};
  for (int rep = 0; rep < 2; ++rep) {
    { const size_t q = L / 4; const double bytes = 32.0 * 8 * q;  // reads 4 q, writes 4 q elements
      float u1 = time_us([&] { hipLaunchKernelGGL(k_bind_eval_quad_stream_sparse, dim3(q / 256), dim3(256), 0, 0, A, B, q, r, (size_t)1000, (size_t)300, lp, nomail); }, 20);
      float u2 = time_us([&] { hipLaunchKernelGGL(k_quad_sparse_c, dim3(q / 256), dim3(256), 0, 0, A, B, q, r, (size_t)1000, (size_t)300, lp); }, 20);
      printf("quad sparse 2^21: library %6.1f us (%5.0f GB/s)   one-trip coalesced stores %6.1f us (%5.0f GB/s)\n", u1, bytes / u1 / 1e3, u2, bytes / u2 / 1e3); }
    { const size_t q = L / 8; const double bytes = 48.0 * 4 * q * 2;
      float u1 = time_us([&] { hipLaunchKernelGGL(k_bind_eval_quad_stream, dim3(q / 256), dim3(256), 0, 0, A, B, q, r, lp, nomail); }, 20);
      float u2 = time_us([&] { hipLaunchKernelGGL(k_quad_c, dim3(q / 256), dim3(256), 0, 0, A, B, q, r, lp); }, 20);
      float u3 = time_us([&] { hipLaunchKernelGGL(k_quad_units<2>, dim3(q / 512), dim3(256), 0, 0, A, B, q, r, lp); }, 20);
      float u4 = time_us([&] { hipLaunchKernelGGL(k_quad_units<3>, dim3(q / 768), dim3(256), 0, 0, A, B, q, r, lp); }, 20);
      float u5 = time_us([&] { hipLaunchKernelGGL(k_quad_units<4>, dim3(q / 1024), dim3(256), 0, 0, A, B, q, r, lp); }, 20);
      float u6 = time_us([&] { hipLaunchKernelGGL((k_quad_units<2, true>), dim3(q / 512), dim3(256), 0, 0, A, B, q, r, lp); }, 20);
      float u7 = time_us([&] { hipLaunchKernelGGL((k_quad_units<4, true>), dim3(q / 1024), dim3(256), 0, 0, A, B, q, r, lp); }, 20);
      {  // equality of the asm-pipelined form with the library's on equal inputs, then its time
        fe_t *A2, *B2;
        hipMalloc(&A2, L * 32); hipMalloc(&B2, L * 32);
        hipMemset(A, 0x11, L * 32); hipMemset(B, 0x22, L * 32); hipMemset(A2, 0x11, L * 32); hipMemset(B2, 0x22, L * 32);
        hipLaunchKernelGGL(k_bind_eval_quad_stream, dim3(q / 256), dim3(256), 0, 0, A, B, q, r, lp, nomail);
        hipLaunchKernelGGL(k_quad_pipe, dim3(q / 512), dim3(256), 0, 0, A2, B2, q, r, lp);
        std::vector<char> x(2 * q * 32), y(2 * q * 32);
        hipMemcpy(x.data(), A, 2 * q * 32, hipMemcpyDeviceToHost); hipMemcpy(y.data(), A2, 2 * q * 32, hipMemcpyDeviceToHost);
        bool same = memcmp(x.data(), y.data(), 2 * q * 32) == 0;
        hipMemcpy(x.data(), B, 2 * q * 32, hipMemcpyDeviceToHost); hipMemcpy(y.data(), B2, 2 * q * 32, hipMemcpyDeviceToHost);
        same = same && memcmp(x.data(), y.data(), 2 * q * 32) == 0;
        float u8 = time_us([&] { hipLaunchKernelGGL(k_quad_pipe, dim3(q / 512), dim3(256), 0, 0, A, B, q, r, lp); }, 20);
        printf("quad 2^20: two units, asm loads up front + hand-placed waits %6.1f us  %s\n", u8, same ? "equal" : "DIFFERENT");
        hipFree(A2); hipFree(B2);
      }
      printf("quad 2^20:        library %6.1f us (%5.0f GB/s)   one-trip coalesced stores %6.1f us (%5.0f GB/s)   2 / 3 / 4 units per block %6.1f / %6.1f / %6.1f us; loads forced up front, 2 / 4 units %6.1f / %6.1f us\n", u1, bytes / u1 / 1e3, u2, bytes / u2 / 1e3, u3, u4, u5, u6, u7); }
  }
  hipFree(A); hipFree(B); hipFree(part);
}
static void cubic() {
  for (int logL : {20, 22}) {
    const size_t L = (size_t)1 << logL, q = L / 4;
    fe_t *A, *B, *C, *A2, *B2, *C2, *eq, *part;
    hipMalloc(&A, L * 32); hipMalloc(&B, L * 32); hipMalloc(&C, L * 32);
    hipMalloc(&A2, L * 32); hipMalloc(&B2, L * 32); hipMalloc(&C2, L * 32);
    hipMalloc(&eq, 1024 * 32); hipMalloc(&part, (q / 64 + 16) * 96);
    hipMemset(eq, 0x05, 1024 * 32);
    fe_t r; for (int i = 0; i < 8; ++i) r.v[i] = 0x01234567u * (i + 1);
    auto fill = [&] { hipMemset(A, 0x11, L * 32); hipMemset(B, 0x22, L * 32); hipMemset(C, 0x33, L * 32); hipMemset(A2, 0x11, L * 32); hipMemset(B2, 0x22, L * 32); hipMemset(C2, 0x33, L * 32); };
    lazy9_t* lp = reinterpret_cast<lazy9_t*>(part);
    const MailRef nomail{nullptr, nullptr, 0u};
    // equality of one launch on equal inputs
    fill();
    hipLaunchKernelGGL((k_bind_eval_cubic_stream<1, false>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp, nomail);
    hipLaunchKernelGGL(k_cubic_c, dim3(q / 256), dim3(256), 0, 0, A2, B2, C2, q, r, eq, 10, lp);
    std::vector<char> x(L * 16), y(L * 16);
    bool same = true;
    for (auto pr : {std::make_pair(A, A2), std::make_pair(B, B2), std::make_pair(C, C2)}) {
      hipMemcpy(x.data(), pr.first, L * 16, hipMemcpyDeviceToHost);
      hipMemcpy(y.data(), pr.second, L * 16, hipMemcpyDeviceToHost);
      same = same && memcmp(x.data(), y.data(), L * 16) == 0;
    }
    const double bytes = 48.0 * L * 3;
    for (int rep = 0; rep < 2; ++rep) {
      float u1 = time_us([&] { hipLaunchKernelGGL((k_bind_eval_cubic_stream<1, false>), dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp, nomail); }, 20);
      float u2 = time_us([&] { hipLaunchKernelGGL(k_cubic_c, dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20);
      float u3 = time_us([&] { hipLaunchKernelGGL(k_cubic_c6, dim3(q / 256), dim3(256), 0, 0, A, B, C, q, r, eq, 10, lp); }, 20);
      printf("L=2^%d cubic stream: library %6.1f us (%5.0f GB/s)   coalesced stores %6.1f us (%5.0f GB/s)  %s   one LDS trip for the six %6.1f us\n", logL, u1, bytes / u1 / 1e3, u2, bytes / u2 / 1e3, same ? "equal" : "DIFFERENT", u3);
    }
    hipFree(A); hipFree(B); hipFree(C); hipFree(A2); hipFree(B2); hipFree(C2); hipFree(eq); hipFree(part);
  }
}
static void floors() {
  for (int lg : {20, 21, 23}) {
    const size_t n = (size_t)1 << lg;
    fe_t *in, *out;
    unsigned* sink;
    hipMalloc(&in, n * 32);
    hipMalloc(&out, n * 32);
    hipMalloc(&sink, 64);
    hipMemset(in, 0x5a, n * 32);
    auto rep = [&](const char* name, double bytes, float us) { printf("2^%d %-40s %7.1f us  %6.0f GB/s\n", lg, name, us, bytes / us / 1e3); };
    rep("store 32 B/lane, strided (AoS)", 32.0 * n, time_us([&] { hipLaunchKernelGGL(k_store_strided, dim3((unsigned)(n / 256)), dim3(256), 0, 0, out, 7u); }, 20));
    rep("store 2 x 16 B/lane, lane-contiguous", 32.0 * n, time_us([&] { hipLaunchKernelGGL(k_store_coalesced, dim3((unsigned)(n / 256)), dim3(256), 0, 0, out, 7u); }, 20));
    rep("load 32 B/lane, strided (AoS)", 32.0 * n, time_us([&] { hipLaunchKernelGGL(k_load_strided, dim3((unsigned)(n / 256)), dim3(256), 0, 0, in, sink); }, 20));
    rep("load 2 x 16 B/lane, lane-contiguous", 32.0 * n, time_us([&] { hipLaunchKernelGGL(k_load_coalesced, dim3((unsigned)(n / 256)), dim3(256), 0, 0, in, sink); }, 20));
    rep("read 2 write 1 (bind traffic), strided", 48.0 * n, time_us([&] { hipLaunchKernelGGL(k_copy_strided, dim3((unsigned)(n / 512)), dim3(256), 0, 0, in, n / 2, out); }, 20));
    rep("read 2 write 1 (bind traffic), lane-contiguous", 48.0 * n, time_us([&] { hipLaunchKernelGGL(k_copy_coalesced, dim3((unsigned)(n / 512)), dim3(256), 0, 0, in, n / 2, out); }, 20));
    hipFree(in);
    hipFree(out);
    hipFree(sink);
  }
}

int main() {
  floors();
  cubic();
  quads();
  const int ell = 20, K = 4, hi_bits = 10;
  const size_t n_hi = (size_t)1 << hi_bits, total = (size_t)1 << ell;
  fe_t *thi, *tlo, *out, *ref;
  hipMalloc(&thi, n_hi * 32);
  hipMalloc(&tlo, 64 * 32);
  hipMalloc(&out, total * 32);
  hipMalloc(&ref, total * 32);
  std::vector<fe_t> h(n_hi);
  for (size_t i = 0; i < n_hi; ++i)
    for (int w = 0; w < 8; ++w) h[i].v[w] = (uint32_t)(0x9e3779b9u * (i * 8 + w + 1)) >> (w == 7 ? 2 : 0);
  hipMemcpy(thi, h.data(), n_hi * 32, hipMemcpyHostToDevice);
  hipMemcpy(tlo, h.data() + 100, 64 * 32, hipMemcpyHostToDevice);
  EqR4 rk;
  for (int i = 0; i < 4; ++i) rk.r[i] = h[200 + i];
  EqW16 wv;  // host-formed weights (Montgomery products on the host side of field.hpp)
  for (unsigned t = 0; t < 16; ++t) {
    const fe_t one = fe_one<S>();
    fe_t w = one;
    for (int i = 0; i < K; ++i) {
      const fe_t f = ((t >> (K - 1 - i)) & 1u) ? rk.r[i] : fe_sub<S>(one, rk.r[i]);
      w = i == 0 ? f : fe_mul<S>(w, f);
    }
    wv.w[t] = w;
  }
  const double bytes = 32.0 * total;
  std::vector<fe_t> a(total), b(total);
  auto report = [&](const char* name, float us, bool check) {
    bool same = true;
    if (check) {
      hipMemcpy(b.data(), out, total * 32, hipMemcpyDeviceToHost);
      same = memcmp(a.data(), b.data(), total * 32) == 0;
    }
    printf("%-52s %7.1f us  %6.0f GB/s  %s\n", name, us, bytes / us / 1e3, check ? (same ? "equal" : "DIFFERENT") : "");
    hipMemset(out, 0, total * 32);
  };
  EqLastK lk;
  for (int t = 0; t < 16; ++t) lk.w[t] = wv.w[t];
  float us = time_us([&] { hipLaunchKernelGGL(k_eq_outer_lastk, dim3((unsigned)(n_hi / EQ_LASTK_HPB * (1024 / EQ_LASTK_BLOCK))), dim3(EQ_LASTK_BLOCK), 0, 0, thi, tlo, K, n_hi, lk, ref); }, 20);
  hipMemcpy(a.data(), ref, total * 32, hipMemcpyDeviceToHost);
  printf("%-52s %7.1f us  %6.0f GB/s\n", "library k_eq_outer_lastk (512 x 8, weights by value)", us, bytes / us / 1e3);
#define RUN(B, H, W)                                                                                                                                 \
  report("block " #B ", " #H " high/block, weights " #W,                                                                                             \
         time_us([&] { hipLaunchKernelGGL((k_var<B, H, W>), dim3((unsigned)(n_hi / H * (1024 / B))), dim3(B), 0, 0, thi, tlo, K, n_hi, rk, wv, out); }, 20), \
         true)
  RUN(1024, 4, false);
  RUN(1024, 4, true);
  RUN(1024, 2, true);
  RUN(1024, 8, true);
  RUN(512, 4, true);
  RUN(512, 8, true);
  RUN(256, 4, false);
  RUN(256, 4, true);
  RUN(256, 8, true);
  RUN(256, 16, true);
  RUN(256, 2, true);
  RUN(256, 1, true);
#define RUNC(B, H) \
  report("coalesced stores, block " #B ", " #H " high/block", time_us([&] { hipLaunchKernelGGL((k_var_c<B, H>), dim3((unsigned)(n_hi / H * (1024 / B))), dim3(B), 0, 0, thi, tlo, K, n_hi, wv, out); }, 20), true)
  RUNC(1024, 4);
  RUNC(512, 4);
  RUNC(512, 8);
  RUNC(256, 4);
  RUNC(256, 8);
#define FLOOR(B, H) \
  report("stores only, block " #B ", " #H " high/block", time_us([&] { hipLaunchKernelGGL((k_store_only<B, H>), dim3((unsigned)(n_hi / H * (1024 / B))), dim3(B), 0, 0, thi, n_hi, out); }, 20), false)
  FLOOR(1024, 4);
  FLOOR(256, 4);
  FLOOR(256, 1);
  FLOOR(256, 16);
  return 0;
}
