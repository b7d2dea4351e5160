// Experiment harness for the read-only streaming reductions (inner round 0 as a dot product of the low halves; the cubic first evaluation): times
// structural variants on tables of 2^20 / 2^22 pairs and prints algorithmic GB/s. Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ispartan2_amd/csrc -Iinclude tools/stream_bench.hip -o tools/stream_bench
#include <cstdio>
#include <vector>

#include "kernels_poly.hpp"

using namespace spk;

// one lazy block partial (single accumulator)
__device__ __forceinline__ void block_partial1(const lazy9_t& s0, lazy9_t* __restrict__ partials) {
  __shared__ lazy9_t sm[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = s0;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = lazy_add(lazy_add(sm[0], sm[1]), lazy_add(sm[2], sm[3]));
}

// V0: PPT loads hoisted, grid covers everything once (what k_eval_quad_stream_lowhi does), one accumulator
template <int PPT>
__global__ void __launch_bounds__(256) k_dot_once(const fe_t* __restrict__ A, const fe_t* __restrict__ B, lazy9_t* __restrict__ partials) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t id0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  fe_t a[PPT], b[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    a[k] = A[id0 + k * stride];
    b[k] = B[id0 + k * stride];
  }
  lazy9_t l0 = lazy_from(fe_zero());
#pragma unroll
  for (int k = 0; k < PPT; ++k) l0 = lazy_add(l0, lazy_from(fe_mul<S>(a[k], b[k])));
  block_partial1(lazy_wave_sum(l0), partials);
}

// V1: persistent grid, BATCH pairs per step, next batch's loads issued before this batch's products (software pipeline)
template <int BATCH>
__global__ void __launch_bounds__(256) k_dot_pipe(const fe_t* __restrict__ A, const fe_t* __restrict__ B, size_t n, lazy9_t* __restrict__ partials) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  fe_t a[BATCH], b[BATCH], na[BATCH], nb[BATCH];
#pragma unroll
  for (int k = 0; k < BATCH; ++k) {
    a[k] = A[id + k * stride];
    b[k] = B[id + k * stride];
  }
  lazy9_t l0 = lazy_from(fe_zero());
  const size_t step = stride * BATCH;
  for (size_t base = id; base < n; base += step) {
    const size_t nxt = base + step;
    if (nxt < n) {
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        na[k] = A[nxt + k * stride];
        nb[k] = B[nxt + k * stride];
      }
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) l0 = lazy_add(l0, lazy_from(fe_mul<S>(a[k], b[k])));
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      a[k] = na[k];
      b[k] = nb[k];
    }
  }
  block_partial1(lazy_wave_sum(l0), partials);
}

// V2: loads only (xor of the words): the read ceiling of this access pattern
template <int PPT>
__global__ void __launch_bounds__(256) k_read_only(const fe_t* __restrict__ A, const fe_t* __restrict__ B, unsigned* __restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t id0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  fe_t a[PPT], b[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    a[k] = A[id0 + k * stride];
    b[k] = B[id0 + k * stride];
  }
  unsigned x = 0;
#pragma unroll
  for (int k = 0; k < PPT; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) x ^= a[k].v[i] ^ b[k].v[i];
  if (x == 0x12345678u) out[0] = x;
}

// cubic first evaluation: V0 = one pair per lane (k_eval_cubic_stream<1>), V1 = two pairs per lane hoisted
template <int PPT>
__global__ void __launch_bounds__(256) k_cubic_once(const fe_t* __restrict__ A, const fe_t* __restrict__ B, const fe_t* __restrict__ C, size_t half,
                                                    const fe_t* __restrict__ eq_in, int s, lazy9_t* __restrict__ partials) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t id0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t mask = ((size_t)1 << s) - 1;
  fe_t a0[PPT], a1[PPT], b0[PPT], b1[PPT], c0[PPT], w[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const size_t id = id0 + k * stride;
    a0[k] = A[id];
    a1[k] = A[id + half];
    b0[k] = B[id];
    b1[k] = B[id + half];
    c0[k] = C[id];
    w[k] = eq_in[id & mask];
  }
  lazy9_t l0 = lazy_from(fe_zero()), l1 = lazy_from(fe_zero());
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const fe_t t0e = fe_sub<S>(fe_mul<S>(a0[k], b0[k]), c0[k]);
    const fe_t tie = fe_mul<S>(fe_sub<S>(a1[k], a0[k]), fe_sub<S>(b1[k], b0[k]));
    l0 = lazy_add(l0, lazy_from(fe_mul<S>(w[k], t0e)));
    l1 = lazy_add(l1, lazy_from(fe_mul<S>(w[k], tie)));
  }
  stream_block_partials(lazy_wave_sum(l0), lazy_wave_sum(l1), partials);
}

#define CK(x)                                                                \
  do {                                                                       \
    hipError_t e_ = (x);                                                     \
    if (e_ != hipSuccess) {                                                  \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
      return 1;                                                              \
    }                                                                        \
  } while (0)

template <typename F>
float time_best(F launch, int reps = 20) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  float best = 1e9f, tot = 0;
  for (int i = 0; i < reps; ++i) {
    (void)hipEventRecord(e0, 0);
    launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
    tot += ms;
  }
  printf("      (avg %.1f us)", tot / reps * 1e3);
  return best * 1e3f;
}

int main() {
  for (int lg : {20, 22}) {
    const size_t half = (size_t)1 << lg, L = 2 * half;
    fe_t *A, *B, *C, *E;
    lazy9_t* P;
    unsigned* O;
    CK(hipMalloc(&A, L * 32));
    CK(hipMalloc(&B, L * 32));
    CK(hipMalloc(&C, L * 32));
    CK(hipMalloc(&E, 4096 * 32));
    CK(hipMalloc(&P, (half / 256) * 2 * sizeof(lazy9_t)));
    CK(hipMalloc(&O, 64));
    std::vector<uint32_t> h(L * 8);
    uint64_t st = 88172645463325252ull;
    for (auto& w : h) {
      st ^= st << 13;
      st ^= st >> 7;
      st ^= st << 17;
      w = (uint32_t)st;
    }
    for (size_t i = 0; i < L; ++i) h[i * 8 + 7] &= 0x7FFFFFFFu;  // < p
    CK(hipMemcpy(A, h.data(), L * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), L * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(C, h.data(), L * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(E, h.data(), 4096 * 32, hipMemcpyHostToDevice));
    const double dot_bytes = 64.0 * half, cub_bytes = 160.0 * half;
    auto rep = [&](const char* name, double bytes, float us) { printf("  2^%d %-34s %7.1f us  %7.1f GB/s  %4.1f %%\n", lg, name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 80.0); };
#define DOT_ONCE(P_) rep("dot once PPT=" #P_, dot_bytes, time_best([&] { hipLaunchKernelGGL((k_dot_once<P_>), dim3((unsigned)(half / 256 / P_)), dim3(256), 0, 0, A, B, P); }))
    DOT_ONCE(1);
    DOT_ONCE(2);
    DOT_ONCE(4);
    DOT_ONCE(8);
#define DOT_PIPE(B_, G_) \
  rep("dot pipe BATCH=" #B_ " blocks=" #G_, dot_bytes, time_best([&] { hipLaunchKernelGGL((k_dot_pipe<B_>), dim3(G_), dim3(256), 0, 0, A, B, half, P); }))
    DOT_PIPE(1, 512);
    DOT_PIPE(1, 1024);
    DOT_PIPE(2, 512);
    DOT_PIPE(2, 1024);
    DOT_PIPE(4, 256);
    DOT_PIPE(4, 512);
    DOT_PIPE(2, 2048);
#define RD(P_) rep("read only PPT=" #P_, dot_bytes, time_best([&] { hipLaunchKernelGGL((k_read_only<P_>), dim3((unsigned)(half / 256 / P_)), dim3(256), 0, 0, A, B, O); }))
    RD(1);
    RD(4);
    RD(8);
#define CUB(P_) \
  rep("cubic once PPT=" #P_, cub_bytes, time_best([&] { hipLaunchKernelGGL((k_cubic_once<P_>), dim3((unsigned)(half / 256 / P_)), dim3(256), 0, 0, A, B, C, half, E, 10, P); }))
    CUB(1);
    CUB(2);
    (void)hipFree(A);
    (void)hipFree(B);
    (void)hipFree(C);
    (void)hipFree(E);
    (void)hipFree(P);
    (void)hipFree(O);
  }
  return 0;
}
