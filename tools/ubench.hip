// Micro-benchmarks that calibrate the design numbers in DESIGN.md: 256-bit modmul throughput on VALU for
// both fields, and streaming-copy bandwidth with the 32-byte-element access pattern the tables use.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ispartan2_amd/csrc tools/ubench.hip -o tools/ubench
#include <cstdio>
#include <vector>

#include "field.hpp"

#define CK(x)                                                      \
  do {                                                             \
    hipError_t e = (x);                                            \
    if (e != hipSuccess) {                                         \
      printf("HIP error %s at %s\n", hipGetErrorString(e), #x);    \
      return 1;                                                    \
    }                                                              \
  } while (0)

template <class FP, int ILP>
__global__ void __launch_bounds__(256) k_mulchain(const fe_t* in, fe_t* out, int iters) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  fe_t x[ILP], y = in[i];
#pragma unroll
  for (int k = 0; k < ILP; ++k) {
    x[k] = in[i + k + 1];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < ILP; ++k) x[k] = fe_mul<FP>(x[k], y);
  }
  fe_t acc = x[0];
#pragma unroll
  for (int k = 1; k < ILP; ++k) acc = fe_add<FP>(acc, x[k]);
  out[i] = acc;
}
template <class FP>
__global__ void __launch_bounds__(256) k_addchain(const fe_t* in, fe_t* out, int iters) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  fe_t x = in[i], y = in[i + 1];
  for (int it = 0; it < iters; ++it) x = fe_add<FP>(x, y);
  out[i] = x;
}
__global__ void __launch_bounds__(256) k_copy32(const fe_t* __restrict__ in, fe_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <class L>
static float time_ms(L&& f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, dev));
  printf("device %s CUs=%d clock=%d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
  const int blocks = p.multiProcessorCount * 8, threads = 256;
  size_t nthreads = (size_t)blocks * threads;
  std::vector<fe_t> h(nthreads + 8);
  for (size_t i = 0; i < h.size(); ++i)
    for (int k = 0; k < 8; ++k) h[i].v[k] = (uint32_t)(i * 2654435761u + k * 40503u) & (k == 7 ? 0x7fffffffu : 0xffffffffu);
  fe_t *d_in, *d_out;
  CK(hipMalloc(&d_in, h.size() * sizeof(fe_t)));
  CK(hipMalloc(&d_out, h.size() * sizeof(fe_t)));
  CK(hipMemcpy(d_in, h.data(), h.size() * sizeof(fe_t), hipMemcpyHostToDevice));
  const int iters = 256;
  {
    float ms = time_ms([&] { hipLaunchKernelGGL((k_mulchain<FqP, 1>), dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters); }, 5);
    printf("Fq modmul ILP1: %.1f Gmul/s\n", (double)nthreads * iters / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_mulchain<FqP, 2>), dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters); }, 5);
    printf("Fq modmul ILP2: %.1f Gmul/s\n", (double)nthreads * iters * 2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_mulchain<FpP, 1>), dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters); }, 5);
    printf("Fp modmul ILP1: %.1f Gmul/s\n", (double)nthreads * iters / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_mulchain<FpP, 2>), dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters); }, 5);
    printf("Fp modmul ILP2: %.1f Gmul/s\n", (double)nthreads * iters * 2 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL((k_addchain<FqP>), dim3(blocks), dim3(threads), 0, 0, d_in, d_out, iters * 4); }, 5);
    printf("Fq modadd: %.1f Gadd/s\n", (double)nthreads * iters * 4 / ms / 1e6);
  }
  for (size_t logn : {20, 24, 26}) {
    size_t n = (size_t)1 << logn;
    fe_t *a, *b;
    CK(hipMalloc(&a, n * sizeof(fe_t)));
    CK(hipMalloc(&b, n * sizeof(fe_t)));
    CK(hipMemset(a, 1, n * sizeof(fe_t)));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_copy32, dim3(4096), dim3(256), 0, 0, a, b, n); }, 10);
    printf("copy 32B/lane  2^%zu elems: %.0f GB/s (read+write)\n", logn, 2.0 * n * 32 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy16, dim3(4096), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n * 2); }, 10);
    printf("copy 16B/lane  2^%zu elems: %.0f GB/s (read+write)\n", logn, 2.0 * n * 32 / ms / 1e6);
    hipFree(a);
    hipFree(b);
  }
  return 0;
}
