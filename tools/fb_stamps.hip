// Where the 36-45 us of a round commitment's table walk go (k_fixed_base_rows_coop_mapped<64, 16, true>, 33 scalars): wall-clock stamps of block 0
// between the stages. Table contents are random field elements (the additions are data-independent apart from the identity / doubling cases).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-sched-strategy=max-ilp -Ispartan2_amd/csrc tools/fb_stamps.hip -o tools/fb_stamps
#define SP_KERNEL_STAMPS 1
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "kernels_msm.hpp"

#define CK(x)                                                                \
  do {                                                                       \
    hipError_t e = (x);                                                      \
    if (e != hipSuccess) {                                                   \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), #x, __LINE__); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

// one wave, a chain of dependent products: the latency of a single product when nothing else runs on the SIMD
template <class FP>
__global__ void __launch_bounds__(64) k_chain(const fe_t* in, fe_t* out, int iters, unsigned long long* t) {
  fe_t x = in[threadIdx.x], y = in[64 + threadIdx.x];
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) x = fe_mul<FP>(x, y);
  const unsigned long long t1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
template <class FP>
__global__ void __launch_bounds__(64) k_chain4(const fe_t* in, fe_t* out, int iters, unsigned long long* t) {  // four independent chains per lane
  fe_t x0 = in[threadIdx.x], x1 = in[64 + threadIdx.x], x2 = in[128 + threadIdx.x], x3 = in[192 + threadIdx.x], y = in[256 + threadIdx.x];
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    x0 = fe_mul<FP>(x0, y);
    x1 = fe_mul<FP>(x1, y);
    x2 = fe_mul<FP>(x2, y);
    x3 = fe_mul<FP>(x3, y);
  }
  const unsigned long long t1 = wall_clock64();
  out[threadIdx.x] = fe_add<FP>(fe_add<FP>(x0, x1), fe_add<FP>(x2, x3));
  if (threadIdx.x == 0) t[0] = t1 - t0;
}

// four waves (one per SIMD), dependent products, a stamp every 5 products: does the issue rate change over the life of a short kernel?
// MODE 0: registers only; 1: operands through LDS with a block barrier per product (the cooperative addition's pattern)
template <int MODE>
__global__ void __launch_bounds__(256) k_chain_profile(const fe_t* in, fe_t* out, unsigned long long* t) {
  __shared__ fe_t sh[256];
  fe_t x = in[threadIdx.x], y = in[64 + (threadIdx.x & 63)];
  const int w = threadIdx.x >> 6;
  for (int i = 0; i < 20; ++i) {
    if ((threadIdx.x & 63) == 0) t[w * 20 + i] = wall_clock64();
    for (int k = 0; k < 5; ++k) {
      if (MODE == 1) {
        sh[threadIdx.x] = x;
        __syncthreads();
        x = sh[(threadIdx.x + 64) & 255];
        __syncthreads();
      }
      x = fe_mul<B>(x, y);
    }
  }
  out[threadIdx.x] = x;
}

// one wave, dependent products, only every STRIDE-th lane enabled: the cost of a sparse EXEC mask
template <class FP>
__global__ void __launch_bounds__(64) k_chain_sparse(const fe_t* in, fe_t* out, int iters, int stride, unsigned long long* t) {
  fe_t x = in[threadIdx.x], y = in[64 + threadIdx.x];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x % stride == 0) {
    for (int i = 0; i < iters; ++i) x = fe_mul<FP>(x, y);
  }
  const unsigned long long t1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}

int main() {
  {
    fe_t *din, *dout;
    unsigned long long* dt;
    CK(hipMalloc((void**)&din, 512 * sizeof(fe_t)));
    std::vector<uint64_t> rnd(512 * 4);
    std::mt19937_64 g0(9);
    for (auto& w : rnd) w = g0() & 0x3fffffffffffffffull;
    CK(hipMemcpy(din, rnd.data(), 512 * sizeof(fe_t), hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&dout, 64 * sizeof(fe_t)));
    CK(hipMalloc((void**)&dt, 8));
    unsigned long long t;
    for (int rep = 0; rep < 3; ++rep)
      for (int stride : {1, 2, 4, 8, 16, 64}) {
        hipLaunchKernelGGL((k_chain_sparse<B>), dim3(1), dim3(64), 0, 0, din, dout, 100, stride, dt);
        CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
        printf("sparse EXEC rep %d: %2d of 64 lanes enabled: base field %.3f us per product", rep, 64 / stride, t / 100.0 / 100);
        hipLaunchKernelGGL((k_chain_sparse<spk::SF>), dim3(1), dim3(64), 0, 0, din, dout, 100, stride, dt);
        CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
        printf(", scalar field %.3f us\n", t / 100.0 / 100);
      }
  }
  {
    fe_t *din, *dout;
    unsigned long long* dt;
    CK(hipMalloc((void**)&din, 512 * sizeof(fe_t)));
    CK(hipMemset(din, 0x21, 512 * sizeof(fe_t)));
    CK(hipMalloc((void**)&dout, 256 * sizeof(fe_t)));
    CK(hipMalloc((void**)&dt, 80 * 8));
    unsigned long long t[80];
    for (int mode = 0; mode < 2; ++mode)
      for (int rep = 0; rep < 5; ++rep) {
        if (mode == 0) hipLaunchKernelGGL((k_chain_profile<0>), dim3(1), dim3(256), 0, 0, din, dout, dt);
        else hipLaunchKernelGGL((k_chain_profile<1>), dim3(1), dim3(256), 0, 0, din, dout, dt);
        CK(hipMemcpy(t, dt, sizeof(t), hipMemcpyDeviceToHost));
        printf("chain profile mode %d rep %d, us per product in each group of 5 (wave 0):", mode, rep);
        for (int i = 1; i < 20; ++i) printf(" %.2f", (t[i] - t[i - 1]) / 100.0 / 5);
        printf("  | wave 3:");
        for (int i = 1; i < 20; ++i) printf(" %.2f", (t[60 + i] - t[60 + i - 1]) / 100.0 / 5);
        printf("\n");
      }
  }
  {
    fe_t *din, *dout;
    unsigned long long* dt;
    CK(hipMalloc((void**)&din, 512 * sizeof(fe_t)));
    CK(hipMemset(din, 0x21, 512 * sizeof(fe_t)));
    CK(hipMalloc((void**)&dout, 64 * sizeof(fe_t)));
    CK(hipMalloc((void**)&dt, 8));
    unsigned long long t;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL((k_chain<B>), dim3(1), dim3(64), 0, 0, din, dout, 200, dt);
      CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
      printf("one wave, dependent products, base field: %.3f us per product", t / 100.0 / 200);
      hipLaunchKernelGGL((k_chain<spk::SF>), dim3(1), dim3(64), 0, 0, din, dout, 200, dt);
      CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
      printf(", scalar field %.3f us", t / 100.0 / 200);
      hipLaunchKernelGGL((k_chain4<B>), dim3(1), dim3(64), 0, 0, din, dout, 200, dt);
      CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
      printf("; four independent chains per lane: base %.3f us per product", t / 100.0 / 800);
      hipLaunchKernelGGL((k_chain4<spk::SF>), dim3(1), dim3(64), 0, 0, din, dout, 200, dt);
      CK(hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost));
      printf(", scalar %.3f us\n", t / 100.0 / 800);
    }
  }
  const size_t nmax = 33, per16 = (size_t)16 * 65535;
  std::mt19937_64 g(5);
  std::vector<uint64_t> tab(per16 * 8);
  for (auto& w : tab) w = g() & 0x3fffffffffffffffull;
  aff_t* d_t;
  CK(hipMalloc((void**)&d_t, per16 * sizeof(aff_t)));
  CK(hipMemcpy(d_t, tab.data(), per16 * sizeof(aff_t), hipMemcpyHostToDevice));
  char* h;
  CK(hipHostMalloc((void**)&h, 128 * spk::FB_SLOT_WORDS * 4 + 128 * 32, hipHostMallocMapped));
  memset(h, 0, 128 * spk::FB_SLOT_WORDS * 4 + 128 * 32);
  char* d;
  CK(hipHostGetDevicePointer((void**)&d, h, 0));
  uint64_t* sc = reinterpret_cast<uint64_t*>(h + 128 * spk::FB_SLOT_WORDS * 4);
  for (size_t i = 0; i < 4 * nmax; ++i) sc[i] = g() & 0x0fffffffffffffffull;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 12; ++rep) {
    const unsigned seq = 100 + rep;
    const int variant = 3;  // 0: events + polling, 1: no events, 2: no polling (stream sync), 3: one block, 4: 1 ms pause before the launch
    const bool events = variant == 0, poll = variant != 2;
    const size_t n = variant == 3 ? 4 : 33;
    if (variant == 4) {
      const auto w0 = std::chrono::steady_clock::now();
      while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count() < 1000) {
      }
    }
    for (size_t i = 0; i < 4 * nmax; ++i) sc[i] = g() & 0x0fffffffffffffffull;
    {
      const unsigned pd = rep >= 6 ? 2500u : 0u;  // second half: every wave waits 25 us before it starts
      CK(hipMemcpyToSymbol(HIP_SYMBOL(sp_predelay), &pd, 4));
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (events) CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((spk::k_fixed_base_rows_coop_mapped<64, 16, true>), dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, reinterpret_cast<const fe_t*>(d + 128 * spk::FB_SLOT_WORDS * 4), n, d_t,
                       (size_t)1, reinterpret_cast<unsigned*>(d), seq);
    if (events) CK(hipEventRecord(e1, st));
    volatile unsigned* tag = reinterpret_cast<volatile unsigned*>(h + (n - 1) * spk::FB_SLOT_WORDS * 4) + spk::FB_SLOT_TAG;
    while (poll && *tag != seq) {
    }
    const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipStreamSynchronize(st));
    float ms = 0;
    if (events) CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long s[64];
    CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(sp_stamps), sizeof(s)));
    printf("variant %d rep %d: launch->last slot seen by the host %.1f us, events %.1f us; block 0: scalars read +%.2f, leaves in LDS +%.2f", variant, rep, host_us, ms * 1e3, (s[1] - s[0]) / 100.0,
           (s[2] - s[0]) / 100.0);
    for (int l = 1; l <= 4; ++l) printf(", level %d +%.2f", l, (s[2 + l] - s[0]) / 100.0);
    printf(", slot stored +%.2f us\n", (s[20] - s[0]) / 100.0);
#ifdef SP_KERNEL_STAGE_STAMPS
    {
      unsigned ctr[4], hw[4];
      CK(hipMemcpyFromSymbol(ctr, HIP_SYMBOL(sp_stage_ctr), 16));
      CK(hipMemcpyFromSymbol(hw, HIP_SYMBOL(sp_hwid), 16));
      unsigned long long s2[192];
      CK(hipMemcpyFromSymbol(s2, HIP_SYMBOL(sp_stamps), sizeof(s2)));
      for (int w = 0; w < 4; ++w) {
        printf("       wave %d (hw_id %08x: simd %u cu %u sh %u se %u) arrives at the barriers (us after leaves):", w, hw[w], (hw[w] >> 4) & 3, (hw[w] >> 8) & 15, (hw[w] >> 12) & 1, (hw[w] >> 13) & 7);
        for (unsigned q = ctr[w] - 20; q < ctr[w]; ++q) printf("%s%.2f", (q - (ctr[w] - 20)) % 5 == 0 ? " | " : " ", (double)(long long)(s2[64 + 32 * w + (q & 31)] - s[2]) / 100.0);
        printf("\n");
      }
    }
#endif
    printf("       shader clock per level (cycles / us = MHz):");
    for (int l = 1; l <= 4; ++l) printf(" %llu / %.2f = %.0f", s[32 + 2 + l] - s[32 + 1 + l], (s[2 + l] - s[1 + l]) / 100.0, (s[32 + 2 + l] - s[32 + 1 + l]) / ((s[2 + l] - s[1 + l]) / 100.0));
    printf("\n");
  }
  return 0;
}
