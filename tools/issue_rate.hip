// Issue rates behind the "52-bit limbs through FP64 FMA" question: how many cycles does a SIMD spend per wave64 instruction of
//   v_mad_u64_u32 (the 32 x 32 -> 64 multiply-accumulate every 256-bit product is made of: 64 per product),
//   v_fma_f64     (the FP64 FMA a 52- or 48-bit-limb product would be made of: 2 per limb product + 3 to 5 exact-split / accumulate operations),
//   v_add_co_u32 / v_addc (the carry chains of both forms)?
// One wave per SIMD would under-report (latency); the kernel runs 8 independent chains per lane and enough waves to fill every SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/issue_rate.hip -o tools/issue_rate && tools/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITERS = 4096, CHAINS = 8;

__global__ void k_mad(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x + c;
  uint32_t x = a + threadIdx.x, y = b;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = (uint64_t)x * (uint32_t)(y + c) + acc[c];  // v_mad_u64_u32
    x += 3;
  }
  uint64_t s = 0;
  for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma(double* out, double a, double b) {
  double acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x + c;
  double x = a + threadIdx.x, y = b;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fma(x, y + c, acc[c]);  // v_fma_f64
    x += 1e-9;
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add(uint32_t* out, uint32_t a) {
  uint32_t acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x + c;
  uint32_t x = a + threadIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = acc[c] + (x ^ acc[(c + 1) % CHAINS]);  // v_xor + v_add_u32
    x += 7;
  }
  uint32_t s = 0;
  for (int c = 0; c < CHAINS; ++c) s ^= acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
static double time_ms(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, simds = 4 * cus;
  const double clk = p.clockRate * 1e3;  // Hz
  const int blocks = cus * 8, threads = 256;  // 8 waves per SIMD
  void* buf;
  hipMalloc(&buf, (size_t)blocks * threads * 8);
  const double waves = (double)blocks * threads / 64;
  auto report = [&](const char* name, double ms, double instr_per_iter) {
    const double wave_instr = waves * ITERS * CHAINS * instr_per_iter;
    const double cycles_per = ms * 1e-3 * clk * simds / wave_instr;
    printf("%-22s %8.3f ms  %7.2f cycles of one SIMD per wave64 instruction (%.1f T lane-ops/s)\n", name, ms, cycles_per, wave_instr * 64 / (ms * 1e-3) / 1e12);
  };
  printf("%s, %d CUs, %.0f MHz\n", p.gcnArchName, cus, clk / 1e6);
  report("v_mad_u64_u32", time_ms([&] { hipLaunchKernelGGL(k_mad, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)buf, 12345u, 678u); }), 1);
  report("v_fma_f64", time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, (double*)buf, 1.5, 2.5); }), 1);
  report("v_xor_b32 + v_add_u32", time_ms([&] { hipLaunchKernelGGL(k_add, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)buf, 99u); }), 2);
  hipFree(buf);
  return 0;
}
