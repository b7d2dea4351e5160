// Does a short latency-bound kernel run at full shader clock when the GPU is otherwise idle? A fixed chain of dependent integer multiply-adds on ONE wave,
// timed with the constant-rate wall clock (wall_clock64, 100 MHz) inside the kernel, launched (a) back to back, (b) after idle gaps of 50 us .. 5 ms,
// (c) beside a background kernel that keeps the other CUs busy. Build: hipcc -O3 --offload-arch=gfx950 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void k_chain(unsigned long long* out, int iters) {
  unsigned long long t0 = wall_clock64();
  unsigned x = threadIdx.x + 1, y = 0x9E3779B1u;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) x = x * y + 12345u;  // dependent chain: 16 v_mad_u32_u24-class ops per iteration
  }
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = x;
  }
}
__global__ void k_busy(volatile int* stop, unsigned* sink) {
  unsigned x = threadIdx.x;
  while (!*stop) {
    for (int i = 0; i < 4096; ++i) x = x * 1664525u + 1013904223u;
  }
  if (x == 0x12345u) sink[0] = x;
}

int main() {
  unsigned long long* d;
  hipHostMalloc(&d, 64, hipHostMallocMapped);
  hipStream_t st, bg;
  hipStreamCreate(&st);
  hipStreamCreate(&bg);
  const int iters = 4096;  // 65536 dependent ops
  auto run = [&](int gap_us, int reps) {
    double sum = 0, mn = 1e30;
    for (int r = 0; r < reps; ++r) {
      if (gap_us) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
      hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, st, d, iters);
      hipStreamSynchronize(st);
      double us = d[0] / 100.0;
      sum += us;
      if (us < mn) mn = us;
    }
    printf("  gap %5d us: chain of %d dependent ops: avg %.1f us, min %.1f us  (%.2f ns per op)\n", gap_us, iters * 16, sum / reps, mn, sum / reps * 1e3 / (iters * 16));
  };
  for (int w = 0; w < 50; ++w) run(0, 1);
  printf("idle GPU:\n");
  for (int gap : {0, 50, 200, 1000, 5000, 20000}) run(gap, 40);
  int* stop;
  hipHostMalloc(&stop, 64, hipHostMallocMapped);
  unsigned* sink;
  hipMalloc(&sink, 64);
  for (int blocks : {8, 64, 248}) {
    *stop = 0;
    hipLaunchKernelGGL(k_busy, dim3(blocks), dim3(256), 0, bg, stop, sink);
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    printf("beside a busy kernel on %d blocks:\n", blocks);
    for (int gap : {0, 200, 5000}) run(gap, 40);
    *stop = 1;
    hipStreamSynchronize(bg);
  }
  return 0;
}
