// Round-trip latency of the host<->resident-kernel mailbox, two ways:
//   A: the kernel polls a word in mapped pinned HOST memory (what k_sumcheck_tail does today)
//   B: the kernel polls a word in DEVICE memory that the host writes through the PCIe BAR (needs a large-BAR system)
// In both the kernel answers by storing the sequence number to mapped pinned host memory.
// build: hipcc -O3 --offload-arch=gfx950 tools/mailbox_bench.hip -o gpurun_out/mailbox_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_pingpong(const unsigned* mailbox, unsigned* answer, unsigned rounds) {
  if (threadIdx.x) return;
  const long long t0 = wall_clock64();
  for (unsigned want = 1; want <= rounds; ++want) {
    while (__hip_atomic_load(mailbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want)
      if (wall_clock64() - t0 > 400000000ll) return;  // 4 s at 100 MHz
    __hip_atomic_store(answer, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double run(const unsigned* d_mail, volatile unsigned* h_mail, unsigned* d_ans, volatile unsigned* h_ans, unsigned rounds) {
  *h_ans = 0;
  *h_mail = 0;
  _mm_sfence();
  hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, 0, d_mail, d_ans, rounds);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned s = 1; s <= rounds; ++s) {
    *h_mail = s;
    _mm_sfence();
    while (*h_ans != s) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { printf("timeout at %u\n", s); return -1; }
    }
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  hipDeviceSynchronize();
  return us / rounds;
}
int main() {
  int large_bar = 0;
  CK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0));
  printf("large BAR: %d\n", large_bar);
  unsigned *h_pin, *d_pin;
  CK(hipHostMalloc(&h_pin, 4096, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&d_pin, h_pin, 0));
  for (int i = 0; i < 3; ++i) printf("A host-memory mailbox : %.2f us / round trip\n", run(d_pin, h_pin, d_pin + 64, h_pin + 64, 2000));
  if (!large_bar) return 0;
  unsigned* d_fg = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&d_fg, 4096, hipDeviceMallocFinegrained);
  printf("fine-grained device alloc: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 0;
  CK(hipMemset(d_fg, 0, 4096));
  CK(hipDeviceSynchronize());
  for (int i = 0; i < 3; ++i) printf("B device-memory mailbox (host writes over the BAR): %.2f us / round trip\n", run(d_fg, d_fg, d_pin + 64, h_pin + 64, 2000));
  return 0;
}
