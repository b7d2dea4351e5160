// FETCH_SIZE / WRITE_SIZE calibration on known byte counts in the access patterns of k_polyabc_short_and_long (MI355X_MICROARCH.md:
// "calibrate on a known byte count in your own access pattern before trusting an absolute"). Each kernel below moves a byte count that is known exactly;
// run once under `rocprofv3 --pmc FETCH_SIZE` and once under `--pmc WRITE_SIZE` (tools/calib_job.sh) and divide.
//   k_stream16      coalesced 16 B / lane streaming read of S bytes (the guide's case: FETCH_SIZE reports S / 2)
//   k_gather32      G random 32-byte element gathers per lane from a table of T bytes (the walk's x[idx] loads; T = 32 MiB is evals_rx at config 2,
//                   T = 1 GiB is past the Infinity Cache); indices are a multiplicative hash of (lane, step): no two gathers of a launch share a line
//                   on purpose when T is large, so the true traffic is G * lanes * (line granule)
//   k_gather32_sorted  the same gathers with every lane's indices increasing by a small random stride (a column's row indices are sorted, and the 64 columns of a
//                   wave are unrelated): the walk's own distribution
//   k_lane_streams  every lane walks its own contiguous run of 4-byte words (the index list of its column): 4 B per step and lane, lanes far apart
//   k_store32       coalesced 32-byte stores of W bytes (WRITE_SIZE)
// Output: one line per kernel with the bytes it moved; tools/calib_report.py joins them with the counter CSVs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct el32 {
  uint4 a, b;
};

__global__ void __launch_bounds__(256) k_stream16(const uint4* __restrict__ src, size_t n16, uint4* __restrict__ sink) {
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc.x ^= v.x;
    acc.y ^= v.y;
    acc.z ^= v.z;
    acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;
}
// coalesced 4 B / lane (a wave reads 256 contiguous bytes per instruction) and coalesced 32 B / lane as the library reads its tables (two dwordx4 per lane,
// a wave reads 2 KiB contiguous): is the halving a property of the request size the L2 sends to the fabric, or of the load width?
__global__ void __launch_bounds__(256) k_stream4(const unsigned* __restrict__ src, size_t n4, uint4* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc ^= src[i];
  if (acc == 0x12345678u) sink[0] = make_uint4(acc, 0, 0, 0);
}
__global__ void __launch_bounds__(256) k_stream32(const el32* __restrict__ src, size_t n32, uint4* __restrict__ sink) {
  uint4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (size_t)gridDim.x * blockDim.x) {
    const el32 v = src[i];
    acc.x ^= v.a.x ^ v.b.x;
    acc.y ^= v.a.y ^ v.b.w;
  }
  if ((acc.x ^ acc.y) == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_gather32(const el32* __restrict__ tab, size_t nel, unsigned steps, uint4* __restrict__ sink) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  for (unsigned s = 0; s < steps; ++s) {
    const size_t h = (t * 0x9E3779B97F4A7C15ull + (size_t)s * 0xD1B54A32D192ED03ull) >> 17;
    const el32 v = tab[h % nel];
    acc.x ^= v.a.x ^ v.b.x;
    acc.y ^= v.a.y ^ v.b.y;
  }
  if ((acc.x ^ acc.y) == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_gather32_sorted(const el32* __restrict__ tab, size_t nel, unsigned steps, unsigned max_stride, uint4* __restrict__ sink) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  size_t idx = ((t * 0x9E3779B97F4A7C15ull) >> 20) % nel;
  unsigned long long r = t * 0xD1B54A32D192ED03ull + 1;
  for (unsigned s = 0; s < steps; ++s) {
    const el32 v = tab[idx];
    acc.x ^= v.a.x ^ v.b.x;
    acc.y ^= v.a.y ^ v.b.y;
    r = r * 6364136223846793005ull + 1442695040888963407ull;
    idx += 1 + (size_t)((r >> 33) % max_stride);
    if (idx >= nel) idx -= nel;
  }
  if ((acc.x ^ acc.y) == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) k_lane_streams(const unsigned* __restrict__ words, size_t run_words, unsigned steps, uint4* __restrict__ sink) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned* p = words + t * run_words;
  unsigned acc = 0;
  for (unsigned s = 0; s < steps; ++s) acc ^= p[s];
  if (acc == 0x12345678u) sink[0] = make_uint4(acc, 0, 0, 0);
}
__global__ void __launch_bounds__(256) k_store32(el32* __restrict__ dst, size_t nel) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  el32 v;
  v.a = make_uint4(1, 2, 3, 4);
  v.b = make_uint4(5, 6, 7, 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nel; i += stride) dst[i] = v;
}

int main() {
  const size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;
  char* big = nullptr;
  uint4* sink = nullptr;
  CK(hipMalloc((void**)&big, 2 * GiB));
  CK(hipMalloc((void**)&sink, 4096));
  CK(hipMemset(big, 1, 2 * GiB));
  CK(hipDeviceSynchronize());
  const unsigned blocks = 4096, lanes = blocks * 256;
  // flush between cases: a 1 GiB streaming read evicts L2 and the Infinity Cache of what the previous case left
  auto flush = [&] {
    hipLaunchKernelGGL(k_stream16, dim3(8192), dim3(256), 0, 0, (const uint4*)(big + GiB), GiB / 16, sink);
    CK(hipDeviceSynchronize());
  };
  for (int rep = 0; rep < 3; ++rep) {
    flush();
    hipLaunchKernelGGL(k_stream16, dim3(8192), dim3(256), 0, 0, (const uint4*)big, (512 * MiB) / 16, sink);
    CK(hipDeviceSynchronize());
    printf("case stream16 bytes %zu\n", 512 * MiB);
    for (size_t T : {32 * MiB, GiB}) {
      flush();
      const unsigned steps = 8;
      hipLaunchKernelGGL(k_gather32, dim3(blocks), dim3(256), 0, 0, (const el32*)big, T / 32, steps, sink);
      CK(hipDeviceSynchronize());
      printf("case gather32_T%zuMiB gathers %zu bytes32 %zu\n", T / MiB, (size_t)lanes * steps, (size_t)lanes * steps * 32);
    }
    for (unsigned ms : {2u, 8u}) {
      flush();
      const unsigned steps = 8;
      hipLaunchKernelGGL(k_gather32_sorted, dim3(blocks), dim3(256), 0, 0, (const el32*)big, (32 * MiB) / 32, steps, ms, sink);
      CK(hipDeviceSynchronize());
      printf("case gather32_sorted_stride%u gathers %zu bytes32 %zu\n", ms, (size_t)lanes * steps, (size_t)lanes * steps * 32);
    }
    {
      flush();
      const unsigned steps = 64;  // 256 B per lane: two lines' worth, the short columns' index lists
      hipLaunchKernelGGL(k_lane_streams, dim3(blocks), dim3(256), 0, 0, (const unsigned*)big, (size_t)64, steps, sink);
      CK(hipDeviceSynchronize());
      printf("case lane_streams bytes %zu\n", (size_t)lanes * steps * 4);
    }
    {
      flush();
      hipLaunchKernelGGL(k_stream4, dim3(8192), dim3(256), 0, 0, (const unsigned*)big, (256 * MiB) / 4, sink);
      CK(hipDeviceSynchronize());
      printf("case stream4 bytes %zu\n", 256 * MiB);
      flush();
      hipLaunchKernelGGL(k_stream32, dim3(8192), dim3(256), 0, 0, (const el32*)big, (512 * MiB) / 32, sink);
      CK(hipDeviceSynchronize());
      printf("case stream32 bytes %zu\n", 512 * MiB);
    }
    {
      flush();
      hipLaunchKernelGGL(k_store32, dim3(8192), dim3(256), 0, 0, (el32*)big, (256 * MiB) / 32);
      CK(hipDeviceSynchronize());
      printf("case store32 bytes %zu\n", 256 * MiB);
    }
  }
  CK(hipFree(big));
  CK(hipFree(sink));
  return 0;
}
