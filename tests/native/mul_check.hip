// The base-field product in its product-scanning form (field.hpp fe_mul_fips on the host; field_fips_device.hpp, the generated column blocks, on the
// device) against the 64-bit CIOS host product, on edge values and seeded random residues. `mul_check host N` runs the host comparison only (no GPU);
// `mul_check device N` launches one wave and several full-occupancy blocks (the carry hand-over between v_mad_u64_u32 and v_addc inside the asm
// blocks is back to back in the lone wave and interleaved with other waves in the full launch) and compares every lane's product and a 64-deep
// dependent chain with the host's. Built and run by tests/test_field_host.py and tests/test_gpu_abi_gaps.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../spartan2_amd/csrc/field.hpp"

__global__ void k_mul(const fe_t* a, const fe_t* b, fe_t* prod, fe_t* chain, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const fe_t x = a[i], y = b[i];
  prod[i] = fe_mul<FpP>(x, y);
  fe_t c = x;
  for (int k = 0; k < 64; ++k) c = fe_mul<FpP>(c, (k & 1) ? y : c);
  chain[i] = c;
}

#if !defined(__HIP_DEVICE_COMPILE__)
static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t next() {
  st ^= st << 13;
  st ^= st >> 7;
  st ^= st << 17;
  return st;
}
static fe_t sample(int k) {
  fe_t x;
  if (k % 97 == 0) return fe_zero();
  if (k % 97 == 1) return fe_one<FpP>();
  if (k % 97 == 2) return fe_neg<FpP>(fe_one<FpP>());
  if (k % 97 == 3) {  // p - 1 as a raw representative
    for (int i = 0; i < 8; ++i) x.v[i] = FpP::P(i);
    x.v[0] -= 1;
    return x;
  }
  if (k % 97 == 4) {  // all low words set
    for (int i = 0; i < 8; ++i) x.v[i] = i < 4 ? 0xffffffffu : 0u;
    return x;
  }
  uint8_t b[64];
  for (int i = 0; i < 8; ++i) {
    const uint64_t w = next();
    memcpy(b + 8 * i, &w, 8);
  }
  return fe_from_uniform<FpP>(b);
}
#endif

int main(int argc, char** argv) {
#if !defined(__HIP_DEVICE_COMPILE__)
  const bool device = argc > 1 && !strcmp(argv[1], "device");
  const int n = argc > 2 ? atoi(argv[2]) : 20000;
  std::vector<fe_t> a(n), b(n);
  for (int k = 0; k < n; ++k) {
    a[k] = sample(k);
    b[k] = sample(k * 7 + 3);
  }
  int bad = 0;
  if (!device) {
    for (int k = 0; k < n; ++k) {
      const fe_t w = fe_mul<FpP>(a[k], b[k]);
      if (!fe_eq(w, fe_mul_fips<FpP>(a[k], b[k])) || !fe_eq(w, fe_mul_limb32<FpP>(a[k], b[k]))) ++bad;
    }
    printf("host: %d mismatches in %d products\n", bad, n);
    return bad != 0;
  }
  fe_t *da, *db, *dp, *dc;
  if (hipMalloc((void**)&da, n * sizeof(fe_t)) || hipMalloc((void**)&db, n * sizeof(fe_t)) || hipMalloc((void**)&dp, n * sizeof(fe_t)) || hipMalloc((void**)&dc, n * sizeof(fe_t))) return 2;
  hipMemcpy(da, a.data(), n * sizeof(fe_t), hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * sizeof(fe_t), hipMemcpyHostToDevice);
  std::vector<fe_t> p(n), c(n);
  for (int pass = 0; pass < 2; ++pass) {
    const int cnt = pass == 0 ? 64 : n;  // a lone wave, then the whole set
    hipLaunchKernelGGL(k_mul, dim3((cnt + 255) / 256), dim3(pass == 0 ? 64 : 256), 0, 0, da, db, dp, dc, (size_t)cnt);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(p.data(), dp, cnt * sizeof(fe_t), hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, cnt * sizeof(fe_t), hipMemcpyDeviceToHost);
    for (int k = 0; k < cnt; ++k) {
      fe_t w = a[k];
      for (int q = 0; q < 64; ++q) w = fe_mul<FpP>(w, (q & 1) ? b[k] : w);
      if (!fe_eq(p[k], fe_mul<FpP>(a[k], b[k])) || !fe_eq(c[k], w)) ++bad;
    }
  }
  printf("device: %d mismatches in %d lanes\n", bad, n + 64);
  return bad != 0;
#else
  return 0;
#endif
}
