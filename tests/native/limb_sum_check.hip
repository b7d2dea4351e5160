// fe_from_limb_sums (spartan2_amd/csrc/field.hpp): the canonical value of a limb-wise sum of canonical elements - what the host's gathering of result slots
// reduces once per round (capi_core.hip reduce_partials_wait) - against a chain of modular additions. Host code only; nothing is launched.
// Usage: limb_sum_check <trials>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../spartan2_amd/csrc/field.hpp"

template <class FP>
static int run(const char* name, int trials) {
  int bad = 0;
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&st]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return (uint32_t)(st >> 16);
  };
  for (int t = 0; t < trials; ++t) {
    const int n = 1 + (int)(rnd() % 64);
    uint64_t limb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    fe_t acc = fe_zero();
    for (int k = 0; k < n; ++k) {
      fe_t v;
      for (int i = 0; i < 8; ++i) v.v[i] = (t % 5 == 0) ? 0xffffffffu : rnd();
      if (t % 7 == 3) v = fe_zero();
      v = fe_add<FP>(v, fe_zero());  // into [0, p): two conditional subtractions cover any 256-bit pattern (p > 2^255)
      v = fe_add<FP>(v, fe_zero());
      if (t % 5 == 0 && k % 2 == 0) v = fe_sub<FP>(fe_zero(), fe_one<FP>());  // p - R mod p ... a large residue
      for (int i = 0; i < 8; ++i) limb[i] += v.v[i];
      acc = fe_add<FP>(acc, v);
    }
    const fe_t r = fe_from_limb_sums<FP>(limb);
    if (memcmp(&r, &acc, sizeof(fe_t)) != 0) ++bad;
  }
  printf("%s: %d sums, %d mismatches\n", name, trials, bad);
  return bad;
}

int main(int argc, char** argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 20000;
  int bad = run<FqP>("scalar field", trials);
  bad += run<FpP>("base field", trials);
  return bad ? 1 : 0;
}
