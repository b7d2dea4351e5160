// Host-only check (no GPU needed): the host side's inversions (field.hpp fe_inv_host_safegcd, fe_inv_host_xgcd) against Fermat's little theorem
// (fe_pow with p - 2) and against x * inv(x) == 1, for both fields: edge values and seeded random residues. Built and run by tests/test_field_host.py.
#include <cstdio>
#include <cstdlib>

#include "../../spartan2_amd/csrc/field.hpp"

#if !defined(__HIP_DEVICE_COMPILE__)  // (the host-side inversion does not exist in the device pass)
template <class FP>
static fe_t fermat(const fe_t& x) {
  uint32_t e[8], bw = 0;
  for (int i = 0; i < 8; ++i) e[i] = sp_subb(FP::P(i), i == 0 ? 2u : 0u, bw);
  return fe_pow<FP>(x, e);
}
template <class FP>
static int check(const char* name, int n) {
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto next = [&] {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return st;
  };
  int bad = 0;
  const fe_t one = fe_one<FP>();
  for (int k = 0; k < n; ++k) {
    fe_t x;
    if (k == 0) x = fe_zero();
    else if (k == 1) x = one;
    else if (k == 2) x = fe_neg<FP>(one);
    else if (k == 3) x = fe_from_u64<FP>(2);
    else if (k < 40) x = fe_from_u64<FP>(next() >> (k & 63));
    else {
      uint8_t b[64];
      for (int i = 0; i < 8; ++i) {
        uint64_t w = next();
        memcpy(b + 8 * i, &w, 8);
      }
      x = fe_from_uniform<FP>(b);
    }
    // three forms must agree: the raw xgcd (public values), the blinded inversion (possibly secret values) and Fermat's exponentiation
    fe_t a;
    const bool in_bound = fe_inv_host_xgcd<FP>(x, &a);
    fe_t sg;
    const bool sg_bound = fe_inv_host_safegcd<FP>(x, &sg);  // the division-step form (what fe_inv / fe_inv_vartime run first)
    const fe_t f = fermat<FP>(x), bl = fe_inv<FP>(x), vt = fe_inv_vartime<FP>(x);
    bool ok = in_bound && sg_bound && fe_eq(a, f) && fe_eq(sg, f) && fe_eq(bl, f) && fe_eq(vt, f);
    if (!fe_is_zero(x)) ok = ok && fe_eq(fe_mul<FP>(a, x), one);
    else ok = ok && fe_is_zero(a);
    if (!ok) {
      if (bad < 5) fprintf(stderr, "%s: mismatch at sample %d\n", name, k);
      ++bad;
    }
  }
  // non-canonical inputs (the C ABI does not reduce caller data): x = p is a zero, x = p + 5 is 5; neither may spin or differ
  {
    fe_t pp, p5, a, b;
    uint32_t c = 0;
    for (int i = 0; i < 8; ++i) pp.v[i] = FP::P(i);
    for (int i = 0; i < 8; ++i) p5.v[i] = sp_addc(pp.v[i], i == 0 ? 5u : 0u, c);
    fe_t five;  // the plain residue 5 read as a Montgomery representative, like p + 5
    for (int i = 0; i < 8; ++i) five.v[i] = i == 0 ? 5u : 0u;
    bool ok = fe_inv_host_xgcd<FP>(pp, &a) && fe_is_zero(a) && fe_is_zero(fe_inv_vartime<FP>(pp));
    ok = ok && fe_inv_host_xgcd<FP>(p5, &a) && fe_inv_host_xgcd<FP>(five, &b) && fe_eq(a, b) && fe_eq(fe_inv_vartime<FP>(p5), b);
    fe_t c0, c5;
    ok = ok && fe_inv_host_safegcd<FP>(pp, &c0) && fe_is_zero(c0) && fe_inv_host_safegcd<FP>(p5, &c5) && fe_eq(c5, b);
    fe_t top;  // 2^256 - 1: the largest non-canonical input
    for (int i = 0; i < 8; ++i) top.v[i] = 0xffffffffu;
    ok = ok && fe_inv_host_safegcd<FP>(top, &c0) && fe_inv_host_xgcd<FP>(top, &c5) && fe_eq(c0, c5);
    if (!ok) {
      fprintf(stderr, "%s: non-canonical input mishandled\n", name);
      ++bad;
    }
  }
  // the blinding masks must differ from call to call (a constant mask would not blind anything)
  {
    uint32_t m1[8], m2[8];
    if (!sp_blind_mask(m1) || !sp_blind_mask(m2) || !memcmp(m1, m2, sizeof(m1))) {
      fprintf(stderr, "%s: blinding mask source is not producing fresh values\n", name);
      ++bad;
    }
  }
  printf("%s: %d samples, %d mismatches\n", name, n, bad);
  return bad;
}
#endif
int main(int argc, char** argv) {
#if !defined(__HIP_DEVICE_COMPILE__)
  const int n = argc > 1 ? atoi(argv[1]) : 20000;
  return (check<FqP>("scalar field", n) + check<FpP>("base field", n)) ? 1 : 0;
#else
  return 0;
#endif
}
