// 256-bit prime-field arithmetic for gfx950 (and for the host-side protocol glue — every function is
// __host__ __device__). Elements are 8 x u32 little-endian limbs in Montgomery form (x * 2^256 mod p),
// canonical < p: byte-identical to the reference's halo2curves 4 x u64 limbs, so
// tables cross the C ABI without conversion.
//
// Two fields:
//   FqP  = T256 scalar field = NIST P-256 prime 2^256 - 2^224 + 2^192 + 2^96 - 1 (src/provider/pt256.rs:55).
//          -p^-1 mod 2^96 == 1, so Montgomery's quotient digits are the running low words themselves and
//          q*p is shifts/adds: the reduction is ~75 add-with-carry instructions, no multiplies. This is
//          what replaces the reference's x86 mulx/adcx REDC (src/big_num/limbs.rs:200-331, montgomery.rs).
//   FpP  = T256 base field (pt256.rs:56), generic modulus -> word-serial Montgomery (CIOS).
//
// The 256x256 product is 64 v_mad_u64_u32 (row-wise: 8 independent mads, then one v_addc chain per row).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <sys/random.h>
#include <sys/types.h>
#endif

#define SP_HD __host__ __device__ __forceinline__

struct alignas(16) fe_t {
  uint32_t v[8];
};

SP_HD uint32_t sp_addc(uint32_t a, uint32_t b, uint32_t& c) {
  unsigned co;
  uint32_t r = __builtin_addc(a, b, c, &co);
  c = co;
  return r;
}
SP_HD uint32_t sp_subb(uint32_t a, uint32_t b, uint32_t& bw) {
  unsigned bo;
  uint32_t r = __builtin_subc(a, b, bw, &bo);
  bw = bo;
  return r;
}

// ---- field parameter packs ------------------------------------------------------------------------
struct FqP {
  static constexpr bool P256_PRIME = true;
  static constexpr uint32_t INV32 = 1u;
  static constexpr uint64_t INV64 = 1ull;
  static constexpr SP_HD uint32_t P(int i) {
    constexpr uint32_t t[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xffffffffu};
    return t[i];
  }
  static constexpr SP_HD uint32_t R1(int i) {  // 2^256 mod p
    constexpr uint32_t t[8] = {0x00000001u, 0x00000000u, 0x00000000u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xfffffffeu, 0x00000000u};
    return t[i];
  }
  static constexpr SP_HD uint32_t R2(int i) {  // 2^512 mod p
    constexpr uint32_t t[8] = {0x00000003u, 0x00000000u, 0xffffffffu, 0xfffffffbu, 0xfffffffeu, 0xffffffffu, 0xfffffffdu, 0x00000004u};
    return t[i];
  }
  static constexpr SP_HD uint32_t R3(int i) {  // 2^768 mod p
    constexpr uint32_t t[8] = {0x0000000au, 0xfffffffdu, 0xfffffff7u, 0xffffffedu, 0xfffffffcu, 0x00000005u, 0x00000001u, 0x00000018u};
    return t[i];
  }
};
struct FpP {
  static constexpr bool P256_PRIME = false;
  static constexpr uint32_t INV32 = 0x0f646959u;  // -p^-1 mod 2^32
  static constexpr uint64_t INV64 = 0xe0a2f6a60f646959ull;
  static constexpr SP_HD uint32_t P(int i) {
    constexpr uint32_t t[8] = {0xb1c4b117u, 0x93135661u, 0x30e73177u, 0x7e72b42bu, 0x00000001u, 0x00000000u, 0x00000001u, 0xffffffffu};
    return t[i];
  }
  static constexpr SP_HD uint32_t R1(int i) {
    constexpr uint32_t t[8] = {0x4e3b4ee9u, 0x6ceca99eu, 0xcf18ce88u, 0x818d4bd4u, 0xfffffffeu, 0xffffffffu, 0xfffffffeu, 0x00000000u};
    return t[i];
  }
  static constexpr SP_HD uint32_t R2(int i) {
    constexpr uint32_t t[8] = {0x910d33ecu, 0xbc7ba9fcu, 0x07eae97fu, 0xd8488344u, 0xa0ed6777u, 0x362b66e6u, 0x21d30291u, 0xa057002au};
    return t[i];
  }
  static constexpr SP_HD uint32_t R3(int i) {
    constexpr uint32_t t[8] = {0x56eb5134u, 0x67cff92eu, 0xd91eb940u, 0x4a49f97du, 0x0a676baeu, 0xb3d51283u, 0x79d924a0u, 0xcae090c9u};
    return t[i];
  }
};

// ---- basic predicates / constants -------------------------------------------------------------------
SP_HD fe_t fe_zero() {
  fe_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
template <class FP>
SP_HD fe_t fe_one() {
  fe_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = FP::R1(i);
  return r;
}
SP_HD bool fe_is_zero(const fe_t& a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i];
  return o == 0;
}
SP_HD bool fe_eq(const fe_t& a, const fe_t& b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// r = (carry:x) - p if (carry:x) >= p else x   (one conditional subtraction, branch-free)
template <class FP>
SP_HD fe_t fe_cond_sub_p(const uint32_t x[8], uint32_t carry) {
  uint32_t d[8], bw = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = sp_subb(x[i], FP::P(i), bw);
  // take the difference when the subtraction did not borrow past the carry word
  bool take = carry >= bw;  // carry:x - p >= 0  <=>  carry - bw >= 0
  fe_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = take ? d[i] : x[i];
  return r;
}

// in-place variant tracking the word above 2^256: (top:x) -= p when (top:x) >= p
template <class FP>
SP_HD void fe_cond_sub_p_top(uint32_t x[8], uint32_t& top) {
  uint32_t d[8], bw = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = sp_subb(x[i], FP::P(i), bw);
  bool take = top >= bw;
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = take ? d[i] : x[i];
  top = take ? top - bw : top;
}

template <class FP>
SP_HD fe_t fe_add(const fe_t& a, const fe_t& b) {
  uint32_t s[8], c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = sp_addc(a.v[i], b.v[i], c);
  return fe_cond_sub_p<FP>(s, c);
}
template <class FP>
SP_HD fe_t fe_sub(const fe_t& a, const fe_t& b) {
  uint32_t d[8], bw = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = sp_subb(a.v[i], b.v[i], bw);
  uint32_t mask = 0u - bw, c = 0;
  fe_t r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = sp_addc(d[i], FP::P(i) & mask, c);
  return r;
}
// The canonical value of sum_i limb[i] 2^(32 i) mod p for limb-wise sums of up to 2^31 canonical elements (each limb < 2^63): carries propagated into
// a ninth and tenth word, the part above 2^256 folded back with 2^256 mod p (twice), then conditional subtractions.
template <class FP>
SP_HD fe_t fe_from_limb_sums(const uint64_t limb[8]) {
  uint32_t t[8];
  uint64_t carry = 0;
  for (int i = 0; i < 8; ++i) {
    const uint64_t x = (limb[i] & 0xffffffffull) + carry;
    t[i] = (uint32_t)x;
    carry = (x >> 32) + (limb[i] >> 32);
  }
  uint64_t hi = carry;  // < 2^32 for the sums this is used on (<= 2^31 elements below 2^256)
  for (int pass = 0; pass < 2; ++pass) {
    uint64_t c2 = 0;
    for (int i = 0; i < 8; ++i) {
      const uint64_t x = hi * FP::R1(i) + t[i] + c2;
      t[i] = (uint32_t)x;
      c2 = x >> 32;
    }
    hi = c2;
  }
  uint32_t top = (uint32_t)hi;  // <= 1 after two folds
  for (int k = 0; k < 3; ++k) fe_cond_sub_p_top<FP>(t, top);
  fe_t r;
  for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  return r;
}
template <class FP>
SP_HD fe_t fe_neg(const fe_t& a) {
  return fe_sub<FP>(fe_zero(), a);
}
template <class FP>
SP_HD fe_t fe_dbl(const fe_t& a) {
  return fe_add<FP>(a, a);
}

// ---- 256 x 256 -> 512 product -------------------------------------------------------------------------
SP_HD void fe_mul_wide(uint32_t t[16], const fe_t& a, const fe_t& b) {
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = (uint64_t)a.v[j] * b.v[i] + t[i + j];  // v_mad_u64_u32, independent
    uint32_t c = 0;
    t[i] = (uint32_t)p[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) t[i + j] = sp_addc((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c);
    t[i + 8] = (uint32_t)(p[7] >> 32) + c;  // cannot overflow: row sum < 2^(32*9)
  }
}

// ---- Montgomery reduction of a 512-bit value ------------------------------------------------------------
// P-256 prime: quotient M solves M = T_lo + (M<<96) + (M<<192) - (M<<224) (mod 2^256); then
// (T + M p) / 2^256 = T_hi + M - (M>>32) + (M>>64) + (M>>160) + k, k = signed carry-out of the M recurrence.
// Output < 2^257; `extra_subs` conditional subtractions bring it to canonical form (1 for a single
// product of canonical inputs, 2 for any 512-bit lazy sum).
template <int EXTRA_SUBS>
SP_HD fe_t fe_redc_p256(const uint32_t t[16]) {
  uint32_t m[8];
  m[0] = t[0];
  m[1] = t[1];
  m[2] = t[2];
  uint32_t c = 0;
  m[3] = sp_addc(t[3], m[0], c);
  m[4] = sp_addc(t[4], m[1], c);
  m[5] = sp_addc(t[5], m[2], c);
  uint32_t x6 = sp_addc(t[6], m[3], c);
  uint32_t x7 = sp_addc(t[7], m[4], c);
  int32_t k = (int32_t)c;
  c = 0;
  x6 = sp_addc(x6, m[0], c);
  x7 = sp_addc(x7, m[1], c);
  k += (int32_t)c;
  uint32_t bw = 0;
  x7 = sp_subb(x7, m[0], bw);
  k -= (int32_t)bw;
  m[6] = x6;
  m[7] = x7;
  // r = T_hi + M            (top word tracked in `top`, signed)
  uint32_t r[8];
  int32_t top = k;
  c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = sp_addc(t[8 + i], m[i], c);
  int32_t hi = (int32_t)c;
  // r -= M >> 32
  bw = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) r[i] = sp_subb(r[i], m[i + 1], bw);
  r[7] = sp_subb(r[7], 0u, bw);
  hi -= (int32_t)bw;
  // r += M >> 64
  c = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) r[i] = sp_addc(r[i], m[i + 2], c);
  r[6] = sp_addc(r[6], 0u, c);
  r[7] = sp_addc(r[7], 0u, c);
  hi += (int32_t)c;
  // r += M >> 160
  c = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) r[i] = sp_addc(r[i], m[i + 5], c);
#pragma unroll
  for (int i = 3; i < 8; ++i) r[i] = sp_addc(r[i], 0u, c);
  hi += (int32_t)c;
  // r += k  (k in [-1, 2]): add the sign-extended constant
  uint32_t kk = (uint32_t)top, ext = (uint32_t)(top >> 31);
  c = 0;
  r[0] = sp_addc(r[0], kk, c);
#pragma unroll
  for (int i = 1; i < 8; ++i) r[i] = sp_addc(r[i], ext, c);
  hi += (int32_t)c + (int32_t)ext;  // ext is 0 or 0xffffffff == -1
  uint32_t topw = (uint32_t)hi;  // value = topw * 2^256 + r, in [0, 2^257)
#pragma unroll
  for (int s = 0; s < EXTRA_SUBS; ++s) fe_cond_sub_p_top<FqP>(r, topw);
  fe_t out;
#pragma unroll
  for (int i = 0; i < 8; ++i) out.v[i] = r[i];
  return out;
}

// Generic word-serial Montgomery reduction (CIOS second half) for FpP: 8 rounds of m = t0 * INV32; t += m*p; t >>= 32.
template <class FP>
SP_HD fe_t fe_redc_generic(const uint32_t tin[16]) {
  uint32_t t[17];
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = tin[i];
  t[16] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t m = t[i] * FP::INV32;
    uint64_t p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = (uint64_t)m * FP::P(j) + t[i + j];
    uint32_t c = 0;
    // t[i] becomes 0; carry the high halves up
#pragma unroll
    for (int j = 1; j < 8; ++j) t[i + j] = sp_addc((uint32_t)p[j], (uint32_t)(p[j - 1] >> 32), c);
    t[i + 8] = sp_addc(t[i + 8], (uint32_t)(p[7] >> 32), c);
#pragma unroll
    for (int j = i + 9; j < 17; ++j) t[j] = sp_addc(t[j], 0u, c);
  }
  return fe_cond_sub_p<FP>(t + 8, t[16]);
}

template <class FP>
SP_HD fe_t fe_redc(const uint32_t t[16]) {
  if constexpr (FP::P256_PRIME) {
    return fe_redc_p256<1>(t);
  } else {
    return fe_redc_generic<FP>(t);
  }
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host-side product for the O(1) protocol glue (claims, UniPoly, inversions): 4 x u64 CIOS with __int128.
// Same canonical results as the device path; never used inside a kernel.
template <class FP>
inline fe_t fe_mul_host64(const fe_t& a, const fe_t& b) {
  typedef unsigned __int128 u128;
  uint64_t A[4], B[4], Pm[4], t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
    B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
    Pm[i] = (uint64_t)FP::P(2 * i) | ((uint64_t)FP::P(2 * i + 1) << 32);
  }
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)A[j] * B[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * FP::INV64;
    c = ((u128)m * Pm[0] + t[0]) >> 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * Pm[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  // conditional subtraction
  uint64_t d[4];
  unsigned __int128 bw = 0;
  for (int i = 0; i < 4; ++i) {
    u128 x = (u128)t[i] - Pm[i] - (uint64_t)bw;
    d[i] = (uint64_t)x;
    bw = (x >> 64) & 1;
  }
  bool take = t[4] >= (uint64_t)bw;
  fe_t r;
  for (int i = 0; i < 4; ++i) {
    uint64_t w = take ? d[i] : t[i];
    r.v[2 * i] = (uint32_t)w;
    r.v[2 * i + 1] = (uint32_t)(w >> 32);
  }
  return r;
}
#endif

// ---- product-scanning Montgomery product for the generic-modulus field (FpP) ----------------------------------------------------------------------
// Column k of the product and of the reduction is accumulated in ONE 96-bit register triple: every term is a v_mad_u64_u32 into the low 64 bits
// whose carry-out (VCC) goes into the third word with a v_addc - two instructions per term, no zero-extended addend pairs to prepare. The compiler
// cannot produce this form (it never uses the multiply-add's carry-out: a 96-bit accumulate written in C costs four instructions per term), and
// its row-wise product + word-serial reduction came to 444 instructions per base-field product (104 multiply-adds, 138 carry adds, 84 register
// moves for the {word, 0} addends, 36 hazard nops) on a part that issues about one instruction of a lone wave every four cycles whatever it is.
// One asm statement per COLUMN (the compiler puts a wait state behind every asm statement whose result the next instruction reads); the terms with
// the base prime's zero word (word 5) are left out.
struct acc96_t {
  uint64_t lo;
  uint32_t hi;
};
SP_HD void acc96_shift(acc96_t& a) {
  a.lo = (a.lo >> 32) | ((uint64_t)a.hi << 32);
  a.hi = 0;
}
#if !defined(__HIP_DEVICE_COMPILE__)
// the algorithm in plain C (host only): what the unit test checks against the 64-bit CIOS product; the device form is generated (tools/gen_fips_mul.py)
template <class FP>
inline fe_t fe_mul_fips(const fe_t& a, const fe_t& b) {
  acc96_t acc = {0, 0};
  auto mac = [&acc](uint32_t x, uint32_t y) {
    const unsigned __int128 t = (((unsigned __int128)acc.hi << 64) | acc.lo) + (uint64_t)x * y;
    acc.lo = (uint64_t)t;
    acc.hi = (uint32_t)(t >> 64);
  };
  uint32_t m[8], r[8];
  for (int i = 0; i < 15; ++i) {
    const int j0 = i < 8 ? 0 : i - 7, j1 = i < 8 ? i : 7;
    for (int j = j0; j <= j1; ++j) mac(a.v[j], b.v[i - j]);
    for (int j = j0; j <= j1; ++j)
      if (!(i < 8 && j == i)) mac(m[j], FP::P(i - j));
    if (i < 8) {
      m[i] = (uint32_t)acc.lo * FP::INV32;
      mac(m[i], FP::P(0));  // the low word becomes 0
    } else {
      r[i - 8] = (uint32_t)acc.lo;
    }
    acc96_shift(acc);
  }
  r[7] = (uint32_t)acc.lo;
  return fe_cond_sub_p<FP>(r, (uint32_t)(acc.lo >> 32));
}
#endif
#include "field_fips_device.hpp"

template <class FP>
SP_HD fe_t fe_mul(const fe_t& a, const fe_t& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
  return fe_mul_host64<FP>(a, b);
#else
  if constexpr (FP::P256_PRIME) {
    uint32_t t[16];
    fe_mul_wide(t, a, b);
    return fe_redc<FP>(t);
  } else {
    return fe_mul_fips_device<FP>(a, b);
  }
#endif
}
// The row-wise product + word-serial reduction whatever the field: the form the compiler schedules itself. The block-cooperative point addition keeps
// it for the base field too - there a level is 4.75 us with this form and 5.4 us with the asm column blocks (which the scheduler cannot interleave
// with the LDS traffic of a stage), while every throughput kernel gains from the shorter instruction stream (tools/ubench: 116 -> 134-140 G products/s,
// the config-4 commitment 10.6 -> 9.2 ms).
template <class FP>
SP_HD fe_t fe_mul_rowwise(const fe_t& a, const fe_t& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
  return fe_mul_host64<FP>(a, b);
#else
  uint32_t t[16];
  fe_mul_wide(t, a, b);
  return fe_redc<FP>(t);
#endif
}
// the 32-bit-limb device algorithm, callable on the host too (unit tests exercise it without a GPU)
template <class FP>
SP_HD fe_t fe_mul_limb32(const fe_t& a, const fe_t& b) {
  uint32_t t[16];
  fe_mul_wide(t, a, b);
  return fe_redc<FP>(t);
}
template <class FP>
SP_HD fe_t fe_sqr(const fe_t& a) {
  return fe_mul<FP>(a, a);
}

// ---- conversions / host-side helpers (O(1) glue; never in a kernel's inner loop) -----------------------
template <class FP>
SP_HD fe_t fe_from_canonical(const fe_t& c) {
  fe_t r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) r2.v[i] = FP::R2(i);
  return fe_mul<FP>(c, r2);
}
template <class FP>
SP_HD fe_t fe_to_canonical(const fe_t& a) {
  fe_t one = fe_zero();
  one.v[0] = 1;
  return fe_mul<FP>(a, one);
}
template <class FP>
SP_HD fe_t fe_from_u64(uint64_t x) {
  fe_t c = fe_zero();
  c.v[0] = (uint32_t)x;
  c.v[1] = (uint32_t)(x >> 32);
  return fe_from_canonical<FP>(c);
}
template <class FP>
SP_HD fe_t fe_from_i64(int64_t x) {
  if (x >= 0) return fe_from_u64<FP>((uint64_t)x);
  return fe_neg<FP>(fe_from_u64<FP>((uint64_t)(-(x + 1)) + 1));
}
// halo2curves from_uniform_bytes (src/provider/traits.rs:275-280): 64 bytes LE mod p
template <class FP>
SP_HD fe_t fe_from_uniform(const uint8_t* b64) {
  fe_t lo, hi, r2, r3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    lo.v[i] = (uint32_t)b64[4 * i] | ((uint32_t)b64[4 * i + 1] << 8) | ((uint32_t)b64[4 * i + 2] << 16) | ((uint32_t)b64[4 * i + 3] << 24);
    hi.v[i] = (uint32_t)b64[32 + 4 * i] | ((uint32_t)b64[32 + 4 * i + 1] << 8) | ((uint32_t)b64[32 + 4 * i + 2] << 16) | ((uint32_t)b64[32 + 4 * i + 3] << 24);
    r2.v[i] = FP::R2(i);
    r3.v[i] = FP::R3(i);
  }
  // lo / hi may be >= p: the Montgomery product still reduces correctly (inputs < 2^256, output canonical after
  // the conditional subtraction because lo*R2 < 2^256 * p).
  return fe_add<FP>(fe_mul<FP>(lo, r2), fe_mul<FP>(hi, r3));
}
// x^e, e given as canonical limbs (not a hot-path op)
template <class FP>
SP_HD fe_t fe_pow(const fe_t& x, const uint32_t e[8]) {
  fe_t acc = fe_one<FP>();
  for (int i = 255; i >= 0; --i) {
    acc = fe_sqr<FP>(acc);
    if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul<FP>(acc, x);
  }
  return acc;
}
#if !defined(__HIP_DEVICE_COMPILE__)
// Host-side VARIABLE-TIME inversion by the binary extended Euclidean algorithm on 4 x u64 (Fermat's 256 squarings + ~150 products cost 13 us, this
// about a third). Its running time depends on the value: call it directly (fe_inv_vartime) only on transcript-public values (1 - r_y[0], rho, acc_eq,
// the tau_k, anything the verifier inverts); fe_inv below blinds its argument first. The input is canonicalised (one conditional subtraction: the
// C ABI's load paths do not reduce caller data, and x == p would never leave the inner loop) and the loop is bounded (2 * 256 + 2 subtract steps are
// enough for coprime values below 2^256); on a bound violation the caller falls back to Fermat. Same canonical result as Fermat; inv(0) == 0.
template <class FP>
inline bool fe_inv_host_xgcd(const fe_t& x, fe_t* out) {
  typedef unsigned __int128 u128;
  struct U4 {
    uint64_t w[4];
  };
  auto is_zero = [](const U4& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; };
  auto is_one = [](const U4& a) { return a.w[0] == 1 && (a.w[1] | a.w[2] | a.w[3]) == 0; };
  auto geq = [](const U4& a, const U4& b) {
    for (int i = 3; i >= 0; --i) {
      if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    }
    return true;
  };
  auto sub = [](U4& a, const U4& b) {  // a -= b (a >= b)
    u128 bw = 0;
    for (int i = 0; i < 4; ++i) {
      u128 d = (u128)a.w[i] - b.w[i] - (uint64_t)bw;
      a.w[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
  };
  auto shr1 = [](U4& a, uint64_t top) {  // (top : a) >> 1
    for (int i = 0; i < 3; ++i) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 63);
    a.w[3] = (a.w[3] >> 1) | (top << 63);
  };
  U4 P, u, v, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
  for (int i = 0; i < 4; ++i) {
    P.w[i] = (uint64_t)FP::P(2 * i) | ((uint64_t)FP::P(2 * i + 1) << 32);
    u.w[i] = (uint64_t)x.v[2 * i] | ((uint64_t)x.v[2 * i + 1] << 32);
  }
  if (geq(u, P)) sub(u, P);  // non-canonical input (p <= x < 2^256 < 2p): reduce
  if (is_zero(u)) {
    *out = fe_zero();
    return true;
  }
  v = P;
  auto halve_mod = [&](U4& a) {  // a / 2 mod p
    uint64_t top = 0;
    if (a.w[0] & 1) {
      u128 c = 0;
      for (int i = 0; i < 4; ++i) {
        c += (u128)a.w[i] + P.w[i];
        a.w[i] = (uint64_t)c;
        c >>= 64;
      }
      top = (uint64_t)c;
    }
    shr1(a, top);
  };
  auto sub_mod = [&](U4& a, const U4& b) {  // a = a - b mod p
    if (geq(a, b)) {
      sub(a, b);
    } else {  // a + p - b
      U4 t = P;
      sub(t, b);  // p - b (b < p)
      u128 c = 0;
      for (int i = 0; i < 4; ++i) {
        c += (u128)a.w[i] + t.w[i];
        a.w[i] = (uint64_t)c;
        c >>= 64;
      }
    }
  };
  int budget = 2 * 256 + 2, halvings = 2 * 256 + 2;  // every subtract step is followed by at least one halving of u or v: bit length u + v falls each time
  while (!is_one(u) && !is_one(v)) {
    if (--budget < 0) return false;
    while (!(u.w[0] & 1)) {
      if (--halvings < 0) return false;
      shr1(u, 0);
      halve_mod(x1);
    }
    while (!(v.w[0] & 1)) {
      if (--halvings < 0) return false;
      shr1(v, 0);
      halve_mod(x2);
    }
    if (geq(u, v)) {
      sub(u, v);
      sub_mod(x1, x2);
    } else {
      sub(v, u);
      sub_mod(x2, x1);
    }
    if (is_zero(u) || is_zero(v)) return false;  // gcd != 1: cannot happen for a prime modulus and 0 < x < p
  }
  const U4& inv_raw = is_one(u) ? x1 : x2;  // (x R)^-1 as a plain residue = x^-1 R^-1
  fe_t y, r2, r3;
  for (int i = 0; i < 4; ++i) {
    y.v[2 * i] = (uint32_t)inv_raw.w[i];
    y.v[2 * i + 1] = (uint32_t)(inv_raw.w[i] >> 32);
  }
  for (int i = 0; i < 8; ++i) r2.v[i] = FP::R2(i);
  r3 = fe_mul_host64<FP>(r2, r2);   // R^2 R^2 / R = R^3
  *out = fe_mul_host64<FP>(y, r3);  // x^-1 R^-1 R^3 / R = x^-1 R
  return true;
}
// Host-side VARIABLE-TIME inversion by Bernstein-Yang division steps ("safegcd", eprint 2019/266) in batches of 62: the decisions of 62 consecutive
// steps depend only on the low 64 bits of (f, g), so a batch is a loop over machine words that yields a 2 x 2 transition matrix with entries below
// 2^62, and only the matrix is applied to the five-limb values (f, g) and (d, e) - ~10 batches for a 256-bit modulus against ~380 multi-limb
// shift / subtract steps of the binary algorithm above (2.1 GHz Xeon: 9.2 -> ~1.6 us; every commitment that leaves the library as an affine point
// pays one). Same contract as fe_inv_host_xgcd: canonical result x^-1 R for x R, inv(0) == 0, non-canonical input reduced first, false only if the
// batch budget is exceeded (the caller then takes Fermat's path; 12 batches = 744 steps cover the proven bound of 590 for 256-bit inputs).
template <class FP>
inline bool fe_inv_host_safegcd(const fe_t& x, fe_t* out) {
  typedef __int128 i128;
  constexpr uint64_t M62 = ~0ull >> 2;
  struct S62 {
    int64_t v[5];
  };
  uint64_t Pw[4], X[4];
  for (int i = 0; i < 4; ++i) {
    Pw[i] = (uint64_t)FP::P(2 * i) | ((uint64_t)FP::P(2 * i + 1) << 32);
    X[i] = (uint64_t)x.v[2 * i] | ((uint64_t)x.v[2 * i + 1] << 32);
  }
  {  // p <= x < 2^256 < 2p: reduce
    uint64_t d[4];
    unsigned __int128 bw = 0;
    for (int i = 0; i < 4; ++i) {
      const unsigned __int128 t = (unsigned __int128)X[i] - Pw[i] - (uint64_t)bw;
      d[i] = (uint64_t)t;
      bw = (t >> 64) & 1;
    }
    if (!bw)
      for (int i = 0; i < 4; ++i) X[i] = d[i];
  }
  if ((X[0] | X[1] | X[2] | X[3]) == 0) {
    *out = fe_zero();
    return true;
  }
  auto to62 = [&](const uint64_t w[4]) {
    S62 r;
    r.v[0] = (int64_t)(w[0] & M62);
    r.v[1] = (int64_t)(((w[0] >> 62) | (w[1] << 2)) & M62);
    r.v[2] = (int64_t)(((w[1] >> 60) | (w[2] << 4)) & M62);
    r.v[3] = (int64_t)(((w[2] >> 58) | (w[3] << 6)) & M62);
    r.v[4] = (int64_t)(w[3] >> 56);
    return r;
  };
  const S62 P = to62(Pw);
  const uint64_t pinv62 = (0 - FP::INV64) & M62;  // p^-1 mod 2^62 (INV64 = -p^-1 mod 2^64)
  S62 f = P, g = to62(X), d = {{0, 0, 0, 0, 0}}, e = {{1, 0, 0, 0, 0}};
  int64_t eta = -1;  // = -delta
  for (int batch = 0;; ++batch) {
    if (batch == 12) return false;
    // 62 division steps on the low words
    uint64_t u = 1, v = 0, q = 0, r = 1, fl = (uint64_t)f.v[0] | ((uint64_t)f.v[1] << 62), gl = (uint64_t)g.v[0] | ((uint64_t)g.v[1] << 62);
    for (int i = 62;;) {
      const int zeros = __builtin_ctzll(gl | (~0ull << i));
      gl >>= zeros;
      u <<= zeros;
      v <<= zeros;
      eta -= zeros;
      i -= zeros;
      if (i == 0) break;
      if (eta < 0) {  // delta > 0 and g odd: (f, g) <- (g, g - f)
        eta = -eta;
        uint64_t t = fl;
        fl = gl;
        gl = 0 - t;
        t = u;
        u = q;
        q = 0 - t;
        t = v;
        v = r;
        r = 0 - t;
      }
      gl += fl;  // g odd: g <- g + f (even; the halving is the next pass's shift)
      q += u;
      r += v;
    }
    const int64_t U = (int64_t)u, V = (int64_t)v, Q = (int64_t)q, R = (int64_t)r;
    {  // (d, e) <- (U d + V e, Q d + R e) / 2^62 mod p, kept in (-2p, p)
      const int64_t sd = d.v[4] >> 63, se = e.v[4] >> 63;
      int64_t md = (U & sd) + (V & se), me = (Q & sd) + (R & se);
      i128 cd = (i128)U * d.v[0] + (i128)V * e.v[0], ce = (i128)Q * d.v[0] + (i128)R * e.v[0];
      md -= (int64_t)((pinv62 * (uint64_t)cd + (uint64_t)md) & M62);
      me -= (int64_t)((pinv62 * (uint64_t)ce + (uint64_t)me) & M62);
      cd += (i128)P.v[0] * md;
      ce += (i128)P.v[0] * me;
      cd >>= 62;
      ce >>= 62;
      for (int i = 1; i < 5; ++i) {
        cd += (i128)U * d.v[i] + (i128)V * e.v[i] + (i128)P.v[i] * md;
        ce += (i128)Q * d.v[i] + (i128)R * e.v[i] + (i128)P.v[i] * me;
        d.v[i - 1] = (int64_t)((uint64_t)cd & M62);
        e.v[i - 1] = (int64_t)((uint64_t)ce & M62);
        cd >>= 62;
        ce >>= 62;
      }
      d.v[4] = (int64_t)cd;
      e.v[4] = (int64_t)ce;
    }
    {  // (f, g) <- (U f + V g, Q f + R g) / 2^62 (exact)
      i128 cf = (i128)U * f.v[0] + (i128)V * g.v[0], cg = (i128)Q * f.v[0] + (i128)R * g.v[0];
      cf >>= 62;
      cg >>= 62;
      for (int i = 1; i < 5; ++i) {
        cf += (i128)U * f.v[i] + (i128)V * g.v[i];
        cg += (i128)Q * f.v[i] + (i128)R * g.v[i];
        f.v[i - 1] = (int64_t)((uint64_t)cf & M62);
        g.v[i - 1] = (int64_t)((uint64_t)cg & M62);
        cf >>= 62;
        cg >>= 62;
      }
      f.v[4] = (int64_t)cf;
      g.v[4] = (int64_t)cg;
    }
    if ((g.v[0] | g.v[1] | g.v[2] | g.v[3] | g.v[4]) == 0) break;
  }
  // f = +-1 (p is prime, 0 < x < p); d = +-x^-1 in (-2p, p): bring it into [0, p) and give it f's sign
  const bool fneg = f.v[4] < 0;
  {
    const bool f_is_pm1 = fneg ? (f.v[0] == (int64_t)M62 && f.v[1] == (int64_t)M62 && f.v[2] == (int64_t)M62 && f.v[3] == (int64_t)M62 && f.v[4] == -1)
                               : (f.v[0] == 1 && (f.v[1] | f.v[2] | f.v[3] | f.v[4]) == 0);
    if (!f_is_pm1) return false;
  }
  auto add_p = [&](S62& a, int64_t sign) {  // a += sign * p, limbs renormalised to 62 bits (top limb signed)
    int64_t c = 0;
    for (int i = 0; i < 5; ++i) {
      const int64_t t = a.v[i] + sign * P.v[i] + c;
      if (i < 4) {
        a.v[i] = t & (int64_t)M62;
        c = t >> 62;
      } else {
        a.v[i] = t;
      }
    }
  };
  auto is_neg = [](const S62& a) { return a.v[4] < 0; };
  for (int k = 0; k < 3 && is_neg(d); ++k) add_p(d, 1);
  if (is_neg(d)) return false;
  if (fneg && (d.v[0] | d.v[1] | d.v[2] | d.v[3] | d.v[4]) != 0) {  // d <- p - d (d in [0, p) first)
    S62 t = d;
    add_p(t, -1);
    if (!is_neg(t)) d = t;  // d was in [p, 2p): cannot happen after the bound above, handled anyway
    for (int i = 0; i < 5; ++i) d.v[i] = -d.v[i];
    add_p(d, 1);  // renormalises the limbs as it adds
  } else {
    S62 t = d;
    add_p(t, -1);
    if (!is_neg(t)) d = t;
  }
  if (is_neg(d)) return false;
  uint64_t w[4];
  w[0] = (uint64_t)d.v[0] | ((uint64_t)d.v[1] << 62);
  w[1] = ((uint64_t)d.v[1] >> 2) | ((uint64_t)d.v[2] << 60);
  w[2] = ((uint64_t)d.v[2] >> 4) | ((uint64_t)d.v[3] << 58);
  w[3] = ((uint64_t)d.v[3] >> 6) | ((uint64_t)d.v[4] << 56);
  fe_t y, r3;
  for (int i = 0; i < 4; ++i) {
    y.v[2 * i] = (uint32_t)w[i];
    y.v[2 * i + 1] = (uint32_t)(w[i] >> 32);
  }
  for (int i = 0; i < 8; ++i) r3.v[i] = FP::R3(i);
  *out = fe_mul_host64<FP>(y, r3);  // x^-1 R^-1 R^3 / R = x^-1 R
  return true;
}
// 256 bits from a per-thread ChaCha20 stream keyed once from the kernel's entropy pool (getrandom(2)): the multiplicative mask of fe_inv. Returns false when
// no entropy could be read (the caller then takes the constant-time path).
inline bool sp_blind_mask(uint32_t out[8]) {
  struct Stream {
    uint32_t key[8];
    uint64_t ctr = 0;
    uint32_t blk[16];
    int left = 0;
    bool ok = false;
    Stream() {
      size_t got = 0;
      uint8_t* p = reinterpret_cast<uint8_t*>(key);
      for (int tries = 0; got < sizeof(key) && tries < 64; ++tries) {
        ssize_t r = getrandom(p + got, sizeof(key) - got, 0);
        if (r > 0) got += (size_t)r;
      }
      ok = got == sizeof(key);
    }
    void refill() {
      uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
      for (int i = 0; i < 8; ++i) s[4 + i] = key[i];
      s[12] = (uint32_t)ctr;
      s[13] = (uint32_t)(ctr >> 32);
      s[14] = s[15] = 0;
      ++ctr;
      uint32_t w[16];
      for (int i = 0; i < 16; ++i) w[i] = s[i];
      auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
      auto qr = [&](int a, int b, int c, int d) {
        w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 16);
        w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 12);
        w[a] += w[b]; w[d] = rotl(w[d] ^ w[a], 8);
        w[c] += w[d]; w[b] = rotl(w[b] ^ w[c], 7);
      };
      for (int r = 0; r < 10; ++r) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
      }
      for (int i = 0; i < 16; ++i) blk[i] = w[i] + s[i];
      left = 16;
    }
  };
  static thread_local Stream st;
  if (!st.ok) return false;
  if (st.left < 8) st.refill();
  for (int i = 0; i < 8; ++i) out[i] = st.blk[16 - st.left + i];
  st.left -= 8;
  return true;
}
#endif
// x^(p-2): constant-time in x (fixed exponent, branch on public exponent bits only); inv(0) == 0
template <class FP>
SP_HD fe_t fe_inv_fermat(const fe_t& x) {
  uint32_t e[8], bw = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = sp_subb(FP::P(i), i == 0 ? 2u : 0u, bw);
  return fe_pow<FP>(x, e);
}
// Inversion of a PUBLIC value (host: variable-time division steps, the binary xgcd behind them; device: Fermat). inv(0) == 0.
template <class FP>
SP_HD fe_t fe_inv_vartime(const fe_t& x) {
#if !defined(__HIP_DEVICE_COMPILE__)
  fe_t y;
  if (fe_inv_host_safegcd<FP>(x, &y) || fe_inv_host_xgcd<FP>(x, &y)) return y;
#endif
  return fe_inv_fermat<FP>(x);
}
// Inversion of a value that may be SECRET (witness-derived: the verifier circuit's blinded evaluations, the Z coordinates of commitments being
// normalised). The reference's ff::Field::invert is constant-time. Host: the argument is multiplied by a fresh random mask m first, so the variable-time
// xgcd runs on x*m — uniformly distributed whatever x is — and the mask is taken off again: (x m)^-1 m = x^-1 exactly; without an entropy source
// (or a zero mask) it is Fermat's fixed exponentiation. Device: Fermat. inv(0) == 0.
// What still shows in the timing: whether x == 0 (x m == 0 leaves the xgcd at once and takes Fermat's path; the reference's callers never invert a
// secret zero - a zero Z coordinate is the identity, a public fact of the proof). The mask is uniform on [1, 2^255), a subset of the field: x m is
// then uniform on a set of 2^255 - 1 field elements that does not depend on x != 0, which is all the blinding needs.
template <class FP>
SP_HD fe_t fe_inv(const fe_t& x) {
  static_assert(FP::P(7) >= 0x80000000u, "fe_inv: the 255-bit mask is canonical only below a modulus of 256 bits");
#if !defined(__HIP_DEVICE_COMPILE__)
  fe_t m;
  if (sp_blind_mask(m.v)) {
    m.v[7] &= 0x7fffffffu;  // < 2^255 < p for both fields: canonical without a comparison
    fe_t y;
    if (!fe_is_zero(m)) {
      const fe_t xm = fe_mul_host64<FP>(x, m);
      if (fe_inv_host_safegcd<FP>(xm, &y) || fe_inv_host_xgcd<FP>(xm, &y)) return fe_mul_host64<FP>(y, m);
    }
  }
#endif
  return fe_inv_fermat<FP>(x);
}
