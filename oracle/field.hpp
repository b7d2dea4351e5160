// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// CPU restatement of the 256-bit prime-field arithmetic the reference gets from the un-vendored
// crate halo2curves 0.10.x (Cargo.toml:41-46) + ff 0.13: 4 x u64 little-endian limbs in
// Montgomery form (x * 2^256 mod p), canonical < p. The moduli are the ones the reference pins
// (src/provider/pt256.rs:47-56, pasta.rs:44-53). Prime-field add/mul/invert are determined by the
// modulus, so parity is anchored on those constants plus the transcript KAT
// (src/provider/keccak.rs:146-152), which exercises from_uniform/to_repr on the Pallas field.
//
// Delayed reduction (src/big_num/delayed_reduction.rs:41-84, montgomery.rs:39-177) is restated by
// its value contract only: reduce(sum a_i*b_i) == sum of Montgomery products, canonical.
#pragma once
#include <cstdint>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <string>
#include <vector>

namespace oracle {

typedef unsigned __int128 u128;

struct U256 {
  uint64_t l[4];
};

inline int cmp256(const uint64_t* a, const uint64_t* b) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}
inline uint64_t add256(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a[i] + b[i];
    r[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
inline uint64_t sub256(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}

inline U256 u256_from_hex(const char* hex) {
  U256 r{{0, 0, 0, 0}};
  size_t n = strlen(hex);
  for (size_t i = 0; i < n; ++i) {
    char c = hex[n - 1 - i];
    uint64_t v = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : c - 'A' + 10;
    r.l[i / 16] |= v << (4 * (i % 16));
  }
  return r;
}

// Runtime Montgomery parameters of one prime field.
struct FieldParams {
  U256 p;
  uint64_t inv;  // -p^{-1} mod 2^64
  U256 r1;       // 2^256 mod p  (Montgomery ONE)
  U256 r2;       // 2^512 mod p
  U256 r3;       // 2^768 mod p
  explicit FieldParams(const char* modulus_hex) {
    p = u256_from_hex(modulus_hex);
    // inv by Newton iteration
    uint64_t x = 1;
    for (int i = 0; i < 6; ++i) x *= 2 - p.l[0] * x;
    inv = (uint64_t)(0 - x);
    // r1 = 2^256 mod p by 256 doublings of 1 (mod p)
    U256 v{{1, 0, 0, 0}};
    auto dbl = [&](U256& a) {
      uint64_t c = add256(a.l, a.l, a.l);
      if (c || cmp256(a.l, p.l) >= 0) sub256(a.l, a.l, p.l);
    };
    for (int i = 0; i < 256; ++i) dbl(v);
    r1 = v;
    for (int i = 0; i < 256; ++i) dbl(v);
    r2 = v;
    for (int i = 0; i < 256; ++i) dbl(v);
    r3 = v;
  }
};

// CIOS Montgomery product on raw limbs.
inline void mont_mul_raw(uint64_t* out, const uint64_t* a, const uint64_t* b, const FieldParams& P) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a[j] * b[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * P.inv;
    c = (u128)m * P.p.l[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * P.p.l[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || cmp256(t, P.p.l) >= 0) sub256(t, t, P.p.l);
  memcpy(out, t, 32);
}

// A field element type bound to a parameter provider tag.
template <class Tag>
struct Fe {
  uint64_t l[4];  // Montgomery form, canonical
  static const FieldParams& P() { return Tag::params(); }

  static Fe zero() { return Fe{{0, 0, 0, 0}}; }
  static Fe one() {
    Fe r;
    memcpy(r.l, P().r1.l, 32);
    return r;
  }
  static Fe from_raw_mont(const uint64_t* limbs) {
    Fe r;
    memcpy(r.l, limbs, 32);
    return r;
  }
  // canonical integer (4 x u64 LE, < p) -> Montgomery
  static Fe from_canonical(const uint64_t* v) {
    Fe r;
    mont_mul_raw(r.l, v, P().r2.l, P());
    return r;
  }
  static Fe from_u64(uint64_t v) {
    uint64_t t[4] = {v, 0, 0, 0};
    return from_canonical(t);
  }
  static Fe from_i64(int64_t v) {
    if (v >= 0) return from_u64((uint64_t)v);
    return from_u64((uint64_t)(-(v + 1)) + 1).neg();
  }
  static Fe from_hex(const char* hex) {
    U256 u = u256_from_hex(hex);
    return from_canonical(u.l);
  }
  // halo2curves from_uniform_bytes (src/provider/traits.rs:275-280): 64 bytes as a 512-bit LE
  // integer reduced mod p. lo + hi*2^256: mont(lo,R2) = lo*R ; mont(hi,R3) = hi*2^256*R.
  static Fe from_uniform(const uint8_t* bytes64) {
    uint64_t lo[4], hi[4];
    memcpy(lo, bytes64, 32);
    memcpy(hi, bytes64 + 32, 32);
    Fe a, b;
    mont_mul_raw(a.l, lo, P().r2.l, P());
    mont_mul_raw(b.l, hi, P().r3.l, P());
    return a + b;
  }
  // to_repr(): canonical value, 4 x u64 LE (bytes LE on a little-endian host)
  void to_canonical(uint64_t* out) const {
    uint64_t one[4] = {1, 0, 0, 0};
    mont_mul_raw(out, l, one, P());
  }
  void to_repr(uint8_t* out32) const {
    uint64_t t[4];
    to_canonical(t);
    memcpy(out32, t, 32);
  }
  // transcript encoding of a scalar: to_repr reversed = big-endian (src/provider/traits.rs:282-286)
  void to_be_bytes(uint8_t* out32) const {
    uint8_t le[32];
    to_repr(le);
    for (int i = 0; i < 32; ++i) out32[i] = le[31 - i];
  }

  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Fe& o) const { return memcmp(l, o.l, 32) == 0; }
  bool operator!=(const Fe& o) const { return !(*this == o); }

  Fe operator+(const Fe& o) const {
    Fe r;
    uint64_t c = add256(r.l, l, o.l);
    if (c || cmp256(r.l, P().p.l) >= 0) sub256(r.l, r.l, P().p.l);
    return r;
  }
  Fe operator-(const Fe& o) const {
    Fe r;
    if (sub256(r.l, l, o.l)) add256(r.l, r.l, P().p.l);
    return r;
  }
  Fe neg() const {
    if (is_zero()) return *this;
    Fe r;
    sub256(r.l, P().p.l, l);
    return r;
  }
  Fe operator*(const Fe& o) const {
    Fe r;
    mont_mul_raw(r.l, l, o.l, P());
    return r;
  }
  Fe& operator+=(const Fe& o) { return *this = *this + o; }
  Fe& operator-=(const Fe& o) { return *this = *this - o; }
  Fe& operator*=(const Fe& o) { return *this = *this * o; }
  Fe dbl() const { return *this + *this; }
  Fe sqr() const { return *this * *this; }

  // x^(e) for a canonical 256-bit exponent
  Fe pow(const uint64_t* e) const {
    Fe acc = one();
    for (int i = 255; i >= 0; --i) {
      acc = acc.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) acc = acc * *this;
    }
    return acc;
  }
  // Fermat inverse; returns zero for zero (callers check, as the reference's CtOption does).
  Fe inv() const {
    uint64_t e[4], two[4] = {2, 0, 0, 0};
    sub256(e, P().p.l, two);
    return pow(e);
  }
  static Fe two_inv() { return from_u64(2).inv(); }
};

// ---- field tags -------------------------------------------------------------------------------
// T256 scalar field = P-256 base prime (src/provider/pt256.rs:55)
struct FqT256Tag {
  static const FieldParams& params() {
    static FieldParams P("ffffffff00000001000000000000000000000000ffffffffffffffffffffffff");
    return P;
  }
};
// T256 base field (src/provider/pt256.rs:56)
struct FpT256Tag {
  static const FieldParams& params() {
    static FieldParams P("ffffffff0000000100000000000000017e72b42b30e7317793135661b1c4b117");
    return P;
  }
};
// Pallas scalar field (src/provider/pasta.rs:44) — only for the transcript KAT.
struct FqPallasTag {
  static const FieldParams& params() {
    static FieldParams P("40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001");
    return P;
  }
};

typedef Fe<FqT256Tag> Fq;  // scalars of the bench engine
typedef Fe<FpT256Tag> Fp;  // curve coordinates of the bench engine
typedef Fe<FqPallasTag> FqPallas;

// Parallel sum of K field accumulators over i in [0, n): body(i, acc) adds item i into acc[0..K). Field addition is exact, so the thread
// count never changes the result (the reference's rayon fold/reduce, e.g. src/sumcheck.rs:1045-1100).
template <class F, int K, class Body>
inline void par_sum(size_t n, size_t min_parallel, Body body, F (&out)[K]) {
  for (int k = 0; k < K; ++k) out[k] = F::zero();
#ifdef _OPENMP
  if (n >= min_parallel && omp_get_max_threads() > 1) {
#pragma omp parallel
    {
      F local[K];
      for (int k = 0; k < K; ++k) local[k] = F::zero();
#pragma omp for schedule(static) nowait
      for (size_t i = 0; i < n; ++i) body(i, local);
#pragma omp critical
      for (int k = 0; k < K; ++k) out[k] = out[k] + local[k];
    }
    return;
  }
#endif
  for (size_t i = 0; i < n; ++i) body(i, out);
}

}  // namespace oracle
