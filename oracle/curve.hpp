// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Short-Weierstrass group arithmetic y^2 = x^3 + a x + b, Jacobian coordinates, restating what the
// reference takes from halo2curves 0.10.x (CurveExt::add_mixed_vartime / double / batch_normalize,
// call sites src/provider/msm.rs:37-38,153, src/provider/traits.rs:194-272).
//
// The bench engine's curve is halo2curves::t256 ("Tom-256": a = -3,
// b = 0xb441071b12f4a0366fb552f8e21ed4ac36b06aceeb354224863e60f20219fc56, G = (3, 0x5a6dd32d...f02d)).
// Those constants are NOT in /root/reference (src/provider/pt256.rs:18-23 imports them); they are
// restated here from the published curve and pinned by: G on curve over the reference's base
// modulus (pt256.rs:56) and order(G) == the reference's order string (pt256.rs:55) — checked in
// tests/test_oracle_kats.py with independent Python integers.
//
// PARITY UNPINNED: generator derivation. The reference derives Hyrax generators with halo2curves'
// hash_to_curve("from_uniform_bytes") over a SHAKE256 stream (src/provider/traits.rs:205-249); that
// map is third-party and absent, so `from_label` below is this build's own documented map
// (SHAKE256 stream -> x candidate -> try-and-increment). All group arithmetic is generator-agnostic.
#pragma once
#include <vector>

#include "field.hpp"
#include "keccak.hpp"

namespace oracle {

struct Affine {
  Fp x, y;  // (0,0) encodes the identity (never on the curve since b != 0)
  bool is_identity() const { return x.is_zero() && y.is_zero(); }
  bool operator==(const Affine& o) const { return x == o.x && y == o.y; }
};

struct T256Curve {
  static const Fp& a() {
    static Fp v = Fp::from_u64(3).neg();
    return v;
  }
  static const Fp& b() {
    static Fp v = Fp::from_hex("b441071b12f4a0366fb552f8e21ed4ac36b06aceeb354224863e60f20219fc56");
    return v;
  }
  static Affine generator() {
    return Affine{Fp::from_u64(3), Fp::from_hex("5a6dd32df58708e64e97345cbe66600decd9d538a351bb3c30b4954925b1f02d")};
  }
};

struct Jac {
  Fp x, y, z;
  static Jac identity() { return Jac{Fp::one(), Fp::one(), Fp::zero()}; }
  static Jac from_affine(const Affine& p) {
    if (p.is_identity()) return identity();
    return Jac{p.x, p.y, Fp::one()};
  }
  bool is_identity() const { return z.is_zero(); }

  Jac dbl() const {
    if (is_identity() || y.is_zero()) return identity();
    Fp xx = x.sqr(), yy = y.sqr(), yyyy = yy.sqr(), zz = z.sqr();
    Fp s = ((x + yy).sqr() - xx - yyyy).dbl();
    Fp m = xx.dbl() + xx + T256Curve::a() * zz.sqr();
    Fp t = m.sqr() - s.dbl();
    Jac r;
    r.x = t;
    r.y = m * (s - t) - yyyy.dbl().dbl().dbl();
    r.z = (y + z).sqr() - yy - zz;
    return r;
  }
  Jac neg() const { return Jac{x, y.neg(), z}; }

  Jac add(const Jac& o) const {
    if (is_identity()) return o;
    if (o.is_identity()) return *this;
    Fp z1z1 = z.sqr(), z2z2 = o.z.sqr();
    Fp u1 = x * z2z2, u2 = o.x * z1z1;
    Fp s1 = y * o.z * z2z2, s2 = o.y * z * z1z1;
    Fp h = u2 - u1, rr = (s2 - s1).dbl();
    if (h.is_zero()) {
      if (rr.is_zero()) return dbl();
      return identity();
    }
    Fp i = h.dbl().sqr(), j = h * i, v = u1 * i;
    Jac r;
    r.x = rr.sqr() - j - v.dbl();
    r.y = rr * (v - r.x) - (s1 * j).dbl();
    r.z = ((z + o.z).sqr() - z1z1 - z2z2) * h;
    return r;
  }
  // add_mixed_vartime (src/provider/msm.rs:37-38): Jacobian + affine
  Jac add_mixed(const Affine& o) const {
    if (o.is_identity()) return *this;
    if (is_identity()) return from_affine(o);
    Fp z1z1 = z.sqr();
    Fp u2 = o.x * z1z1, s2 = o.y * z * z1z1;
    Fp h = u2 - x, rr = (s2 - y).dbl();
    if (h.is_zero()) {
      if (rr.is_zero()) return dbl();
      return identity();
    }
    Fp hh = h.sqr(), i = hh.dbl().dbl(), j = h * i, v = x * i;
    Jac r;
    r.x = rr.sqr() - j - v.dbl();
    r.y = rr * (v - r.x) - (y * j).dbl();
    r.z = (z + h).sqr() - z1z1 - hh;
    return r;
  }
  Affine to_affine() const {
    if (is_identity()) return Affine{Fp::zero(), Fp::zero()};
    Fp zi = z.inv(), zi2 = zi.sqr();
    return Affine{x * zi2, y * zi2 * zi};
  }
};

inline Affine affine_neg(const Affine& p) { return Affine{p.x, p.y.neg()}; }

inline bool on_curve(const Affine& p) {
  if (p.is_identity()) return true;
  return p.y.sqr() == p.x.sqr() * p.x + T256Curve::a() * p.x + T256Curve::b();
}

// Montgomery's trick (DlogGroup::batch_affine, src/provider/traits.rs:194-198)
inline std::vector<Affine> batch_affine(const std::vector<Jac>& pts) {
  size_t n = pts.size();
  std::vector<Affine> out(n);
  std::vector<Fp> pref(n);
  Fp acc = Fp::one();
  for (size_t i = 0; i < n; ++i) {
    pref[i] = acc;
    if (!pts[i].is_identity()) acc = acc * pts[i].z;
  }
  Fp inv = acc.inv();
  for (size_t i = n; i-- > 0;) {
    if (pts[i].is_identity()) {
      out[i] = Affine{Fp::zero(), Fp::zero()};
      continue;
    }
    Fp zi = inv * pref[i];
    inv = inv * pts[i].z;
    Fp zi2 = zi.sqr();
    out[i] = Affine{pts[i].x * zi2, pts[i].y * zi2 * zi};
  }
  return out;
}

// Scalar multiplication by a canonical 256-bit integer (double-and-add; oracle clarity over speed).
inline Jac scalar_mul_canonical(const Jac& p, const uint64_t k[4]) {
  Jac acc = Jac::identity();
  for (int i = 255; i >= 0; --i) {
    acc = acc.dbl();
    if ((k[i / 64] >> (i % 64)) & 1) acc = acc.add(p);
  }
  return acc;
}
inline Jac scalar_mul(const Jac& p, const Fq& k) {
  uint64_t c[4];
  k.to_canonical(c);
  return scalar_mul_canonical(p, c);
}

// Transcript encoding of a point: affine x BE || y BE (src/provider/traits.rs:288-305).
inline void point_to_transcript_bytes(const Affine& a, uint8_t out[64]) {
  a.x.to_be_bytes(out);
  a.y.to_be_bytes(out + 32);
}

// This build's generator derivation (see PARITY UNPINNED above). Mirrors the *shape* of
// src/provider/traits.rs:205-249: one SHAKE256(label) stream, 32 uniform bytes per generator.
// Map: x = bytes as LE integer mod p; while x^3+ax+b is a non-residue, x += 1;
// y = rhs^((p+1)/4) (p = 3 mod 4), take the root whose canonical value is even.
inline std::vector<Affine> from_label(const char* label, size_t n) {
  Shake256 sh;
  sh.update((const uint8_t*)label, strlen(label));
  std::vector<Affine> out(n);
  const FieldParams& P = Fp::P();
  uint64_t e[4], one[4] = {1, 0, 0, 0};
  add256(e, P.p.l, one);  // p+1 (no overflow: p < 2^256 - 1)
  for (int i = 0; i < 3; ++i) e[i] = (e[i] >> 2) | (e[i + 1] << 62);
  e[3] >>= 2;
  for (size_t i = 0; i < n; ++i) {
    uint8_t buf[64];
    memset(buf, 0, 64);
    sh.read(buf, 32);
    Fp x = Fp::from_uniform(buf);
    for (;;) {
      Fp rhs = x.sqr() * x + T256Curve::a() * x + T256Curve::b();
      Fp y = rhs.pow(e);
      if (y.sqr() == rhs && !rhs.is_zero()) {
        uint64_t c[4];
        y.to_canonical(c);
        if (c[0] & 1) y = y.neg();
        out[i] = Affine{x, y};
        break;
      }
      x = x + Fp::one();
    }
  }
  return out;
}

}  // namespace oracle
