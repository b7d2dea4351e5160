// Short-Weierstrass group arithmetic for the bench engine's curve T256 (halo2curves::t256 = "Tom-256":
// y^2 = x^3 - 3x + b over the field of src/provider/pt256.rs:56, group order = the P-256 base prime).
// Jacobian coordinates, __host__ __device__: kernels use it for bucket/tree sums, the host side of the library
// for the short sequential tails (window Horner, normalisation), where one CPU core beats one GPU lane.
// Replaces CurveExt::{add_mixed_vartime, double, +} and Curve::batch_normalize as used by src/provider/msm.rs:24-57,
// :150-175 and src/provider/traits.rs:194-198.
#pragma once
#include "field.hpp"

typedef FpP B;  // base field

struct aff_t {
  fe_t x, y;  // (0,0) == identity
};
struct jac_t {
  fe_t x, y, z;  // z == 0 -> identity
};

struct T256 {
  static SP_HD fe_t b() {  // 0xb441071b12f4a0366fb552f8e21ed4ac36b06aceeb354224863e60f20219fc56 in Montgomery form
    fe_t c;
    c.v[0] = 0x0219fc56u; c.v[1] = 0x863e60f2u; c.v[2] = 0xeb354224u; c.v[3] = 0x36b06aceu;
    c.v[4] = 0xe21ed4acu; c.v[5] = 0x6fb552f8u; c.v[6] = 0x12f4a036u; c.v[7] = 0xb441071bu;
    return fe_from_canonical<B>(c);
  }
};

SP_HD bool aff_is_identity(const aff_t& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
SP_HD bool jac_is_identity(const jac_t& p) { return fe_is_zero(p.z); }
SP_HD jac_t jac_identity() {
  jac_t r;
  r.x = fe_one<B>();
  r.y = fe_one<B>();
  r.z = fe_zero();
  return r;
}
SP_HD jac_t jac_from_affine(const aff_t& p) {
  if (aff_is_identity(p)) return jac_identity();
  jac_t r;
  r.x = p.x;
  r.y = p.y;
  r.z = fe_one<B>();
  return r;
}
SP_HD aff_t aff_neg(const aff_t& p) {
  aff_t r;
  r.x = p.x;
  r.y = fe_neg<B>(p.y);  // neg(0) == 0 keeps the identity encoding
  return r;
}

// a = -3 doubling (dbl-2001-b): 3M + 5S
SP_HD jac_t jac_dbl(const jac_t& p) {
  if (jac_is_identity(p) || fe_is_zero(p.y)) return jac_identity();
  fe_t delta = fe_sqr<B>(p.z), gamma = fe_sqr<B>(p.y), beta = fe_mul<B>(p.x, gamma);
  fe_t t = fe_mul<B>(fe_sub<B>(p.x, delta), fe_add<B>(p.x, delta));
  fe_t alpha = fe_add<B>(fe_dbl<B>(t), t);
  fe_t beta4 = fe_dbl<B>(fe_dbl<B>(beta));
  jac_t r;
  r.x = fe_sub<B>(fe_sqr<B>(alpha), fe_dbl<B>(beta4));
  r.z = fe_sub<B>(fe_sub<B>(fe_sqr<B>(fe_add<B>(p.y, p.z)), gamma), delta);
  fe_t g2 = fe_sqr<B>(gamma);
  fe_t g8 = fe_dbl<B>(fe_dbl<B>(fe_dbl<B>(g2)));
  r.y = fe_sub<B>(fe_mul<B>(alpha, fe_sub<B>(beta4, r.x)), g8);
  return r;
}

// Jacobian + affine (madd-2007-bl): 7M + 4S, all special cases handled
SP_HD jac_t jac_add_mixed(const jac_t& p, const aff_t& q) {
  if (aff_is_identity(q)) return p;
  if (jac_is_identity(p)) return jac_from_affine(q);
  fe_t z1z1 = fe_sqr<B>(p.z);
  fe_t u2 = fe_mul<B>(q.x, z1z1), s2 = fe_mul<B>(fe_mul<B>(q.y, p.z), z1z1);
  fe_t h = fe_sub<B>(u2, p.x), rr = fe_dbl<B>(fe_sub<B>(s2, p.y));
  if (fe_is_zero(h)) {
    if (fe_is_zero(rr)) return jac_dbl(p);
    return jac_identity();
  }
  fe_t hh = fe_sqr<B>(h), i = fe_dbl<B>(fe_dbl<B>(hh)), j = fe_mul<B>(h, i), v = fe_mul<B>(p.x, i);
  jac_t r;
  r.x = fe_sub<B>(fe_sub<B>(fe_sqr<B>(rr), j), fe_dbl<B>(v));
  r.y = fe_sub<B>(fe_mul<B>(rr, fe_sub<B>(v, r.x)), fe_dbl<B>(fe_mul<B>(p.y, j)));
  r.z = fe_sub<B>(fe_sub<B>(fe_sqr<B>(fe_add<B>(p.z, h)), z1z1), hh);
  return r;
}

// Jacobian + Jacobian (add-2007-bl): 11M + 5S
SP_HD jac_t jac_add(const jac_t& p, const jac_t& q) {
  if (jac_is_identity(p)) return q;
  if (jac_is_identity(q)) return p;
  fe_t z1z1 = fe_sqr<B>(p.z), z2z2 = fe_sqr<B>(q.z);
  fe_t u1 = fe_mul<B>(p.x, z2z2), u2 = fe_mul<B>(q.x, z1z1);
  fe_t s1 = fe_mul<B>(fe_mul<B>(p.y, q.z), z2z2), s2 = fe_mul<B>(fe_mul<B>(q.y, p.z), z1z1);
  fe_t h = fe_sub<B>(u2, u1), rr = fe_dbl<B>(fe_sub<B>(s2, s1));
  if (fe_is_zero(h)) {
    if (fe_is_zero(rr)) return jac_dbl(p);
    return jac_identity();
  }
  fe_t i = fe_sqr<B>(fe_dbl<B>(h)), j = fe_mul<B>(h, i), v = fe_mul<B>(u1, i);
  jac_t r;
  r.x = fe_sub<B>(fe_sub<B>(fe_sqr<B>(rr), j), fe_dbl<B>(v));
  r.y = fe_sub<B>(fe_mul<B>(rr, fe_sub<B>(v, r.x)), fe_dbl<B>(fe_mul<B>(s1, j)));
  r.z = fe_mul<B>(fe_sub<B>(fe_sub<B>(fe_sqr<B>(fe_add<B>(p.z, q.z)), z1z1), z2z2), h);
  return r;
}

SP_HD aff_t jac_to_affine(const jac_t& p) {
  aff_t r;
  if (jac_is_identity(p)) {
    r.x = fe_zero();
    r.y = fe_zero();
    return r;
  }
  fe_t zi = fe_inv<B>(p.z), zi2 = fe_sqr<B>(zi);
  r.x = fe_mul<B>(p.x, zi2);
  r.y = fe_mul<B>(fe_mul<B>(p.y, zi2), zi);
  return r;
}
SP_HD bool aff_on_curve(const aff_t& p) {
  if (aff_is_identity(p)) return true;
  fe_t x3 = fe_mul<B>(fe_sqr<B>(p.x), p.x);
  fe_t rhs = fe_add<B>(fe_sub<B>(x3, fe_add<B>(fe_dbl<B>(p.x), p.x)), T256::b());
  return fe_eq(fe_sqr<B>(p.y), rhs);
}

// ---- XYZZ coordinates: (X, Y, ZZ, ZZZ) with x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; identity = (., ., 0, 0) ------------------------------------------
// Chains of additions inside the kernels run in this form (add-2008-s: 12M + 2S in four dependency levels; madd-2008-s: 8M + 2S) instead of
// Jacobian (add-2007-bl 11M + 5S in five levels; madd-2007-bl 7M + 4S); points enter and leave through two products each. The group element is the
// same, so every normalised byte downstream is too.
struct xyzz_t {
  fe_t x, y, zz, zzz;
};
SP_HD bool xyzz_is_identity(const xyzz_t& p) { return fe_is_zero(p.zz); }
SP_HD xyzz_t xyzz_identity() {
  xyzz_t r;
  r.x = r.y = r.zz = r.zzz = fe_zero();
  return r;
}
SP_HD xyzz_t xyzz_from_jac(const jac_t& p) {
  if (jac_is_identity(p)) return xyzz_identity();
  xyzz_t r;
  r.x = p.x;
  r.y = p.y;
  r.zz = fe_sqr<B>(p.z);
  r.zzz = fe_mul<B>(r.zz, p.z);
  return r;
}
SP_HD xyzz_t xyzz_from_affine(const aff_t& p) {
  if (aff_is_identity(p)) return xyzz_identity();
  xyzz_t r;
  r.x = p.x;
  r.y = p.y;
  r.zz = r.zzz = fe_one<B>();
  return r;
}
// Jacobian (X ZZ, Y ZZZ, ZZ): x = X ZZ / ZZ^2, y = Y ZZZ / ZZ^3 = Y / ZZZ since ZZ^3 = ZZZ^2
SP_HD jac_t xyzz_to_jac(const xyzz_t& p) {
  if (xyzz_is_identity(p)) return jac_identity();
  jac_t r;
  r.x = fe_mul<B>(p.x, p.zz);
  r.y = fe_mul<B>(p.y, p.zzz);
  r.z = p.zz;
  return r;
}
// dbl-2008-s-1 with a = -3 (M = 3 (X - ZZ)(X + ZZ)): 9 products. Only ever on the P = Q path of an addition, so it is written for SIZE, not speed: one
// product that a nine-step loop runs on operands picked per step (~5 KB of code). Inlined as nine products (or as the Jacobian round trip it used to be)
// it put ~30 KB of never-executed code into every addition of every kernel, against a 64 KB instruction cache.
SP_HD xyzz_t xyzz_dbl(const xyzz_t& p) {
  if (xyzz_is_identity(p)) return xyzz_identity();
  const fe_t u = fe_dbl<B>(p.y);
  fe_t v = u, w = u, sv = u, m = u, x3 = u, y3 = u, wy = u, zz = u, a = u, b = u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int k = 0; k < 9; ++k) {
    switch (k) {
      case 0: a = u; b = u; break;                                          // V = U^2
      case 1: a = u; b = v; break;                                          // W = U V
      case 2: a = p.x; b = v; break;                                        // S = X V
      case 3: a = fe_sub<B>(p.x, p.zz); b = fe_add<B>(p.x, p.zz); break;    // (X - ZZ)(X + ZZ)
      case 4: a = m; b = m; break;                                          // M^2
      case 5: a = m; b = fe_sub<B>(sv, x3); break;                          // M (S - X3)
      case 6: a = w; b = p.y; break;                                        // W Y
      case 7: a = v; b = p.zz; break;                                       // ZZ3 = V ZZ
      default: a = w; b = p.zzz; break;                                     // ZZZ3 = W ZZZ
    }
    const fe_t r = fe_mul<B>(a, b);
    switch (k) {
      case 0: v = r; break;
      case 1: w = r; break;
      case 2: sv = r; break;
      case 3: m = fe_add<B>(fe_dbl<B>(r), r); break;
      case 4: x3 = fe_sub<B>(r, fe_dbl<B>(sv)); break;
      case 5: y3 = r; break;
      case 6: wy = r; break;
      case 7: zz = r; break;
      default: a = r; break;
    }
  }
  xyzz_t o;
  o.x = x3;
  o.y = fe_sub<B>(y3, wy);
  o.zz = zz;
  o.zzz = a;
  return o;
}
// madd-2008-s
SP_HD xyzz_t xyzz_add_mixed(const xyzz_t& p, const aff_t& q) {
  if (aff_is_identity(q)) return p;
  if (xyzz_is_identity(p)) return xyzz_from_affine(q);
  const fe_t u2 = fe_mul<B>(q.x, p.zz), s2 = fe_mul<B>(q.y, p.zzz);
  const fe_t pd = fe_sub<B>(u2, p.x), rd = fe_sub<B>(s2, p.y);
  if (fe_is_zero(pd)) return fe_is_zero(rd) ? xyzz_dbl(p) : xyzz_identity();
  const fe_t pp = fe_sqr<B>(pd), ppp = fe_mul<B>(pd, pp), qv = fe_mul<B>(p.x, pp);
  xyzz_t r;
  r.x = fe_sub<B>(fe_sub<B>(fe_sqr<B>(rd), ppp), fe_dbl<B>(qv));
  r.y = fe_sub<B>(fe_mul<B>(rd, fe_sub<B>(qv, r.x)), fe_mul<B>(p.y, ppp));
  r.zz = fe_mul<B>(p.zz, pp);
  r.zzz = fe_mul<B>(p.zzz, ppp);
  return r;
}
// add-2008-s
SP_HD xyzz_t xyzz_add(const xyzz_t& p, const xyzz_t& q) {
  if (xyzz_is_identity(p)) return q;
  if (xyzz_is_identity(q)) return p;
  const fe_t u1 = fe_mul<B>(p.x, q.zz), u2 = fe_mul<B>(q.x, p.zz), s1 = fe_mul<B>(p.y, q.zzz), s2 = fe_mul<B>(q.y, p.zzz);
  const fe_t pd = fe_sub<B>(u2, u1), rd = fe_sub<B>(s2, s1);
  if (fe_is_zero(pd)) return fe_is_zero(rd) ? xyzz_dbl(p) : xyzz_identity();
  const fe_t pp = fe_sqr<B>(pd), ppp = fe_mul<B>(pd, pp), qv = fe_mul<B>(u1, pp);
  xyzz_t r;
  r.x = fe_sub<B>(fe_sub<B>(fe_sqr<B>(rd), ppp), fe_dbl<B>(qv));
  r.y = fe_sub<B>(fe_mul<B>(rd, fe_sub<B>(qv, r.x)), fe_mul<B>(s1, ppp));
  r.zz = fe_mul<B>(fe_mul<B>(p.zz, q.zz), pp);
  r.zzz = fe_mul<B>(fe_mul<B>(p.zzz, q.zzz), ppp);
  return r;
}
