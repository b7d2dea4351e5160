// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Restates src/polys/{eq,multilinear,univariate}.rs of the reference.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>

#include "field.hpp"

namespace oracle {

static const size_t EFF_MAX = (size_t)-1;

// EqPolynomial::evals_from_points (src/polys/eq.rs:59-92): r[0] ends up on the index MSB.
template <class F>
std::vector<F> eq_evals_from_points(const std::vector<F>& r) {
  size_t ell = r.size();
  std::vector<F> evals((size_t)1 << ell, F::zero());
  size_t size = 1;
  evals[0] = F::one();
  for (size_t k = ell; k-- > 0;) {
#pragma omp parallel for schedule(static) if (size >= 8192)
    for (size_t i = 0; i < size; ++i) {
      F y = evals[i] * r[k];
      evals[size + i] = y;
      evals[i] = evals[i] - y;
    }
    size *= 2;
  }
  return evals;
}

// EqPolynomial::evaluate (src/polys/eq.rs:42-47)
template <class F>
F eq_evaluate(const std::vector<F>& r, const std::vector<F>& rx) {
  F acc = F::one();
  for (size_t i = 0; i < r.size(); ++i) acc = acc * (rx[i] * r[i] + (F::one() - rx[i]) * (F::one() - r[i]));
  return acc;
}

// MultilinearPolynomial (src/polys/multilinear.rs:34-164)
template <class F>
struct MultilinearPolynomial {
  std::vector<F> Z;
  size_t lo_eff = EFF_MAX, hi_eff = EFF_MAX;
  MultilinearPolynomial() {}
  explicit MultilinearPolynomial(std::vector<F> z) : Z(std::move(z)) {}
  MultilinearPolynomial(std::vector<F> z, size_t lo, size_t hi) : Z(std::move(z)), lo_eff(lo), hi_eff(hi) {}

  size_t eff_pairs() const {  // multilinear.rs:78-84
    size_t n = Z.size() / 2;
    return std::max(std::min(lo_eff, n), std::min(hi_eff, n));
  }
  // multilinear.rs:95-164, all three zero-structure branches
  void bind_poly_var_top(const F& r) {
    if (Z.size() < 2) throw std::runtime_error("bind_poly_var_top: need >= 2 elements");
    size_t n = Z.size() / 2;
    size_t lo = std::min(lo_eff, n), hi = std::min(hi_eff, n);
    size_t eff = std::max(lo, hi);
    F one_minus_r = F::one() - r;
    // (the OpenMP loops mirror the reference's rayon par_iter_mut, :130-136; element-wise, so the result is thread-count independent)
    if (hi == 0) {
#pragma omp parallel for schedule(static) if (lo >= 4096)
      for (size_t i = 0; i < lo; ++i) Z[i] = Z[i] * one_minus_r;
    } else if (hi <= lo) {
#pragma omp parallel for schedule(static) if (hi >= 4096)
      for (size_t i = 0; i < hi; ++i) Z[i] = Z[i] + r * (Z[n + i] - Z[i]);
#pragma omp parallel for schedule(static) if (lo - hi >= 4096)
      for (size_t i = hi; i < lo; ++i) Z[i] = Z[i] * one_minus_r;
    } else {
#pragma omp parallel for schedule(static) if (lo >= 4096)
      for (size_t i = 0; i < lo; ++i) Z[i] = Z[i] + r * (Z[n + i] - Z[i]);
#pragma omp parallel for schedule(static) if (hi - lo >= 4096)
      for (size_t i = lo; i < hi; ++i) Z[i] = r * Z[n + i];
    }
    Z.resize(n);
    lo_eff = std::min(eff, n / 2);
    hi_eff = eff > n / 2 ? eff - n / 2 : 0;
  }
};

// test-side helper of the reference (src/polys/multilinear.rs:218-231): direct evaluation
template <class F>
F multilinear_evaluate(const std::vector<F>& Z, const std::vector<F>& r) {
  std::vector<F> chis = eq_evals_from_points(r);
  F acc = F::zero();
  for (size_t i = 0; i < Z.size(); ++i) acc = acc + chis[i] * Z[i];
  return acc;
}

// SparsePolynomial::evaluate (src/polys/multilinear.rs:190-207)
template <class F>
F sparse_poly_evaluate(size_t num_vars, const std::vector<F>& Z, const std::vector<F>& r) {
  if (num_vars != r.size()) throw std::runtime_error("sparse_poly_evaluate: arity");
  size_t npow = 1, num_vars_z = 0;
  while (npow < Z.size()) {
    npow <<= 1;
    ++num_vars_z;
  }
  std::vector<F> tail(r.begin() + (num_vars - 1 - num_vars_z), r.end());
  std::vector<F> chis = eq_evals_from_points(tail);
  F partial = F::zero();
  for (size_t i = 0; i < Z.size(); ++i) partial = partial + Z[i] * chis[i];
  F common = F::one();
  for (size_t i = 0; i < num_vars - 1 - num_vars_z; ++i) common = common * (F::one() - r[i]);
  return common * partial;
}

// UniPoly / CompressedUniPoly (src/polys/univariate.rs:30-190)
template <class F>
struct UniPoly {
  std::vector<F> coeffs;
  static UniPoly from_evals(const std::vector<F>& e) {
    UniPoly p;
    if (e.size() == 3) {  // univariate.rs:84-93
      F c = e[0];
      F a = (e[0] - e[1].dbl() + e[2]) * F::two_inv();
      F b = e[1] - c - a;
      p.coeffs = {c, b, a};
    } else if (e.size() == 4) {  // univariate.rs:102-118
      F d = e[0];
      F e1_3 = e[1].dbl() + e[1], e2_3 = e[2].dbl() + e[2];
      F delta3 = e[3] - e2_3 + e1_3 - e[0];
      F a = delta3 * F::from_u64(6).inv();
      F delta2 = e[2] - e[1].dbl() + e[0];
      F b = delta2 * F::two_inv() - (a.dbl() + a);
      F c = e[1] - d - b - a;
      p.coeffs = {d, c, b, a};
    } else {
      throw std::runtime_error("UniPoly::from_evals: only degree 2/3 restated");
    }
    return p;
  }
  size_t degree() const { return coeffs.size() - 1; }
  F evaluate(const F& r) const {  // univariate.rs:136-144
    F eval = coeffs[0], power = r;
    for (size_t i = 1; i < coeffs.size(); ++i) {
      eval = eval + power * coeffs[i];
      power = power * r;
    }
    return eval;
  }
  std::vector<F> compress() const {  // univariate.rs:147-153 (linear term dropped)
    std::vector<F> c;
    c.push_back(coeffs[0]);
    for (size_t i = 2; i < coeffs.size(); ++i) c.push_back(coeffs[i]);
    return c;
  }
  static UniPoly decompress(const std::vector<F>& c, const F& hint) {  // univariate.rs:166-179
    F lin = hint - c[0] - c[0];
    for (size_t i = 1; i < c.size(); ++i) lin = lin - c[i];
    UniPoly p;
    p.coeffs.push_back(c[0]);
    p.coeffs.push_back(lin);
    for (size_t i = 1; i < c.size(); ++i) p.coeffs.push_back(c[i]);
    return p;
  }
  // to_transcript_bytes (univariate.rs:182-190): compressed coeffs, each to_repr() LITTLE-endian
  std::vector<uint8_t> to_transcript_bytes() const {
    std::vector<F> c = compress();
    std::vector<uint8_t> b(32 * c.size());
    for (size_t i = 0; i < c.size(); ++i) c[i].to_repr(b.data() + 32 * i);
    return b;
  }
};

}  // namespace oracle
