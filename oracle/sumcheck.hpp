// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Restates src/sumcheck.rs: verify (:67-114), compute_eval_points_quad (:128-174), prove_quad
// (:190-247), prove_cubic_with_three_inputs (:502-571) and eq_sumcheck::EqSumCheckInstance
// (:920-1429: split-eq tables after Gruen, claim-derived evaluations after BDDT).
// Delayed reduction is restated by value (sum of Montgomery products, canonical).
#pragma once
#include <vector>

#include "keccak.hpp"
#include "polys.hpp"

namespace oracle {

template <class F>
struct SumcheckProof {
  std::vector<std::vector<F>> compressed_polys;

  // src/sumcheck.rs:67-114
  bool verify(const F& claim, size_t num_rounds, size_t degree_bound, Transcript& tr, F* e_out, std::vector<F>* r_out) const {
    F e = claim;
    std::vector<F> r;
    if (compressed_polys.size() != num_rounds) return false;
    for (size_t i = 0; i < num_rounds; ++i) {
      UniPoly<F> poly = UniPoly<F>::decompress(compressed_polys[i], e);
      if (poly.degree() != degree_bound) return false;
      std::vector<uint8_t> b = poly.to_transcript_bytes();
      tr.absorb_bytes("p", b.data(), b.size());
      F r_i = tr.squeeze<F>("c");
      r.push_back(r_i);
      e = poly.evaluate(r_i);
    }
    *e_out = e;
    *r_out = r;
    return true;
  }
};

// src/sumcheck.rs:128-174
template <class F>
void compute_eval_points_quad(const MultilinearPolynomial<F>& A, const MultilinearPolynomial<F>& B, F* eval0, F* tinf) {
  size_t full_len = A.Z.size() / 2;
  size_t len = std::min(std::min(A.eff_pairs(), B.eff_pairs()), full_len);
  F tot[2];
  par_sum<F, 2>(len, 4096, [&](size_t i, F* acc) {
    acc[0] = acc[0] + A.Z[i] * B.Z[i];
    acc[1] = acc[1] + (A.Z[full_len + i] - A.Z[i]) * (B.Z[full_len + i] - B.Z[i]);
  }, tot);
  *eval0 = tot[0];
  *tinf = tot[1];
}

// src/sumcheck.rs:190-247
template <class F>
void prove_quad(const F& claim, size_t num_rounds, MultilinearPolynomial<F>& A, MultilinearPolynomial<F>& B, Transcript& tr,
                SumcheckProof<F>* proof, std::vector<F>* r_out, std::vector<F>* final_evals) {
  F claim_per_round = claim;
  proof->compressed_polys.clear();
  r_out->clear();
  for (size_t round = 0; round < num_rounds; ++round) {
    F e0, tinf;
    compute_eval_points_quad(A, B, &e0, &tinf);
    F e2 = claim_per_round + claim_per_round - (e0 + e0 + e0) + tinf + tinf;
    UniPoly<F> poly = UniPoly<F>::from_evals({e0, claim_per_round - e0, e2});
    std::vector<uint8_t> b = poly.to_transcript_bytes();
    tr.absorb_bytes("p", b.data(), b.size());
    F r_i = tr.squeeze<F>("c");
    r_out->push_back(r_i);
    proof->compressed_polys.push_back(poly.compress());
    claim_per_round = poly.evaluate(r_i);
    A.bind_poly_var_top(r_i);
    B.bind_poly_var_top(r_i);
  }
  *final_evals = {A.Z[0], B.Z[0]};
}

// src/sumcheck.rs:934-1429
template <class F>
struct EqSumCheckInstance {
  size_t init_num_vars, first_half, second_half, round;
  std::vector<F> taus;
  F eval_eq_left;
  std::vector<std::vector<F>> poly_eq_left, poly_eq_right;
  struct Triple {
    F eq0, slope, eqm1;
  };
  std::vector<Triple> eq_tau;

  static std::vector<std::vector<F>> compute_eq_polynomials(const std::vector<F>& ts) {  // :960-979
    std::vector<std::vector<F>> result;
    result.push_back({F::one()});
    for (size_t i = 0; i < ts.size(); ++i) {
      const std::vector<F>& prev = result[i];
      std::vector<F> next(prev.size() * 2);
      for (size_t k = 0; k < prev.size(); ++k) {
        F hi = prev[k] * ts[i];
        next[prev.size() + k] = hi;
        next[k] = prev[k] - hi;
      }
      result.push_back(next);
    }
    return result;
  }
  explicit EqSumCheckInstance(const std::vector<F>& taus_) : taus(taus_) {  // :956-1016
    size_t l = taus.size();
    init_num_vars = l;
    first_half = l / 2;
    second_half = l - first_half;
    round = 1;
    eval_eq_left = F::one();
    std::vector<F> left, right;
    for (size_t i = first_half; i-- > 1;) left.push_back(taus[i]);  // taus[1..first_half] reversed
    for (size_t i = l; i-- > first_half;) right.push_back(taus[i]);
    poly_eq_left = compute_eq_polynomials(left);
    poly_eq_right = compute_eq_polynomials(right);
    for (size_t i = 0; i < l; ++i) {
      F om = F::one() - taus[i];
      F sl = taus[i] - om;
      eq_tau.push_back(Triple{om, sl, om - sl});
    }
  }
  // weight table lookup shared by the sums: E~(id)
  void sums(const MultilinearPolynomial<F>& A, const MultilinearPolynomial<F>& B, const MultilinearPolynomial<F>& C, F* t0, F* tinf,
            F* tm1 /*nullable*/) const {
    size_t half_p = A.Z.size() / 2;
    F a0 = F::zero(), ai = F::zero(), am = F::zero();
    auto elem = [&](size_t id, F* e0, F* ei, F* em) {
      *e0 = A.Z[id] * B.Z[id] - C.Z[id];
      *ei = (A.Z[id + half_p] - A.Z[id]) * (B.Z[id + half_p] - B.Z[id]);
      if (tm1) {
        F ma = A.Z[id].dbl() - A.Z[id + half_p], mb = B.Z[id].dbl() - B.Z[id + half_p], mc = C.Z[id].dbl() - C.Z[id + half_p];
        *em = ma * mb - mc;
      }
    };
    F tot[3];
    if (round < first_half) {  // :1041-1105
      const std::vector<F>& el = poly_eq_left[first_half - round];
      const std::vector<F>& er = poly_eq_right[second_half];
      par_sum<F, 3>(el.size(), 4, [&](size_t x_out, F* acc) {
        F i0 = F::zero(), ii = F::zero(), im = F::zero();
        for (size_t x_in = 0; x_in < er.size(); ++x_in) {
          size_t id = (x_out << second_half) | x_in;
          F e0, ei, em;
          elem(id, &e0, &ei, &em);
          i0 = i0 + er[x_in] * e0;
          ii = ii + er[x_in] * ei;
          if (tm1) im = im + er[x_in] * em;
        }
        acc[0] = acc[0] + el[x_out] * i0;
        acc[1] = acc[1] + el[x_out] * ii;
        if (tm1) acc[2] = acc[2] + el[x_out] * im;
      }, tot);
    } else {  // :1107-1147
      const std::vector<F>& er = poly_eq_right[init_num_vars - round];
      par_sum<F, 3>(half_p, 4096, [&](size_t id, F* acc) {
        F e0, ei, em;
        elem(id, &e0, &ei, &em);
        acc[0] = acc[0] + er[id] * e0;
        acc[1] = acc[1] + er[id] * ei;
        if (tm1) acc[2] = acc[2] + er[id] * em;
      }, tot);
    }
    a0 = tot[0];
    ai = tot[1];
    am = tot[2];
    *t0 = a0;
    *tinf = ai;
    if (tm1) *tm1 = am;
  }
  static void finish(const F& s_0, const F& e1, const F& s_leading, const F& s_m1, F* ev0, F* ev2, F* ev3) {  // :1303-1320
    F half = F::two_inv();
    F c1 = (e1 - s_m1) * half - s_leading;
    F c2 = (e1 + s_m1) * half - s_0;
    F inner_2 = c2 + s_leading.dbl();
    *ev2 = s_0 + (c1 + inner_2.dbl()).dbl();
    F c3_3 = s_leading.dbl() + s_leading;
    F inner_3 = c2 + c3_3;
    F mid_3 = c1 + inner_3.dbl() + inner_3;
    *ev3 = s_0 + mid_3.dbl() + mid_3;
    *ev0 = s_0;
  }
  // evaluation_points_zero_check_round0 (:1163-1271): only t_inf is summed (C is not read), t(0) = 0 and the claim is 0
  void evaluation_points_zero_check_round0(const MultilinearPolynomial<F>& A, const MultilinearPolynomial<F>& B, F* ev0, F* ev2, F* ev3) const {
    if (round != 1) throw std::runtime_error("zero-check round-0 skip is only valid at round 0");
    size_t half_p = A.Z.size() / 2;
    F tinf = F::zero();
    if (round < first_half) {
      const std::vector<F>& el = poly_eq_left[first_half - round];
      const std::vector<F>& er = poly_eq_right[second_half];
      for (size_t x_out = 0; x_out < el.size(); ++x_out) {
        F inner = F::zero();
        for (size_t x_in = 0; x_in < er.size(); ++x_in) {
          size_t id = (x_out << second_half) | x_in;
          inner = inner + er[x_in] * ((A.Z[id + half_p] - A.Z[id]) * (B.Z[id + half_p] - B.Z[id]));
        }
        tinf = tinf + el[x_out] * inner;
      }
    } else {
      const std::vector<F>& er = poly_eq_right[init_num_vars - round];
      for (size_t id = 0; id < half_p; ++id) tinf = tinf + er[id] * ((A.Z[id + half_p] - A.Z[id]) * (B.Z[id + half_p] - B.Z[id]));
    }
    F p = eval_eq_left;
    const Triple& T = eq_tau[round - 1];
    F l_1_p = (T.eq0 + T.slope) * p;
    F s_0 = F::zero(), s_1 = F::zero(), s_leading = T.slope * p * tinf, s_m1;
    if (!l_1_p.is_zero()) {  // derive_from_claim with t_0 = 0, claim = 0
      F t_1 = s_1 * l_1_p.inv();
      F t_m1 = tinf.dbl() + F::zero() - t_1;
      s_m1 = T.eqm1 * p * t_m1;
    } else {  // :1244-1268
      s_m1 = T.eqm1 * p * tinf.dbl();
    }
    finish(s_0, s_1, s_leading, s_m1, ev0, ev2, ev3);
  }
  // :1025-1156 + derive_from_claim :1277-1324 + fallback :1327-1396
  void evaluation_points_cubic_with_three_inputs(const MultilinearPolynomial<F>& A, const MultilinearPolynomial<F>& B,
                                                 const MultilinearPolynomial<F>& C, const F& claim, F* ev0, F* ev2, F* ev3) const {
    F t0, tinf;
    sums(A, B, C, &t0, &tinf, nullptr);
    F p = eval_eq_left;
    const Triple& T = eq_tau[round - 1];
    F l_0_p = T.eq0 * p;
    F l_1_p = (T.eq0 + T.slope) * p;
    if (!l_1_p.is_zero()) {
      F l_1_p_inv = l_1_p.inv();
      F s_0 = l_0_p * t0;
      F s_1 = claim - s_0;
      F t_1 = s_1 * l_1_p_inv;
      F s_leading = T.slope * p * tinf;
      F t_m1 = tinf.dbl() + t0.dbl() - t_1;
      F s_m1 = T.eqm1 * p * t_m1;
      finish(s_0, s_1, s_leading, s_m1, ev0, ev2, ev3);
    } else {
      F t0b, tinfb, tm1;
      sums(A, B, C, &t0b, &tinfb, &tm1);
      F s_0 = T.eq0 * p * t0;
      F s_leading = T.slope * p * tinf;
      F s_m1 = T.eqm1 * p * tm1;
      finish(s_0, claim - s_0, s_leading, s_m1, ev0, ev2, ev3);
    }
  }
  void bound(const F& r) {  // :1399-1405
    F tau = taus[round - 1];
    eval_eq_left = eval_eq_left * (F::one() - tau - r + (r * tau).dbl());
    round += 1;
  }
};

// src/sumcheck.rs:502-571
template <class F>
void prove_cubic_with_three_inputs(const F& claim, const std::vector<F>& taus, MultilinearPolynomial<F>& A, MultilinearPolynomial<F>& B,
                                   MultilinearPolynomial<F>& C, Transcript& tr, SumcheckProof<F>* proof, std::vector<F>* r_out,
                                   std::vector<F>* final_evals) {
  F claim_per_round = claim;
  size_t num_rounds = taus.size();
  EqSumCheckInstance<F> eq(taus);
  proof->compressed_polys.clear();
  r_out->clear();
  for (size_t round = 0; round < num_rounds; ++round) {
    F e0, e2, e3;
    eq.evaluation_points_cubic_with_three_inputs(A, B, C, claim_per_round, &e0, &e2, &e3);
    UniPoly<F> poly = UniPoly<F>::from_evals({e0, claim_per_round - e0, e2, e3});
    std::vector<uint8_t> b = poly.to_transcript_bytes();
    tr.absorb_bytes("p", b.data(), b.size());
    F r_i = tr.squeeze<F>("c");
    r_out->push_back(r_i);
    proof->compressed_polys.push_back(poly.compress());
    claim_per_round = poly.evaluate(r_i);
    A.bind_poly_var_top(r_i);
    B.bind_poly_var_top(r_i);
    C.bind_poly_var_top(r_i);
    eq.bound(r_i);
  }
  *final_evals = {A.Z[0], B.Z[0], C.Z[0]};
}

}  // namespace oracle
