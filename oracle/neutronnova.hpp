// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Kernel-level restatements of the NeutronNova rows of SURVEY.md 8(a) that sit on the data-parallel path:
//   weights_from_r, R1CSWitness::fold_multiple          src/r1cs/mod.rs:153-166, 570-660            (a22)
//   msm_shared_weights                                   src/provider/msm.rs:228-356                 (a16)
//   HyraxPCS::fold_commitments / fold_blinds             src/provider/pcs/hyrax_pc.rs:737-818        (a20)
//   compute_eval_points_cubic_with_additive_term[_with_outer_pow]   src/sumcheck.rs:262-498          (a6)
// The NIFS round logic and the ZK wrapper around them (src/neutronnova_zk.rs) are not restated yet.
#pragma once
#include <vector>

#include "hyrax.hpp"

namespace oracle {

inline std::vector<Fq> weights_from_r(const std::vector<Fq>& r_bs, size_t n) {  // src/r1cs/mod.rs:153-166
  std::vector<Fq> w(n);
  for (size_t i = 0; i < n; ++i) {
    Fq wi = Fq::one();
    size_t k = i;
    for (const Fq& r : r_bs) {
      wi = wi * ((k & 1) ? r : (Fq::one() - r));  // eq01
      k >>= 1;
    }
    w[i] = wi;
  }
  return w;
}

// acc[j] = sum_i w_i * W_i[j]  (both the small-value fast path :615-631 and the delayed-reduction path :632-645, by value)
inline std::vector<Fq> fold_witnesses(const std::vector<Fq>& w, const std::vector<const Fq*>& Ws, size_t dim) {
  std::vector<Fq> acc(dim, Fq::zero());
  for (size_t i = 0; i < Ws.size(); ++i)
    for (size_t j = 0; j < dim; ++j) acc[j] = acc[j] + w[i] * Ws[i][j];
  return acc;
}

// fold_blinds (hyrax_pc.rs:795-818)
inline HyraxBlind fold_blinds(const std::vector<HyraxBlind>& blinds, const std::vector<Fq>& w) {
  HyraxBlind acc(blinds[0].size(), Fq::zero());
  for (size_t k = 0; k < blinds.size(); ++k)
    for (size_t i = 0; i < acc.size(); ++i) acc[i] = acc[i] + blinds[k][i] * w[k];
  return acc;
}

// msm_shared_weights (msm.rs:228-356): one scalar vector, many base rows; result per row (by value: sum_k w_k * bases[row][k])
inline std::vector<Jac> msm_shared_weights(const std::vector<Fq>& weights, const std::vector<std::vector<Affine>>& bases_rows) {
  std::vector<Jac> out;
  for (const auto& row : bases_rows) out.push_back(msm(weights.data(), row.data(), weights.size()));
  return out;
}

// fold_commitments (hyrax_pc.rs:737-793): row-wise weighted sum of the instances' commitments
inline HyraxCommitment fold_commitments(const std::vector<HyraxCommitment>& comms, const std::vector<Fq>& weights) {
  size_t n = comms[0].size();
  std::vector<std::vector<Affine>> rows(n);
  for (size_t row = 0; row < n; ++row) {
    std::vector<Jac> pts;
    for (const auto& c : comms) pts.push_back(c[row]);
    rows[row] = batch_affine(pts);
  }
  return msm_shared_weights(weights, rows);
}

// compute_eval_points_cubic_with_additive_term_with_outer_pow (src/sumcheck.rs:366-498), incl. the len < left fallback (:262-342)
inline void eval_points_cubic_outer_pow(const std::vector<Fq>& pow_left, const std::vector<Fq>& pow_right, const std::vector<Fq>& A,
                                        const std::vector<Fq>& B, const std::vector<Fq>& C, Fq* e0, Fq* e2, Fq* e3) {
  size_t len = A.size() / 2, left = pow_left.size();
  Fq a0 = Fq::zero(), a2 = Fq::zero(), a3 = Fq::zero();
  auto term = [&](const Fq& tl, const Fq& th, size_t low, Fq* t0, Fq* t2, Fq* t3) {
    size_t high = low + len;
    *t0 = tl * (A[low] * B[low] - C[low]);
    Fq tb = th + th - tl, ab = A[high] + A[high] - A[low], bb = B[high] + B[high] - B[low], cb = C[high] + C[high] - C[low];
    *t2 = tb * (ab * bb - cb);
    tb = tb + th - tl;
    ab = ab + A[high] - A[low];
    bb = bb + B[high] - B[low];
    cb = cb + C[high] - C[low];
    *t3 = tb * (ab * bb - cb);
  };
  if (len < left) {  // compute_eval_points_cubic_with_additive_term: the pow table itself is the fourth bound table
    for (size_t i = 0; i < len; ++i) {
      Fq t0, t2, t3;
      term(pow_left[i], pow_left[i + len], i, &t0, &t2, &t3);
      a0 = a0 + t0;
      a2 = a2 + t2;
      a3 = a3 + t3;
    }
  } else {
    size_t right = len / left;
    for (size_t i = 0; i < left; ++i) {
      Fq i0 = Fq::zero(), i2 = Fq::zero(), i3 = Fq::zero();
      for (size_t j = 0; j < right; ++j) {
        Fq t0, t2, t3;
        term(pow_right[j], pow_right[j + right], i + j * left, &t0, &t2, &t3);
        i0 = i0 + t0;
        i2 = i2 + t2;
        i3 = i3 + t3;
      }
      a0 = a0 + pow_left[i] * i0;
      a2 = a2 + pow_left[i] * i2;
      a3 = a3 + pow_left[i] * i3;
    }
  }
  *e0 = a0;
  *e2 = a2;
  *e3 = a3;
}

}  // namespace oracle
