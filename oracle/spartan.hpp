// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Restates the protocol driver src/spartan.rs: setup (:146-173), prep_prove (:176-216),
// prove (:219-466), verify (:469-578), with the witness staging of src/bellpepper/r1cs.rs
// (shared_witness :306-357, precommitted_witness :359-409, r1cs_instance_and_witness :411-538): any mix of shared / precommitted / rest
// variables (the bench circuits are precommitted-only and take the `skip_synthesize` + commit_zeros path :443,:468; the reference's own e2e test
// circuit, src/spartan.rs:587-651, is rest-only), and circuits with verifier challenges (:429-431, :443-461): the challenges are squeezed after
// the precommitted commitment and the rest of the witness is re-synthesized from them — `circuit.synthesize` is the caller's callback here.
//
// The verifier-key digest is the reference's: SHA-256 over bincode(vk_ee) || bincode(ck_s) || S.write_bytes() (src/spartan.rs:73-104,
// src/digest.rs:49-77), absorbed as 32 raw bytes under the label "vk". The byte layout of the third-party point / field types inside it is
// the one documented assumption of wire.hpp (halo2curves derive_serde).
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <stdexcept>
#include <vector>

#include "hyrax.hpp"
#include "sparse.hpp"
#include "sumcheck.hpp"
#include "wire.hpp"

namespace oracle {

struct SpartanProverKey {  // src/spartan.rs:30-58 (pk and vk share everything the oracle needs)
  HyraxKey ck, ck_s;
  SplitR1CSShape<Fq> S;
  uint8_t vk_digest[32];
};

// SpartanVerifierKey::write_bytes (src/spartan.rs:73-90) hashed by DigestComputer::digest (src/digest.rs:62-76)
inline void spartan_vk_digest(const SpartanProverKey& vk, uint8_t out[32]) {
  Sha256 h;
  WireWriter w(&h);
  w.hyrax_key(vk.ck);    // vk_ee: the HyraxVerifierKey of the witness key (same num_cols, ck, h)
  w.hyrax_key(vk.ck_s);  // ck_s: HyraxCommitmentKey of width 1
  w.shape_digest_bytes(vk.S);
  h.finalize(out);
}

inline SpartanProverKey spartan_setup(SplitR1CSShape<Fq> S) {  // src/spartan.rs:146-173
  SpartanProverKey pk;
  pk.S = std::move(S);
  pk.ck = HyraxKey::setup("ck", DEFAULT_COMMITMENT_WIDTH);  // src/r1cs/mod.rs:1031-1043
  pk.ck_s = HyraxKey::setup("ck_s", 1);
  spartan_vk_digest(pk, pk.vk_digest);
  return pk;
}

struct SpartanPrep {  // SpartanPrepSNARK, src/spartan.rs:107-124 (+ PrecommittedState, bellpepper/r1cs.rs:290-301)
  std::vector<Fq> W;  // padded witness (shared | precommitted | rest); the rest values are known up front (no challenges)
  HyraxCommitment comm_W_shared, comm_W_precommitted;  // empty when the segment is empty
  HyraxBlind r_W_shared, r_W_precommitted;
  std::vector<Fq> cached_az, cached_bz, cached_cz;
  bool is_small = true;
};

// src/spartan.rs:176-216 + bellpepper/r1cs.rs:306-409 (shared_witness, precommitted_witness).
// `witness` = unpadded aux assignment, shared | precommitted | rest. Circuits with verifier challenges are not restated.
inline SpartanPrep spartan_prep_prove(const SpartanProverKey& pk, const std::vector<Fq>& witness, bool is_small, Tape& tape) {
  const SplitR1CSShape<Fq>& S = pk.S;
  if (witness.size() != S.num_shared_unpadded + S.num_precommitted_unpadded + S.num_rest_unpadded) throw std::runtime_error("InvalidWitnessLength");
  SpartanPrep ps;
  ps.is_small = is_small;
  ps.W.assign(S.num_vars(), Fq::zero());
  std::copy(witness.begin(), witness.begin() + S.num_shared_unpadded, ps.W.begin());
  std::copy(witness.begin() + S.num_shared_unpadded, witness.begin() + S.num_shared_unpadded + S.num_precommitted_unpadded, ps.W.begin() + S.num_shared);
  std::copy(witness.begin() + S.num_shared_unpadded + S.num_precommitted_unpadded, witness.end(), ps.W.begin() + S.num_shared + S.num_precommitted);
  if (S.num_shared_unpadded > 0) {  // r1cs.rs:337-344
    ps.r_W_shared = hyrax_blind(pk.ck, S.num_shared, tape);
    ps.comm_W_shared = hyrax_commit(pk.ck, ps.W.data(), S.num_shared, ps.r_W_shared, is_small);
  }
  if (S.num_precommitted_unpadded > 0) {  // r1cs.rs:388-399
    ps.r_W_precommitted = hyrax_blind(pk.ck, S.num_precommitted, tape);
    ps.comm_W_precommitted = hyrax_commit(pk.ck, ps.W.data() + S.num_shared, S.num_precommitted, ps.r_W_precommitted, is_small);
  }
  std::vector<Fq> zc(ps.W.begin(), ps.W.begin() + S.num_shared + S.num_precommitted);
  S.multiply_vec_precommitted(zc, &ps.cached_az, &ps.cached_bz, &ps.cached_cz);
  return ps;
}

struct SpartanProof {  // SpartanSNARK, src/spartan.rs:130-138
  HyraxCommitment comm_W;  // shared rows, precommitted rows, rest rows (to_regular_instance, src/r1cs/mod.rs:1535-1550)
  size_t rows_shared = 0, rows_precommitted = 0;
  std::vector<Fq> public_values, challenges;  // SplitR1CSInstance (src/r1cs/mod.rs:1423-1437)
  SumcheckProof<Fq> sc_proof_outer, sc_proof_inner;
  Fq claims_outer[3];
  Fq eval_W, blind_eval_W;
  IpaProof eval_arg;

  // Canonical flat layout shared with the HIP library's proof buffer (DESIGN.md "proof layout"):
  // affine points as (x,y) Montgomery limbs, scalars as Montgomery limbs.
  std::vector<uint64_t> serialize() const {
    std::vector<uint64_t> out;
    auto pf = [&](const Fq& f) { out.insert(out.end(), f.l, f.l + 4); };
    auto pp = [&](const Affine& a) {
      out.insert(out.end(), a.x.l, a.x.l + 4);
      out.insert(out.end(), a.y.l, a.y.l + 4);
    };
    for (const Affine& a : batch_affine(comm_W)) pp(a);
    for (const Fq& f : public_values) pf(f);
    for (const Fq& f : challenges) pf(f);
    for (const auto& p : sc_proof_outer.compressed_polys)
      for (const Fq& f : p) pf(f);
    for (int i = 0; i < 3; ++i) pf(claims_outer[i]);
    for (const auto& p : sc_proof_inner.compressed_polys)
      for (const Fq& f : p) pf(f);
    pf(eval_W);
    pf(blind_eval_W);
    pp(eval_arg.delta.to_affine());
    pp(eval_arg.beta.to_affine());
    for (const Fq& f : eval_arg.z_vec) pf(f);
    pf(eval_arg.z_delta);
    pf(eval_arg.z_beta);
    return out;
  }
};

// circuit.synthesize(.., Some(&challenges)) of a circuit with verifier challenges (bellpepper/r1cs.rs:443-461): challenges -> the unpadded rest
// segment of the witness (num_rest_unpadded values)
typedef std::function<std::vector<Fq>(const std::vector<Fq>&)> RestSynth;

// src/spartan.rs:219-466
inline SpartanProof spartan_prove(const SpartanProverKey& pk, const SpartanPrep& ps_in, const std::vector<Fq>& public_values, Tape& tape,
                                  const RestSynth& synth = RestSynth()) {
  const SplitR1CSShape<Fq>& S = pk.S;
  SpartanPrep ps_local;
  const SpartanPrep* psp = &ps_in;
  static const bool trace = getenv("ORACLE_TRACE") != nullptr;  // phase timeline on stderr
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "oracle lap %-22s %9.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
    t_prev = t;
  };
  Transcript tr("SpartanSNARK");
  tr.absorb_bytes("vk", pk.vk_digest, 32);
  tr.absorb_scalars("public_values", public_values.data(), public_values.size());
  // r1cs_instance_and_witness (bellpepper/r1cs.rs:411-538)
  if (!ps_in.comm_W_shared.empty()) {
    std::vector<uint8_t> b = commitment_transcript_bytes(ps_in.comm_W_shared);
    tr.absorb_bytes("comm_W_shared", b.data(), b.size());
  }
  if (!ps_in.comm_W_precommitted.empty()) {
    std::vector<uint8_t> b = commitment_transcript_bytes(ps_in.comm_W_precommitted);
    tr.absorb_bytes("comm_W_precommitted", b.data(), b.size());
  }
  // challenges (r1cs.rs:429-431) and the re-synthesized rest segment (:443-461)
  std::vector<Fq> challenges(S.num_challenges);
  for (auto& c : challenges) c = tr.squeeze<Fq>("challenge");
  if (S.num_challenges != 0) {
    if (!synth) throw std::runtime_error("a circuit with verifier challenges needs its synthesize callback");
    std::vector<Fq> rest = synth(challenges);
    if (rest.size() != S.num_rest_unpadded) throw std::runtime_error("synthesize returned the wrong number of rest variables");
    ps_local = ps_in;
    std::copy(rest.begin(), rest.end(), ps_local.W.begin() + S.num_shared + S.num_precommitted);
    psp = &ps_local;
  }
  const SpartanPrep& ps = *psp;
  HyraxBlind r_W_rest = hyrax_blind(pk.ck, S.num_rest, tape);
  HyraxCommitment comm_W_rest;
  if (S.num_rest_unpadded == 0) {
    comm_W_rest = hyrax_commit_zeros(pk.ck, S.num_rest, r_W_rest);  // r1cs.rs:468-470
  } else {  // r1cs.rs:471-489: is_small hint, or detection over the non-padding part (hyrax_commit detects per row: same value)
    comm_W_rest = hyrax_commit(pk.ck, ps.W.data() + S.num_shared + S.num_precommitted, S.num_rest, r_W_rest, ps.is_small);
  }
  {
    std::vector<uint8_t> b = commitment_transcript_bytes(comm_W_rest);
    tr.absorb_bytes("comm_W_rest", b.data(), b.size());
  }
  HyraxBlind r_W = ps.r_W_shared;  // combine_blinds (r1cs.rs:515-524)
  r_W.insert(r_W.end(), ps.r_W_precommitted.begin(), ps.r_W_precommitted.end());
  r_W.insert(r_W.end(), r_W_rest.begin(), r_W_rest.end());
  HyraxCommitment comm_W = ps.comm_W_shared;
  comm_W.insert(comm_W.end(), ps.comm_W_precommitted.begin(), ps.comm_W_precommitted.end());
  comm_W.insert(comm_W.end(), comm_W_rest.begin(), comm_W_rest.end());

  size_t num_vars = S.num_vars();
  std::vector<Fq> z = ps.W;
  z.push_back(Fq::one());
  z.insert(z.end(), public_values.begin(), public_values.end());
  z.insert(z.end(), challenges.begin(), challenges.end());
  size_t num_rounds_x = log2_exact(S.num_cons), num_rounds_y = log2_exact(num_vars) + 1;

  std::vector<Fq> tau(num_rounds_x);
  for (auto& t : tau) t = tr.squeeze<Fq>("t");

  lap("witness_commit");
  std::vector<Fq> az, bz, cz;
  S.multiply_vec_incremental_into(z, ps.cached_az, ps.cached_bz, ps.cached_cz, &az, &bz, &cz);
  MultilinearPolynomial<Fq> pAz(az), pBz(bz), pCz(cz);
  lap("matvec");

  SpartanProof proof;
  proof.comm_W = comm_W;
  proof.rows_shared = ps.comm_W_shared.size();
  proof.rows_precommitted = ps.comm_W_precommitted.size();
  proof.public_values = public_values;
  proof.challenges = challenges;
  std::vector<Fq> r_x, claims_outer;
  prove_cubic_with_three_inputs(Fq::zero(), tau, pAz, pBz, pCz, tr, &proof.sc_proof_outer, &r_x, &claims_outer);
  for (int i = 0; i < 3; ++i) proof.claims_outer[i] = claims_outer[i];
  tr.absorb_scalars("claims_outer", claims_outer.data(), 3);
  lap("outer_sumcheck");

  Fq r = tr.squeeze<Fq>("r");
  Fq claim_inner_joint = claims_outer[0] + r * claims_outer[1] + r * r * claims_outer[2];
  std::vector<Fq> evals_rx = eq_evals_from_points(r_x);
  std::vector<Fq> poly_ABC = S.bind_and_prepare_poly_ABC(evals_rx, r);
  lap("eq + poly_ABC");

  // manual inner round 0 (src/spartan.rs:323-384)
  size_t num_extra = S.num_extra();
  Fq eval0 = Fq::zero();
  for (size_t j = 0; j < num_vars; ++j) eval0 = eval0 + poly_ABC[j] * z[j];
  Fq correction_low = Fq::zero(), correction_cross = Fq::zero();
  for (size_t j = 0; j < num_extra; ++j) {
    correction_low = correction_low + poly_ABC[j] * z[j];
    correction_cross = correction_cross + (poly_ABC[num_vars + j] - poly_ABC[j]) * (z[num_vars + j] - z[j]);
  }
  Fq t_inf = eval0 - correction_low + correction_cross;
  Fq eval2 = claim_inner_joint + claim_inner_joint - (eval0 + eval0 + eval0) + t_inf + t_inf;
  UniPoly<Fq> inner_r0 = UniPoly<Fq>::from_evals({eval0, claim_inner_joint - eval0, eval2});
  {
    std::vector<uint8_t> b = inner_r0.to_transcript_bytes();
    tr.absorb_bytes("p", b.data(), b.size());
  }
  Fq r0 = tr.squeeze<Fq>("c");
  Fq claim_after_r0 = inner_r0.evaluate(r0);
  Fq one_minus_r0 = Fq::one() - r0;
  for (size_t j = 0; j < num_extra; ++j) {
    poly_ABC[j] = poly_ABC[j] + r0 * (poly_ABC[num_vars + j] - poly_ABC[j]);
    z[j] = z[j] + r0 * (z[num_vars + j] - z[j]);
  }
  for (size_t j = num_extra; j < num_vars; ++j) {
    poly_ABC[j] = poly_ABC[j] * one_minus_r0;
    z[j] = z[j] * one_minus_r0;
  }
  poly_ABC.resize(num_vars);
  z.resize(num_vars);
  MultilinearPolynomial<Fq> pABC(poly_ABC), pz(z);
  std::vector<Fq> r_y_rest, claims_inner;
  prove_quad(claim_after_r0, num_rounds_y - 1, pABC, pz, tr, &proof.sc_proof_inner, &r_y_rest, &claims_inner);
  proof.sc_proof_inner.compressed_polys.insert(proof.sc_proof_inner.compressed_polys.begin(), inner_r0.compress());
  std::vector<Fq> r_y;
  r_y.push_back(r0);
  r_y.insert(r_y.end(), r_y_rest.begin(), r_y_rest.end());
  Fq eval_Z = claims_inner[1];

  std::vector<Fq> X;
  X.push_back(Fq::one());
  X.insert(X.end(), public_values.begin(), public_values.end());
  X.insert(X.end(), challenges.begin(), challenges.end());  // to_regular_instance: X = public_values ++ challenges (src/r1cs/mod.rs:1546-1549)
  std::vector<Fq> r_y_tail(r_y.begin() + 1, r_y.end());
  Fq eval_X = sparse_poly_evaluate(num_rounds_y - 1, X, r_y_tail);
  Fq denom = Fq::one() - r_y[0];
  if (denom.is_zero()) throw std::runtime_error("DivisionByZero");
  proof.eval_W = (eval_Z - r_y[0] * eval_X) * denom.inv();

  lap("inner_sumcheck");
  HyraxBlind blind_eval_W = hyrax_blind(pk.ck_s, 1, tape);
  proof.blind_eval_W = blind_eval_W[0];
  HyraxCommitment comm_eval_W = hyrax_commit(pk.ck_s, &proof.eval_W, 1, blind_eval_W, false);
  proof.eval_arg = hyrax_prove(pk.ck, pk.ck_s, tr, comm_W, ps.W, r_W, r_y_tail, comm_eval_W, blind_eval_W, tape);
  lap("pcs_prove");
  return proof;
}

// src/spartan.rs:469-578. Returns 0 on accept, else a small code naming the failed check.
inline int spartan_verify(const SpartanProverKey& vk, const SpartanProof& pf) {
  const SplitR1CSShape<Fq>& S = vk.S;
  Transcript tr("SpartanSNARK");
  tr.absorb_bytes("vk", vk.vk_digest, 32);
  tr.absorb_scalars("public_values", pf.public_values.data(), pf.public_values.size());
  // SplitR1CSInstance::validate (src/r1cs/mod.rs:1490-1533): commitment lengths + transcript absorption
  size_t rows_sh = div_ceil(S.num_shared, vk.ck.num_cols), rows_pre = div_ceil(S.num_precommitted, vk.ck.num_cols), rows_rest = div_ceil(S.num_rest, vk.ck.num_cols);
  if (pf.rows_shared != rows_sh || pf.rows_precommitted != rows_pre || pf.comm_W.size() != rows_sh + rows_pre + rows_rest) return 1;
  if (pf.public_values.size() != S.num_public) return 1;
  {
    HyraxCommitment sh(pf.comm_W.begin(), pf.comm_W.begin() + rows_sh), pre(pf.comm_W.begin() + rows_sh, pf.comm_W.begin() + rows_sh + rows_pre),
        rest(pf.comm_W.begin() + rows_sh + rows_pre, pf.comm_W.end());
    std::vector<uint8_t> b;
    if (S.num_shared > 0) {
      b = commitment_transcript_bytes(sh);
      tr.absorb_bytes("comm_W_shared", b.data(), b.size());
    }
    if (S.num_precommitted > 0) {
      b = commitment_transcript_bytes(pre);
      tr.absorb_bytes("comm_W_precommitted", b.data(), b.size());
    }
    // challenges are re-derived and must match the instance's (validate, src/r1cs/mod.rs:1516-1526)
    if (pf.challenges.size() != S.num_challenges) return 1;
    for (size_t i = 0; i < S.num_challenges; ++i)
      if (tr.squeeze<Fq>("challenge") != pf.challenges[i]) return 1;
    b = commitment_transcript_bytes(rest);
    tr.absorb_bytes("comm_W_rest", b.data(), b.size());
  }
  size_t num_vars = S.num_vars();
  size_t num_rounds_x = log2_exact(S.num_cons), num_rounds_y = log2_exact(num_vars) + 1;
  std::vector<Fq> tau(num_rounds_x);
  for (auto& t : tau) t = tr.squeeze<Fq>("t");
  Fq claim_outer_final;
  std::vector<Fq> r_x;
  if (!pf.sc_proof_outer.verify(Fq::zero(), num_rounds_x, 3, tr, &claim_outer_final, &r_x)) return 2;
  Fq taus_bound_rx = eq_evaluate(tau, r_x);
  if (claim_outer_final != taus_bound_rx * (pf.claims_outer[0] * pf.claims_outer[1] - pf.claims_outer[2])) return 3;
  tr.absorb_scalars("claims_outer", pf.claims_outer, 3);
  Fq r = tr.squeeze<Fq>("r");
  Fq claim_inner_joint = pf.claims_outer[0] + r * pf.claims_outer[1] + r * r * pf.claims_outer[2];
  Fq claim_inner_final;
  std::vector<Fq> r_y;
  if (!pf.sc_proof_inner.verify(claim_inner_joint, num_rounds_y, 2, tr, &claim_inner_final, &r_y)) return 4;
  std::vector<Fq> X;
  X.push_back(Fq::one());
  X.insert(X.end(), pf.public_values.begin(), pf.public_values.end());
  X.insert(X.end(), pf.challenges.begin(), pf.challenges.end());
  std::vector<Fq> r_y_tail(r_y.begin() + 1, r_y.end());
  Fq eval_X = sparse_poly_evaluate(log2_exact(num_vars), X, r_y_tail);
  Fq eval_Z = (Fq::one() - r_y[0]) * pf.eval_W + r_y[0] * eval_X;
  std::vector<Fq> T_x = eq_evals_from_points(r_x), T_y = eq_evals_from_points(r_y);
  Fq ea, eb, ec;
  S.evaluate_with_tables(T_x, T_y, &ea, &eb, &ec);
  if (claim_inner_final != (ea + r * eb + r * r * ec) * eval_Z) return 5;
  HyraxBlind be{pf.blind_eval_W};
  HyraxCommitment comm_eval_W = hyrax_commit(vk.ck_s, &pf.eval_W, 1, be, false);
  if (!hyrax_verify(vk.ck, vk.ck_s, tr, pf.comm_W, r_y_tail, comm_eval_W, pf.eval_arg)) return 6;
  return 0;
}

}  // namespace oracle
