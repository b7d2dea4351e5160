// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Typed wire codecs on top of wire.hpp (see its header for the framing, the one third-party assumption and what is pinned):
// SpartanSNARK / NeutronNovaZkSNARK proofs to and from bincode bytes, the verifier keys as serde values.
#pragma once
#include "neutronnova_zk.hpp"
#include "spartan.hpp"
#include "wire.hpp"

namespace oracle {

// the key as a serde value (derive(Serialize) on SpartanVerifierKey, src/spartan.rs:62-70): vk_ee, ck_s, S — all through bincode
inline std::vector<uint8_t> spartan_vk_to_bytes(const SpartanProverKey& vk) {
  std::vector<uint8_t> out;
  WireWriter w(&out);
  w.hyrax_key(vk.ck);
  w.hyrax_key(vk.ck_s);
  w.shape_bincode(vk.S);
  return out;
}
// ---- SpartanSNARK (src/spartan.rs:125-137) ------------------------------------------------------------------------------------------------
inline std::vector<uint8_t> spartan_proof_to_bytes(const SpartanProof& pf) {
  std::vector<uint8_t> out;
  WireWriter w(&out);
  const size_t rs = pf.rows_shared, rp = pf.rows_precommitted;
  // U: SplitR1CSInstance { comm_W_shared: Option, comm_W_precommitted: Option, comm_W_rest, public_values, challenges }
  w.option_commitment(HyraxCommitment(pf.comm_W.begin(), pf.comm_W.begin() + rs));
  w.option_commitment(HyraxCommitment(pf.comm_W.begin() + rs, pf.comm_W.begin() + rs + rp));
  w.commitment(HyraxCommitment(pf.comm_W.begin() + rs + rp, pf.comm_W.end()));
  w.scalars(pf.public_values);
  w.scalars(pf.challenges);
  w.sumcheck(pf.sc_proof_outer);
  for (int i = 0; i < 3; ++i) w.fe(pf.claims_outer[i]);  // (Scalar, Scalar, Scalar)
  w.sumcheck(pf.sc_proof_inner);
  w.fe(pf.eval_W);
  w.scalars({pf.blind_eval_W});  // Blind<E> = HyraxBlind { blind: Vec<Scalar> } of one row
  w.ipa(pf.eval_arg);
  return out;
}
inline SpartanProof spartan_proof_from_bytes(const uint8_t* b, size_t n) {
  WireReader r(b, n);
  SpartanProof pf;
  HyraxCommitment sh = r.option_commitment(), pre = r.option_commitment(), rest = r.commitment();
  pf.rows_shared = sh.size();
  pf.rows_precommitted = pre.size();
  pf.comm_W = sh;
  pf.comm_W.insert(pf.comm_W.end(), pre.begin(), pre.end());
  pf.comm_W.insert(pf.comm_W.end(), rest.begin(), rest.end());
  pf.public_values = r.scalars();
  pf.challenges = r.scalars();
  pf.sc_proof_outer = r.sumcheck();
  for (int i = 0; i < 3; ++i) pf.claims_outer[i] = r.fe<Fq>();
  pf.sc_proof_inner = r.sumcheck();
  pf.eval_W = r.fe<Fq>();
  std::vector<Fq> bl = r.scalars();
  if (bl.size() != 1) throw std::runtime_error("wire: blind_eval_W must hold one row");
  pf.blind_eval_W = bl[0];
  pf.eval_arg = r.ipa();
  r.done();
  return pf;
}

// ---- NeutronNovaZkSNARK (src/neutronnova_zk.rs:1373-1385) ---------------------------------------------------------------------------------
inline std::vector<uint8_t> nn_proof_to_bytes(const NNProof& pf) {
  std::vector<uint8_t> out;
  WireWriter w(&out);
  w.option_commitment(pf.comm_W_shared);
  auto inst = [&](const NNSplitInstance& u) {  // SplitR1CSInstance with comm_W_shared = None (:2069-2078), no challenges
    w.u8(0);
    w.option_commitment(u.comm_pre);
    w.commitment(u.comm_rest);
    w.scalars(u.publics);
    w.scalars({});
  };
  w.u64(pf.step_instances.size());
  for (const auto& u : pf.step_instances) inst(u);
  inst(pf.core_instance);
  w.ipa(pf.eval_arg);
  // U_verifier: SplitMultiRoundR1CSInstance { comm_w_per_round: Vec<Commitment>, public_values, challenges_per_round: Vec<Vec<Scalar>> }
  w.u64(pf.U_verifier.comm_w_per_round.size());
  for (const auto& c : pf.U_verifier.comm_w_per_round) w.commitment(c);
  w.scalars(pf.U_verifier.public_values);
  w.u64(pf.U_verifier.challenges_per_round.size());
  for (const auto& c : pf.U_verifier.challenges_per_round) w.scalars(c);
  w.commitment(pf.nifs_comm_T);  // NovaNIFS { comm_T }
  // random_U: RelaxedR1CSInstance { comm_W, comm_E, X, u }
  w.commitment(pf.random_U.comm_W);
  w.commitment(pf.random_U.comm_E);
  w.scalars(pf.random_U.X);
  w.fe(pf.random_U.u);
  // RelaxedR1CSSpartanProof { sc_proof_outer, claims_outer, sc_proof_inner, v_W, blind_W, v_E, blind_E }
  w.sumcheck(pf.relaxed.sc_outer);
  for (int i = 0; i < 3; ++i) w.fe(pf.relaxed.claims_outer[i]);
  w.sumcheck(pf.relaxed.sc_inner);
  w.scalars(pf.relaxed.v_W);
  w.fe(pf.relaxed.blind_W);
  w.scalars(pf.relaxed.v_E);
  w.fe(pf.relaxed.blind_E);
  return out;
}
inline NNProof nn_proof_from_bytes(const uint8_t* b, size_t n) {
  WireReader r(b, n);
  NNProof pf;
  pf.comm_W_shared = r.option_commitment();
  auto inst = [&]() {
    NNSplitInstance u;
    if (r.u8() != 0) throw std::runtime_error("wire: a NeutronNova instance carries its own shared commitment");
    u.comm_pre = r.option_commitment();
    u.comm_rest = r.commitment();
    u.publics = r.scalars();
    if (!r.scalars().empty()) throw std::runtime_error("wire: NeutronNova instances have no challenges");
    return u;
  };
  size_t steps = r.len(1 + 1 + 8 + 8 + 8);
  for (size_t i = 0; i < steps; ++i) pf.step_instances.push_back(inst());
  pf.core_instance = inst();
  pf.eval_arg = r.ipa();
  size_t rounds = r.len(8);
  for (size_t i = 0; i < rounds; ++i) pf.U_verifier.comm_w_per_round.push_back(r.commitment());
  pf.U_verifier.public_values = r.scalars();
  size_t cr = r.len(8);
  for (size_t i = 0; i < cr; ++i) pf.U_verifier.challenges_per_round.push_back(r.scalars());
  pf.nifs_comm_T = r.commitment();
  pf.random_U.comm_W = r.commitment();
  pf.random_U.comm_E = r.commitment();
  pf.random_U.X = r.scalars();
  pf.random_U.u = r.fe<Fq>();
  pf.relaxed.sc_outer = r.sumcheck();
  for (int i = 0; i < 3; ++i) pf.relaxed.claims_outer[i] = r.fe<Fq>();
  pf.relaxed.sc_inner = r.sumcheck();
  pf.relaxed.v_W = r.scalars();
  pf.relaxed.blind_W = r.fe<Fq>();
  pf.relaxed.v_E = r.scalars();
  pf.relaxed.blind_E = r.fe<Fq>();
  r.done();
  return pf;
}

}  // namespace oracle
