// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Restates src/provider/pcs/hyrax_pc.rs (setup :152-177, blind :192-205, commit :207-303,
// commit_zeros :305-319, prove :387-478, bind_with_delayed :38-54, verify :480-531, transcript
// encoding :714-729) and src/provider/pcs/ipa.rs (prove :125-170, verify :173-221,
// instance encoding :58-70).
//
// Randomness: the reference draws blinds / IPA masks from rand::thread_rng() (hyrax_pc.rs:192-205,
// ipa.rs:139-145). Here every draw comes from an explicit tape of 64-byte uniform blocks, each
// reduced with from_uniform, consumed in the reference's call order.
#pragma once
#include <stdexcept>
#include <vector>

#include "msm.hpp"
#include "polys.hpp"

namespace oracle {

struct Tape {
  const uint8_t* bytes;
  size_t blocks, pos;
  Tape(const uint8_t* b, size_t n) : bytes(b), blocks(n), pos(0) {}
  Fq next() {
    if (pos >= blocks) throw std::runtime_error("random tape exhausted");
    return Fq::from_uniform(bytes + 64 * pos++);
  }
};

struct HyraxKey {  // HyraxCommitmentKey / HyraxVerifierKey (hyrax_pc.rs:56-108)
  size_t num_cols;
  std::vector<Affine> ck;
  Jac h;
  FixedBaseMul h_table;
  static HyraxKey setup(const char* label, size_t width) {  // hyrax_pc.rs:152-177
    HyraxKey k;
    k.num_cols = width;
    std::vector<Affine> gens = from_label(label, width + 1);
    k.ck.assign(gens.begin(), gens.begin() + width);
    k.h = Jac::from_affine(gens[width]);
    k.h_table = FixedBaseMul::precompute(k.h, 8);
    return k;
  }
};

typedef std::vector<Jac> HyraxCommitment;  // one group element per row
typedef std::vector<Fq> HyraxBlind;

inline size_t div_ceil(size_t a, size_t b) { return (a + b - 1) / b; }

inline HyraxBlind hyrax_blind(const HyraxKey& ck, size_t n, Tape& tape) {  // hyrax_pc.rs:192-205
  HyraxBlind b(div_ceil(n, ck.num_cols));
  for (auto& x : b) x = tape.next();
  return b;
}

// hyrax_pc.rs:207-303 (ck_tables path for widths <= 64 is a NeutronNova "next" row, not restated)
inline HyraxCommitment hyrax_commit(const HyraxKey& ck, const Fq* v, size_t n, const HyraxBlind& r, bool is_small) {
  size_t num_rows = div_ceil(n, ck.num_cols);
  HyraxCommitment comm(num_rows);
#pragma omp parallel for schedule(dynamic) if (num_rows >= 4)  // rows are independent (hyrax_pc.rs:230: par_iter over rows)
  for (size_t i = 0; i < num_rows; ++i) {
    size_t lower = i * ck.num_cols, upper = std::min(lower + ck.num_cols, n);
    const Fq* sc = v + lower;
    size_t len = upper - lower;
    bool all_zero = true;
    for (size_t k = 0; k < len; ++k) all_zero &= sc[k].is_zero();
    if (all_zero) {
      comm[i] = ck.h_table.mul(r[i]);
      continue;
    }
    size_t eff = len;
    while (eff > 0 && sc[eff - 1].is_zero()) --eff;
    Jac m;
    if (eff <= 16) {
      m = msm(sc, ck.ck.data(), eff);
    } else {
      bool all_small = is_small;
      std::vector<uint64_t> small(eff);
      if (!is_small) {
        all_small = true;
        for (size_t k = 0; k < eff && all_small; ++k) {
          uint64_t c[4];
          sc[k].to_canonical(c);
          if (c[1] | c[2] | c[3]) all_small = false;
        }
      }
      if (all_small) {
        for (size_t k = 0; k < eff; ++k) {
          uint64_t c[4];
          sc[k].to_canonical(c);
          small[k] = c[0];
        }
        m = msm_small(small.data(), ck.ck.data(), eff);
      } else {
        m = msm(sc, ck.ck.data(), eff);
      }
    }
    comm[i] = m.add(ck.h_table.mul(r[i]));
  }
  return comm;
}

// commit_without_blind (hyrax_pc.rs:533-568): the per-row MSMs alone, no blinding term; an all-zero row is the identity
inline std::vector<Jac> hyrax_commit_without_blind(const HyraxKey& ck, const Fq* v, size_t n, bool is_small) {
  const size_t num_cols = ck.ck.size(), num_rows = div_ceil(n, num_cols);
  std::vector<Jac> raw(num_rows);
#pragma omp parallel for schedule(dynamic) if (num_rows >= 4)
  for (size_t i = 0; i < num_rows; ++i) {
    const size_t lower = i * num_cols, upper = std::min(lower + num_cols, n), len = upper - lower;
    const Fq* row = v + lower;
    bool all_zero = true;
    for (size_t k = 0; k < len; ++k) all_zero &= row[k].is_zero();
    if (all_zero) {
      raw[i] = Jac::identity();
    } else if (is_small) {  // the caller's hint is trusted: the low 8 bytes of to_repr (:553-560)
      std::vector<uint64_t> small(len);
      for (size_t k = 0; k < len; ++k) {
        uint64_t c[4];
        row[k].to_canonical(c);
        small[k] = c[0];
      }
      raw[i] = msm_small(small.data(), ck.ck.data(), len);
    } else {
      raw[i] = msm(row, ck.ck.data(), len);
    }
  }
  return raw;
}
// commit_incremental (hyrax_pc.rs:570-607): cached raw row MSMs + MSM of the changed entries + h * blind
inline HyraxCommitment hyrax_commit_incremental(const HyraxKey& ck, const std::vector<Jac>& raw, const Fq* delta, size_t n, const HyraxBlind& blind) {
  const size_t num_cols = ck.ck.size(), num_rows = div_ceil(n, num_cols);
  if (blind.size() < num_rows) throw std::runtime_error("commit_incremental: too few blinds");
  HyraxCommitment comm(num_rows);
#pragma omp parallel for schedule(dynamic) if (num_rows >= 4)
  for (size_t i = 0; i < num_rows; ++i) {
    const size_t lower = i * num_cols, upper = std::min(lower + num_cols, n), len = upper - lower;
    const Fq* row = delta + lower;
    bool all_zero = true;
    for (size_t k = 0; k < len; ++k) all_zero &= row[k].is_zero();
    Jac point = i < raw.size() ? raw[i] : Jac::identity();
    if (!all_zero) point = point.add(msm(row, ck.ck.data(), len));
    comm[i] = point.add(ck.h_table.mul(blind[i]));
  }
  return comm;
}

inline HyraxCommitment hyrax_commit_zeros(const HyraxKey& ck, size_t n, const HyraxBlind& r) {  // hyrax_pc.rs:305-319
  size_t num_rows = div_ceil(n, ck.num_cols);
  HyraxCommitment comm(num_rows);
  for (size_t i = 0; i < num_rows; ++i) comm[i] = ck.h_table.mul(r[i]);
  return comm;
}

inline std::vector<uint8_t> commitment_transcript_bytes(const HyraxCommitment& c) {  // hyrax_pc.rs:714-729
  std::vector<uint8_t> v;
  const char* b = "poly_commitment_begin";
  const char* e = "poly_commitment_end";
  v.insert(v.end(), b, b + strlen(b));
  std::vector<Affine> aff = batch_affine(c);
  for (const Affine& a : aff) {
    uint8_t buf[64];
    point_to_transcript_bytes(a, buf);
    v.insert(v.end(), buf, buf + 64);
  }
  v.insert(v.end(), e, e + strlen(e));
  return v;
}

struct IpaProof {  // ipa.rs:105-115
  Jac delta, beta;
  std::vector<Fq> z_vec;
  Fq z_delta, z_beta;
};

inline void absorb_point(Transcript& tr, const char* label, const Jac& p) {
  uint8_t buf[64];
  point_to_transcript_bytes(p.to_affine(), buf);
  tr.absorb_bytes(label, buf, 64);
}

static const char* IPA_PROTOCOL_NAME = "inner product argument (linear)";

// ipa.rs:125-170
inline IpaProof ipa_prove(const std::vector<Affine>& ck, const Jac& h, const Affine& ck_c, const Jac& h_c, const Jac& comm_a, const std::vector<Fq>& b_vec,
                          const Jac& comm_c, const std::vector<Fq>& a_vec, const Fq& r_a, const Fq& r_c, Transcript& tr, Tape& tape) {
  tr.dom_sep(IPA_PROTOCOL_NAME);
  {
    uint8_t buf[128];
    point_to_transcript_bytes(comm_a.to_affine(), buf);
    point_to_transcript_bytes(comm_c.to_affine(), buf + 64);
    tr.absorb_bytes("U", buf, 128);
  }
  size_t n = b_vec.size();
  std::vector<Fq> d(n);
  for (auto& x : d) x = tape.next();
  Fq r_delta = tape.next(), r_beta = tape.next();
  IpaProof pf;
  pf.delta = msm(d.data(), ck.data(), n).add(scalar_mul(h, r_delta));
  Fq ip = Fq::zero();
  for (size_t i = 0; i < n; ++i) ip = ip + b_vec[i] * d[i];
  pf.beta = scalar_mul(Jac::from_affine(ck_c), ip).add(scalar_mul(h_c, r_beta));
  absorb_point(tr, "delta", pf.delta);
  absorb_point(tr, "beta", pf.beta);
  Fq r = tr.squeeze<Fq>("r");
  pf.z_vec.resize(n);
  for (size_t i = 0; i < n; ++i) pf.z_vec[i] = r * a_vec[i] + d[i];
  pf.z_delta = r * r_a + r_delta;
  pf.z_beta = r * r_c + r_beta;
  return pf;
}

inline bool jac_eq(const Jac& a, const Jac& b) { return a.to_affine() == b.to_affine(); }

// ipa.rs:173-221
inline bool ipa_verify(const IpaProof& pf, const std::vector<Affine>& ck, const Jac& h, const Affine& ck_c, const Jac& h_c, size_t n, const Jac& comm_a,
                       const std::vector<Fq>& b_vec, const Jac& comm_c, Transcript& tr) {
  tr.dom_sep(IPA_PROTOCOL_NAME);
  {
    uint8_t buf[128];
    point_to_transcript_bytes(comm_a.to_affine(), buf);
    point_to_transcript_bytes(comm_c.to_affine(), buf + 64);
    tr.absorb_bytes("U", buf, 128);
  }
  absorb_point(tr, "delta", pf.delta);
  absorb_point(tr, "beta", pf.beta);
  Fq r = tr.squeeze<Fq>("r");
  if (pf.z_vec.size() != n || ck.size() < n) return false;
  Jac lhs1 = scalar_mul(comm_a, r).add(pf.delta);
  Jac rhs1 = msm(pf.z_vec.data(), ck.data(), n).add(scalar_mul(h, pf.z_delta));
  if (!jac_eq(lhs1, rhs1)) return false;
  Fq ip = Fq::zero();
  for (size_t i = 0; i < n; ++i) ip = ip + pf.z_vec[i] * b_vec[i];
  Jac lhs2 = scalar_mul(comm_c, r).add(pf.beta);
  Jac rhs2 = scalar_mul(Jac::from_affine(ck_c), ip).add(scalar_mul(h_c, pf.z_beta));
  return jac_eq(lhs2, rhs2);
}

// hyrax_pc.rs:38-54
inline std::vector<Fq> bind_with_delayed(const Fq* poly, const std::vector<Fq>& l, size_t r_len) {
  std::vector<Fq> acc(r_len, Fq::zero());
#pragma omp parallel for schedule(static) if (r_len * l.size() >= 65536)
  for (size_t i = 0; i < r_len; ++i) {
    Fq a = Fq::zero();
    for (size_t j = 0; j < l.size(); ++j) a = a + l[j] * poly[j * r_len + i];
    acc[i] = a;
  }
  return acc;
}

inline size_t log2_exact(size_t n) {
  size_t l = 0;
  while (((size_t)1 << l) < n) ++l;
  return l;
}

// hyrax_pc.rs:387-478
inline IpaProof hyrax_prove(const HyraxKey& ck, const HyraxKey& ck_eval, Transcript& tr, const HyraxCommitment& comm, const std::vector<Fq>& poly,
                            const HyraxBlind& blind, const std::vector<Fq>& point, const HyraxCommitment& comm_eval, const HyraxBlind& blind_eval,
                            Tape& tape) {
  size_t n = poly.size();
  if (n != ((size_t)1 << point.size())) throw std::runtime_error("Hyrax prove: InvalidInputLength");
  std::vector<uint8_t> cb = commitment_transcript_bytes(comm);
  tr.absorb_bytes("poly_com", cb.data(), cb.size());
  size_t num_cols = ck.num_cols, num_rows = div_ceil(n, num_cols);
  size_t nvr = log2_exact(num_rows);
  Jac comm_LZ;
  std::vector<Fq> R, LZ;
  Fq r_LZ;
  if (nvr == 0) {
    comm_LZ = comm[0];
    R = eq_evals_from_points(point);
    LZ = poly;
    r_LZ = blind[0];
  } else {
    std::vector<Fq> L = eq_evals_from_points(std::vector<Fq>(point.begin(), point.begin() + nvr));
    R = eq_evals_from_points(std::vector<Fq>(point.begin() + nvr, point.end()));
    LZ = bind_with_delayed(poly.data(), L, R.size());
    r_LZ = Fq::zero();
    for (size_t i = 0; i < L.size(); ++i) r_LZ = r_LZ + L[i] * blind[i];
    comm_LZ = msm(LZ.data(), ck.ck.data(), LZ.size()).add(ck.h_table.mul(r_LZ));
  }
  return ipa_prove(ck.ck, ck.h, ck_eval.ck[0], ck_eval.h, comm_LZ, R, comm_eval[0], LZ, r_LZ, blind_eval[0], tr, tape);
}

// hyrax_pc.rs:480-531
inline bool hyrax_verify(const HyraxKey& vk, const HyraxKey& ck_eval, Transcript& tr, const HyraxCommitment& comm, const std::vector<Fq>& point,
                         const HyraxCommitment& comm_eval, const IpaProof& arg) {
  std::vector<uint8_t> cb = commitment_transcript_bytes(comm);
  tr.absorb_bytes("poly_com", cb.data(), cb.size());
  size_t n = (size_t)1 << point.size();
  size_t num_cols = vk.num_cols, num_rows = div_ceil(n, num_cols);
  size_t nvr = log2_exact(num_rows);
  Jac comm_LZ;
  std::vector<Fq> R;
  if (nvr == 0) {
    R = eq_evals_from_points(point);
    comm_LZ = comm[0];
  } else {
    std::vector<Fq> L = eq_evals_from_points(std::vector<Fq>(point.begin(), point.begin() + nvr));
    R = eq_evals_from_points(std::vector<Fq>(point.begin() + nvr, point.end()));
    if (comm.size() < L.size()) return false;
    std::vector<Affine> bases = batch_affine(comm);
    // bases may contain the identity (not the case for blinded commitments); msm handles it by value
    comm_LZ = msm(L.data(), bases.data(), L.size());
  }
  return ipa_verify(arg, vk.ck, vk.h, ck_eval.ck[0], ck_eval.h, R.size(), comm_LZ, R, comm_eval[0], tr);
}

}  // namespace oracle
