// Pieces shared by the host-side protocol drivers above the C ABI (spartan_snark.cpp, sharded_snark.cpp): the integer R1CS view and
// SplitR1CSShape::new padding, this build's generator derivation, the vk digest (the reference's SHA-256 stream), the O(sqrt N) host-side eq tables.
#pragma once
#include "host_common.hpp"

namespace spartan2 {

// ---- integer R1CS as produced by the frontend (arguments of SplitR1CSShape::new) ---------------------------------------
struct CsrIntView {
  const int64_t* data;
  const uint32_t* indices;
  const uint64_t* indptr;
};
struct R1CSIntView {
  size_t num_cons, num_shared, num_precommitted, num_rest, num_public, num_challenges;
  CsrIntView m[3];
};

struct PaddedShape {
  sp_dims dims;
  std::vector<fe_t> data[3];
  std::vector<uint32_t> idx[3];
  std::vector<uint64_t> ptr[3];
  size_t num_vars() const { return dims.num_shared + dims.num_precommitted + dims.num_rest; }
  size_t num_cols() const { return num_vars() + 1 + dims.num_public + dims.num_challenges; }
};

inline size_t pad_to_width(size_t w, size_t n) { return (n + w - 1) / w * w; }
inline size_t next_pow2(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
inline size_t log2_ceil(size_t n) {
  size_t l = 0;
  while (((size_t)1 << l) < n) ++l;
  return l;
}

// SplitR1CSShape::new (src/r1cs/mod.rs:810-911)
inline PaddedShape pad_shape(const R1CSIntView& R) {
  const size_t width = DEFAULT_COMMITMENT_WIDTH;
  size_t sp_ = pad_to_width(width, R.num_shared), pp = pad_to_width(width, R.num_precommitted), rp = pad_to_width(width, R.num_rest);
  size_t nvp = sp_ + pp + rp;
  if (nvp < R.num_public + R.num_challenges + 1) rp = std::max(R.num_public + R.num_challenges + 1, nvp) - (sp_ + pp);
  nvp = sp_ + pp + rp;
  if (next_pow2(nvp) != nvp) rp = next_pow2(nvp) - (sp_ + pp);
  nvp = sp_ + pp + rp;
  const size_t num_vars = R.num_shared + R.num_precommitted + R.num_rest;
  const size_t ncp = next_pow2(R.num_cons);
  PaddedShape P;
  P.dims.num_cons = ncp;
  P.dims.num_cons_unpadded = R.num_cons;
  P.dims.num_shared = sp_;
  P.dims.num_precommitted = pp;
  P.dims.num_rest = rp;
  P.dims.num_shared_unpadded = R.num_shared;
  P.dims.num_precommitted_unpadded = R.num_precommitted;
  P.dims.num_rest_unpadded = R.num_rest;
  P.dims.num_public = R.num_public;
  P.dims.num_challenges = R.num_challenges;
  // small coefficient cache: the SHA circuits only use +-2^k
  for (int m = 0; m < 3; ++m) {
    const size_t nnz = R.m[m].indptr[R.num_cons];
    P.data[m].resize(nnz);
    P.idx[m].resize(nnz);
    for (size_t k = 0; k < nnz; ++k) {
      P.data[m][k] = fe_from_i64<S>(R.m[m].data[k]);
      size_t c = R.m[m].indices[k];
      if (c >= R.num_shared && c < R.num_shared + R.num_precommitted) c += sp_ - R.num_shared;
      else if (c >= R.num_shared + R.num_precommitted && c < num_vars) c += sp_ + pp - R.num_shared - R.num_precommitted;
      else if (c >= num_vars) c += nvp - num_vars;
      P.idx[m][k] = (uint32_t)c;
    }
    P.ptr[m].assign(R.m[m].indptr, R.m[m].indptr + R.num_cons + 1);
    P.ptr[m].resize(ncp + 1, nnz);
  }
  return P;
}

// SplitR1CSShape::equalize (src/r1cs/mod.rs:913-971): the larger constraint count (row pointers extended with the last offset) and the larger variable
// count (growth into num_rest; the columns of 1 | public | challenges move up by it) for both shapes
inline void equalize(PaddedShape& A, PaddedShape& B) {
  const size_t cons = std::max<size_t>(A.dims.num_cons, B.dims.num_cons), vars = std::max(A.num_vars(), B.num_vars());
  auto grow = [&](PaddedShape& S) {
    const size_t orig_cons = S.dims.num_cons, nv = S.num_vars();
    S.dims.num_cons = cons;
    if (nv != vars) S.dims.num_rest = vars - (S.dims.num_shared + S.dims.num_precommitted);
    for (int m = 0; m < 3; ++m) {
      for (uint32_t& c : S.idx[m])
        if (c >= nv) c += (uint32_t)(vars - nv);
      const uint64_t nnz = S.ptr[m].empty() ? 0 : S.ptr[m].back();
      S.ptr[m].resize(S.ptr[m].size() + (cons - orig_cons), nnz);
    }
  };
  grow(A);
  grow(B);
}

// DigestHelperTrait::digest of SpartanVerifierKey (src/spartan.rs:73-104): SHA-256 over bincode(vk_ee) || bincode(ck_s) || S.write_bytes(), through the
// library's wire sink (include/spartan_hip.h "wire formats"). gens / gens_s = the num_cols + 1 generators of each key, h last.
inline void padded_csr(const PaddedShape& P, sp_csr cs[3]) {
  for (int m = 0; m < 3; ++m) cs[m] = sp_csr{u64p(P.data[m].data()), P.idx[m].data(), P.ptr[m].data()};
}
inline void spartan_vk_digest(const PaddedShape& P, const std::vector<aff_t>& gens, const std::vector<aff_t>& gens_s, uint8_t out[32]) {
  sp_csr cs[3];
  padded_csr(P, cs);
  ck(sp_vk_digest(&P.dims, &cs[0], &cs[1], &cs[2], u64p(&gens[0].x), gens.size() - 1, u64p(&gens.back().x), u64p(&gens_s[0].x), gens_s.size() - 1,
                  u64p(&gens_s.back().x), out),
     "vk digest");
}

// This build's generator derivation (documented in DESIGN.md; shape of src/provider/traits.rs:205-249):
// SHAKE256(label) stream, 32 bytes per generator -> x = bytes LE mod p; increment x until x^3 - 3x + b is a non-zero square;
// y = rhs^((p+1)/4), take the root with even canonical value.
inline std::vector<aff_t> from_label(const char* label, size_t n) {
  sp::Shake256State sh;
  sh.init();
  sh.update((const uint8_t*)label, strlen(label));
  uint32_t e[8], c = 0;
  for (int i = 0; i < 8; ++i) e[i] = sp_addc(FpP::P(i), i == 0 ? 1u : 0u, c);  // p + 1
  for (int i = 0; i < 7; ++i) e[i] = (e[i] >> 2) | (e[i + 1] << 30);
  e[7] >>= 2;
  std::vector<aff_t> out(n);
  const fe_t b = T256::b(), one = fe_one<B>();
  for (size_t i = 0; i < n; ++i) {
    uint8_t buf[64];
    memset(buf, 0, 64);
    sh.read(buf, 32);
    fe_t x = fe_from_uniform<B>(buf);
    for (;;) {
      fe_t rhs = fe_add<B>(fe_sub<B>(fe_mul<B>(fe_sqr<B>(x), x), fe_add<B>(fe_dbl<B>(x), x)), b);
      fe_t y = fe_pow<B>(rhs, e);
      if (!fe_is_zero(rhs) && fe_eq(fe_sqr<B>(y), rhs)) {
        fe_t yc = fe_to_canonical<B>(y);
        if (yc.v[0] & 1u) y = fe_neg<B>(y);
        out[i].x = x;
        out[i].y = y;
        break;
      }
      x = fe_add<B>(x, one);
    }
  }
  return out;
}

// EqPolynomial::evals_from_points on the host for the O(sqrt N) tables of the opening (src/polys/eq.rs:59-92)
inline std::vector<fe_t> eq_evals_host(const fe_t* r, size_t ell) {
  std::vector<fe_t> ev((size_t)1 << ell, fe_zero());
  ev[0] = fe_one<S>();
  size_t size = 1;
  for (size_t k = ell; k-- > 0;) {
    for (size_t i = 0; i < size; ++i) {
      fe_t y = fe_mul<S>(ev[i], r[k]);
      ev[size + i] = y;
      ev[i] = fe_sub<S>(ev[i], y);
    }
    size *= 2;
  }
  return ev;
}
// SparsePolynomial::evaluate (src/polys/multilinear.rs:190-207)
inline fe_t sparse_poly_evaluate(size_t num_vars, const std::vector<fe_t>& Z, const fe_t* r) {
  size_t nvz = log2_ceil(Z.size());
  // the reference asserts the same relation (SparsePolynomial::evaluate, multilinear.rs:196); an instance with log2(M) == log2_ceil(1 + num_public)
  // has no room for the (1, X) block beside W
  if (num_vars < nvz + 1) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "SparsePolynomial::evaluate: too many public values for this many variables");
  std::vector<fe_t> chis = eq_evals_host(r + (num_vars - 1 - nvz), nvz + 1);
  fe_t partial = fe_zero();
  for (size_t i = 0; i < Z.size(); ++i) partial = fe_add<S>(partial, fe_mul<S>(Z[i], chis[i]));
  fe_t common = fe_one<S>();
  for (size_t i = 0; i < num_vars - 1 - nvz; ++i) common = fe_mul<S>(common, fe_sub<S>(fe_one<S>(), r[i]));
  return fe_mul<S>(common, partial);
}

// ---- verifier-side pieces shared by SpartanSNARK::verify and NeutronNovaZkSNARK::verify ------------------------------------------------------
inline bool same_point(const jac_t& a, const jac_t& b) {
  const aff_t x = jac_to_affine(a), y = jac_to_affine(b);
  return fe_eq(x.x, y.x) && fe_eq(x.y, y.y);
}
// SumcheckProof::verify (src/sumcheck.rs:67-114) on compressed polynomials of `deg` + 1 coefficients minus the linear one
inline bool sumcheck_verify(Tr& tr, const fe_t& claim, size_t rounds, size_t deg, const fe_t* cpolys, fe_t* e_out, std::vector<fe_t>* r_out) {
  fe_t e = claim;
  r_out->clear();
  for (size_t i = 0; i < rounds; ++i) {
    const fe_t* c = cpolys + i * deg;  // c[0] = constant, c[1..] = degree 2.. coefficients
    fe_t lin = fe_sub<S>(fe_sub<S>(e, c[0]), c[0]);  // CompressedUniPoly::decompress (univariate.rs:166-179)
    for (size_t k = 1; k < deg; ++k) lin = fe_sub<S>(lin, c[k]);
    std::vector<uint8_t> b(32 * deg);
    for (size_t k = 0; k < deg; ++k) sp::fe_to_le_bytes<S>(c[k], b.data() + 32 * k);
    tr.absorb("p", b.data(), b.size());
    const fe_t r = tr.squeeze("c");
    r_out->push_back(r);
    fe_t eval = c[0], power = r;  // UniPoly::evaluate on (c0, lin, c[1], ...)
    eval = fe_add<S>(eval, fe_mul<S>(power, lin));
    for (size_t k = 1; k < deg; ++k) {
      power = fe_mul<S>(power, r);
      eval = fe_add<S>(eval, fe_mul<S>(power, c[k]));
    }
    e = eval;
  }
  *e_out = e;
  return true;
}

// every scalar / coordinate of an untrusted proof must be a canonical residue: the reference's deserialisation rejects anything >= the modulus,
// and x and x + p would otherwise be two encodings of one proof (same transcript bytes, same group elements)
template <class F>
inline bool limbs_canonical(const fe_t& v) {
  for (int i = 7; i >= 0; --i) {
    if (v.v[i] < F::P(i)) return true;
    if (v.v[i] > F::P(i)) return false;
  }
  return false;  // == p
}


struct SpartanProofBuf {  // SpartanSNARK (src/spartan.rs:130-138) in the canonical flat layout of DESIGN.md
  std::vector<uint64_t> words;
  void pf(const fe_t& f) { words.insert(words.end(), u64p(&f), u64p(&f) + 4); }
  void pp(const aff_t& a) {
    pf(a.x);
    pf(a.y);
  }
};

inline R1CSIntView make_view(size_t num_cons, size_t num_shared, size_t num_precommitted, size_t num_rest, size_t num_public, size_t num_challenges,
                             const int64_t* Ad, const uint32_t* Ai, const uint64_t* Ap, const int64_t* Bd, const uint32_t* Bi, const uint64_t* Bp,
                             const int64_t* Cd, const uint32_t* Ci, const uint64_t* Cp) {
  R1CSIntView R{num_cons, num_shared, num_precommitted, num_rest, num_public, num_challenges, {{Ad, Ai, Ap}, {Bd, Bi, Bp}, {Cd, Ci, Cp}}};
  return R;
}

}  // namespace spartan2
