// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// extern "C" surface of the CPU restatement, loaded with ctypes by tests/, smoke() and the
// cpu_baseline leg of bench.py. Field elements cross as uint64[4] Montgomery limbs (the reference's
// in-memory form, SURVEY.md section 8 conventions); affine points as uint64[8] = x limbs | y limbs,
// (0,0) for the identity.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "neutronnova.hpp"
#include "nifs.hpp"
#include "spartan.hpp"
#include "neutronnova_zk.hpp"
#include "wire_formats.hpp"

using namespace oracle;

static thread_local std::string g_err;
#define ORC_TRY try {
#define ORC_CATCH                   \
  }                                 \
  catch (const std::exception& e) { \
    g_err = e.what();               \
    return -1;                      \
  }                                 \
  return 0;

template <class F>
static std::vector<F> load(const uint64_t* p, size_t n) {
  std::vector<F> v(n);
  for (size_t i = 0; i < n; ++i) memcpy(v[i].l, p + 4 * i, 32);
  return v;
}
template <class F>
static void store(uint64_t* p, const std::vector<F>& v) {
  for (size_t i = 0; i < v.size(); ++i) memcpy(p + 4 * i, v[i].l, 32);
}
static Affine load_aff(const uint64_t* p) {
  Affine a;
  memcpy(a.x.l, p, 32);
  memcpy(a.y.l, p + 4, 32);
  return a;
}
static void store_aff(uint64_t* p, const Affine& a) {
  memcpy(p, a.x.l, 32);
  memcpy(p + 4, a.y.l, 32);
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
// thread count of the OpenMP loops (0 = all host cores); returns the count in effect. Results never depend on it (exact arithmetic).
int orc_set_threads(int n) {
#ifdef _OPENMP
  static const int all = omp_get_max_threads() < 64 ? omp_get_max_threads() : 64;  // beyond this the short loops only pay for the fork/join
  omp_set_num_threads(n > 0 ? n : all);
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// ---- field ------------------------------------------------------------------------------------
// field_id: 0 = T256 scalar (P-256 base prime), 1 = T256 base, 2 = Pallas scalar
#define FIELD_DISPATCH(id, ...)                        \
  switch (id) {                                        \
    case 0: { typedef Fq F; __VA_ARGS__; break; }       \
    case 1: { typedef Fp F; __VA_ARGS__; break; }       \
    case 2: { typedef FqPallas F; __VA_ARGS__; break; } \
    default: return -1;                                \
  }

int orc_field_modulus(int id, uint64_t* out) {
  FIELD_DISPATCH(id, memcpy(out, F::P().p.l, 32));
  return 0;
}
int orc_field_binop(int id, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
  FIELD_DISPATCH(id, {
    F x = F::from_raw_mont(a), y = F::from_raw_mont(b), r;
    switch (op) {
      case 0: r = x + y; break;
      case 1: r = x - y; break;
      case 2: r = x * y; break;
      case 3: r = x.inv(); break;
      case 4: r = x.neg(); break;
      default: return -1;
    }
    memcpy(out, r.l, 32);
  });
  return 0;
}
int orc_field_from_canonical(int id, const uint64_t* v, uint64_t* out) {
  FIELD_DISPATCH(id, { F r = F::from_canonical(v); memcpy(out, r.l, 32); });
  return 0;
}
int orc_field_to_canonical(int id, const uint64_t* v, uint64_t* out) {
  FIELD_DISPATCH(id, { F::from_raw_mont(v).to_canonical(out); });
  return 0;
}
int orc_field_from_uniform(int id, const uint8_t* bytes64, uint64_t* out) {
  FIELD_DISPATCH(id, { F r = F::from_uniform(bytes64); memcpy(out, r.l, 32); });
  return 0;
}
// value contract of DelayedReduction (src/big_num/delayed_reduction.rs:41-84): reduce(sum a_i b_i)
int orc_field_dot(int id, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  FIELD_DISPATCH(id, {
    F acc = F::zero();
    for (size_t i = 0; i < n; ++i) acc = acc + F::from_raw_mont(a + 4 * i) * F::from_raw_mont(b + 4 * i);
    memcpy(out, acc.l, 32);
  });
  return 0;
}

// ---- hashing / transcript ---------------------------------------------------------------------
int orc_keccak256(const uint8_t* data, size_t n, uint8_t* out32) {
  Keccak256 h;
  h.update(data, n);
  h.finalize(out32);
  return 0;
}
int orc_shake256(const uint8_t* data, size_t n, uint8_t* out, size_t outlen) {
  Shake256 s;
  s.update(data, n);
  s.read(out, outlen);
  return 0;
}
void* orc_transcript_new(const char* label) { return new Transcript(label); }
void orc_transcript_free(void* t) { delete (Transcript*)t; }
int orc_transcript_absorb(void* t, const char* label, const uint8_t* bytes, size_t n) {
  ((Transcript*)t)->absorb_bytes(label, bytes, n);
  return 0;
}
int orc_transcript_dom_sep(void* t, const char* bytes) {
  ((Transcript*)t)->dom_sep(bytes);
  return 0;
}
int orc_transcript_squeeze(void* t, const char* label, int field_id, uint64_t* out) {
  uint8_t b[64];
  ((Transcript*)t)->squeeze_bytes(label, b);
  return orc_field_from_uniform(field_id, b, out);
}
// scalar absorbed with its transcript encoding (BE), field_id as above
int orc_transcript_absorb_scalar(void* t, const char* label, int field_id, const uint64_t* s) {
  FIELD_DISPATCH(field_id, ((Transcript*)t)->absorb_scalar(label, F::from_raw_mont(s)));
  return 0;
}

// ---- polynomials ------------------------------------------------------------------------------
int orc_eq_evals(const uint64_t* r, size_t ell, uint64_t* out) {
  store(out, eq_evals_from_points(load<Fq>(r, ell)));
  return 0;
}
int orc_bind_top(uint64_t* Z, size_t len, size_t* lo_eff, size_t* hi_eff, const uint64_t* r) {
  ORC_TRY
  MultilinearPolynomial<Fq> p(load<Fq>(Z, len), *lo_eff, *hi_eff);
  p.bind_poly_var_top(Fq::from_raw_mont(r));
  store(Z, p.Z);
  *lo_eff = p.lo_eff;
  *hi_eff = p.hi_eff;
  ORC_CATCH
}
int orc_multilinear_evaluate(const uint64_t* Z, size_t len, const uint64_t* r, size_t ell, uint64_t* out) {
  Fq v = multilinear_evaluate(load<Fq>(Z, len), load<Fq>(r, ell));
  memcpy(out, v.l, 32);
  return 0;
}
int orc_sparse_poly_evaluate(size_t num_vars, const uint64_t* Z, size_t zlen, const uint64_t* r, uint64_t* out) {
  ORC_TRY
  Fq v = sparse_poly_evaluate(num_vars, load<Fq>(Z, zlen), load<Fq>(r, num_vars));
  memcpy(out, v.l, 32);
  ORC_CATCH
}
int orc_unipoly_from_evals(const uint64_t* evals, size_t n, uint64_t* coeffs) {
  ORC_TRY
  store(coeffs, UniPoly<Fq>::from_evals(load<Fq>(evals, n)).coeffs);
  ORC_CATCH
}
int orc_unipoly_evaluate(const uint64_t* coeffs, size_t n, const uint64_t* r, uint64_t* out) {
  UniPoly<Fq> p;
  p.coeffs = load<Fq>(coeffs, n);
  Fq v = p.evaluate(Fq::from_raw_mont(r));
  memcpy(out, v.l, 32);
  return 0;
}

// ---- sum-check --------------------------------------------------------------------------------
int orc_sumcheck_cubic3(const uint64_t* claim, const uint64_t* taus, size_t ell, uint64_t* A, uint64_t* B, uint64_t* C, void* tr,
                        uint64_t* out_polys /*ell*3*/, uint64_t* out_r /*ell*/, uint64_t* out_final /*3*/) {
  ORC_TRY
  size_t len = (size_t)1 << ell;
  MultilinearPolynomial<Fq> a(load<Fq>(A, len)), b(load<Fq>(B, len)), c(load<Fq>(C, len));
  SumcheckProof<Fq> pf;
  std::vector<Fq> r, fin;
  prove_cubic_with_three_inputs(Fq::from_raw_mont(claim), load<Fq>(taus, ell), a, b, c, *(Transcript*)tr, &pf, &r, &fin);
  for (size_t i = 0; i < ell; ++i) store(out_polys + 12 * i, pf.compressed_polys[i]);
  store(out_r, r);
  store(out_final, fin);
  ORC_CATCH
}
int orc_sumcheck_quad(const uint64_t* claim, size_t rounds, uint64_t* A, size_t loA, size_t hiA, uint64_t* B, size_t loB, size_t hiB, void* tr,
                      uint64_t* out_polys /*rounds*2*/, uint64_t* out_r, uint64_t* out_final /*2*/) {
  ORC_TRY
  size_t len = (size_t)1 << rounds;
  MultilinearPolynomial<Fq> a(load<Fq>(A, len), loA, hiA), b(load<Fq>(B, len), loB, hiB);
  SumcheckProof<Fq> pf;
  std::vector<Fq> r, fin;
  prove_quad(Fq::from_raw_mont(claim), rounds, a, b, *(Transcript*)tr, &pf, &r, &fin);
  for (size_t i = 0; i < rounds; ++i) store(out_polys + 8 * i, pf.compressed_polys[i]);
  store(out_r, r);
  store(out_final, fin);
  ORC_CATCH
}
// src/sumcheck.rs:67-114. polys: rounds x (degree) compressed coefficients. returns 0 ok / 1 reject.
int orc_sumcheck_verify(const uint64_t* claim, size_t rounds, size_t degree, const uint64_t* polys, void* tr, uint64_t* out_e, uint64_t* out_r) {
  SumcheckProof<Fq> pf;
  for (size_t i = 0; i < rounds; ++i) pf.compressed_polys.push_back(load<Fq>(polys + 4 * degree * i, degree));
  Fq e;
  std::vector<Fq> r;
  if (!pf.verify(Fq::from_raw_mont(claim), rounds, degree, *(Transcript*)tr, &e, &r)) return 1;
  memcpy(out_e, e.l, 32);
  store(out_r, r);
  return 0;
}

// ---- curve / msm ------------------------------------------------------------------------------
int orc_curve_generator(uint64_t* out) {
  store_aff(out, T256Curve::generator());
  return 0;
}
int orc_on_curve(const uint64_t* p) { return on_curve(load_aff(p)) ? 1 : 0; }
int orc_point_add(const uint64_t* a, const uint64_t* b, uint64_t* out) {
  store_aff(out, Jac::from_affine(load_aff(a)).add(Jac::from_affine(load_aff(b))).to_affine());
  return 0;
}
int orc_point_mul(const uint64_t* a, const uint64_t* k, uint64_t* out) {
  store_aff(out, scalar_mul(Jac::from_affine(load_aff(a)), Fq::from_raw_mont(k)).to_affine());
  return 0;
}
int orc_from_label(const char* label, size_t n, uint64_t* out) {
  std::vector<Affine> g = from_label(label, n);
  for (size_t i = 0; i < n; ++i) store_aff(out + 8 * i, g[i]);
  return 0;
}
static std::vector<Affine> load_bases(const uint64_t* b, size_t n) {
  std::vector<Affine> v(n);
  for (size_t i = 0; i < n; ++i) v[i] = load_aff(b + 8 * i);
  return v;
}
int orc_msm(const uint64_t* scalars, const uint64_t* bases, size_t n, size_t threads, uint64_t* out) {
  std::vector<Fq> s = load<Fq>(scalars, n);
  std::vector<Affine> b = load_bases(bases, n);
  store_aff(out, msm(s.data(), b.data(), n, threads).to_affine());
  return 0;
}
int orc_msm_naive(const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t* out) {  // test-side naive sum (msm.rs:878-901)
  Jac acc = Jac::identity();
  for (size_t i = 0; i < n; ++i) acc = acc.add(scalar_mul(Jac::from_affine(load_aff(bases + 8 * i)), Fq::from_raw_mont(scalars + 4 * i)));
  store_aff(out, acc.to_affine());
  return 0;
}
int orc_msm_small(const uint64_t* scalars_u64, const uint64_t* bases, size_t n, uint64_t* out) {
  std::vector<Affine> b = load_bases(bases, n);
  store_aff(out, msm_small(scalars_u64, b.data(), n).to_affine());
  return 0;
}
int orc_fixed_base_mul(const uint64_t* base, const uint64_t* scalars, size_t n, uint64_t* out) {
  FixedBaseMul t = FixedBaseMul::precompute(Jac::from_affine(load_aff(base)), 8);
  for (size_t i = 0; i < n; ++i) store_aff(out + 8 * i, t.mul(Fq::from_raw_mont(scalars + 4 * i)).to_affine());
  return 0;
}

// ---- Hyrax ------------------------------------------------------------------------------------
void* orc_hyrax_setup(const char* label, size_t width) { return new HyraxKey(HyraxKey::setup(label, width)); }
void orc_hyrax_free(void* k) { delete (HyraxKey*)k; }
int orc_hyrax_key_export(void* k, uint64_t* ck_out /*width*8*/, uint64_t* h_out /*8*/) {
  HyraxKey* key = (HyraxKey*)k;
  for (size_t i = 0; i < key->ck.size(); ++i) store_aff(ck_out + 8 * i, key->ck[i]);
  store_aff(h_out, key->h.to_affine());
  return 0;
}
int orc_hyrax_commit(void* k, const uint64_t* v, size_t n, const uint64_t* blinds, int is_small, uint64_t* out_rows) {
  ORC_TRY
  HyraxKey* key = (HyraxKey*)k;
  std::vector<Fq> vv = load<Fq>(v, n);
  HyraxBlind b = load<Fq>(blinds, div_ceil(n, key->num_cols));
  HyraxCommitment c = hyrax_commit(*key, vv.data(), n, b, is_small != 0);
  std::vector<Affine> a = batch_affine(c);
  for (size_t i = 0; i < a.size(); ++i) store_aff(out_rows + 8 * i, a[i]);
  ORC_CATCH
}
// PCS::commit_without_blind / commit_incremental (hyrax_pc.rs:533-607): raw rows cross as affine points, (0,0) = identity
int orc_hyrax_commit_without_blind(void* k, const uint64_t* v, size_t n, int is_small, uint64_t* out_rows) {
  ORC_TRY
  HyraxKey* key = (HyraxKey*)k;
  std::vector<Fq> vv = load<Fq>(v, n);
  std::vector<Affine> a = batch_affine(hyrax_commit_without_blind(*key, vv.data(), n, is_small != 0));
  for (size_t i = 0; i < a.size(); ++i) store_aff(out_rows + 8 * i, a[i]);
  ORC_CATCH
}
int orc_hyrax_commit_incremental(void* k, const uint64_t* raw_rows, size_t nraw, const uint64_t* delta, size_t n, const uint64_t* blinds, uint64_t* out_rows) {
  ORC_TRY
  HyraxKey* key = (HyraxKey*)k;
  std::vector<Jac> raw(nraw);
  for (size_t i = 0; i < nraw; ++i) raw[i] = Jac::from_affine(load_aff(raw_rows + 8 * i));
  std::vector<Fq> dv = load<Fq>(delta, n);
  HyraxBlind b = load<Fq>(blinds, div_ceil(n, key->num_cols));
  std::vector<Affine> a = batch_affine(hyrax_commit_incremental(*key, raw, dv.data(), n, b));
  for (size_t i = 0; i < a.size(); ++i) store_aff(out_rows + 8 * i, a[i]);
  ORC_CATCH
}
// HyraxPCS::prove (hyrax_pc.rs:387-478) + the linear IPA (ipa.rs:125-170) as one call, for the ABI-level parity test of sp_hyrax_prove. The tape holds
// d_vec (cols blocks), r_delta, r_beta in the reference's draw order. out = delta (8) | beta (8) | z_vec (4 * cols) | z_delta (4) | z_beta (4).
int orc_hyrax_prove(void* k, void* k_eval, void* tr, const uint64_t* comm_rows, size_t rows, const uint64_t* poly, size_t n, const uint64_t* blinds, const uint64_t* point,
                    size_t npt, const uint64_t* comm_eval, const uint64_t* blind_eval, const uint8_t* tape, size_t tape_blocks, uint64_t* out) {
  ORC_TRY
  HyraxKey *key = (HyraxKey*)k, *key_eval = (HyraxKey*)k_eval;
  HyraxCommitment comm(rows), ce(1);
  for (size_t i = 0; i < rows; ++i) comm[i] = Jac::from_affine(load_aff(comm_rows + 8 * i));
  ce[0] = Jac::from_affine(load_aff(comm_eval));
  Tape tp(tape, tape_blocks);
  IpaProof p = hyrax_prove(*key, *key_eval, *(Transcript*)tr, comm, load<Fq>(poly, n), load<Fq>(blinds, rows), load<Fq>(point, npt), ce, load<Fq>(blind_eval, 1), tp);
  store_aff(out, p.delta.to_affine());
  store_aff(out + 8, p.beta.to_affine());
  store(out + 16, p.z_vec);
  std::vector<Fq> tail{p.z_delta, p.z_beta};
  store(out + 16 + 4 * p.z_vec.size(), tail);
  ORC_CATCH
}
int orc_rowmat_vec(const uint64_t* poly, const uint64_t* l, size_t rows, size_t cols, uint64_t* out) {  // bind_with_delayed
  std::vector<Fq> p = load<Fq>(poly, rows * cols), L = load<Fq>(l, rows);
  store(out, bind_with_delayed(p.data(), L, cols));
  return 0;
}

// ---- NeutronNova kernel-level rows (oracle/neutronnova.hpp) ---------------------------------------------------
int orc_weights_from_r(const uint64_t* r_bs, size_t ell, size_t n, uint64_t* out) {
  store(out, weights_from_r(load<Fq>(r_bs, ell), n));
  return 0;
}
// Ws: n_inst tables of `dim` elements, concatenated
int orc_fold_witnesses(const uint64_t* weights, const uint64_t* Ws, size_t n_inst, size_t dim, uint64_t* out) {
  std::vector<Fq> w = load<Fq>(weights, n_inst), all = load<Fq>(Ws, n_inst * dim);
  std::vector<const Fq*> ptrs;
  for (size_t i = 0; i < n_inst; ++i) ptrs.push_back(all.data() + i * dim);
  store(out, fold_witnesses(w, ptrs, dim));
  return 0;
}
// bases_rows: rows x n affine points, row-major; out: rows affine points
int orc_msm_shared_weights(const uint64_t* weights, size_t n, const uint64_t* bases_rows, size_t rows, uint64_t* out) {
  std::vector<Fq> w = load<Fq>(weights, n);
  std::vector<std::vector<Affine>> br(rows);
  for (size_t r = 0; r < rows; ++r) br[r] = load_bases(bases_rows + 8 * r * n, n);
  std::vector<Jac> res = msm_shared_weights(w, br);
  std::vector<Affine> a = batch_affine(res);
  for (size_t r = 0; r < rows; ++r) store_aff(out + 8 * r, a[r]);
  return 0;
}
int orc_eval_cubic_outer_pow(const uint64_t* pow_left, size_t nleft, const uint64_t* pow_right, size_t nright, const uint64_t* A, const uint64_t* B,
                             const uint64_t* C, size_t len2, uint64_t* out3) {
  Fq e0, e2, e3;
  eval_points_cubic_outer_pow(load<Fq>(pow_left, nleft), load<Fq>(pow_right, nright), load<Fq>(A, len2), load<Fq>(B, len2), load<Fq>(C, len2), &e0, &e2, &e3);
  memcpy(out3, e0.l, 32);
  memcpy(out3 + 4, e2.l, 32);
  memcpy(out3 + 8, e3.l, 32);
  return 0;
}

// ---- NeutronNova NIFS data path (oracle/nifs.hpp) ------------------------------------------------------------
// out_i64[n], out_large[n] (0/1 flags); returns the number of large positions
long orc_to_small_vec_or_zero(const uint64_t* v, size_t n, int64_t* out_i64, uint8_t* out_large) {
  std::vector<Fq> f = load<Fq>(v, n);
  std::vector<int64_t> o;
  std::vector<size_t> l;
  to_small_vec_or_zero(f.data(), n, o, l);
  memcpy(out_i64, o.data(), 8 * n);
  memset(out_large, 0, n);
  for (size_t k : l) out_large[k] = 1;
  return (long)l.size();
}
// sum_i f_i * (a_i * b_i) through SmallAccumulator (small_value.rs:254-403 property tests)
int orc_small_acc_dot(const uint64_t* f, const int64_t* a, const int64_t* b, size_t n, uint64_t* out) {
  ORC_TRY
  std::vector<Fq> fv = load<Fq>(f, n);
  SmallAccumulator acc;
  for (size_t i = 0; i < n; ++i) acc.accumulate(fv[i], (__int128)a[i] * (__int128)b[i]);
  Fq r = acc.reduce();
  memcpy(out, r.l, 32);
  ORC_CATCH
}
int orc_tensor_decomp(size_t n, size_t* ell, size_t* left, size_t* right) {
  compute_tensor_decomp(n, ell, left, right);
  return 0;
}
int orc_pow_split_evals(const uint64_t* tau, size_t ell, size_t left, size_t right, uint64_t* out) {
  ORC_TRY
  store(out, pow_split_evals(load<Fq>(tau, 1)[0], ell, left, right));
  ORC_CATCH
}
typedef void (*orc_nifs_hook)(void* user, size_t t, const uint64_t* coeffs16, uint64_t* r_b);
// A, B, C: n_padded layers of left*right elements, concatenated. out_polys: ell_b x 4 coefficients [d, c, b, a]; out_tail = T_out | eq_rho_at_rb
int orc_nifs_prove_core(size_t n_padded, size_t left, size_t right, const uint64_t* E_eq, const uint64_t* rhos, size_t ell_b, const uint64_t* A,
                        const uint64_t* B, const uint64_t* C, int use_i64, orc_nifs_hook hook, void* user, uint64_t* out_polys, uint64_t* out_r_bs,
                        uint64_t* out_A, uint64_t* out_B, uint64_t* out_C, uint64_t* out_tail) {
  ORC_TRY
  size_t total = left * right;
  std::vector<Layer> a(n_padded), b(n_padded), c(n_padded);
  for (size_t i = 0; i < n_padded; ++i) {
    a[i] = load<Fq>(A + 4 * i * total, total);
    b[i] = load<Fq>(B + 4 * i * total, total);
    c[i] = load<Fq>(C + 4 * i * total, total);
  }
  NifsRoundHook h = [&](size_t t, const std::array<Fq, 4>& co) {
    uint64_t buf[16], r[4];
    for (int i = 0; i < 4; ++i) memcpy(buf + 4 * i, co[i].l, 32);
    hook(user, t, buf, r);
    return Fq::from_raw_mont(r);
  };
  NifsCoreOutput o = nifs_prove_core(left, right, load<Fq>(E_eq, left + right), load<Fq>(rhos, ell_b), std::move(a), std::move(b), std::move(c), use_i64 != 0, h);
  for (size_t t = 0; t < ell_b; ++t)
    for (int i = 0; i < 4; ++i) memcpy(out_polys + 16 * t + 4 * i, o.polys[t][i].l, 32);
  store(out_r_bs, o.r_bs);
  store(out_A, o.A);
  store(out_B, o.B);
  store(out_C, o.C);
  memcpy(out_tail, o.T_out.l, 32);
  memcpy(out_tail + 4, o.eq_rho_at_rb.l, 32);
  ORC_CATCH
}

// ---- R1CS shape -------------------------------------------------------------------------------
struct IntCsr {
  const int64_t* data;
  const uint32_t* indices;
  const uint64_t* indptr;
};
static SparseMatrix<Fq> to_matrix(const IntCsr& m, size_t rows, size_t cols) {
  SparseMatrix<Fq> M;
  size_t nnz = m.indptr[rows];
  M.data.resize(nnz);
  M.indices.resize(nnz);
  for (size_t i = 0; i < nnz; ++i) {
    M.data[i] = Fq::from_i64(m.data[i]);
    M.indices[i] = m.indices[i];
  }
  M.indptr.assign(m.indptr, m.indptr + rows + 1);
  M.cols = cols;
  return M;
}
// arguments == SplitR1CSShape::new (src/r1cs/mod.rs:810-820) with int64 coefficients
void* orc_shape_new(size_t num_cons, size_t num_shared, size_t num_precommitted, size_t num_rest, size_t num_public, size_t num_challenges,
                    const int64_t* Ad, const uint32_t* Ai, const uint64_t* Ap, const int64_t* Bd, const uint32_t* Bi, const uint64_t* Bp,
                    const int64_t* Cd, const uint32_t* Ci, const uint64_t* Cp) {
  try {
    size_t cols = num_shared + num_precommitted + num_rest + 1 + num_public + num_challenges;
    auto* S = new SplitR1CSShape<Fq>(SplitR1CSShape<Fq>::make(num_cons, num_shared, num_precommitted, num_rest, num_public, num_challenges,
                                                              to_matrix(IntCsr{Ad, Ai, Ap}, num_cons, cols), to_matrix(IntCsr{Bd, Bi, Bp}, num_cons, cols),
                                                              to_matrix(IntCsr{Cd, Ci, Cp}, num_cons, cols)));
    return S;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void orc_shape_free(void* s) { delete (SplitR1CSShape<Fq>*)s; }
// SplitR1CSShape::equalize (src/r1cs/mod.rs:913-971) on two shapes in place; orc_shape_digest = SHA-256 over SplitR1CSShape::write_bytes (:775-794)
int orc_shape_equalize(void* a, void* b) {
  try {
    SplitR1CSShape<Fq>::equalize(*(SplitR1CSShape<Fq>*)a, *(SplitR1CSShape<Fq>*)b);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int orc_shape_digest(void* s, uint8_t out[32]) {
  Sha256 h;
  WireWriter w(&h);
  w.shape_digest_bytes(*(SplitR1CSShape<Fq>*)s);
  h.finalize(out);
  return 0;
}
int orc_shape_sizes(void* s, uint64_t* out10) {  // SplitR1CSShape::sizes (src/r1cs/mod.rs:1008-1021)
  auto* S = (SplitR1CSShape<Fq>*)s;
  uint64_t v[10] = {S->num_cons_unpadded, S->num_shared_unpadded, S->num_precommitted_unpadded, S->num_rest_unpadded, S->num_cons,
                    S->num_shared,        S->num_precommitted,    S->num_rest,                  S->num_public,        S->num_challenges};
  memcpy(out10, v, sizeof v);
  return 0;
}
int orc_shape_multiply_vec(void* s, const uint64_t* z, uint64_t* az, uint64_t* bz, uint64_t* cz) {
  ORC_TRY
  auto* S = (SplitR1CSShape<Fq>*)s;
  std::vector<Fq> a, b, c;
  S->multiply_vec(load<Fq>(z, S->num_vars() + S->num_extra()), &a, &b, &c);
  store(az, a);
  store(bz, b);
  store(cz, c);
  ORC_CATCH
}
int orc_shape_multiply_vec_incremental(void* s, const uint64_t* z, const uint64_t* caz, const uint64_t* cbz, const uint64_t* ccz, uint64_t* az,
                                       uint64_t* bz, uint64_t* cz) {
  ORC_TRY
  auto* S = (SplitR1CSShape<Fq>*)s;
  std::vector<Fq> a, b, c;
  S->multiply_vec_incremental_into(load<Fq>(z, S->num_vars() + S->num_extra()), load<Fq>(caz, S->num_cons), load<Fq>(cbz, S->num_cons),
                                   load<Fq>(ccz, S->num_cons), &a, &b, &c);
  store(az, a);
  store(bz, b);
  store(cz, c);
  ORC_CATCH
}
int orc_shape_poly_abc(void* s, const uint64_t* rx, const uint64_t* r, size_t out_len, uint64_t* out) {
  ORC_TRY
  auto* S = (SplitR1CSShape<Fq>*)s;
  store(out, S->bind_and_prepare_poly_ABC_inner(load<Fq>(rx, S->num_cons), Fq::from_raw_mont(r), out_len));
  ORC_CATCH
}

int orc_zero_check_round0(const uint64_t* taus, size_t ell, const uint64_t* A, const uint64_t* B, uint64_t* out3) {
  ORC_TRY
  std::vector<Fq> t = load<Fq>(taus, ell);
  MultilinearPolynomial<Fq> pa(load<Fq>(A, (size_t)1 << ell)), pb(load<Fq>(B, (size_t)1 << ell));
  EqSumCheckInstance<Fq> eq(t);
  Fq e0, e2, e3;
  eq.evaluation_points_zero_check_round0(pa, pb, &e0, &e2, &e3);
  store(out3, std::vector<Fq>{e0, e2, e3});
  ORC_CATCH
}

// ---- Spartan ----------------------------------------------------------------------------------
void* orc_spartan_setup(void* shape) {
  try {
    return new SpartanProverKey(spartan_setup(*(SplitR1CSShape<Fq>*)shape));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void orc_spartan_pk_free(void* pk) { delete (SpartanProverKey*)pk; }
int orc_spartan_pk_export(void* pk, uint64_t* ck /*2048*8*/, uint64_t* h /*8*/, uint64_t* ck_s /*8*/, uint64_t* h_s /*8*/, uint8_t* digest32) {
  auto* k = (SpartanProverKey*)pk;
  for (size_t i = 0; i < k->ck.ck.size(); ++i) store_aff(ck + 8 * i, k->ck.ck[i]);
  store_aff(h, k->ck.h.to_affine());
  store_aff(ck_s, k->ck_s.ck[0]);
  store_aff(h_s, k->ck_s.h.to_affine());
  memcpy(digest32, k->vk_digest, 32);
  return 0;
}
static std::vector<Fq> from_u64s(const uint64_t* v, size_t n) {
  std::vector<Fq> out(n);
  for (size_t i = 0; i < n; ++i) out[i] = Fq::from_u64(v[i]);
  return out;
}
void* orc_spartan_prep_prove(void* pk, const uint64_t* witness_u64, size_t n, int is_small, const uint8_t* tape, size_t tape_blocks, size_t* tape_used) {
  try {
    Tape t(tape, tape_blocks);
    auto* ps = new SpartanPrep(spartan_prep_prove(*(SpartanProverKey*)pk, from_u64s(witness_u64, n), is_small != 0, t));
    if (tape_used) *tape_used = t.pos;
    return ps;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void orc_spartan_prep_free(void* ps) { delete (SpartanPrep*)ps; }
int orc_spartan_prep_export(void* ps_, uint64_t* comm_rows, uint64_t* caz, uint64_t* cbz, uint64_t* ccz) {
  auto* ps = (SpartanPrep*)ps_;
  if (comm_rows) {
    HyraxCommitment both = ps->comm_W_shared;
    both.insert(both.end(), ps->comm_W_precommitted.begin(), ps->comm_W_precommitted.end());
    std::vector<Affine> a = batch_affine(both);
    for (size_t i = 0; i < a.size(); ++i) store_aff(comm_rows + 8 * i, a[i]);
  }
  if (caz) store(caz, ps->cached_az);
  if (cbz) store(cbz, ps->cached_bz);
  if (ccz) store(ccz, ps->cached_cz);
  return 0;
}
// synthesize callback of a circuit with verifier challenges: (user, challenges (nch x 4 limbs), nch, out rest witness (num_rest_unpadded x 4 limbs))
typedef void (*orc_rest_hook)(void* user, const uint64_t* challenges, size_t nch, uint64_t* out_rest);
void* orc_spartan_prove_hook(void* pk, void* ps, const uint64_t* publics_u64, size_t npub, const uint8_t* tape, size_t tape_blocks, size_t* tape_used,
                             double* seconds, orc_rest_hook hook, void* user);
void* orc_spartan_prove(void* pk, void* ps, const uint64_t* publics_u64, size_t npub, const uint8_t* tape, size_t tape_blocks, size_t* tape_used,
                        double* seconds) {
  return orc_spartan_prove_hook(pk, ps, publics_u64, npub, tape, tape_blocks, tape_used, seconds, nullptr, nullptr);
}
void* orc_spartan_prove_hook(void* pk, void* ps, const uint64_t* publics_u64, size_t npub, const uint8_t* tape, size_t tape_blocks, size_t* tape_used,
                             double* seconds, orc_rest_hook hook, void* user) {
  try {
    Tape t(tape, tape_blocks);
    auto t0 = std::chrono::steady_clock::now();
    const size_t nrest = ((SpartanProverKey*)pk)->S.num_rest_unpadded;
    RestSynth synth;
    if (hook)
      synth = [hook, user, nrest](const std::vector<Fq>& ch) {
        std::vector<uint64_t> raw(4 * ch.size() + 4), out(4 * nrest + 4);
        for (size_t i = 0; i < ch.size(); ++i) memcpy(&raw[4 * i], ch[i].l, 32);
        hook(user, raw.data(), ch.size(), out.data());
        return load<Fq>(out.data(), nrest);
      };
    auto* pf = new SpartanProof(spartan_prove(*(SpartanProverKey*)pk, *(SpartanPrep*)ps, from_u64s(publics_u64, npub), t, synth));
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (tape_used) *tape_used = t.pos;
    return pf;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void orc_spartan_proof_free(void* pf) { delete (SpartanProof*)pf; }
size_t orc_spartan_proof_words(void* pf) { return ((SpartanProof*)pf)->serialize().size(); }
int orc_spartan_proof_serialize(void* pf, uint64_t* out) {
  std::vector<uint64_t> v = ((SpartanProof*)pf)->serialize();
  memcpy(out, v.data(), v.size() * 8);
  return 0;
}
int orc_spartan_verify(void* pk, void* pf) {
  try {
    return spartan_verify(*(SpartanProverKey*)pk, *(SpartanProof*)pf);
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
// Rebuild a proof from the canonical flat layout (SpartanProof::serialize) — used to run the
// restated verifier on proofs produced by the HIP library.
void* orc_spartan_proof_from_words(void* pk_, const uint64_t* w, size_t nwords) {
  try {
    auto* pk = (SpartanProverKey*)pk_;
    const SplitR1CSShape<Fq>& S = pk->S;
    size_t rows_sh = div_ceil(S.num_shared, pk->ck.num_cols);
    size_t rows_pre = div_ceil(S.num_precommitted, pk->ck.num_cols), rows_rest = div_ceil(S.num_rest, pk->ck.num_cols);
    size_t lx = log2_exact(S.num_cons), ly = log2_exact(S.num_vars()) + 1, nz = pk->ck.num_cols;
    if (S.num_vars() < nz) nz = S.num_vars();
    size_t expect = 8 * (rows_sh + rows_pre + rows_rest) + 4 * (S.num_public + S.num_challenges) + 12 * lx + 12 + 8 * ly + 8 + 16 + 4 * nz + 8;
    if (nwords != expect) throw std::runtime_error("proof_from_words: length mismatch");
    auto* pf = new SpartanProof();
    size_t o = 0;
    auto gf = [&]() { Fq f = Fq::from_raw_mont(w + o); o += 4; return f; };
    auto gp = [&]() { Affine a = load_aff(w + o); o += 8; return Jac::from_affine(a); };
    for (size_t i = 0; i < rows_sh + rows_pre + rows_rest; ++i) pf->comm_W.push_back(gp());
    pf->rows_shared = rows_sh;
    pf->rows_precommitted = rows_pre;
    for (size_t i = 0; i < S.num_public; ++i) pf->public_values.push_back(gf());
    for (size_t i = 0; i < S.num_challenges; ++i) pf->challenges.push_back(gf());
    for (size_t i = 0; i < lx; ++i) pf->sc_proof_outer.compressed_polys.push_back({gf(), gf(), gf()});
    for (int i = 0; i < 3; ++i) pf->claims_outer[i] = gf();
    for (size_t i = 0; i < ly; ++i) pf->sc_proof_inner.compressed_polys.push_back({gf(), gf()});
    pf->eval_W = gf();
    pf->blind_eval_W = gf();
    pf->eval_arg.delta = gp();
    pf->eval_arg.beta = gp();
    for (size_t i = 0; i < nz; ++i) pf->eval_arg.z_vec.push_back(gf());
    pf->eval_arg.z_delta = gf();
    pf->eval_arg.z_beta = gf();
    return pf;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}

// NeutronNovaNIFS::prove as a whole (oracle/nifs.hpp). Instances: comm rows (n x rows x affine), X (n x d), W (n x num_vars), r_W (n x rows).
// Outputs: polys (ell_b x 4), r_bs, E_eq (left + right), final layers A, B, C (num_cons each), tail = T_out | eq_rho_at_rb,
// folded W (num_vars), folded r_W (rows), folded X (d), folded comm (rows affine).
int orc_nifs_prove(void* shape, void* key, size_t n, size_t rows, size_t d, const uint64_t* comms, const uint64_t* X, const uint64_t* W, const uint64_t* r_W,
                   int use_i64, void* transcript, orc_nifs_hook hook, void* user, uint64_t* out_polys, uint64_t* out_r_bs, uint64_t* out_E, uint64_t* out_A,
                   uint64_t* out_B, uint64_t* out_C, uint64_t* out_tail, uint64_t* out_W, uint64_t* out_rW, uint64_t* out_X, uint64_t* out_comm) {
  ORC_TRY
  auto* S = (SplitR1CSShape<Fq>*)shape;
  auto* ck = (HyraxKey*)key;
  size_t nv = S->num_vars();
  std::vector<NifsInstance> Us(n);
  std::vector<NifsWitness> Ws(n);
  for (size_t i = 0; i < n; ++i) {
    for (size_t r = 0; r < rows; ++r) Us[i].comm_W.push_back(Jac::from_affine(load_aff(comms + 8 * (i * rows + r))));
    Us[i].X = load<Fq>(X + 4 * i * d, d);
    Ws[i].W = load<Fq>(W + 4 * i * nv, nv);
    Ws[i].r_W = load<Fq>(r_W + 4 * i * rows, rows);
  }
  NifsRoundHook h = [&](size_t t, const std::array<Fq, 4>& co) {
    uint64_t buf[16], r[4] = {0, 0, 0, 0};
    for (int q = 0; q < 4; ++q) memcpy(buf + 4 * q, co[q].l, 32);
    hook(user, t, buf, r);
    return Fq::from_raw_mont(r);
  };
  NifsProveOutput o = nifs_prove(*S, *ck, std::move(Us), std::move(Ws), use_i64 != 0, *(Transcript*)transcript, h);
  for (size_t t = 0; t < o.core.polys.size(); ++t)
    for (int q = 0; q < 4; ++q) memcpy(out_polys + 16 * t + 4 * q, o.core.polys[t][q].l, 32);
  store(out_r_bs, o.core.r_bs);
  store(out_E, o.E_eq);
  store(out_A, o.core.A);
  store(out_B, o.core.B);
  store(out_C, o.core.C);
  memcpy(out_tail, o.core.T_out.l, 32);
  memcpy(out_tail + 4, o.core.eq_rho_at_rb.l, 32);
  store(out_W, o.folded_W.W);
  store(out_rW, o.folded_W.r_W);
  store(out_X, o.folded_U.X);
  std::vector<Affine> a = batch_affine(o.folded_U.comm_W);
  for (size_t r = 0; r < a.size(); ++r) store_aff(out_comm + 8 * r, a[r]);
  ORC_CATCH
}

// ---- batched ZK sum-check drivers (oracle/neutronnova.hpp) -----------------------------------------------------------------
typedef void (*orc_batched_hook)(void* user, size_t round, const uint64_t* coeffs_step, const uint64_t* coeffs_core, size_t ncoeffs, uint64_t* r_out);
static BatchedRoundHook wrap_batched(orc_batched_hook hook, void* user) {
  return [=](size_t round, const std::vector<Fq>& cs, const std::vector<Fq>& cc) {
    uint64_t a[16], b[16], r[4];
    for (size_t i = 0; i < cs.size(); ++i) {
      memcpy(a + 4 * i, cs[i].l, 32);
      memcpy(b + 4 * i, cc[i].l, 32);
    }
    hook(user, round, a, b, cs.size(), r);
    return Fq::from_raw_mont(r);
  };
}
// tables: A0 | A1 | B0 | B1, each 2^num_rounds elements; out_finals: 4 F
int orc_prove_quad_batched(const uint64_t* claims2, size_t num_rounds, const uint64_t* tables, size_t start_round, orc_batched_hook hook, void* user,
                           uint64_t* out_r, uint64_t* out_finals) {
  ORC_TRY
  size_t n = (size_t)1 << num_rounds;
  MultilinearPolynomial<Fq> A0(load<Fq>(tables, n)), A1(load<Fq>(tables + 4 * n, n)), B0(load<Fq>(tables + 8 * n, n)), B1(load<Fq>(tables + 12 * n, n));
  Fq claims[2] = {load<Fq>(claims2, 2)[0], load<Fq>(claims2, 2)[1]};
  std::vector<Fq> r, fin;
  prove_quad_batched(claims, num_rounds, A0, A1, B0, B1, start_round, wrap_batched(hook, user), &r, &fin);
  store(out_r, r);
  store(out_finals, fin);
  ORC_CATCH
}
// tables: A_step | B_step | C_step | A_core | B_core | C_core, each 2^num_rounds; out_finals: those six at index 0; out_base_tau: pow_left[0] at the end
int orc_prove_cubic_outer_pow_batched(size_t num_rounds, const uint64_t* pow_left, size_t nleft, const uint64_t* pow_right, size_t nright, const uint64_t* tables,
                                      const uint64_t* t_out_step, size_t start_round, orc_batched_hook hook, void* user, uint64_t* out_r, uint64_t* out_finals,
                                      uint64_t* out_base_tau) {
  ORC_TRY
  size_t n = (size_t)1 << num_rounds;
  std::vector<MultilinearPolynomial<Fq>> t;
  for (int q = 0; q < 6; ++q) t.emplace_back(load<Fq>(tables + 4 * q * n, n));
  MultilinearPolynomial<Fq>* step[3] = {&t[0], &t[1], &t[2]};
  MultilinearPolynomial<Fq>* core[3] = {&t[3], &t[4], &t[5]};
  std::vector<Fq> pl = load<Fq>(pow_left, nleft), pr = load<Fq>(pow_right, nright);
  std::vector<Fq> r = prove_cubic_outer_pow_batched(num_rounds, pl, pr, step, core, load<Fq>(t_out_step, 1)[0], start_round, wrap_batched(hook, user));
  store(out_r, r);
  for (int q = 0; q < 6; ++q) memcpy(out_finals + 4 * q, t[q].Z[0].l, 32);
  memcpy(out_base_tau, pl[0].l, 32);
  ORC_CATCH
}


// ---- NeutronNovaZkSNARK (oracle/neutronnova_zk.hpp) ---------------------------------------------------------------------------------------------
void* orc_nn_setup(void* shape_step, void* shape_core, size_t num_steps) {
  try {
    return nn_setup(*(SplitR1CSShape<Fq>*)shape_step, *(SplitR1CSShape<Fq>*)shape_core, num_steps).release();
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void orc_nn_free(void* k) { delete (NNKey*)k; }
// out: nb, nx, ny, vc rounds, vc total vars, vc num_cons, vc num_cons_unpadded, vc num_public
void orc_nn_info(void* k, uint64_t out[8]) {
  auto* pk = (NNKey*)k;
  uint64_t v[8] = {pk->nb, pk->nx, pk->ny, pk->vc_shape->num_rounds, pk->vc_shape->total_vars(), pk->vc_shape->num_cons, pk->vc_shape->num_cons_unpadded, pk->vc_shape->num_public};
  memcpy(out, v, sizeof v);
}
void orc_nn_digest(void* k, uint8_t out[32]) { memcpy(out, ((NNKey*)k)->vk_digest, 32); }
// prep_prove + prove. step_wit: n x wit_len u64 (unpadded aux: shared | precommitted), step_pub: n x npub u64; tape_used[0] = blocks used by prep_prove, [1] = by prove
void* orc_nn_prove(void* k, size_t n, const uint64_t* step_wit, size_t wit_len, const uint64_t* step_pub, size_t npub, const uint64_t* core_wit, const uint64_t* core_pub,
                   int is_small, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, double* seconds) {
  try {
    auto* pk = (NNKey*)k;
    std::vector<std::vector<Fq>> sw(n), sp(n);
    for (size_t i = 0; i < n; ++i) {
      sw[i] = from_u64s(step_wit + i * wit_len, wit_len);
      sp[i] = from_u64s(step_pub + i * npub, npub);
    }
    Tape t(tape, tape_blocks);
    // the core circuit's own lengths (after SplitR1CSShape::equalize the core may have more or fewer variables and public values than a step)
    const SplitR1CSShape<Fq>& Sc = pk->S_core;
    NNPrep ps = nn_prep_prove(*pk, sw, sp, from_u64s(core_wit, Sc.num_shared_unpadded + Sc.num_precommitted_unpadded + Sc.num_rest_unpadded), from_u64s(core_pub, Sc.num_public),
                              is_small != 0, t);
    if (tape_used) tape_used[0] = t.pos;
    auto t0 = std::chrono::steady_clock::now();
    auto* pf = new NNProof(nn_prove(*pk, ps, is_small != 0, t));
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (tape_used) tape_used[1] = t.pos - tape_used[0];
    return pf;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void orc_nn_proof_free(void* pf) { delete (NNProof*)pf; }
size_t orc_nn_proof_words(void* pf) { return ((NNProof*)pf)->serialize().size(); }
int orc_nn_proof_serialize(void* pf, uint64_t* out) {
  std::vector<uint64_t> v = ((NNProof*)pf)->serialize();
  memcpy(out, v.data(), v.size() * 8);
  return 0;
}
int orc_nn_verify(void* k, void* pf) {
  try {
    return nn_verify(*(NNKey*)k, *(NNProof*)pf);
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
// rebuild a proof from the canonical flat layout (NNProof::serialize), for the verifier-side tests of device-produced proofs
void* orc_nn_proof_from_words(void* k, const uint64_t* w, size_t nwords) {
  try {
    auto* pk = (NNKey*)k;
    const SplitR1CSShape<Fq>& S = pk->S_step;
    const MultiRoundShape& vs = *pk->vc_shape;
    const size_t CW = DEFAULT_COMMITMENT_WIDTH, rows_sh = div_ceil(S.num_shared, CW), rows_pre = div_ceil(S.num_precommitted, CW), rows_rest = div_ceil(S.num_rest, CW);
    size_t o = 0;
    auto need = [&](size_t k_) {
      if (o + k_ > nwords) throw std::runtime_error("nn_proof_from_words: buffer too short");
    };
    auto gf = [&]() { need(4); Fq f = Fq::from_raw_mont(w + o); o += 4; return f; };
    auto gc = [&](size_t rows) {
      HyraxCommitment c;
      for (size_t i = 0; i < rows; ++i) {
        need(8);
        c.push_back(Jac::from_affine(load_aff(w + o)));
        o += 8;
      }
      return c;
    };
    auto* pf = new NNProof();
    std::unique_ptr<NNProof> guard(pf);
    pf->comm_W_shared = gc(rows_sh);
    (void)rows_pre;
    (void)rows_rest;
    auto ginst = [&](const SplitR1CSShape<Fq>& Sh) {  // step and core may split their rows differently (same total after equalize)
      NNSplitInstance u;
      u.comm_pre = gc(div_ceil(Sh.num_precommitted, CW));
      u.comm_rest = gc(div_ceil(Sh.num_rest, CW));
      for (size_t i = 0; i < Sh.num_public; ++i) u.publics.push_back(gf());
      return u;
    };
    for (size_t i = 0; i < pk->num_steps; ++i) pf->step_instances.push_back(ginst(S));
    pf->core_instance = ginst(pk->S_core);
    HyraxCommitment db = gc(2);
    pf->eval_arg.delta = db[0];
    pf->eval_arg.beta = db[1];
    for (size_t i = 0; i < CW; ++i) pf->eval_arg.z_vec.push_back(gf());
    pf->eval_arg.z_delta = gf();
    pf->eval_arg.z_beta = gf();
    for (size_t r = 0; r < vs.num_rounds; ++r) pf->U_verifier.comm_w_per_round.push_back(gc(vs.vars_padded[r] / vs.width));
    for (size_t i = 0; i < vs.num_public; ++i) pf->U_verifier.public_values.push_back(gf());
    for (size_t r = 0; r < vs.num_rounds; ++r) {
      std::vector<Fq> c;
      for (size_t i = 0; i < vs.chals_per_round[r]; ++i) c.push_back(gf());
      pf->U_verifier.challenges_per_round.push_back(c);
    }
    pf->nifs_comm_T = gc(vs.num_cons / vs.width);
    pf->random_U.comm_W = gc(vs.total_vars() / vs.width);
    pf->random_U.comm_E = gc(vs.num_cons / vs.width);
    pf->random_U.u = gf();
    for (size_t i = 0; i < vs.num_io(); ++i) pf->random_U.X.push_back(gf());
    const size_t lx = log2_exact(vs.num_cons), ly = log2_exact(next_pow2(vs.total_vars())) + 1;
    for (size_t i = 0; i < lx; ++i) pf->relaxed.sc_outer.compressed_polys.push_back({gf(), gf(), gf()});
    for (int i = 0; i < 3; ++i) pf->relaxed.claims_outer[i] = gf();
    for (size_t i = 0; i < ly; ++i) pf->relaxed.sc_inner.compressed_polys.push_back({gf(), gf()});
    for (size_t i = 0; i < vs.width; ++i) pf->relaxed.v_W.push_back(gf());
    pf->relaxed.blind_W = gf();
    for (size_t i = 0; i < vs.width; ++i) pf->relaxed.v_E.push_back(gf());
    pf->relaxed.blind_E = gf();
    if (o != nwords) throw std::runtime_error("nn_proof_from_words: trailing words");
    return guard.release();
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}

// ---- wire formats (oracle/wire.hpp, oracle/wire_formats.hpp) ------------------------------------------------------------------------------
int orc_sha256(const uint8_t* data, size_t n, uint8_t* out32) {
  sha256(data, n, out32);
  return 0;
}
static long copy_out(const std::vector<uint8_t>& b, uint8_t* out, size_t cap) {
  if (out && b.size() <= cap) memcpy(out, b.data(), b.size());
  return (long)b.size();
}
// each returns the byte length (call with out = NULL to size the buffer), -1 on error
long orc_spartan_vk_to_bytes(void* pk, uint8_t* out, size_t cap) {
  try {
    return copy_out(spartan_vk_to_bytes(*(SpartanProverKey*)pk), out, cap);
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
long orc_spartan_proof_to_bytes(void* pf, uint8_t* out, size_t cap) {
  try {
    return copy_out(spartan_proof_to_bytes(*(SpartanProof*)pf), out, cap);
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
void* orc_spartan_proof_from_bytes(const uint8_t* b, size_t n) {
  try {
    return new SpartanProof(spartan_proof_from_bytes(b, n));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
long orc_nn_proof_to_bytes(void* pf, uint8_t* out, size_t cap) {
  try {
    return copy_out(nn_proof_to_bytes(*(NNProof*)pf), out, cap);
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
void* orc_nn_proof_from_bytes(const uint8_t* b, size_t n) {
  try {
    return new NNProof(nn_proof_from_bytes(b, n));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
}  // extern "C"
