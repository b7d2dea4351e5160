// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Restates src/r1cs/sparse.rs (CSR SparseMatrix :385-394, PrecomputedSparseMatrix :29-233,
// FilteredSpmv :305-380) and the SplitR1CSShape pieces of src/r1cs/mod.rs that sit on the path
// (new/padding :810-911, multiply_vec :1075-1107, multiply_vec_precommitted :1112-1128,
// multiply_vec_incremental_into :1170-1211, bind_and_prepare_poly_ABC :1235-1398,
// evaluate_with_tables_fast :1216-1226 by value).
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>

#include "field.hpp"

namespace oracle {

template <class F>
struct SparseMatrix {  // sparse.rs:385-394
  std::vector<F> data;
  std::vector<size_t> indices;
  std::vector<size_t> indptr;
  size_t cols = 0;
  size_t rows() const { return indptr.size() - 1; }
};

template <class F>
struct FilteredSpmv {  // sparse.rs:365-380
  std::vector<uint32_t> rows, cols;
  std::vector<F> vals;
  void multiply_vec_add(const std::vector<F>& v, std::vector<F>& out) const {
    for (size_t i = 0; i < rows.size(); ++i) out[rows[i]] = out[rows[i]] + vals[i] * v[cols[i]];
  }
};

template <class F>
struct PrecomputedSparseMatrix {  // sparse.rs:29-45
  size_t num_rows = 0, num_cols = 0;
  std::vector<uint32_t> off_unit_pos, off_unit_neg, off_small, off_general;
  std::vector<uint32_t> unit_pos_cols, unit_neg_cols, small_cols, general_cols;
  std::vector<int8_t> small_coeffs;
  std::vector<F> general_vals;

  static PrecomputedSparseMatrix from_sparse(const SparseMatrix<F>& m) {  // sparse.rs:49-134
    PrecomputedSparseMatrix p;
    p.num_rows = m.indptr.size() - 1;
    p.num_cols = m.cols;
    F one = F::one(), neg_one = F::one().neg();
    F small_pos[6], small_neg[6];
    for (int k = 0; k < 6; ++k) {
      small_pos[k] = F::from_u64(k + 2);
      small_neg[k] = small_pos[k].neg();
    }
    for (size_t r = 0; r < p.num_rows; ++r) {
      p.off_unit_pos.push_back((uint32_t)p.unit_pos_cols.size());
      p.off_unit_neg.push_back((uint32_t)p.unit_neg_cols.size());
      p.off_small.push_back((uint32_t)p.small_cols.size());
      p.off_general.push_back((uint32_t)p.general_cols.size());
      for (size_t k = m.indptr[r]; k < m.indptr[r + 1]; ++k) {
        const F& val = m.data[k];
        uint32_t col = (uint32_t)m.indices[k];
        if (val == one) {
          p.unit_pos_cols.push_back(col);
        } else if (val == neg_one) {
          p.unit_neg_cols.push_back(col);
        } else {
          int found = 0;
          for (int s = 0; s < 6 && !found; ++s) {
            if (val == small_pos[s]) {
              p.small_cols.push_back(col);
              p.small_coeffs.push_back((int8_t)(s + 2));
              found = 1;
            } else if (val == small_neg[s]) {
              p.small_cols.push_back(col);
              p.small_coeffs.push_back((int8_t)-(s + 2));
              found = 1;
            }
          }
          if (!found) {
            p.general_cols.push_back(col);
            p.general_vals.push_back(val);
          }
        }
      }
    }
    p.off_unit_pos.push_back((uint32_t)p.unit_pos_cols.size());
    p.off_unit_neg.push_back((uint32_t)p.unit_neg_cols.size());
    p.off_small.push_back((uint32_t)p.small_cols.size());
    p.off_general.push_back((uint32_t)p.general_cols.size());
    return p;
  }
  static F small_mul(int8_t coeff, const F& x) {  // sparse.rs:137-155
    int a = coeff < 0 ? -coeff : coeff;
    F r;
    switch (a) {
      case 2: r = x.dbl(); break;
      case 3: r = x.dbl() + x; break;
      case 4: r = x.dbl().dbl(); break;
      case 5: r = x.dbl().dbl() + x; break;
      case 6: { F d = x.dbl(); r = d.dbl() + d; break; }
      case 7: { F d = x.dbl(); r = d.dbl() + d + x; break; }
      default: throw std::runtime_error("small_mul");
    }
    return coeff < 0 ? r.neg() : r;
  }
  F compute_row_single(size_t row, const std::vector<F>& v) const {  // sparse.rs:194-218
    F sum = F::zero();
    for (uint32_t i = off_unit_pos[row]; i < off_unit_pos[row + 1]; ++i) sum = sum + v[unit_pos_cols[i]];
    for (uint32_t i = off_unit_neg[row]; i < off_unit_neg[row + 1]; ++i) sum = sum - v[unit_neg_cols[i]];
    for (uint32_t i = off_small[row]; i < off_small[row + 1]; ++i) sum = sum + small_mul(small_coeffs[i], v[small_cols[i]]);
    for (uint32_t i = off_general[row]; i < off_general[row + 1]; ++i) sum = sum + general_vals[i] * v[general_cols[i]];
    return sum;
  }
  std::vector<F> multiply_vec(const std::vector<F>& v) const {  // sparse.rs:221-233
    if (v.size() != num_cols) throw std::runtime_error("multiply_vec: invalid shape");
    std::vector<F> out(num_rows);
#pragma omp parallel for schedule(static) if (num_rows >= 4096)  // rows parallel above 4096 (sparse.rs:223)
    for (size_t r = 0; r < num_rows; ++r) out[r] = compute_row_single(r, v);
    return out;
  }
  FilteredSpmv<F> build_filtered(size_t col_min, size_t num_rows_used) const {  // sparse.rs:305-358
    FilteredSpmv<F> f;
    size_t nr = std::min(num_rows_used, num_rows);
    F one = F::one(), neg_one = one.neg();
    for (size_t row = 0; row < nr; ++row) {
      for (uint32_t i = off_unit_pos[row]; i < off_unit_pos[row + 1]; ++i)
        if (unit_pos_cols[i] >= col_min) { f.rows.push_back((uint32_t)row); f.cols.push_back(unit_pos_cols[i]); f.vals.push_back(one); }
      for (uint32_t i = off_unit_neg[row]; i < off_unit_neg[row + 1]; ++i)
        if (unit_neg_cols[i] >= col_min) { f.rows.push_back((uint32_t)row); f.cols.push_back(unit_neg_cols[i]); f.vals.push_back(neg_one); }
      for (uint32_t i = off_small[row]; i < off_small[row + 1]; ++i)
        if (small_cols[i] >= col_min) { f.rows.push_back((uint32_t)row); f.cols.push_back(small_cols[i]); f.vals.push_back(small_mul(small_coeffs[i], one)); }
      for (uint32_t i = off_general[row]; i < off_general[row + 1]; ++i)
        if (general_cols[i] >= col_min) { f.rows.push_back((uint32_t)row); f.cols.push_back(general_cols[i]); f.vals.push_back(general_vals[i]); }
    }
    return f;
  }
};

static const size_t DEFAULT_COMMITMENT_WIDTH = 2048;  // src/lib.rs:63

inline size_t pad_to_width(size_t width, size_t n) { return ((n + width - 1) / width) * width; }
inline size_t next_pow2(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

template <class F>
struct SplitR1CSShape {  // src/r1cs/mod.rs:743-773
  size_t num_cons, num_cons_unpadded, num_shared_unpadded, num_precommitted_unpadded, num_rest_unpadded;
  size_t num_shared, num_precommitted, num_rest, num_public, num_challenges;
  SparseMatrix<F> A, B, C;
  PrecomputedSparseMatrix<F> pa, pb, pc;
  FilteredSpmv<F> fa, fb, fc;

  size_t num_vars() const { return num_shared + num_precommitted + num_rest; }
  size_t num_extra() const { return 1 + num_public + num_challenges; }

  // src/r1cs/mod.rs:810-911 (column remap + row padding)
  static SplitR1CSShape make(size_t num_cons, size_t num_shared, size_t num_precommitted, size_t num_rest, size_t num_public,
                             size_t num_challenges, SparseMatrix<F> A, SparseMatrix<F> B, SparseMatrix<F> C) {
    size_t width = DEFAULT_COMMITMENT_WIDTH;
    size_t sp = pad_to_width(width, num_shared), pp = pad_to_width(width, num_precommitted), rp = pad_to_width(width, num_rest);
    size_t nvp = sp + pp + rp;
    if (nvp < num_public + num_challenges + 1) rp = std::max(num_public + num_challenges + 1, nvp) - (sp + pp);
    nvp = sp + pp + rp;
    if (next_pow2(nvp) != nvp) rp = next_pow2(nvp) - (sp + pp);
    size_t num_vars = num_shared + num_precommitted + num_rest;
    nvp = sp + pp + rp;
    size_t ncp = next_pow2(num_cons);
    auto apply_pad = [&](SparseMatrix<F>& M) {
      for (size_t& c : M.indices) {
        if (c >= num_shared && c < num_shared + num_precommitted) c += sp - num_shared;
        else if (c >= num_shared + num_precommitted && c < num_vars) c += sp + pp - num_shared - num_precommitted;
        else if (c >= num_vars) c += nvp - num_vars;
      }
      M.cols += nvp - num_vars;
      size_t nnz = M.indptr.empty() ? 0 : M.indptr.back();
      M.indptr.resize(M.indptr.size() + (ncp - num_cons), nnz);
    };
    apply_pad(A);
    apply_pad(B);
    apply_pad(C);
    SplitR1CSShape S;
    S.num_cons = ncp;
    S.num_shared = sp;
    S.num_precommitted = pp;
    S.num_rest = rp;
    S.num_cons_unpadded = num_cons;
    S.num_shared_unpadded = num_shared;
    S.num_precommitted_unpadded = num_precommitted;
    S.num_rest_unpadded = num_rest;
    S.num_public = num_public;
    S.num_challenges = num_challenges;
    S.A = std::move(A);
    S.B = std::move(B);
    S.C = std::move(C);
    S.precompute();
    return S;
  }
  // SplitR1CSShape::equalize (src/r1cs/mod.rs:913-971): both shapes get the larger number of constraints (row pointers extended) and the larger number
  // of variables (the growth goes to num_rest; columns of the constant, the public values and the challenges move up by it)
  static void equalize(SplitR1CSShape& SA, SplitR1CSShape& SB) {
    const size_t cons = std::max(SA.num_cons, SB.num_cons), vars = std::max(SA.num_vars(), SB.num_vars());
    auto grow = [&](SplitR1CSShape& S) {
      const size_t orig_cons = S.num_cons, nv = S.num_vars();
      S.num_cons = cons;
      if (nv != vars) S.num_rest = vars - (S.num_shared + S.num_precommitted);
      for (SparseMatrix<F>* M : {&S.A, &S.B, &S.C}) {
        for (size_t& c : M->indices)
          if (c >= nv) c += vars - nv;
        M->cols += vars - nv;
        const size_t nnz = M->indptr.empty() ? 0 : M->indptr.back();
        M->indptr.resize(M->indptr.size() + (cons - orig_cons), nnz);
      }
      S.precompute();  // (the reference's precomputed forms are built lazily, after this point)
    };
    grow(SA);
    grow(SB);
  }
  void precompute() {  // src/r1cs/mod.rs:1059-1073
    pa = PrecomputedSparseMatrix<F>::from_sparse(A);
    pb = PrecomputedSparseMatrix<F>::from_sparse(B);
    pc = PrecomputedSparseMatrix<F>::from_sparse(C);
    size_t col_min = num_shared + num_precommitted;
    fa = pa.build_filtered(col_min, num_cons_unpadded);
    fb = pb.build_filtered(col_min, num_cons_unpadded);
    fc = pc.build_filtered(col_min, num_cons_unpadded);
  }
  void multiply_vec(const std::vector<F>& z, std::vector<F>* az, std::vector<F>* bz, std::vector<F>* cz) const {  // :1075-1107
    if (z.size() != num_vars() + num_extra()) throw std::runtime_error("InvalidWitnessLength");
    *az = pa.multiply_vec(z);
    *bz = pb.multiply_vec(z);
    *cz = pc.multiply_vec(z);
  }
  void multiply_vec_precommitted(const std::vector<F>& z_cached, std::vector<F>* az, std::vector<F>* bz, std::vector<F>* cz) const {  // :1112-1128
    size_t cached_len = num_shared + num_precommitted;
    if (z_cached.size() != cached_len) throw std::runtime_error("multiply_vec_precommitted: length");
    std::vector<F> z(num_vars() + num_extra(), F::zero());
    std::copy(z_cached.begin(), z_cached.end(), z.begin());
    multiply_vec(z, az, bz, cz);
  }
  void multiply_vec_incremental_into(const std::vector<F>& z, const std::vector<F>& caz, const std::vector<F>& cbz, const std::vector<F>& ccz,
                                     std::vector<F>* az, std::vector<F>* bz, std::vector<F>* cz) const {  // :1170-1211
    *az = caz;
    *bz = cbz;
    *cz = ccz;
    fa.multiply_vec_add(z, *az);
    fb.multiply_vec_add(z, *bz);
    fc.multiply_vec_add(z, *cz);
  }
  // accumulate_rows (:1324-1398) for one matrix and one row-coefficient
  static void accumulate_matrix(const PrecomputedSparseMatrix<F>& p, size_t row, const F& coef, std::vector<F>& out) {
    for (uint32_t i = p.off_unit_pos[row]; i < p.off_unit_pos[row + 1]; ++i) out[p.unit_pos_cols[i]] = out[p.unit_pos_cols[i]] + coef;
    for (uint32_t i = p.off_unit_neg[row]; i < p.off_unit_neg[row + 1]; ++i) out[p.unit_neg_cols[i]] = out[p.unit_neg_cols[i]] - coef;
    for (uint32_t i = p.off_small[row]; i < p.off_small[row + 1]; ++i)
      out[p.small_cols[i]] = out[p.small_cols[i]] + PrecomputedSparseMatrix<F>::small_mul(p.small_coeffs[i], coef);
    for (uint32_t i = p.off_general[row]; i < p.off_general[row + 1]; ++i) out[p.general_cols[i]] = out[p.general_cols[i]] + p.general_vals[i] * coef;
  }
  std::vector<F> bind_and_prepare_poly_ABC_inner(const std::vector<F>& rx, const F& r, size_t out_len) const {  // :1271-1321
    if (rx.size() != num_cons) throw std::runtime_error("poly_ABC: rx length");
    F r2 = r * r;
    std::vector<F> out(out_len, F::zero());
#ifdef _OPENMP
    const int T = num_cons_unpadded >= 8192 ? std::min(omp_get_max_threads(), 16) : 1;  // full-length accumulator per thread: keep it bounded
    if (T > 1) {  // per-thread full-length accumulators, then a vector reduce — the reference's structure (:1299-1319)
      std::vector<std::vector<F>> part(T, std::vector<F>(out_len, F::zero()));
#pragma omp parallel num_threads(T)
      {
        std::vector<F>& mine = part[omp_get_thread_num()];
#pragma omp for schedule(static)
        for (size_t row = 0; row < num_cons_unpadded; ++row) {
          F rx_row = rx[row];
          accumulate_matrix(pa, row, rx_row, mine);
          accumulate_matrix(pb, row, rx_row * r, mine);
          accumulate_matrix(pc, row, rx_row * r2, mine);
        }
      }
#pragma omp parallel for schedule(static)
      for (size_t j = 0; j < out_len; ++j) {
        F a = F::zero();
        for (int t = 0; t < T; ++t) a = a + part[t][j];
        out[j] = a;
      }
      return out;
    }
#endif
    for (size_t row = 0; row < num_cons_unpadded; ++row) {
      F rx_row = rx[row];
      accumulate_matrix(pa, row, rx_row, out);
      accumulate_matrix(pb, row, rx_row * r, out);
      accumulate_matrix(pc, row, rx_row * r2, out);
    }
    return out;
  }
  std::vector<F> bind_and_prepare_poly_ABC(const std::vector<F>& rx, const F& r) const {  // :1235-1244
    return bind_and_prepare_poly_ABC_inner(rx, r, num_vars() + num_extra());
  }
  // evaluate_with_tables_fast (:1216-1226), by value: M(rx,ry) = sum_{(i,j)} M[i,j] T_x[i] T_y[j]
  void evaluate_with_tables(const std::vector<F>& T_x, const std::vector<F>& T_y, F* ea, F* eb, F* ec) const {
    auto ev = [&](const SparseMatrix<F>& M) {
      F acc = F::zero();
      for (size_t row = 0; row + 1 < M.indptr.size(); ++row) {
        F inner = F::zero();
        for (size_t k = M.indptr[row]; k < M.indptr[row + 1]; ++k) inner = inner + M.data[k] * T_y[M.indices[k]];
        acc = acc + inner * T_x[row];
      }
      return acc;
    };
    *ea = ev(A);
    *eb = ev(B);
    *ec = ev(C);
  }
};

}  // namespace oracle
