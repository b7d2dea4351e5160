// Host side ABOVE the C ABI: NeutronNovaNIFS::prove (src/neutronnova_zk.rs:511-1273) restated in C++ over include/spartan_hip.h.
// What stays with the caller, exactly as in the reference: the per-round `process_round` of the ZK verifier circuit
// (SatisfyingAssignment::process_round, called from `finish_round!` :723-727 and once more after the rounds :1207-1210) — a
// verifier-circuit witness commit + transcript step, not data-parallel work. It enters as a callback
// `hook(user, t, coeffs[4 F], r_b out)`; for t == ell_b the coefficients are {T_out, eq_rho_at_rb, 0, 0} and r_b is ignored.
// Everything that scales with the instances runs on the device: the matrix-vector products into the NIFS layers, the rounds, the witness
// fold (fold_multiple), the commitment fold (fold_commitments[_partial]).
#include "neutronnova_nifs.hpp"

#include "comm.hpp"

namespace spartan2 {

void compute_tensor_decomp(size_t n, size_t* ell, size_t* left, size_t* right) {  // src/neutronnova_zk.rs:56-67
  size_t l = 0;
  while ((size_t(1) << l) < n) ++l;
  *ell = l;
  *left = size_t(1) << ((l + 1) / 2);
  *right = size_t(1) << (l / 2);
}

// R1CSInstance transcript bytes = comm_W || X (src/r1cs/mod.rs:728-736)
static std::vector<uint8_t> instance_bytes(const aff_t* comm, size_t rows, const fe_t* X, size_t d) {
  std::vector<uint8_t> b = commitment_bytes(comm, rows);
  const size_t off = b.size();
  b.resize(off + 32 * d);
  for (size_t j = 0; j < d; ++j) sp::fe_to_be_bytes<S>(X[j], b.data() + off + 32 * j);
  return b;
}
// sum_i w[i] * M[i][.] for a host matrix of n rows x cols (fold_blinds, the X fold: hyrax_pc.rs:795-818, neutronnova_zk.rs:1236-1245): on the device
// (bind_with_delayed's kernel) once the matrix is big enough to pay for its upload
static void fold_rows(sp_ctx* ctx, const fe_t* M, size_t n, size_t cols, const fe_t* w, fe_t* out) {
  if (cols == 0) return;
  if (n * cols < 16384) {
    for (size_t j = 0; j < cols; ++j) out[j] = fe_zero();
    for (size_t i = 0; i < n; ++i)
      for (size_t j = 0; j < cols; ++j) out[j] = fe_add<S>(out[j], fe_mul<S>(M[i * cols + j], w[i]));
    return;
  }
  sp_table* t = nullptr;
  ck(sp_table_from_host(ctx, u64p(M), n * cols, (size_t)-1, (size_t)-1, &t), "upload");
  int rc = sp_rowmat_vec(ctx, t, n, cols, u64p(w), u64p(out));
  sp_table_free(t);
  ck(rc, "fold rows");
}
// Us: comm rows (n x rows affine) + X (n x d); Ws: resident witness tables (num_vars each) + blinds (n x rows)
// Layers Az_b, Bz_b, Cz_b of every (padded) instance and, for small_values, their i64 mirrors: the transcript-independent part that the reference
// caches in prep_prove (cached_step_matvec / cached_step_i64, src/neutronnova_zk.rs:1520-1600).
sp_nifs* nifs_prepare(sp_ctx* ctx, const sp_shape* shape, const sp_dims& dims, size_t n, const fe_t* X, const sp_table* const* Ws, bool small_values) {
  if (n == 0) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "NIFS prepare: no instances");
  const size_t d = dims.num_public, num_vars = dims.num_shared + dims.num_precommitted + dims.num_rest;
  size_t n_padded = 2;
  while (n_padded < n) n_padded <<= 1;
  auto inst = [&](size_t i) { return i < n ? i : 0; };
  size_t ell_cons, left, right;
  compute_tensor_decomp(dims.num_cons, &ell_cons, &left, &right);
  sp_nifs* nifs = nullptr;
  ck(sp_nifs_create(ctx, n_padded, left, right, &nifs), "nifs_create");
  sp_table* z = nullptr;
  try {
    ck(sp_table_zeros(ctx, num_vars + 1 + d, (size_t)-1, (size_t)-1, &z), "z alloc");
    const fe_t one = fe_one<S>();
    for (size_t i = 0; i < n_padded; ++i) {  // z = [W | 1 | X]; (Az, Bz, Cz) straight into layer i (:583-596)
      ck(sp_table_copy(ctx, z, 0, Ws[inst(i)], 0, num_vars), "z <- W");
      std::vector<fe_t> tail(1 + d);
      tail[0] = one;
      for (size_t j = 0; j < d; ++j) tail[1 + j] = X[inst(i) * d + j];
      ck(sp_table_write(ctx, z, num_vars, u64p(tail.data()), 1 + d), "z tail");
      sp_table* v[3];
      for (int q = 0; q < 3; ++q) ck(sp_nifs_layer(nifs, q, i, &v[q]), "nifs_layer");
      int rc = sp_multiply_vec(ctx, shape, z, v[0], v[1], v[2]);
      for (int q = 0; q < 3; ++q) sp_table_free(v[q]);
      ck(rc, "multiply_vec");
    }
    if (small_values) ck(sp_nifs_prepare_small(nifs), "prepare_small");
  } catch (...) {
    sp_table_free(z);
    sp_nifs_free(nifs);
    throw;
  }
  sp_table_free(z);
  return nifs;
}

void nifs_prove(sp_ctx* ctx, const sp_shape* shape, const sp_dims& dims, const sp_ck* ckey, size_t n, size_t rows, const aff_t* comms, const fe_t* X,
                       const sp_table* const* Ws, const fe_t* r_W, bool small_values, sp_nifs* prepared, sp_transcript* tr, nn_round_hook hook, void* user,
                       NifsOutputs& out) {
  if (n == 0) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "NIFS prove: no instances");
  static const bool trace = [] {
    const char* e = getenv("SPARTAN_HOST_LAPS");
    return e && e[0] == '1';
  }();
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_lap = now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    sp_ctx_synchronize(ctx);
    const double t = now();
    fprintf(stderr, "nifs lap %-28s %8.3f ms\n", what, t - t_lap);
    t_lap = t;
  };
  const size_t d = dims.num_public, num_vars = dims.num_shared + dims.num_precommitted + dims.num_rest;
  size_t n_padded = 2;
  while (n_padded < n) n_padded <<= 1;
  size_t ell_b = 0;
  while ((size_t(1) << ell_b) < n_padded) ++ell_b;
  auto inst = [&](size_t i) { return i < n ? i : 0; };  // padding clones instance 0 (:549-552)

  auto absorb = [&](const char* label, const uint8_t* b, size_t len) { ck(sp_transcript_absorb(tr, (const uint8_t*)label, strlen(label), b, len), "absorb"); };
  auto squeeze = [&](const char* label) {
    fe_t f;
    ck(sp_transcript_squeeze(tr, (const uint8_t*)label, strlen(label), u64p(&f)), "squeeze");
    return f;
  };
  {  // transcript.absorb(b"U", U) (:553-555; R1CSInstance bytes = comm_W || X, src/r1cs/mod.rs:728-736): encodings in parallel, the sponge in order
    std::vector<std::vector<uint8_t>> ub(n);
    parallel_for(n, rows + d, [&](size_t i) { ub[i] = instance_bytes(comms + i * rows, rows, X + i * d, d); });
    for (size_t i = 0; i < n_padded; ++i) absorb("U", ub[inst(i)].data(), ub[inst(i)].size());
  }
  {
    uint8_t zero_be[32] = {0};  // T = 0 (:556-557)
    absorb("T", zero_be, 32);
  }
  size_t ell_cons, left, right;
  compute_tensor_decomp(dims.num_cons, &ell_cons, &left, &right);
  const fe_t tau = squeeze("tau");
  ck(sp_pow_split_evals(u64p(&tau), ell_cons, left, right, out.E_eq), "split_evals");
  std::vector<fe_t> rhos(ell_b);
  for (auto& r : rhos) r = squeeze("rho");

  lap("transcript preamble");
  sp_nifs* nifs = prepared;
  struct Guard {
    sp_nifs* n;
    ~Guard() { sp_nifs_free(n); }
  } guard{prepared ? nullptr : (nifs = nifs_prepare(ctx, shape, dims, n, X, Ws, small_values))};
  lap("layers");
  ck(sp_nifs_begin(nifs, out.E_eq, u64p(rhos.data()), ell_b, small_values ? 2 : 0), "nifs_begin");
  lap("begin (mirrors, c_vals)");
  std::vector<fe_t> r_bs(ell_b);
  for (size_t t = 0; t < ell_b; ++t) {
    uint64_t* co = out.polys + 16 * t;
    ck(sp_nifs_round(nifs, t, co), "nifs_round");
    hook(user, t, co, u64p(&r_bs[t]));
    ck(sp_nifs_challenge(nifs, u64p(&r_bs[t])), "nifs_challenge");
  }
  lap("rounds");
  memcpy(out.r_bs, r_bs.data(), ell_b * sizeof(fe_t));
  ck(sp_nifs_finish(nifs, out.A, out.B, out.C, out.tail, out.tail + 4), "nifs_finish");
  {
    uint64_t fin[16] = {0}, ignored[4];
    memcpy(fin, out.tail, 64);
    hook(user, ell_b, fin, ignored);
  }
  lap("finish");
  // fold_witnesses (:1212-1231): truncated to shared + precommitted when that prefix is non-empty, rest re-zeroed
  const size_t effective_len = dims.num_shared + dims.num_precommitted;
  const bool truncated = effective_len > 0;
  const size_t dim = truncated ? effective_len : num_vars;
  std::vector<fe_t> w(n_padded);
  ck(sp_weights_from_r(u64p(r_bs.data()), ell_b, n_padded, u64p(w.data())), "weights_from_r");
  std::vector<const sp_table*> wt(n_padded);
  for (size_t i = 0; i < n_padded; ++i) wt[i] = Ws[inst(i)];
  ck(sp_fold_tables(ctx, wt.data(), n_padded, u64p(w.data()), dim, out.folded_W), "fold_multiple");
  if (dim < num_vars) ck(sp_table_zero(ctx, out.folded_W, dim, num_vars - dim), "zero rest");
  ck(sp_table_set_len(out.folded_W, num_vars, (size_t)-1, (size_t)-1), "set_len");
  lap("fold_multiple");
  std::vector<fe_t> f_rW(rows, fe_zero()), f_X(d, fe_zero());  // fold_blinds (hyrax_pc.rs:795-818), X fold (:1236-1245)
  {
    std::vector<fe_t> wi(n, fe_zero());  // padding clones instance 0: its weights add up
    for (size_t i = 0; i < n_padded; ++i) wi[inst(i)] = fe_add<S>(wi[inst(i)], w[i]);
    fold_rows(ctx, r_W, n, rows, wi.data(), f_rW.data());
    fold_rows(ctx, X, n, d, wi.data(), f_X.data());
  }
  memcpy(out.folded_rW, f_rW.data(), rows * sizeof(fe_t));
  memcpy(out.folded_X, f_X.data(), d * sizeof(fe_t));
  // fold_commitments_partial / fold_commitments (:1247-1261, hyrax_pc.rs:737-793, 820-874)
  size_t data_rows = truncated ? (effective_len + DEFAULT_COMMITMENT_WIDTH - 1) / DEFAULT_COMMITMENT_WIDTH : rows;
  if (data_rows > rows) data_rows = rows;
  std::vector<aff_t> bases(data_rows * n_padded);
  for (size_t r = 0; r < data_rows; ++r)
    for (size_t i = 0; i < n_padded; ++i) bases[r * n_padded + i] = comms[inst(i) * rows + r];
  auto fold = [w = std::move(w), bases = std::move(bases), rest = std::vector<fe_t>(f_rW.begin() + data_rows, f_rW.end()), n_padded, data_rows, rows, ckey,
               dst = out.folded_comm](sp_ctx* on) {
    if (data_rows) ck(sp_msm_shared_weights(on, u64p(w.data()), n_padded, (const uint64_t*)bases.data(), data_rows, dst), "fold_commitments");
    if (data_rows < rows)  // rest rows: folded_blind[row] * h
      ck(sp_fixed_base_mul_h(on, ckey, u64p(rest.data()), rows - data_rows, dst + 8 * data_rows), "rest rows");
  };
  if (out.deferred_fold_commitments) {
    *out.deferred_fold_commitments = std::move(fold);
    lap("fold_commitments (deferred)");
    return;
  }
  fold(ctx);
  lap("fold_commitments");
}


// NeutronNovaNIFS::prove with the 2^ell_b instances sharded over the ranks of `comm` (BASELINE config 5): rank g holds the
// n_local = n / world consecutive instances [g n_local, (g + 1) n_local) - their witness tables, commitments, publics, blinds - and builds only
// their (Az, Bz, Cz) layers. Exchanges: the instances' commitments / publics / blinds once (the transcript absorbs every U), c_vals once, TWO field
// elements per round for the first log2(n_local) rounds (all-gather + modular adds in rank order: every rank holds identical totals and runs the
// O(1) finish and the deterministic hook redundantly, no broadcast), then ONE bulk exchange - each rank's single remaining A / B layer, gathered
// by every rank (device to device over the collective) - after which all ranks run the last log2(world) rounds on identical data, and the
// partial C / witness folds (one layer / one witness per rank) summed on every rank. Results are identical on all ranks and bit-identical to
// nifs_prove on the whole batch. n must be a power of two and n_local >= 2.
void nifs_prove_sharded(sp_ctx* ctx, Comm& comm, const sp_shape* shape, const sp_dims& dims, const sp_ck* ckey, size_t n_local, size_t rows, const aff_t* comms_local,
                        const fe_t* X_local, const sp_table* const* Ws_local, const fe_t* r_W_local, bool small_values, sp_nifs* prepared, sp_transcript* tr,
                        nn_round_hook hook, void* user, NifsOutputs& out) {
  const size_t world = (size_t)comm.world, rank = (size_t)comm.rank, n = n_local * world;
  if (n_local < 2 || (n_local & (n_local - 1)) || (world & (world - 1)))
    throw Error(SP_ERR_INVALID_INPUT_LENGTH, "sharded NIFS: instances per rank and ranks must be powers of two, at least two instances per rank");
  static const bool trace = [] {
    const char* e = getenv("SPARTAN_HOST_LAPS");
    return e && e[0] == '1';
  }();
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_lap = now();
  auto lap = [&](const char* what) {
    if (!trace) return;
    sp_ctx_synchronize(ctx);
    const double t = now();
    fprintf(stderr, "nifs(sharded) lap %-28s %8.3f ms\n", what, t - t_lap);
    t_lap = t;
  };
  const size_t d = dims.num_public, num_vars = dims.num_shared + dims.num_precommitted + dims.num_rest;
  size_t ell_b = 0, local_rounds = 0;
  while ((size_t(1) << ell_b) < n) ++ell_b;
  while ((size_t(1) << local_rounds) < n_local) ++local_rounds;

  // every instance's U = (comm_W, X) and blinds on every rank (64 rows + 32 d + 32 rows bytes per instance)
  const size_t rec = rows * sizeof(aff_t) + d * sizeof(fe_t) + rows * sizeof(fe_t);
  std::vector<aff_t> comms_all;
  std::vector<fe_t> X_all, r_W_all;
  const aff_t* comms = comms_local;  // one rank: its own instances are all there are, nothing to gather (12.6 MB through the collective and three
  const fe_t* X = X_local;           // copies took 4-5 ms of a config-5 NIFS at world 1)
  const fe_t* r_W = r_W_local;
  if (world > 1) {
    std::vector<uint8_t> mine(n_local * rec), all(n * rec);
    for (size_t i = 0; i < n_local; ++i) {
      uint8_t* p = mine.data() + i * rec;
      memcpy(p, comms_local + i * rows, rows * sizeof(aff_t));
      if (d) memcpy(p + rows * sizeof(aff_t), X_local + i * d, d * sizeof(fe_t));
      memcpy(p + rows * sizeof(aff_t) + d * sizeof(fe_t), r_W_local + i * rows, rows * sizeof(fe_t));
    }
    comm.allgather(mine.data(), mine.size(), all.data());
    comms_all.resize(n * rows);
    X_all.resize(n * d);
    r_W_all.resize(n * rows);
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* p = all.data() + i * rec;
      memcpy(&comms_all[i * rows], p, rows * sizeof(aff_t));
      if (d) memcpy(&X_all[i * d], p + rows * sizeof(aff_t), d * sizeof(fe_t));
      memcpy(&r_W_all[i * rows], p + rows * sizeof(aff_t) + d * sizeof(fe_t), rows * sizeof(fe_t));
    }
    comms = comms_all.data();
    X = X_all.data();
    r_W = r_W_all.data();
  }
  lap("gather instance data");

  auto absorb = [&](const char* label, const uint8_t* b, size_t len) { ck(sp_transcript_absorb(tr, (const uint8_t*)label, strlen(label), b, len), "absorb"); };
  auto squeeze = [&](const char* label) {
    fe_t f;
    ck(sp_transcript_squeeze(tr, (const uint8_t*)label, strlen(label), u64p(&f)), "squeeze");
    return f;
  };
  {  // transcript.absorb(b"U", U) (:553-555)
    std::vector<std::vector<uint8_t>> ub(n);
    parallel_for(n, rows + d, [&](size_t i) { ub[i] = instance_bytes(comms + i * rows, rows, X + i * d, d); });
    for (size_t i = 0; i < n; ++i) absorb("U", ub[i].data(), ub[i].size());
  }
  {
    uint8_t zero_be[32] = {0};  // T = 0 (:556-557)
    absorb("T", zero_be, 32);
  }
  size_t ell_cons, left, right;
  compute_tensor_decomp(dims.num_cons, &ell_cons, &left, &right);
  const size_t total = left * right;
  const fe_t tau = squeeze("tau");
  ck(sp_pow_split_evals(u64p(&tau), ell_cons, left, right, out.E_eq), "split_evals");
  std::vector<fe_t> rhos(ell_b);
  for (auto& r : rhos) r = squeeze("rho");
  lap("transcript preamble");

  struct Objs {
    sp_nifs *loc = nullptr, *root = nullptr;
    std::vector<sp_table*> t;
    ~Objs() {
      for (sp_table* x : t) sp_table_free(x);
      sp_nifs_free(loc);
      sp_nifs_free(root);
    }
  } o;
  auto keep = [&](sp_table* t) {
    o.t.push_back(t);
    return t;
  };
  auto dev = [&](const sp_table* t) {
    void* p = nullptr;
    ck(sp_table_device_ptr(t, &p, nullptr), "device_ptr");
    return p;
  };
  // prepared: the rank's layers (+ mirrors) built at prep time (cached_step_matvec, :1520-1590); consumed by the rounds, owned by the caller
  sp_nifs* const loc = prepared ? prepared : (o.loc = nifs_prepare(ctx, shape, dims, n_local, X_local, Ws_local, small_values));
  ck(sp_nifs_begin_shard(loc, out.E_eq, u64p(rhos.data()), ell_b, rank * n_local, small_values ? 2 : 0), "nifs_begin_shard");
  std::vector<fe_t> cv_loc(n_local), cv(n);
  ck(sp_nifs_cvals(loc, u64p(cv_loc.data())), "nifs_cvals");
  comm.allgather(cv_loc.data(), n_local * sizeof(fe_t), cv.data());
  ck(sp_nifs_set_cvals(loc, u64p(cv.data()), n), "nifs_set_cvals");
  lap("layers, begin, c_vals");

  std::vector<fe_t> r_bs(ell_b);
  for (size_t t = 0; t < local_rounds; ++t) {  // the data-parallel rounds: own pairs, two field elements exchanged
    fe_t sums[2];
    ck(sp_nifs_round_sums(loc, t, u64p(sums)), "nifs_round_sums");
    comm.field_sum(sums, 2);
    uint64_t* co = out.polys + 16 * t;
    ck(sp_nifs_round_finish(loc, t, u64p(sums), co), "nifs_round_finish");
    hook(user, t, co, u64p(&r_bs[t]));
    ck(sp_nifs_challenge(loc, u64p(&r_bs[t])), "nifs_challenge");
  }
  sp_nifs* fin = loc;
  if (world > 1) {
    // hand-off: apply the pending fold; every rank gathers every rank's remaining A / B layer and continues on identical data
    ck(sp_nifs_fold_pending(loc), "nifs_fold_pending");
    fe_t T_cur, acc_eq;
    ck(sp_nifs_state(loc, u64p(&T_cur), u64p(&acc_eq)), "nifs_state");
    ck(sp_nifs_create(ctx, world, left, right, &o.root), "nifs_create");
    ck(sp_ctx_synchronize(ctx), "synchronize");
    for (int which = 0; which < 2; ++which) {
      sp_table *src = nullptr, *dst = nullptr;
      ck(sp_nifs_current_layer(loc, which, 0, &src), "nifs_current_layer");
      keep(src);
      ck(sp_nifs_layer(o.root, which, 0, &dst), "nifs_layer");  // the root object's layers are contiguous: layer b at offset b * total
      keep(dst);
      comm.allgather_device(dev(src), total * sizeof(fe_t), dev(dst));
    }
    ck(sp_nifs_resume(o.root, out.E_eq, u64p(rhos.data()), ell_b, local_rounds, u64p(r_bs.data()), u64p(&T_cur), u64p(&acc_eq), u64p(cv.data())), "nifs_resume");
    for (size_t t = local_rounds; t < ell_b; ++t) {
      uint64_t* co = out.polys + 16 * t;
      ck(sp_nifs_round(o.root, t, co), "nifs_round");
      hook(user, t, co, u64p(&r_bs[t]));
      ck(sp_nifs_challenge(o.root, u64p(&r_bs[t])), "nifs_challenge");
    }
    fin = o.root;
  }
  memcpy(out.r_bs, r_bs.data(), ell_b * sizeof(fe_t));
  lap("rounds");
  std::vector<fe_t> w(n);
  ck(sp_weights_from_r(u64p(r_bs.data()), ell_b, n, u64p(w.data())), "weights_from_r");
  // fold_blinds, X fold, fold_commitments on the gathered instance data (O(n rows) work, done redundantly on every rank) need the weights alone: they run on a
  // helper thread and the context's AUXILIARY stream (sp_msm_shared_weights_aux) beside this thread's layer and witness folds on the main stream - a bucket
  // MSM per commitment row and its host-side Horner under 3 ms of bandwidth-bound fold kernels (round 6; the two used to run one after the other)
  const size_t effective_len = dims.num_shared + dims.num_precommitted;
  const bool truncated = effective_len > 0;
  struct CommFold {
    std::thread th;
    std::exception_ptr err;
    ~CommFold() {
      if (th.joinable()) th.join();
    }
  } cf;
  {
    const fe_t* wp = w.data();
    NifsOutputs* op = &out;
    cf.th = std::thread([&cf, ctx, wp, n, rows, comms, truncated, effective_len, op] {
      try {
        ck(sp_ctx_bind_thread(ctx), "helper thread: device");
        size_t data_rows = truncated ? (effective_len + DEFAULT_COMMITMENT_WIDTH - 1) / DEFAULT_COMMITMENT_WIDTH : rows;
        if (data_rows > rows) data_rows = rows;
        std::vector<aff_t> bases(data_rows * n);
        for (size_t r = 0; r < data_rows; ++r)
          for (size_t i = 0; i < n; ++i) bases[r * n + i] = comms[i * rows + r];
        if (data_rows) ck(sp_msm_shared_weights_aux(ctx, u64p(wp), n, (const uint64_t*)bases.data(), data_rows, op->folded_comm), "fold_commitments");
      } catch (...) {
        cf.err = std::current_exception();
      }
    });
  }
  // (the two small row folds use the main stream's bind_with_delayed kernel and workspaces: they stay on this thread)
  std::vector<fe_t> f_rW(rows, fe_zero()), f_X(d, fe_zero());
  fold_rows(ctx, r_W, n, rows, w.data(), f_rW.data());
  fold_rows(ctx, X, n, d, w.data(), f_X.data());
  memcpy(out.folded_rW, f_rW.data(), rows * sizeof(fe_t));
  memcpy(out.folded_X, f_X.data(), d * sizeof(fe_t));
  std::vector<fe_t> ones(world, fe_one<S>());
  // sum over ranks of one device vector per rank: gather into `buf` (world x len), fold the windows with unit weights
  auto sum_over_ranks = [&](const sp_table* part, size_t len, sp_table* dst) {
    sp_table* buf = nullptr;
    ck(sp_table_zeros(ctx, world * len, (size_t)-1, (size_t)-1, &buf), "alloc");
    keep(buf);
    ck(sp_ctx_synchronize(ctx), "synchronize");
    comm.allgather_device(dev(part), len * sizeof(fe_t), dev(buf));
    std::vector<const sp_table*> views(world);
    for (size_t g = 0; g < world; ++g) {
      sp_table* v = nullptr;
      ck(sp_table_view(buf, g * len, len, &v), "view");
      views[g] = keep(v);
    }
    ck(sp_fold_tables(ctx, views.data(), world, u64p(ones.data()), len, dst), "sum over ranks");
  };
  if (world == 1) {
    ck(sp_nifs_finish(fin, out.A, out.B, out.C, out.tail, out.tail + 4), "nifs_finish");
  } else {
    ck(sp_nifs_finish(fin, out.A, out.B, nullptr, out.tail, out.tail + 4), "nifs_finish");
    // Cz = sum_b w_b Cz_b (:1168-1203): the C layers never move; each rank folds its own with its slice of the weights
    std::vector<const sp_table*> cl(n_local);
    for (size_t i = 0; i < n_local; ++i) {
      sp_table* v = nullptr;
      ck(sp_nifs_layer(loc, 2, i, &v), "nifs_layer");
      cl[i] = keep(v);
    }
    sp_table* part = nullptr;
    ck(sp_table_zeros(ctx, total, (size_t)-1, (size_t)-1, &part), "alloc");
    keep(part);
    ck(sp_fold_tables(ctx, cl.data(), n_local, u64p(w.data() + rank * n_local), total, part), "fold C");
    sum_over_ranks(part, total, out.C);
  }
  {
    uint64_t finw[16] = {0}, ignored[4];
    memcpy(finw, out.tail, 64);
    hook(user, ell_b, finw, ignored);
  }
  // fold_witnesses (:1212-1231)
  const size_t dim = truncated ? effective_len : num_vars;
  if (world == 1) {
    ck(sp_fold_tables(ctx, Ws_local, n_local, u64p(w.data()), dim, out.folded_W), "fold_multiple");
  } else {
    sp_table* part = nullptr;
    ck(sp_table_zeros(ctx, dim, (size_t)-1, (size_t)-1, &part), "alloc");
    keep(part);
    ck(sp_fold_tables(ctx, Ws_local, n_local, u64p(w.data() + rank * n_local), dim, part), "fold_multiple");
    sum_over_ranks(part, dim, out.folded_W);
  }
  if (dim < num_vars) ck(sp_table_zero(ctx, out.folded_W, dim, num_vars - dim), "zero rest");
  ck(sp_table_set_len(out.folded_W, num_vars, (size_t)-1, (size_t)-1), "set_len");
  lap("finish, C and witness folds");
  cf.th.join();
  if (cf.err) std::rethrow_exception(cf.err);
  {
    size_t data_rows = truncated ? (effective_len + DEFAULT_COMMITMENT_WIDTH - 1) / DEFAULT_COMMITMENT_WIDTH : rows;
    if (data_rows > rows) data_rows = rows;
    if (data_rows < rows) ck(sp_fixed_base_mul_h(ctx, ckey, u64p(f_rW.data() + data_rows), rows - data_rows, out.folded_comm + 8 * data_rows), "rest rows");
  }
  lap("blind / X / commitment folds");
}

}  // namespace spartan2

using namespace spartan2;

extern "C" {
const char* ss_last_error();
void ss_set_error(const char* msg);

// the prep_prove part of the NIFS: layers (and i64 mirrors) of the instances; free with sp_nifs_free
int nn_nifs_prepare(sp_ctx* ctx, const sp_shape* S, const uint64_t dims10[10], size_t n, const uint64_t* X, const sp_table* const* Ws, int small_values, sp_nifs** out) {
  try {
    sp_dims dims;
    memcpy(&dims, dims10, sizeof(sp_dims));
    *out = nifs_prepare(ctx, S, dims, n, (const fe_t*)X, Ws, small_values != 0);
    return SP_OK;
  } catch (const Error& e) {
    ss_set_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    ss_set_error(e.what());
    return SP_ERR_INTERNAL;
  }
}

// dims10 = sp_dims as ten uint64 (ss_padded_dims order). comms: n x rows affine; X: n x num_public; Ws: n resident witness tables; r_W: n x rows.
// out_* buffers: polys ell_b x 16, r_bs ell_b x 4, E_eq (left + right) x 4, tail 8, folded_rW rows x 4, folded_X d x 4, folded_comm rows x 8.
// prepared: the object nn_nifs_prepare returned for these instances (consumed by the rounds; the caller frees it), or NULL to build the layers here
int nn_nifs_prove(sp_ctx* ctx, const sp_shape* S, const uint64_t dims10[10], const sp_ck* ckey, size_t n, size_t rows, const uint64_t* comms, const uint64_t* X,
                  const sp_table* const* Ws, const uint64_t* r_W, int small_values, sp_nifs* prepared, sp_transcript* tr, nn_round_hook hook, void* user, uint64_t* out_polys,
                  uint64_t* out_r_bs, uint64_t* out_E, uint64_t* out_tail, uint64_t* out_folded_rW, uint64_t* out_folded_X, uint64_t* out_folded_comm,
                  sp_table* out_A, sp_table* out_B, sp_table* out_C, sp_table* out_folded_W) {
  try {
    sp_dims dims;
    memcpy(&dims, dims10, sizeof(sp_dims));
    NifsOutputs o{out_polys, out_r_bs, out_E, out_tail, out_folded_rW, out_folded_X, out_folded_comm, out_A, out_B, out_C, out_folded_W};
    nifs_prove(ctx, S, dims, ckey, n, rows, (const aff_t*)comms, (const fe_t*)X, Ws, (const fe_t*)r_W, small_values != 0, prepared, tr, hook, user, o);
    return SP_OK;
  } catch (const Error& e) {
    ss_set_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    ss_set_error(e.what());
    return SP_ERR_INTERNAL;
  }
}

// The sharded form: `comm` is an ssc_comm_* handle; *_local arguments describe this rank's n_local instances; outputs as nn_nifs_prove, identical on
// every rank (polys / r_bs: ell_b = log2(n_local * world) rounds). prepared: nn_nifs_prepare's object for the rank's instances (or NULL).
int nn_nifs_prove_sharded(sp_ctx* ctx, void* comm, const sp_shape* S, const uint64_t dims10[10], const sp_ck* ckey, size_t n_local, size_t rows, const uint64_t* comms_local,
                          const uint64_t* X_local, const sp_table* const* Ws_local, const uint64_t* r_W_local, int small_values, sp_nifs* prepared, sp_transcript* tr,
                          nn_round_hook hook, void* user,
                          uint64_t* out_polys, uint64_t* out_r_bs, uint64_t* out_E, uint64_t* out_tail, uint64_t* out_folded_rW, uint64_t* out_folded_X,
                          uint64_t* out_folded_comm, sp_table* out_A, sp_table* out_B, sp_table* out_C, sp_table* out_folded_W) {
  try {
    sp_dims dims;
    memcpy(&dims, dims10, sizeof(sp_dims));
    NifsOutputs o{out_polys, out_r_bs, out_E, out_tail, out_folded_rW, out_folded_X, out_folded_comm, out_A, out_B, out_C, out_folded_W};
    nifs_prove_sharded(ctx, *(Comm*)comm, S, dims, ckey, n_local, rows, (const aff_t*)comms_local, (const fe_t*)X_local, Ws_local, (const fe_t*)r_W_local, small_values != 0, prepared,
                       tr, hook, user, o);
    return SP_OK;
  } catch (const Error& e) {
    ss_set_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    ss_set_error(e.what());
    return SP_ERR_INTERNAL;
  }
}
}
