// SpartanSNARK::{setup, prep_prove, prove} for ONE proof sharded over the GPUs of a node — SURVEY.md 8(e), north_star: "MSM shards by point
// range and sum-check by evaluation-table slice across the 8 GPUs of one node with one RCCL reduce per round". One process per GPU; every rank
// calls these functions with the same arguments and obtains the same proof (bit-identical to the unsharded prover's, tests/test_gpu_sharded_snark.py).
//
// World = 2^k ranks, rank g:
//   * Hyrax witness commitment (hyrax_pc.rs:230-300): BY ROW — rank g commits the rows [g R/2^k, (g+1) R/2^k); one all-gather of 64-byte rows.
//   * Az, Bz, Cz (src/r1cs/mod.rs:1170-1211): BY ROW, interleaved — rank g computes the rows i = (j << k) | g, which is exactly its slice of the
//     three sum-check tables (a "row-slice" sp_shape built from the rows of the CSR matrices; z replicated).
//   * outer sum-check (src/sumcheck.rs:502-571): BY TABLE SLICE on the last k variables — the pairs of the first ell - k rounds are rank-local;
//     per round the slice's two sums, scaled by eq(tau[ell-k..), bits of g), are all-gathered and added in rank order; after ell - k rounds the
//     ranks gather their three final values into 2^k-element tables and all finish the last k rounds redundantly.
//   * poly_ABC (src/r1cs/mod.rs:1235-1321): BY COLUMN, interleaved — rank g computes the columns c = (j << k) | g ("column-slice" sp_shape,
//     evals_rx replicated), its slice of the inner sum-check's first table; the second one is the same slice of z (strided device copy).
//   * inner sum-check (src/sumcheck.rs:190-247): by slice as above.
//   * opening (hyrax_pc.rs:387-478, ipa.rs:125-170): L . W BY ROW BLOCK (partial 2048-vectors all-gathered and added), comm_LZ = sum_i L_i comm_W[i]
//     and delta = <d, ck> BY POINT RANGE (partial points all-gathered and added with the group law: RCCL has no such reduction).
// Everything that does not scale with the instance (transcript, claims, eq tables of the opening) runs redundantly on every rank.
#include <unistd.h>

#include <thread>
#include <condition_variable>
#include <mutex>

#include "comm.hpp"
#include "snark_common.hpp"

namespace spartan2 {

struct ShardedKey {
  sp_ctx* ctx = nullptr;
  Comm* comm = nullptr;
  int k = 0;  // log2(world)
  sp_shape *S_rows = nullptr, *S_cols = nullptr;
  sp_ck *ck = nullptr, *ck_s = nullptr;
  sp_dims dims;  // of the whole instance
  size_t num_vars = 0, num_extra = 0, num_cols = 0;
  uint8_t vk_digest[32];
  std::vector<aff_t> gens, gens_s;
  ~ShardedKey() {
    sp_shape_free(S_rows);
    sp_shape_free(S_cols);
    sp_ck_free(ck);
    sp_ck_free(ck_s);
  }
};

struct ShardedPrep {
  sp_table *W = nullptr, *Wblk = nullptr;                        // replicated witness; this rank's row block (for bind_with_delayed)
  sp_table *caz = nullptr, *cbz = nullptr, *ccz = nullptr;      // cached partial products of this rank's rows
  sp_table *az = nullptr, *bz = nullptr, *cz = nullptr;
  sp_table *p0 = nullptr, *p1 = nullptr;  // per-pair products of the outer sum-check's first round on this rank's rows (sp_multiply_vec_incremental_round0)
  sp_table *z = nullptr, *zs = nullptr, *abc = nullptr, *rx = nullptr;  // z replicated (2M), its slice, poly_ABC slice, evals_rx
  std::vector<aff_t> comm_W;  // all rows; the fixed ones filled at prep time
  std::vector<fe_t> r_W_fixed;
  size_t rows_shared = 0, rows_precommitted = 0;
  std::vector<uint8_t> comm_shared_bytes, comm_pre_bytes;  // transcript encodings of the rows committed at prep time (hyrax_pc.rs:714-729)
  sp_table *gS = nullptr, *gT[3] = {nullptr, nullptr, nullptr};  // staging + gathered tables of the sum-checks' hand-over (2^gather_log2 elements)
  size_t g_cap = 0;
  bool is_small = true;
  ~ShardedPrep() {
    for (sp_table* t : {W, Wblk, caz, cbz, ccz, az, bz, cz, p0, p1, z, zs, abc, rx, gS, gT[0], gT[1], gT[2]}) sp_table_free(t);
  }
};

static size_t ceil_slice(size_t n, size_t g, size_t world) { return n > g ? (n - g + world - 1) / world : 0; }  // #{i < n : i = g mod world}

// rows [lo, hi) of this rank's block that fall into the row range [a, b)
static void block_overlap(size_t lo, size_t hi, size_t a, size_t b, size_t* first, size_t* count) {
  const size_t s = std::max(lo, a), e = std::min(hi, b);
  *first = s;
  *count = e > s ? e - s : 0;
}

ShardedKey* sharded_setup(sp_ctx* ctx, Comm* comm, const R1CSIntView& R) {
  auto* pk = new ShardedKey();
  try {
    pk->ctx = ctx;
    pk->comm = comm;
    const size_t world = (size_t)comm->world, g = (size_t)comm->rank;
    while (((size_t)1 << pk->k) < world) ++pk->k;
    if (((size_t)1 << pk->k) != world) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "sharded prover: the number of ranks must be a power of two");
    if (R.num_challenges != 0) throw Error(SP_ERR_INTERNAL, "sharded prover: circuits with verifier challenges are not driven by this layer");
    PaddedShape P = pad_shape(R);
    pk->dims = P.dims;
    pk->num_vars = P.num_vars();
    pk->num_extra = 1 + P.dims.num_public + P.dims.num_challenges;
    pk->num_cols = P.num_cols();
    const size_t N = P.dims.num_cons, M = pk->num_vars;
    if (N < 2 * world || M < DEFAULT_COMMITMENT_WIDTH * world) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "sharded prover: instance too small for this many ranks");
    // row slice: rows i = (j << k) | g, all columns
    {
      std::vector<fe_t> data[3];
      std::vector<uint32_t> idx[3];
      std::vector<uint64_t> ptr[3];
      for (int m = 0; m < 3; ++m) {
        ptr[m].push_back(0);
        for (size_t j = 0; j < N / world; ++j) {
          const size_t i = (j << pk->k) | g;
          for (uint64_t e = P.ptr[m][i]; e < P.ptr[m][i + 1]; ++e) {
            data[m].push_back(P.data[m][e]);
            idx[m].push_back(P.idx[m][e]);
          }
          ptr[m].push_back(data[m].size());
        }
      }
      sp_dims d = P.dims;
      d.num_cons = N / world;
      d.num_cons_unpadded = ceil_slice(P.dims.num_cons_unpadded, g, world);
      sp_csr cs[3];
      for (int m = 0; m < 3; ++m) cs[m] = sp_csr{u64p(data[m].data()), idx[m].data(), ptr[m].data()};
      ck(sp_shape_from_csr(ctx, &cs[0], &cs[1], &cs[2], &d, &pk->S_rows), "shape_from_csr (row slice)");
    }
    // column slice: all rows, columns c = (j << k) | g renumbered j
    {
      std::vector<fe_t> data[3];
      std::vector<uint32_t> idx[3];
      std::vector<uint64_t> ptr[3];
      for (int m = 0; m < 3; ++m) {
        ptr[m].push_back(0);
        for (size_t i = 0; i < N; ++i) {
          for (uint64_t e = P.ptr[m][i]; e < P.ptr[m][i + 1]; ++e) {
            const size_t c = P.idx[m][e];
            if ((c & (world - 1)) != g) continue;
            data[m].push_back(P.data[m][e]);
            idx[m].push_back((uint32_t)(c >> pk->k));
          }
          ptr[m].push_back(data[m].size());
        }
      }
      sp_dims d;
      memset(&d, 0, sizeof d);
      d.num_cons = N;
      d.num_cons_unpadded = P.dims.num_cons_unpadded;
      d.num_precommitted = d.num_precommitted_unpadded = M / world;
      const size_t extra = ceil_slice(pk->num_extra, g, world);  // the slice's share of (1, public...) — M is a multiple of world
      d.num_public = extra > 0 ? extra - 1 : 0;
      sp_csr cs[3];
      for (int m = 0; m < 3; ++m) cs[m] = sp_csr{u64p(data[m].data()), idx[m].data(), ptr[m].data()};
      ck(sp_shape_from_csr(ctx, &cs[0], &cs[1], &cs[2], &d, &pk->S_cols), "shape_from_csr (column slice)");
    }
    pk->gens = from_label("ck", DEFAULT_COMMITMENT_WIDTH + 1);
    pk->gens_s = from_label("ck_s", 2);
    ck(sp_ck_create(ctx, u64p(&pk->gens[0].x), DEFAULT_COMMITMENT_WIDTH, u64p(&pk->gens[DEFAULT_COMMITMENT_WIDTH].x), &pk->ck), "ck_create");
    ck(sp_ck_create(ctx, u64p(&pk->gens_s[0].x), 1, u64p(&pk->gens_s[1].x), &pk->ck_s), "ck_s_create");
    spartan_vk_digest(P, pk->gens, pk->gens_s, pk->vk_digest);
  } catch (...) {
    delete pk;
    throw;
  }
  return pk;
}

// all-gather of per-rank row blocks of a commitment: `mine` holds this rank's rows_per_rank rows (identity where a row belongs to another phase)
static void gather_rows(Comm& comm, const std::vector<aff_t>& mine, std::vector<aff_t>* all) {
  all->resize(mine.size() * comm.world);
  comm.allgather(mine.data(), mine.size() * sizeof(aff_t), all->data());
}

ShardedPrep* sharded_prep_prove(const ShardedKey& pk, const uint64_t* witness_u64, size_t n_witness, bool is_small, Tape& tape) {
  const sp_dims& d = pk.dims;
  if (n_witness != d.num_shared_unpadded + d.num_precommitted_unpadded + d.num_rest_unpadded) throw Error(SP_ERR_INVALID_WITNESS_LENGTH, "InvalidWitnessLength");
  auto* ps = new ShardedPrep();
  try {
    sp_ctx* ctx = pk.ctx;
    Comm& comm = *pk.comm;
    const size_t world = (size_t)comm.world, g = (size_t)comm.rank;
    const size_t M = pk.num_vars, N = d.num_cons, CW = DEFAULT_COMMITMENT_WIDTH;
    ps->is_small = is_small;
    // machine words in, Montgomery-form elements formed on the device (sp_table_write_u64; spartan_snark.cpp prep_prove)
    ck(sp_table_zeros(ctx, M, (size_t)-1, (size_t)-1, &ps->W), "alloc W");
    auto put = [&](size_t dst, size_t src, size_t cnt) { ck(sp_table_write_u64(ctx, ps->W, dst, witness_u64 + src, cnt), "upload W"); };
    put(0, 0, d.num_shared_unpadded);
    put(d.num_shared, d.num_shared_unpadded, d.num_precommitted_unpadded);
    put(d.num_shared + d.num_precommitted, d.num_shared_unpadded + d.num_precommitted_unpadded, d.num_rest_unpadded);
    const size_t blk = M / world;
    ck(sp_table_zeros(ctx, blk, (size_t)-1, (size_t)-1, &ps->Wblk), "alloc W block");
    ck(sp_table_copy(ctx, ps->Wblk, 0, ps->W, g * blk, blk), "W block");
    // commitments fixed at prep time: shared rows then precommitted rows (blinds for ALL of them are drawn by every rank: same tape positions)
    ps->rows_shared = d.num_shared_unpadded ? (d.num_shared + CW - 1) / CW : 0;
    ps->rows_precommitted = d.num_precommitted_unpadded ? (d.num_precommitted + CW - 1) / CW : 0;
    ps->r_W_fixed.resize(ps->rows_shared + ps->rows_precommitted);
    for (auto& b : ps->r_W_fixed) b = tape.next();
    const size_t rows_all = M / CW, rpr = rows_all / world, lo = g * rpr, hi = lo + rpr;
    ps->comm_W.assign(rows_all, aff_t{fe_zero(), fe_zero()});
    std::vector<aff_t> mine(rpr, aff_t{fe_zero(), fe_zero()});
    // segment row ranges in the padded layout: shared [0, num_shared / CW), precommitted [num_shared / CW, (num_shared + num_precommitted) / CW)
    struct Seg {
      size_t a, b, blind0;
      bool on;
    } segs[2] = {{0, d.num_shared / CW, 0, ps->rows_shared != 0},
                 {d.num_shared / CW, (d.num_shared + d.num_precommitted) / CW, ps->rows_shared, ps->rows_precommitted != 0}};
    for (const Seg& sgm : segs) {
      if (!sgm.on) continue;
      size_t first, cnt;
      block_overlap(lo, hi, sgm.a, sgm.b, &first, &cnt);
      if (!cnt) continue;
      ck(sp_hyrax_commit(ctx, pk.ck, ps->W, first * CW, cnt * CW, u64p(ps->r_W_fixed.data() + sgm.blind0 + (first - sgm.a)), is_small ? 1 : 0,
                         u64p(&mine[first - lo].x)),
         "commit (row block)");
    }
    std::vector<aff_t> all;
    gather_rows(comm, mine, &all);
    for (const Seg& sgm : segs)
      if (sgm.on)
        for (size_t r = sgm.a; r < sgm.b; ++r) ps->comm_W[r] = all[r];
    // the rows' transcript encodings are fixed with them; every prove still hashes them (the reference re-absorbs them in every prove)
    if (ps->rows_shared) ps->comm_shared_bytes = commitment_bytes(ps->comm_W.data(), ps->rows_shared);
    if (ps->rows_precommitted) ps->comm_pre_bytes = commitment_bytes(ps->comm_W.data() + d.num_shared / CW, ps->rows_precommitted);
    // z (2M, zero padded so that the inner sum-check's slices exist) and the cached products of this rank's rows on z = [W_shared+precommitted | 0]
    ck(sp_table_zeros(ctx, 2 * M, (size_t)-1, (size_t)-1, &ps->z), "alloc z");
    ck(sp_table_copy(ctx, ps->z, 0, ps->W, 0, d.num_shared + d.num_precommitted), "copy W");
    ck(sp_table_set_len(ps->z, pk.num_cols, (size_t)-1, (size_t)-1), "z len");
    for (sp_table** t : {&ps->caz, &ps->cbz, &ps->ccz, &ps->az, &ps->bz, &ps->cz}) ck(sp_table_zeros(ctx, N / world, (size_t)-1, (size_t)-1, t), "alloc Az");
    ck(sp_multiply_vec(ctx, pk.S_rows, ps->z, ps->caz, ps->cbz, ps->ccz), "multiply_vec_precommitted (row slice)");
    ck(sp_table_zeros(ctx, N, (size_t)-1, (size_t)-1, &ps->rx), "alloc rx");
    if (N / world >= 2) {
      ck(sp_table_zeros(ctx, N / world / 2, (size_t)-1, (size_t)-1, &ps->p0), "alloc round-0 products");
      ck(sp_table_zeros(ctx, N / world / 2, (size_t)-1, (size_t)-1, &ps->p1), "alloc round-0 products");
    }
    ck(sp_table_zeros(ctx, 2 * M / world, (size_t)-1, (size_t)-1, &ps->abc), "alloc poly_ABC slice");
    ck(sp_table_zeros(ctx, 2 * M / world, (size_t)-1, (size_t)-1, &ps->zs), "alloc z slice");
    ck(sp_ctx_synchronize(ctx), "sync");
  } catch (...) {
    delete ps;
    throw;
  }
  return ps;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int reduce_hook(void* user, uint64_t* sums, size_t count) {
  try {
    ((Comm*)user)->field_sum(reinterpret_cast<fe_t*>(sums), count);
    return 0;
  } catch (...) {
    return SP_ERR_INTERNAL;
  }
}

// SpartanSNARK::prove (src/spartan.rs:219-466), one proof over all ranks
// Where the sharded sum-checks hand over: the slices exchange sums only while the GATHERED tables would still have more than 2^gather_log2 elements
// (the bandwidth-bound rounds, where sharding pays); then every rank gathers all slices and finishes alone - one bulk exchange instead of one small
// exchange in each of the remaining ~16 latency-bound rounds. SPARTAN_SHARD_GATHER_LOG2 (default 16; k = one element per rank, the old hand-over).
static size_t gather_log2() {
  static const size_t v = [] {
    const char* e = getenv("SPARTAN_SHARD_GATHER_LOG2");
    const long x = e ? atol(e) : 16;
    return (size_t)(x < 0 ? 0 : (x > 24 ? 24 : x));
  }();
  return v;
}
// rounds the slice runs before the hand-over, for a sum-check of `ell` variables over 2^k ranks
static size_t local_rounds(size_t ell, size_t k) {
  if (k == 0) return ell;
  size_t g2 = gather_log2();
  if (g2 < k) g2 = k;
  const size_t loc = ell - k, keep = g2 - k;  // slice variables left unbound at the hand-over
  return loc > keep ? loc - keep : 0;
}
static void* table_dev(const sp_table* t) {
  void* p = nullptr;
  ck(sp_table_device_ptr(t, &p, nullptr), "device_ptr");
  return p;
}
// gathered[q][(j << k) | g] = rank g's slice[q][j], j < m: the bound tables in natural order on every rank
static void gather_slices(sp_ctx* ctx, Comm& comm, ShardedPrep& ps, sp_table* const* slices, int ntab, size_t m) {
  const size_t world = (size_t)comm.world, total = m * world;
  if (total > ps.g_cap) {
    for (sp_table** t : {&ps.gS, &ps.gT[0], &ps.gT[1], &ps.gT[2]}) {
      sp_table_free(*t);
      *t = nullptr;
    }
    ps.g_cap = total;
    for (sp_table** t : {&ps.gS, &ps.gT[0], &ps.gT[1], &ps.gT[2]}) ck(sp_table_zeros(ctx, total, (size_t)-1, (size_t)-1, t), "hand-over tables");
  }
  for (int q = 0; q < ntab; ++q) {
    ck(sp_ctx_synchronize(ctx), "synchronize");
    comm.allgather_device(table_dev(slices[q]), m * sizeof(fe_t), table_dev(ps.gS));
    for (size_t r = 0; r < world; ++r) ck(sp_table_scatter_strided(ctx, ps.gT[q], r, world, ps.gS, r * m, m), "interleave");
    ck(sp_table_set_len(ps.gT[q], total, (size_t)-1, (size_t)-1), "gathered len");
  }
  ck(sp_ctx_synchronize(ctx), "synchronize");  // gS is reused by the next table / call
}

SpartanProofBuf sharded_prove(const ShardedKey& pk, ShardedPrep& ps, const uint64_t* publics_u64, size_t npub, Tape& tape, double phase_ms[8]) {
  const sp_dims& d = pk.dims;
  sp_ctx* ctx = pk.ctx;
  Comm& comm = *pk.comm;
  const size_t world = (size_t)comm.world, g = (size_t)comm.rank, k = (size_t)pk.k;
  const size_t M = pk.num_vars, N = d.num_cons, CW = DEFAULT_COMMITMENT_WIDTH;
  if (npub != d.num_public) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "public_values length");
  ck(sp_ctx_bind_thread(ctx), "device");
  const double t_start = now_ms();
  static const bool laps_on = getenv("SPARTAN_HOST_LAPS") != nullptr;  // statement laps on stderr (diagnostics, as the unsharded drivers')
  double t_lap = t_start;
  auto lap = [&](const char* name) {
    if (!laps_on) return;
    const double t = now_ms();
    fprintf(stderr, "sharded lap %-28s %.3f ms\n", name, t - t_lap);
    t_lap = t;
  };
  std::vector<fe_t> publics(npub);
  for (size_t i = 0; i < npub; ++i) publics[i] = fe_from_u64<S>(publics_u64[i]);

  // The rest rows of this rank's block when the segment is all padding (commit_zeros = h * blind, hyrax_pc.rs:305-319): their blinds are drawn and the
  // fixed-base job launched before anything else, as the unsharded driver does - the chain commitment -> absorb -> tau is what the outer sum-check waits
  // for, and queued behind the matrix-vector product below it waited for that too (0.44 ms at config 4).
  const size_t rows_all = M / CW, rpr = rows_all / world, lo = g * rpr, hi = lo + rpr;
  const size_t rows_fixed = (d.num_shared + d.num_precommitted) / CW, rows_rest = d.num_rest / CW;
  std::vector<fe_t> r_W_rest(rows_rest);
  for (auto& b : r_W_rest) b = tape.next();
  size_t rest_first = 0, rest_cnt = 0;
  block_overlap(lo, hi, rows_fixed, rows_all, &rest_first, &rest_cnt);
  sp_fb_job* rest_job = nullptr;
  struct RestGuard {  // an error exit must not leave the context's one asynchronous fixed-base job outstanding
    sp_ctx* ctx;
    sp_fb_job*& job;
    size_t n;
    ~RestGuard() {
      if (job) {
        std::vector<uint64_t> sink(8 * n + 8);
        sp_fixed_base_mul_h_finish(ctx, job, sink.data());
      }
    }
  } rest_guard{ctx, rest_job, rest_cnt};
  if (rest_cnt && d.num_rest_unpadded == 0)
    ck(sp_fixed_base_mul_h_begin(ctx, pk.ck, u64p(r_W_rest.data() + (rest_first - rows_fixed)), rest_cnt, &rest_job), "commit_zeros (row block, begin)");
  // z = [W | 1 | public | 0 ...] replicated; Az, Bz, Cz of this rank's rows and the round-0 products of its pairs. Issued next: the products depend on
  // the witness and the publics alone and run under the rest rows' commitment, the transcript's absorbs and the helper's start (round 6: the product sat
  // behind all of that, 0.23 ms of the phase clock at config 4); the slice of z the inner sum-check binds is cut out right behind it, off its path too.
  ck(sp_table_set_len(ps.z, 2 * M, (size_t)-1, (size_t)-1), "z len");
  ck(sp_table_copy(ctx, ps.z, 0, ps.W, 0, M), "z <- W");
  {
    ck(sp_table_zero(ctx, ps.z, M, M), "clear z high half");
    std::vector<fe_t> tail(pk.num_extra);
    tail[0] = fe_one<S>();
    std::copy(publics.begin(), publics.end(), tail.begin() + 1);
    if (tail.size() <= 2048) ck(sp_table_write_async(ctx, ps.z, M, u64p(tail.data()), tail.size()), "z tail");
    else ck(sp_table_write(ctx, ps.z, M, u64p(tail.data()), tail.size()), "z tail");
  }
  ck(sp_table_set_len(ps.z, pk.num_cols, (size_t)-1, (size_t)-1), "z len");
  if (ps.p0) ck(sp_multiply_vec_incremental_round0(ctx, pk.S_rows, ps.z, ps.caz, ps.cbz, ps.ccz, ps.az, ps.bz, ps.cz, ps.p0, ps.p1), "multiply_vec_incremental (row slice)");
  else ck(sp_multiply_vec_incremental(ctx, pk.S_rows, ps.z, ps.caz, ps.cbz, ps.ccz, ps.az, ps.bz, ps.cz), "multiply_vec_incremental (row slice)");
  // the inner sum-check's z: one rank binds z itself (it is rebuilt from W by the next prove), several take their interleaved slice
  sp_table* const z_inner = world == 1 ? ps.z : ps.zs;
  if (world > 1) {
    ck(sp_table_set_len(ps.z, 2 * M, (size_t)-1, (size_t)-1), "z len");
    ck(sp_table_gather_strided(ctx, ps.zs, 0, ps.z, g, world, 2 * M / world), "z slice");
    ck(sp_table_set_len(ps.z, pk.num_cols, (size_t)-1, (size_t)-1), "z len");
  }

  lap("z_and_spmv_issue");
  Tr tr(ctx, "SpartanSNARK");
  tr.absorb("vk", pk.vk_digest, 32);
  tr.absorb_scalars("public_values", publics.data(), npub);
  if (ps.rows_shared) tr.absorb("comm_W_shared", ps.comm_shared_bytes.data(), ps.comm_shared_bytes.size());
  if (ps.rows_precommitted) tr.absorb("comm_W_precommitted", ps.comm_pre_bytes.data(), ps.comm_pre_bytes.size());
  // rest rows (bellpepper/r1cs.rs:463-491): every rank draws all blinds, commits the rest rows of its block (commit_zeros = h * blind when the
  // segment is all padding), one all-gather
  {
    std::vector<aff_t> mine(rpr, aff_t{fe_zero(), fe_zero()});
    const size_t first = rest_first, cnt = rest_cnt;
    if (cnt) {
      if (rest_job) {
        sp_fb_job* j = rest_job;
        rest_job = nullptr;
        ck(sp_fixed_base_mul_h_finish(ctx, j, u64p(&mine[first - lo].x)), "commit_zeros (row block)");
      } else
        ck(sp_hyrax_commit(ctx, pk.ck, ps.W, first * CW, cnt * CW, u64p(r_W_rest.data() + (first - rows_fixed)), ps.is_small ? 1 : 0, u64p(&mine[first - lo].x)),
           "commit rest (row block)");
    }
    lap("commit_rest_rows");
    std::vector<aff_t> all;
    gather_rows(comm, mine, &all);
    lap("gather_rows");
    for (size_t r = rows_fixed; r < rows_all; ++r) ps.comm_W[r] = all[r];
    const std::vector<uint8_t> b = commitment_bytes(ps.comm_W.data() + rows_fixed, rows_rest);
    tr.absorb("comm_W_rest", b.data(), b.size());
    lap("absorb_rest_rows");
  }
  // the proof's comm_W omits the rows of empty segments (there are none in the padded layout: a segment with no variables has no rows)
  std::vector<aff_t> comm_W;
  std::vector<fe_t> r_W;
  {
    size_t bi = 0;
    if (ps.rows_shared)
      for (size_t r = 0; r < d.num_shared / CW; ++r) {
        comm_W.push_back(ps.comm_W[r]);
        r_W.push_back(ps.r_W_fixed[bi++]);
      }
    if (ps.rows_precommitted)
      for (size_t r = d.num_shared / CW; r < rows_fixed; ++r) {
        comm_W.push_back(ps.comm_W[r]);
        r_W.push_back(ps.r_W_fixed[bi++]);
      }
    for (size_t r = rows_fixed; r < rows_all; ++r) {
      comm_W.push_back(ps.comm_W[r]);
      r_W.push_back(r_W_rest[r - rows_fixed]);
    }
  }
  if (comm_W.size() != rows_all) throw Error(SP_ERR_INTERNAL, "sharded prover: every witness segment must be non-empty or absent");
  const double t_wit = now_ms();
  // What the opening needs that depends on nothing later than the witness commitment runs on a helper thread under the sum-checks: the sponge state
  // of absorb("poly_com", comm_W) (the first absorb after the inner sum-check's last squeeze), the IPA mask d and its blinds (peeked at their tape
  // positions: blind_eval_W precedes them), and this rank's point range of delta = <d, ck> as an MSM on the auxiliary stream.
  struct Ahead {
    std::thread th;
    std::exception_ptr err;
    sp_absorb_state* poly_com = nullptr;
    std::vector<fe_t> dvec;
    fe_t r_delta, r_beta;
    sp_msm_job* delta_job = nullptr;
    sp_ctx* ctx = nullptr;
    const sp_ck* key = nullptr;
    // comm_LZ ahead: this rank's rows are a contiguous block, i.e. the top k row bits are fixed to the rank: its part of <L, comm_W> is
    // eq(r[0..k), rank) * sum_j eq(r[k..nvr), j) comm_W[lo + j] - an eq-weighted MSM the helper starts the moment the observer has the row challenges
    std::mutex mu;
    std::condition_variable cv;
    int rows_state = 0;  // 0: not drawn yet, 1: drawn, 2: abandoned
    fe_t r_rows[32];
    size_t nvr = 0;
    sp_points* pts = nullptr;
    sp_msm_job* lz_job = nullptr;
    // behind the row challenges too (round 6): this rank's part of L^T W (bind_with_delayed with eq(r[k..nvr), .) formed on the device, a stream of its own)
    // and r_LZ = <eq(r_rows, .), r_W> (2 x 2^nvr host products) - both used to sit behind the LAST challenge on the proving thread
    sp_vec_job* vec_job = nullptr;
    const sp_table* Wblk = nullptr;
    const fe_t* r_W = nullptr;
    size_t n_rW = 0;
    fe_t r_LZ;
    bool have_r_LZ = false;
    // ... and behind the COLUMN challenges (state 3, published after the last round): <R, d> of ipa.rs:148 (2 x 2048 host products) beside the proving
    // thread's own end-of-prove work
    int cols_state = 0;
    bool rows_stage_done = false;  // everything the helper starts before it waits for the column challenges is in place (or it has given up): set under mu
    void mark_rows_stage_done() {
      {
        std::lock_guard<std::mutex> l(mu);
        rows_stage_done = true;
      }
      cv.notify_all();
    }
    void wait_rows_stage() {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return rows_stage_done; });
    }
    const fe_t* col_point = nullptr;
    size_t n_col = 0;
    fe_t ip;
    bool have_ip = false;
    void publish_cols(int v) {
      {
        std::lock_guard<std::mutex> l(mu);
        if (cols_state == 0) cols_state = v;
      }
      cv.notify_one();
    }
    void publish(int v) {
      {
        std::lock_guard<std::mutex> l(mu);
        if (rows_state == 0) rows_state = v;
      }
      cv.notify_one();
    }
    void join() {
      if (th.joinable()) th.join();
    }
    ~Ahead() {  // error exits: the helper's products are still owned here
      publish(2);
      publish_cols(2);
      join();
      uint64_t sink[8];
      if (vec_job) {
        std::vector<uint64_t> vs(4 * 4096);
        sp_rowmat_vec_eq_finish(ctx, vec_job, vs.data());
      }
      if (delta_job) sp_msm_ck_finish(ctx, key, delta_job, nullptr, sink);
      if (lz_job) sp_msm_job_finish(ctx, lz_job, sink);
      if (poly_com) sp_absorb_state_free(poly_com);
      sp_points_free(pts);
    }
  } ahead;
  ahead.ctx = ctx;
  ahead.key = pk.ck;
  ahead.Wblk = ps.Wblk;
  ahead.r_W = r_W.data();
  ahead.n_rW = r_W.size();
  const size_t ncols_ipa = (size_t)1 << ((log2_ceil(M)) - log2_ceil(rows_all));
  const size_t nvr_rows = log2_ceil(rows_all), ly_all = log2_ceil(M) + 1;
  // the row challenges are r_y[1 ..= nvr] (the observer sees every round, slice or gathered); the block must hold at least two rows
  const bool lz_on = nvr_rows > k && nvr_rows + 1 <= ly_all && nvr_rows <= 31 && rpr == ((size_t)1 << (nvr_rows - k));
  ahead.nvr = nvr_rows;
  {
    Tape peek = tape;
    const aff_t* rows_ptr = comm_W.data();
    const size_t nrows = comm_W.size(), cpr0 = ncols_ipa / world;
    ahead.th = std::thread([&ahead, peek, rows_ptr, nrows, ncols_ipa, cpr0, g, ctx, &pk, lz_on, lo, rpr, k]() mutable {
      try {
        const std::vector<uint8_t> b = commitment_bytes(rows_ptr, nrows);
        ck(sp_transcript_preabsorb((const uint8_t*)"poly_com", 8, b.data(), b.size(), &ahead.poly_com), "poly_com (prepare)");
        peek.skip(1);  // blind_eval_W
        ahead.dvec.resize(ncols_ipa);
        for (auto& x : ahead.dvec) x = peek.next();
        ahead.r_delta = peek.next();
        ahead.r_beta = peek.next();
        ck(sp_ctx_bind_thread(ctx), "helper thread: device");
        if (cpr0 * (size_t)pk.comm->world == ncols_ipa)
          ck(sp_msm_ck_range_begin(ctx, pk.ck, u64p(ahead.dvec.data() + g * cpr0), g * cpr0, cpr0, &ahead.delta_job), "delta (begin)");
        if (lz_on) {
          ck(sp_points_upload(ctx, u64p(&rows_ptr[lo].x), rpr, &ahead.pts), "comm_W block (upload)");
          int st;
          {
            std::unique_lock<std::mutex> lk(ahead.mu);
            ahead.cv.wait(lk, [&] { return ahead.rows_state != 0; });
            st = ahead.rows_state;
          }
          if (st == 1) {
            ck(sp_msm_eq_begin(ctx, ahead.pts, u64p(ahead.r_rows + k), ahead.nvr - k, &ahead.lz_job), "comm_LZ (begin)");
            if (ncols_ipa <= 4096 && ahead.nvr - k <= 20)
              ck(sp_rowmat_vec_eq_begin(ctx, ahead.Wblk, u64p(ahead.r_rows + k), ahead.nvr - k, ncols_ipa, &ahead.vec_job), "bind_with_delayed (begin)");
            const std::vector<fe_t> Lh = eq_evals_host(ahead.r_rows, ahead.nvr);
            if (Lh.size() == ahead.n_rW) {
              fe_t acc = fe_zero();
              for (size_t i = 0; i < Lh.size(); ++i) acc = fe_add<S>(acc, fe_mul<S>(Lh[i], ahead.r_W[i]));
              ahead.r_LZ = acc;
              ahead.have_r_LZ = true;
            }
            ahead.mark_rows_stage_done();
            int cs;
            {
              std::unique_lock<std::mutex> lk(ahead.mu);
              ahead.cv.wait(lk, [&] { return ahead.cols_state != 0; });
              cs = ahead.cols_state;
            }
            if (cs == 1) {
              const std::vector<fe_t> Rh = eq_evals_host(ahead.col_point, ahead.n_col);
              if (Rh.size() == ahead.dvec.size()) {
                fe_t acc = fe_zero();
                for (size_t i = 0; i < Rh.size(); ++i) acc = fe_add<S>(acc, fe_mul<S>(Rh[i], ahead.dvec[i]));
                ahead.ip = acc;
                ahead.have_ip = true;
              }
            }
          }
        }
      } catch (...) {
        ahead.err = std::current_exception();
      }
      ahead.mark_rows_stage_done();
    });
  }

  const size_t lx = log2_ceil(N), ly = log2_ceil(M) + 1;
  std::vector<fe_t> tau(lx);
  for (auto& t : tau) t = tr.squeeze("t");
  const double t_mv = now_ms();
  lap("helper_start_and_tau");

  SpartanProofBuf proof;
  for (const aff_t& a : comm_W) proof.pp(a);
  for (const fe_t& f : publics) proof.pf(f);
  // outer sum-check on slices: ell - k local rounds with one exchange each, then k rounds on the gathered 2^k-element tables
  std::vector<fe_t> outer_polys(3 * lx), r_x(lx);
  fe_t claims_outer[3];
  {
    fe_t claim = fe_zero(), p = fe_one<S>();
    fe_t scale = fe_one<S>();  // eq(tau[lx - k ..), bits of g), MSB of g = first of those variables
    for (size_t i = 0; i < k; ++i) {
      const fe_t t = tau[lx - k + i];
      scale = fe_mul<S>(scale, ((g >> (k - 1 - i)) & 1) ? t : fe_sub<S>(fe_one<S>(), t));
    }
    fe_t fin[3];
    const size_t loc = lx - k, R = local_rounds(lx, k);
    if (R == loc) {  // unsharded, or the hand-over at one element per rank
      if (ps.p0)
        ck(sp_sumcheck_cubic3_sharded_round0(ctx, u64p(&claim), u64p(&p), u64p(tau.data()), loc, 0, ps.az, ps.bz, ps.cz, ps.p0, ps.p1, tr.t, k ? u64p(&scale) : nullptr,
                                             k ? reduce_hook : nullptr, &comm, u64p(outer_polys.data()), u64p(r_x.data()), u64p(fin)),
           "outer sum-check (local rounds)");
      else
        ck(sp_sumcheck_cubic3_sharded(ctx, u64p(&claim), u64p(&p), u64p(tau.data()), loc, ps.az, ps.bz, ps.cz, tr.t, k ? u64p(&scale) : nullptr,
                                      k ? reduce_hook : nullptr, &comm, u64p(outer_polys.data()), u64p(r_x.data()), u64p(fin)),
           "outer sum-check (local rounds)");
      if (k) {
        std::vector<fe_t> all(3 * world);
        comm.allgather(fin, sizeof fin, all.data());
        sp_table* T[3] = {nullptr, nullptr, nullptr};
        struct Guard {
          sp_table** t;
          ~Guard() {
            for (int i = 0; i < 3; ++i) sp_table_free(t[i]);
          }
        } guard{T};
        for (int q = 0; q < 3; ++q) {
          std::vector<fe_t> col(world);
          for (size_t r = 0; r < world; ++r) col[r] = all[3 * r + q];
          ck(sp_table_from_host(ctx, u64p(col.data()), world, (size_t)-1, (size_t)-1, &T[q]), "gathered table");
        }
        ck(sp_sumcheck_cubic3_sharded(ctx, u64p(&claim), u64p(&p), u64p(tau.data() + loc), k, T[0], T[1], T[2], tr.t, nullptr, nullptr, nullptr,
                                      u64p(outer_polys.data() + 3 * loc), u64p(r_x.data() + loc), u64p(fin)),
           "outer sum-check (last rounds)");
      }
    } else {  // R slice rounds with one exchange each, ONE bulk hand-over, then lx - R rounds on the gathered tables
      if (R && ps.p0)
        ck(sp_sumcheck_cubic3_sharded_round0(ctx, u64p(&claim), u64p(&p), u64p(tau.data()), loc, R, ps.az, ps.bz, ps.cz, ps.p0, ps.p1, tr.t, u64p(&scale), reduce_hook, &comm,
                                             u64p(outer_polys.data()), u64p(r_x.data()), nullptr),
           "outer sum-check (slice rounds)");
      else if (R)
        ck(sp_sumcheck_cubic3_sharded_partial(ctx, u64p(&claim), u64p(&p), u64p(tau.data()), loc, R, ps.az, ps.bz, ps.cz, tr.t, u64p(&scale), reduce_hook, &comm,
                                              u64p(outer_polys.data()), u64p(r_x.data())),
           "outer sum-check (slice rounds)");
      sp_table* const sl[3] = {ps.az, ps.bz, ps.cz};
      gather_slices(ctx, comm, ps, sl, 3, (size_t)1 << (loc - R));
      ck(sp_sumcheck_cubic3_sharded(ctx, u64p(&claim), u64p(&p), u64p(tau.data() + R), lx - R, ps.gT[0], ps.gT[1], ps.gT[2], tr.t, nullptr, nullptr, nullptr,
                                    u64p(outer_polys.data() + 3 * R), u64p(r_x.data() + R), u64p(fin)),
         "outer sum-check (gathered rounds)");
    }
    for (int i = 0; i < 3; ++i) claims_outer[i] = fin[i];
  }
  tr.absorb_scalars("claims_outer", claims_outer, 3);
  for (const fe_t& f : outer_polys) proof.pf(f);
  for (int i = 0; i < 3; ++i) proof.pf(claims_outer[i]);
  const double t_outer = now_ms();
  lap("outer");

  const fe_t r = tr.squeeze("r");
  const fe_t claim_inner_joint = fe_add<S>(fe_add<S>(claims_outer[0], fe_mul<S>(r, claims_outer[1])), fe_mul<S>(fe_mul<S>(r, r), claims_outer[2]));
  ck(sp_eq_table_into(ctx, u64p(r_x.data()), lx, ps.rx), "evals_rx");
  ck(sp_poly_abc(ctx, pk.S_cols, ps.rx, u64p(&r), 2 * M / world, ps.abc), "poly_ABC (column slice)");
  const double t_abc = now_ms();

  // inner sum-check on slices of the 2M-long tables: (lo_eff, hi_eff) = (M / world, this slice's share of the num_extra entries)
  const size_t extra_here = ceil_slice(pk.num_extra, g, world);
  ck(sp_table_set_len(ps.abc, 2 * M / world, M / world, extra_here), "abc len");
  ck(sp_table_set_len(z_inner, 2 * M / world, M / world, extra_here), "z len");
  std::vector<fe_t> inner_polys(2 * ly), r_y(ly);
  fe_t claims_inner[2];
  {
    fe_t claim = claim_inner_joint;
    fe_t fin[2];
    struct Obs {
      decltype(ahead)* a;
      bool on;
      size_t base;  // rounds already run by an earlier call of this sum-check
      static void fn(void* u, size_t round, const uint64_t r[4]) {  // r_y[base + round]; the row variables are r_y[1 ..= nvr] (MSB first)
        Obs* o = (Obs*)u;
        round += o->base;
        if (!o->on || round == 0 || round > o->a->nvr) return;
        memcpy(&o->a->r_rows[round - 1], r, 32);
        if (round == o->a->nvr) o->a->publish(1);
      }
    } obs{&ahead, lz_on, 0};
    if (ly != ly_all) throw Error(SP_ERR_INTERNAL, "sharded prover: inner round count");
    const size_t loc = ly - k, R = local_rounds(ly, k);
    if (R == loc) {
      ck(sp_sumcheck_quad_sharded_observed(ctx, u64p(&claim), loc, ps.abc, z_inner, tr.t, k ? reduce_hook : nullptr, &comm, &Obs::fn, &obs, u64p(inner_polys.data()),
                                           u64p(r_y.data()), u64p(fin)),
         "inner sum-check (local rounds)");
      if (k) {
        std::vector<fe_t> all(2 * world);
        comm.allgather(fin, sizeof fin, all.data());
        sp_table* T[2] = {nullptr, nullptr};
        struct Guard {
          sp_table** t;
          ~Guard() {
            for (int i = 0; i < 2; ++i) sp_table_free(t[i]);
          }
        } guard{T};
        for (int q = 0; q < 2; ++q) {
          std::vector<fe_t> col(world);
          for (size_t rr = 0; rr < world; ++rr) col[rr] = all[2 * rr + q];
          ck(sp_table_from_host(ctx, u64p(col.data()), world, (size_t)-1, (size_t)-1, &T[q]), "gathered table");
        }
        obs.base = loc;
        ck(sp_sumcheck_quad_sharded_observed(ctx, u64p(&claim), k, T[0], T[1], tr.t, nullptr, nullptr, &Obs::fn, &obs, u64p(inner_polys.data() + 2 * loc),
                                             u64p(r_y.data() + loc), u64p(fin)),
           "inner sum-check (last rounds)");
      }
    } else {
      if (R)
        ck(sp_sumcheck_quad_sharded_partial(ctx, u64p(&claim), loc, R, ps.abc, z_inner, tr.t, reduce_hook, &comm, &Obs::fn, &obs, u64p(inner_polys.data()),
                                            u64p(r_y.data())),
           "inner sum-check (slice rounds)");
      sp_table* const sl[2] = {ps.abc, z_inner};
      gather_slices(ctx, comm, ps, sl, 2, (size_t)1 << (loc - R));
      obs.base = R;
      ck(sp_sumcheck_quad_sharded_observed(ctx, u64p(&claim), ly - R, ps.gT[0], ps.gT[1], tr.t, nullptr, nullptr, &Obs::fn, &obs, u64p(inner_polys.data() + 2 * R),
                                           u64p(r_y.data() + R), u64p(fin)),
         "inner sum-check (gathered rounds)");
    }
    claims_inner[0] = fin[0];
    claims_inner[1] = fin[1];
  }
  for (const fe_t& f : inner_polys) proof.pf(f);
  const fe_t eval_Z = claims_inner[1];
  std::vector<fe_t> X;
  X.push_back(fe_one<S>());
  X.insert(X.end(), publics.begin(), publics.end());
  const fe_t eval_X = sparse_poly_evaluate(ly - 1, X, r_y.data() + 1);
  const fe_t denom = fe_sub<S>(fe_one<S>(), r_y[0]);
  if (fe_is_zero(denom)) throw Error(SP_ERR_DIVISION_BY_ZERO, "DivisionByZero");
  const fe_t eval_W = fe_mul<S>(fe_sub<S>(eval_Z, fe_mul<S>(r_y[0], eval_X)), fe_inv_vartime<S>(denom));
  const double t_inner = now_ms();
  lap("abc_and_inner");

  // HyraxPCS::prove (hyrax_pc.rs:387-478) + InnerProductArgumentLinear::prove (ipa.rs:125-170)
  const fe_t blind_eval_W = tape.next();
  proof.pf(eval_W);
  proof.pf(blind_eval_W);
  const fe_t* point = r_y.data() + 1;
  const size_t npoint = ly - 1, nvr = log2_ceil(rows_all);
  ahead.col_point = point + nvr;
  ahead.n_col = npoint - nvr;
  ahead.publish_cols(1);  // the helper forms <R, d> from here on (when it got as far as the row challenges; otherwise below)
  const size_t ncols = (size_t)1 << (npoint - nvr);  // 2048
  aff_t comm_eval_W;
  ck(sp_hyrax_commit_small(ctx, pk.ck_s, u64p(&eval_W), 1, u64p(&blind_eval_W), u64p(&comm_eval_W.x)), "commit eval_W");
  lap("comm_eval_W");
  ahead.wait_rows_stage();  // (the helper may still be forming <R, d>: joined in front of beta)
  if (ahead.err) {
    ahead.publish_cols(2);
    ahead.join();
    std::rethrow_exception(ahead.err);
  }
  if (ncols != ncols_ipa) throw Error(SP_ERR_INTERNAL, "sharded prover: IPA width mismatch");
  ck(sp_transcript_absorb_prepared(tr.t, ahead.poly_com), "poly_com");  // the state itself is released by `ahead`
  tr.dom_sep("inner product argument (linear)");
  const std::vector<fe_t>& dvec = ahead.dvec;
  tape.skip(ncols + 2);  // d, r_delta, r_beta: drawn by the helper from these very positions
  const fe_t r_delta = ahead.r_delta, r_beta = ahead.r_beta;
  // one exchange: [partial L.W (ncols F) | partial comm_LZ (point) | partial <d, ck> (point)]
  std::vector<fe_t> LZ(ncols);
  aff_t comm_LZ, delta;
  {
    const size_t rec = ncols + 4;  // in field-element units (a point = 2 elements)
    std::vector<fe_t> mine(rec), all(rec * world);
    std::vector<fe_t> L;  // eq(r_rows, .): only the paths the helper did not take ahead need it on this thread
    if (ahead.vec_job) {  // begun by the helper behind the row challenges, weights eq(r[k..nvr), .): finish, then the rank's factor
      sp_vec_job* j = ahead.vec_job;
      ahead.vec_job = nullptr;
      ck(sp_rowmat_vec_eq_finish(ctx, j, u64p(mine.data())), "bind_with_delayed (finish)");
      if (k) {
        fe_t sc = fe_one<S>();
        for (size_t i = 0; i < k; ++i) sc = fe_mul<S>(sc, ((g >> (k - 1 - i)) & 1) ? point[i] : fe_sub<S>(fe_one<S>(), point[i]));
        for (size_t i = 0; i < ncols; ++i) mine[i] = fe_mul<S>(mine[i], sc);
      }
    } else {
      L = eq_evals_host(point, nvr);
      ck(sp_rowmat_vec(ctx, ps.Wblk, rpr, ncols, u64p(L.data() + lo), u64p(mine.data())), "bind_with_delayed (row block)");
    }
    lap("LtW_finish");
    if (ahead.lz_job) {  // started by the helper at round nvr: finish, then the rank's factor eq(r[0..k), rank bits)
      sp_msm_job* j = ahead.lz_job;
      ahead.lz_job = nullptr;
      aff_t part;
      ck(sp_msm_job_finish(ctx, j, u64p(&part.x)), "comm_LZ (finish)");
      if (k) {
        fe_t sc = fe_one<S>();
        for (size_t i = 0; i < k; ++i) sc = fe_mul<S>(sc, ((g >> (k - 1 - i)) & 1) ? point[i] : fe_sub<S>(fe_one<S>(), point[i]));
        ck(sp_vartime_scalar_mul(ctx, u64p(&part.x), 1, u64p(&sc), u64p(&mine[ncols])), "comm_LZ (rank factor)");
      } else {
        memcpy(&mine[ncols], &part, sizeof(aff_t));
      }
    } else {
      if (L.empty()) L = eq_evals_host(point, nvr);
      ck(sp_msm(ctx, u64p(L.data() + lo), u64p(&comm_W[lo].x), rpr, u64p(&mine[ncols])), "comm_LZ (point range)");
    }
    lap("comm_LZ_finish");
    const size_t cpr = ncols / world;
    if (cpr * world != ncols) throw Error(SP_ERR_INTERNAL, "sharded prover: key width not divisible by the number of ranks");
    {
      sp_msm_job* j = ahead.delta_job;
      ahead.delta_job = nullptr;
      if (j) ck(sp_msm_ck_finish(ctx, pk.ck, j, nullptr, u64p(&mine[ncols + 2])), "delta (point range)");
      else ck(sp_msm(ctx, u64p(dvec.data() + g * cpr), u64p(&pk.gens[g * cpr].x), cpr, u64p(&mine[ncols + 2])), "delta (point range)");
    }
    lap("lz_delta_finish");
    comm.allgather(mine.data(), rec * sizeof(fe_t), all.data());
    lap("pcs_exchange");
    std::vector<aff_t> p1(world), p2(world + 1);
    for (size_t rr = 0; rr < world; ++rr) {
      const fe_t* rp = all.data() + rr * rec;
      for (size_t i = 0; i < ncols; ++i) LZ[i] = rr == 0 ? rp[i] : fe_add<S>(LZ[i], rp[i]);
      memcpy(&p1[rr], rp + ncols, sizeof(aff_t));
      memcpy(&p2[rr], rp + ncols + 2, sizeof(aff_t));
    }
    ck(sp_fixed_base_mul_h(ctx, pk.ck, u64p(&r_delta), 1, u64p(&p2[world].x)), "h * r_delta");
    ck(sp_point_sum(u64p(&p1[0].x), world, u64p(&comm_LZ.x)), "comm_LZ (sum)");
    ck(sp_point_sum(u64p(&p2[0].x), world + 1, u64p(&delta.x)), "delta (sum)");
  }
  lap("point_sums");
  ahead.join();
  if (ahead.err) std::rethrow_exception(ahead.err);
  lap("join_helper");
  fe_t r_LZ = fe_zero(), ip = fe_zero();
  if (ahead.have_r_LZ) {
    r_LZ = ahead.r_LZ;
  } else {
    const std::vector<fe_t> L = eq_evals_host(point, nvr);
    for (size_t i = 0; i < L.size(); ++i) r_LZ = fe_add<S>(r_LZ, fe_mul<S>(L[i], r_W[i]));
  }
  if (ahead.have_ip) {
    ip = ahead.ip;
  } else {
    const std::vector<fe_t> Rv = eq_evals_host(point + nvr, npoint - nvr);
    for (size_t i = 0; i < ncols; ++i) ip = fe_add<S>(ip, fe_mul<S>(Rv[i], dvec[i]));
  }
  aff_t beta;
  ck(sp_hyrax_commit_small(ctx, pk.ck_s, u64p(&ip), 1, u64p(&r_beta), u64p(&beta.x)), "beta");
  {
    uint8_t b[128];
    point_bytes(comm_LZ, b);
    point_bytes(comm_eval_W, b + 64);
    tr.absorb("U", b, 128);
    point_bytes(delta, b);
    tr.absorb("delta", b, 64);
    point_bytes(beta, b);
    tr.absorb("beta", b, 64);
  }
  const fe_t rr = tr.squeeze("r");
  lap("beta_and_transcript");
  proof.pp(delta);
  proof.pp(beta);
  for (size_t i = 0; i < ncols; ++i) proof.pf(fe_add<S>(fe_mul<S>(rr, LZ[i]), dvec[i]));
  proof.pf(fe_add<S>(fe_mul<S>(rr, r_LZ), r_delta));
  proof.pf(fe_add<S>(fe_mul<S>(rr, blind_eval_W), r_beta));
  const double t_end = now_ms();
  lap("z_vec");
  if (phase_ms) {
    phase_ms[0] = t_wit - t_start;
    phase_ms[1] = t_mv - t_wit;
    phase_ms[2] = t_outer - t_mv;
    phase_ms[3] = t_abc - t_outer;
    phase_ms[4] = t_inner - t_abc;
    phase_ms[5] = t_end - t_inner;
    phase_ms[6] = t_end - t_start;
    phase_ms[7] = (double)comm.calls;
  }
  return proof;
}

// PCS::commit (hyrax_pc.rs:207-303) of a vector whose rows are sharded BY ROW: this rank's `n_local` elements (rows_local = n_local / width rows)
// are resident in `v`; all ranks obtain all rows (rank-major). The MSM leg of BASELINE config 4.
void sharded_commit(sp_ctx* ctx, Comm& comm, const sp_ck* ckey, const sp_table* v, size_t n_local, const uint64_t* blinds_local, bool is_small, uint64_t* out_all_rows) {
  const size_t rows_local = (n_local + DEFAULT_COMMITMENT_WIDTH - 1) / DEFAULT_COMMITMENT_WIDTH;
  std::vector<aff_t> mine(rows_local);
  ck(sp_hyrax_commit(ctx, ckey, v, 0, n_local, blinds_local, is_small ? 1 : 0, u64p(&mine[0].x)), "commit (row block)");
  comm.allgather(mine.data(), rows_local * sizeof(aff_t), out_all_rows);
}

}  // namespace spartan2

// ---- C surface for the harness ---------------------------------------------------------------------------------------------------------
using namespace spartan2;
extern "C" void ss_set_error(const char* msg);
static int catch_all_sd() {
  try {
    throw;
  } catch (const Error& e) {
    ss_set_error(e.what());
    return e.code;
  } catch (const std::exception& e) {
    ss_set_error(e.what());
    return SP_ERR_INTERNAL;
  }
}

extern "C" {
// ncclGetUniqueId: rank 0 calls this and hands the 128 bytes to every rank (the launcher's job: torch.distributed broadcast in bench.py)
int ssc_rccl_unique_id(uint8_t out[128]) {
  try {
    RcclApi& api = RcclApi::get();
    if (!api.ok()) throw Error(SP_ERR_NO_DEVICE, "librccl could not be loaded");
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) throw Error(SP_ERR_INTERNAL, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    memcpy(out, &id, 128);
    return 0;
  } catch (...) {
    return catch_all_sd();
  }
}
int ssc_comm_rccl(int device, int rank, int world, const uint8_t id_bytes[128], void** out) {
  try {
    RcclApi& api = RcclApi::get();
    if (!api.ok()) throw Error(SP_ERR_NO_DEVICE, "librccl could not be loaded");
    auto c = std::make_unique<Comm>();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->hipck(hipSetDevice(device), "comm: hipSetDevice");
    c->hipck(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking), "comm: stream");
    ncclUniqueId id;
    memcpy(&id, id_bytes, 128);
    // RCCL prints a version banner on stdout when a communicator is created; stdout belongs to the caller (bench.py prints one JSON line there),
    // so the banner is sent to stderr
    fflush(stdout);
    const int saved_stdout = dup(1);
    if (saved_stdout >= 0) dup2(2, 1);
    ncclResult_t r = api.CommInitRank(&c->nc, world, id, rank);
    fflush(stdout);
    if (saved_stdout >= 0) {
      dup2(saved_stdout, 1);
      close(saved_stdout);
    }
    if (r != ncclSuccess) throw Error(SP_ERR_INTERNAL, std::string("ncclCommInitRank: ") + api.GetErrorString(r));
    *out = c.release();
    return 0;
  } catch (...) {
    return catch_all_sd();
  }
}
int ssc_comm_callback(int rank, int world, ssc_allgather_fn fn, void* user, void** out) {
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  c->fn = fn;
  c->user = user;
  *out = c;
  return 0;
}
int ssc_comm_allgather(void* comm, const void* send, size_t bytes, void* recv) {
  try {
    ((Comm*)comm)->allgather(send, bytes, recv);
    return 0;
  } catch (...) {
    return catch_all_sd();
  }
}
void ssc_comm_stats(void* comm, uint64_t out[2]) {
  out[0] = ((Comm*)comm)->calls;
  out[1] = ((Comm*)comm)->bytes_moved;
}
// exchanges that took the copy-free small-record form (BAR-written send buffer, publish kernel, polled slot)
uint64_t ssc_comm_small_calls(void* comm) { return ((Comm*)comm)->small_calls; }
void ssc_comm_free(void* comm) { delete (Comm*)comm; }

int ssd_setup(sp_ctx* ctx, void* comm, size_t num_cons, size_t num_shared, size_t num_precommitted, size_t num_rest, size_t num_public, size_t num_challenges,
              const int64_t* Ad, const uint32_t* Ai, const uint64_t* Ap, const int64_t* Bd, const uint32_t* Bi, const uint64_t* Bp, const int64_t* Cd,
              const uint32_t* Ci, const uint64_t* Cp, void** out_pk) {
  try {
    *out_pk = sharded_setup(ctx, (Comm*)comm, make_view(num_cons, num_shared, num_precommitted, num_rest, num_public, num_challenges, Ad, Ai, Ap, Bd, Bi, Bp, Cd, Ci, Cp));
    return 0;
  } catch (...) {
    return catch_all_sd();
  }
}
void ssd_pk_free(void* pk) { delete (ShardedKey*)pk; }
void ssd_pk_info(void* pk_, uint64_t dims_out[10]) { memcpy(dims_out, &((ShardedKey*)pk_)->dims, sizeof(sp_dims)); }
int ssd_prep_prove(void* pk, const uint64_t* witness_u64, size_t n, int is_small, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, void** out_ps) {
  try {
    Tape t{tape, tape_blocks};
    *out_ps = sharded_prep_prove(*(ShardedKey*)pk, witness_u64, n, is_small != 0, t);
    if (tape_used) *tape_used = t.pos;
    return 0;
  } catch (...) {
    return catch_all_sd();
  }
}
void ssd_prep_free(void* ps) { delete (ShardedPrep*)ps; }
int ssd_prove(void* pk, void* ps, const uint64_t* publics_u64, size_t npub, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, uint64_t* out_words,
              size_t out_cap, double* phase_ms /* 8 */) {
  try {
    Tape t{tape, tape_blocks};
    SpartanProofBuf pf = sharded_prove(*(ShardedKey*)pk, *(ShardedPrep*)ps, publics_u64, npub, t, phase_ms);
    if (pf.words.size() > out_cap) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "proof buffer too small");
    memcpy(out_words, pf.words.data(), pf.words.size() * 8);
    if (tape_used) *tape_used = t.pos;
    return 0;
  } catch (...) {
    return catch_all_sd();
  }
}
int ssd_commit(sp_ctx* ctx, void* comm, const sp_ck* ckey, const sp_table* v, size_t n_local, const uint64_t* blinds_local, int is_small, uint64_t* out_all_rows) {
  try {
    sharded_commit(ctx, *(Comm*)comm, ckey, v, n_local, blinds_local, is_small != 0, out_all_rows);
    return 0;
  } catch (...) {
    return catch_all_sd();
  }
}
}
