// Host side ABOVE the C ABI: the protocol driver of the reference restated in C++ and calling ONLY the entry points of
// include/spartan_hip.h for everything that scales with the instance — exactly what the reference's Rust drivers would do
// through an `extern "C"` block. Mirrors
//   SplitR1CSShape::new            src/r1cs/mod.rs:810-911      (padding / column remap)
//   SpartanSNARK::setup            src/spartan.rs:146-173
//   SpartanSNARK::prep_prove       src/spartan.rs:176-216       (+ bellpepper/r1cs.rs:359-409 precommitted_witness)
//   SpartanSNARK::prove            src/spartan.rs:219-466       (+ bellpepper/r1cs.rs:411-538, hyrax_pc.rs:387-478, ipa.rs:125-170)
//   SpartanSNARK::verify           src/spartan.rs:469-578       (matrix evaluations and opening MSMs on the device)
// for circuits without verifier challenges: any mix of shared / precommitted / rest variables (bench circuits: precommitted-only,
// skip_synthesize + commit_zeros path :443; the reference's e2e CubicCircuit src/spartan.rs:587-651: rest-only).
//
// Randomness is injected: every blind / mask is the next 64-byte block of a caller-supplied tape reduced with from_uniform,
// in the reference's call order.
// Generators: this build's own derivation (the reference's is a third-party hash-to-curve): PARITY UNPINNED, see DESIGN.md.
#include "snark_common.hpp"

namespace spartan2 {

// ---- keys / state ----------------------------------------------------------------------------------------------------------
struct SpartanProverKey {  // src/spartan.rs:30-58
  sp_ctx* ctx = nullptr;
  sp_shape* S = nullptr;
  sp_ck *ck = nullptr, *ck_s = nullptr;
  sp_dims dims;
  size_t num_vars = 0, num_extra = 0, num_cols = 0;
  uint8_t vk_digest[32];
  std::vector<aff_t> gens, gens_s;
  // verify()'s device workspaces (T_x, T_y, the three products M T_y): allocated by the first verify on this key and kept - five hipMalloc / hipFree
  // pairs of 32-64 MB were half of a 3.3 ms verify. One verify at a time per key (as for every handle of the ABI).
  mutable sp_table *v_Tx = nullptr, *v_Ty = nullptr, *v_mv[3] = {nullptr, nullptr, nullptr};
  ~SpartanProverKey() {
    sp_table_free(v_Tx);
    sp_table_free(v_Ty);
    for (sp_table* t : v_mv) sp_table_free(t);
    sp_shape_free(S);
    sp_ck_free(ck);
    sp_ck_free(ck_s);
  }
};

// How a helper thread waits for its owner. It sleeps on a condition variable that the owner notifies at each state change: measured at config 2, a
// lone prove is as fast this way as with a spinning helper (1.29-1.31 ms either way; wake-up ~10 us, off the critical path or under the MSM it
// starts), and a spinning helper costs one CPU per prove in flight on top of the owner's polling thread - under the 16-CPU CFS quota of the bench
// boxes eight such pairs got the whole cgroup throttled and resident kernels ran into their watchdog.
// A helper that waits INSIDE a prove for the owner's next notification (the opening's helper: delta once the outer sum-check is in its resident rounds,
// comm_LZ once the row challenges are drawn) may spin while few proves are in flight in the process: a sleeping thread's wake-up is the scheduler's
// to time - usually tens of microseconds, now and then milliseconds (one prove in 50 - 2000 was 2 - 5 ms long for it on the bench boxes). With many
// proves in flight (the 8-context throughput mode under a 16-CPU quota) the helpers sleep: a CPU per helper is then worth more than the odd late one.
// delta's MSM (ipa.rs:147; ~175 us of bucket kernels on the auxiliary stream) is issued by the opening's helper when the outer sum-check has this many
// rounds left (SPARTAN_DELTA_ROUNDS_LEFT, A/B): late enough to stay clear of the streaming rounds, early enough to be done before poly_ABC starts
static size_t delta_rounds_before_end() {
  static const size_t v = [] {
    const char* e = getenv("SPARTAN_DELTA_ROUNDS_LEFT");
    const int k = e ? atoi(e) : 17;
    return (size_t)(k < 1 ? 1 : k);
  }();
  return v;
}
static std::atomic<int> g_active_proves{0};
static constexpr int HELPER_SPIN_MAX_PROVES = 4;
static bool helper_may_spin() { return g_active_proves.load(std::memory_order_relaxed) <= HELPER_SPIN_MAX_PROVES; }
struct ActiveProve {
  ActiveProve() { g_active_proves.fetch_add(1, std::memory_order_relaxed); }
  ~ActiveProve() { g_active_proves.fetch_sub(1, std::memory_order_relaxed); }
};

// Per-prep-state driver options (ss_prep_set_flags; defaults from the environment at prep_prove time).
//   FLAG_PREFIX_CACHE: the transcript prefix new + vk + public_values + comm_W_shared / comm_W_precommitted is the same for every prove on one prep
//     state; with this flag its sponge state is computed once and cloned (Keccak256Transcript is Clone, keccak.rs:25) — an API-level optimisation
//     the reference does not make (it re-hashes the prefix in every prove, src/spartan.rs:226-236, r1cs.rs:422-427). Default OFF: the prefix is
//     re-hashed inside every prove (on a helper thread, under commit_zeros), so the timed region does the reference's work.
//   FLAG_LZ_DIRECT: the opening in the reference's own order (bind W with L, then the MSM over the key) instead of the MSM over the row commitments.
//   FLAG_REFERENCE_ORDER: prove_reference_order below — ONE thread, no helper jobs, only include/spartan_hip.h entry points (the one prep-time product it uses
//     is the FixedBaseMul table set of the committed rows, handed to sp_hyrax_prove_announce_tables), called
//     in the order of the statements of src/spartan.rs:226-466 (what an unchanged spartan.rs bound to the ABI does). bench.py reports it beside the headline.
enum : unsigned { FLAG_PREFIX_CACHE = 1u, FLAG_LZ_DIRECT = 2u, FLAG_REFERENCE_ORDER = 4u };

struct SpartanPrepSNARK {  // src/spartan.rs:107-124
  sp_table *W = nullptr, *caz = nullptr, *cbz = nullptr, *ccz = nullptr;      // witness + cached partial products
  sp_table *az = nullptr, *bz = nullptr, *cz = nullptr, *z = nullptr;          // scratch reused across prove calls
  sp_table *rx = nullptr, *abc = nullptr;
  sp_table *p0 = nullptr, *p1 = nullptr;  // per-pair products of the outer sum-check's first round (sp_multiply_vec_incremental_round0)
  std::vector<aff_t> comm_W_fixed;  // rows committed at prep time: shared rows, then precommitted rows
  std::vector<fe_t> r_W_fixed;      // their blinds, same order
  size_t rows_shared = 0, rows_precommitted = 0;
  std::vector<uint8_t> comm_shared_bytes, comm_pre_bytes;  // transcript encodings (hyrax_pc.rs:714-729)
  bool is_small = true;
  sp_transcript* tr_prefix = nullptr;  // FLAG_PREFIX_CACHE: transcript state after the per-instance prefix (see prove)
  std::vector<fe_t> tr_publics;
  sp_transcript* tr_fresh = nullptr;   // default: the prefix re-hashed for this prove by the second helper
  Background bg2;
  unsigned flags = 0;
  Background bg;                        // hashes comm_W for the PCS transcript step while the sum-checks run, then starts comm_LZ's MSM
  sp_absorb_state* poly_com = nullptr;  // its result
  sp_points* comm_pts = nullptr;        // comm_W on the device: the bases of comm_LZ's MSM
  // FixedBaseMul tables of the rows committed here (shared, precommitted) and of h (msm.rs:653-689): with no rest variables the remaining rows of
  // comm_W are h * blind (commit_zeros), so comm_LZ = sum_fixed L_i comm_W[i] + (sum_rest L_i blind_i) h is one multi_mul over these tables
  // (sp_fbtables_multi_mul: 14 levels of additions in one launch instead of a 512-point MSM behind the last row challenge).
  // Up to 512 rows (2^20 variables): measured at 1024 / 2048 rows the walk (a 17-level chain over 256 / 512 blocks) loses to the MSM it would replace
  // (1.77 vs 1.70 ms, 2.58 vs 2.51 ms): there the sum-check's last rounds no longer cover it and its blocks compete with the streaming rounds.
  sp_fbtables* lz_tables = nullptr;
  std::vector<aff_t> lz_points;  // their bases (the committed rows, then h) while the build is still to be queued
  int lz_mode = 0;               // 1: queue the build behind the first prove on this state (prep_prove, SPARTAN_PREP_TABLES)
  void queue_lz_tables(sp_ctx* ctx) {  // end of a prove
    if (lz_mode != 1 || lz_tables || lz_points.empty()) return;
    lz_mode = 0;
    if (sp_fbtables_create_async(ctx, u64p(&lz_points[0].x), lz_points.size(), &lz_tables) != SP_OK) lz_tables = nullptr;  // (the MSM over the rows stays)
  }
  // host wall-clock of prep_prove's phases, ms (ss_prep_phases): witness (u64 -> elements on the device), commit (PCS::commit of the shared and
  // precommitted rows), tables (FixedBaseMul tables of the committed rows), matvec (multiply_vec_precommitted), scratch (allocations), total
  double prep_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ~SpartanPrepSNARK() {
    sp_fbtables_free(lz_tables);
    bg.wait_nothrow();
    bg2.wait_nothrow();
    sp_transcript_free(tr_fresh);
    sp_points_free(comm_pts);
    sp_absorb_state_free(poly_com);
    sp_transcript_free(tr_prefix);
    for (sp_table* t : {W, caz, cbz, ccz, az, bz, cz, z, rx, abc, p0, p1}) sp_table_free(t);
  }
};

struct PhaseTimes {
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // witness_commit, matvec, outer, poly_abc, inner, pcs, total, (spare)
  std::vector<std::pair<std::string, double>> laps;  // fine-grained host-side laps (diagnostics)
};

// SpartanSNARK::setup (src/spartan.rs:146-173)
SpartanProverKey* setup(sp_ctx* ctx, const R1CSIntView& R) {
  auto* pk = new SpartanProverKey();
  try {
    pk->ctx = ctx;
    PaddedShape P = pad_shape(R);
    pk->dims = P.dims;
    pk->num_vars = P.num_vars();
    pk->num_extra = 1 + P.dims.num_public + P.dims.num_challenges;
    pk->num_cols = P.num_cols();
    sp_csr cs[3];
    for (int m = 0; m < 3; ++m) cs[m] = sp_csr{u64p(P.data[m].data()), P.idx[m].data(), P.ptr[m].data()};
    ck(sp_shape_from_csr(ctx, &cs[0], &cs[1], &cs[2], &P.dims, &pk->S), "shape_from_csr");
    pk->gens = from_label("ck", DEFAULT_COMMITMENT_WIDTH + 1);  // commitment_key (src/r1cs/mod.rs:1031-1043) -> PCS::setup(b"ck", ., 2048)
    pk->gens_s = from_label("ck_s", 2);                          // PCS::setup(b"ck_s", 1, 1)
    ck(sp_ck_create(ctx, u64p(&pk->gens[0].x), DEFAULT_COMMITMENT_WIDTH, u64p(&pk->gens[DEFAULT_COMMITMENT_WIDTH].x), &pk->ck), "ck_create");
    ck(sp_ck_create(ctx, u64p(&pk->gens_s[0].x), 1, u64p(&pk->gens_s[1].x), &pk->ck_s), "ck_s_create");
    spartan_vk_digest(P, pk->gens, pk->gens_s, pk->vk_digest);
  } catch (...) {
    delete pk;
    throw;
  }
  return pk;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// SpartanSNARK::prep_prove (src/spartan.rs:176-216)
SpartanPrepSNARK* prep_prove(const SpartanProverKey& pk, const uint64_t* witness_u64, size_t n_witness, bool is_small, Tape& tape) {
  const sp_dims& d = pk.dims;
  if (n_witness != d.num_shared_unpadded + d.num_precommitted_unpadded + d.num_rest_unpadded) throw Error(SP_ERR_INVALID_WITNESS_LENGTH, "InvalidWitnessLength");
  auto* ps = new SpartanPrepSNARK();
  ps->bg.set_spin_policy(&helper_may_spin);
  ps->bg2.set_spin_policy(&helper_may_spin);
  try {
    sp_ctx* ctx = pk.ctx;
    const size_t M = pk.num_vars, N = d.num_cons;
    ps->is_small = is_small;
    const double t_begin = now_ms();
    double t_last = t_begin;
    auto phase = [&](int k) {
      const double t = now_ms();
      ps->prep_ms[k] += t - t_last;
      t_last = t;
    };
    {
      const char* e = getenv("SPARTAN_LZ_DIRECT");
      if (e && e[0] == '1') ps->flags |= FLAG_LZ_DIRECT;
      e = getenv("SPARTAN_PREFIX_CACHE");
      if (e && e[0] == '1') ps->flags |= FLAG_PREFIX_CACHE;
    }
    // shared_witness / precommitted_witness (bellpepper/r1cs.rs:306-409): each segment starts at its padded offset. The rest
    // segment is filled here as well: without verifier challenges it does not change between proves.
    // The words go to the device as they are (8 bytes a value) and become Montgomery-form elements there: sp_table_write_u64 (the reference hands the same
    // machine words to msm_small, hyrax_pc.rs:266-292, and never builds wide scalars for the commitment either).
    ck(sp_table_zeros(ctx, M, (size_t)-1, (size_t)-1, &ps->W), "alloc W");
    auto put = [&](size_t dst, size_t src, size_t cnt) { ck(sp_table_write_u64(ctx, ps->W, dst, witness_u64 + src, cnt), "upload W"); };
    put(0, 0, d.num_shared_unpadded);
    put(d.num_shared, d.num_shared_unpadded, d.num_precommitted_unpadded);
    // (with verifier challenges the rest segment is synthesized inside every prove, after the challenges are drawn: bellpepper/r1cs.rs:443-461)
    if (d.num_challenges == 0) put(d.num_shared + d.num_precommitted, d.num_shared_unpadded + d.num_precommitted_unpadded, d.num_rest_unpadded);
    phase(0);
    const size_t CW = DEFAULT_COMMITMENT_WIDTH;
    ps->rows_shared = d.num_shared_unpadded ? (d.num_shared + CW - 1) / CW : 0;
    ps->rows_precommitted = d.num_precommitted_unpadded ? (d.num_precommitted + CW - 1) / CW : 0;
    ps->r_W_fixed.resize(ps->rows_shared + ps->rows_precommitted);  // PCS::blind (hyrax_pc.rs:192-205), shared first
    for (auto& b : ps->r_W_fixed) b = tape.next();
    ps->comm_W_fixed.resize(ps->r_W_fixed.size());
    if (ps->rows_shared) {
      ck(sp_hyrax_commit(ctx, pk.ck, ps->W, 0, d.num_shared, u64p(ps->r_W_fixed.data()), is_small ? 1 : 0, u64p(&ps->comm_W_fixed[0].x)), "commit shared");
      ps->comm_shared_bytes = commitment_bytes(ps->comm_W_fixed.data(), ps->rows_shared);
    }
    if (ps->rows_precommitted) {
      ck(sp_hyrax_commit(ctx, pk.ck, ps->W, d.num_shared, d.num_precommitted, u64p(ps->r_W_fixed.data() + ps->rows_shared), is_small ? 1 : 0,
                         u64p(&ps->comm_W_fixed[ps->rows_shared].x)),
         "commit precommitted");
      ps->comm_pre_bytes = commitment_bytes(ps->comm_W_fixed.data() + ps->rows_shared, ps->rows_precommitted);
    }
    phase(1);
    {
      const size_t fixed = ps->comm_W_fixed.size(), rows_all = (M + CW - 1) / CW;
      if (!(ps->flags & FLAG_LZ_DIRECT) && d.num_rest_unpadded == 0 && d.num_challenges == 0 && fixed >= 1 && fixed + 1 <= 512 && rows_all > 1) {
        // WHEN they are built (SPARTAN_PREP_TABLES): the build is ~3 ms of device work (profiles/r06_prep_prove.md) that pays for itself from the second prove
        // on a state onwards (50-100 us a prove) and slows whatever proves beside it. "lazy" (default): queued behind the FIRST prove on the state - a state
        // that is proven once (the reference's own use: prep_prove, prove, drop) never pays, the bench's repeated proves get them from the third on;
        // "prep": queued here; "sync": built here and waited for (round 5); "off": never.
        const char* mode = getenv("SPARTAN_PREP_TABLES");
        ps->lz_points.assign(ps->comm_W_fixed.begin(), ps->comm_W_fixed.end());
        ps->lz_points.push_back(pk.gens[CW]);  // h
        ps->lz_mode = !mode || !strcmp(mode, "lazy") ? 1 : (!strcmp(mode, "off") ? 0 : 2);
        if (mode && !strcmp(mode, "sync")) ck(sp_fbtables_create(ctx, u64p(&ps->lz_points[0].x), ps->lz_points.size(), &ps->lz_tables), "tables of the committed rows");
        else if (ps->lz_mode == 2) ck(sp_fbtables_create_async(ctx, u64p(&ps->lz_points[0].x), ps->lz_points.size(), &ps->lz_tables), "tables of the committed rows");
      }
    }
    phase(2);
    // multiply_vec_precommitted (src/r1cs/mod.rs:1112-1128): z = [W_cached | 0 ...]
    ck(sp_table_zeros(ctx, 2 * M, (size_t)-1, (size_t)-1, &ps->z), "alloc z");
    ck(sp_table_copy(ctx, ps->z, 0, ps->W, 0, d.num_shared + d.num_precommitted), "copy W");
    ck(sp_table_set_len(ps->z, pk.num_cols, (size_t)-1, (size_t)-1), "z len");
    for (sp_table** t : {&ps->caz, &ps->cbz, &ps->ccz, &ps->az, &ps->bz, &ps->cz}) ck(sp_table_zeros(ctx, N, (size_t)-1, (size_t)-1, t), "alloc Az");
    ck(sp_multiply_vec(ctx, pk.S, ps->z, ps->caz, ps->cbz, ps->ccz), "multiply_vec_precommitted");
    if (getenv("SPARTAN_PREP_TRACE")) ck(sp_ctx_synchronize(ctx), "sync");
    phase(3);
    ck(sp_table_zeros(ctx, N, (size_t)-1, (size_t)-1, &ps->rx), "alloc rx");
    const char* no_p0 = getenv("SPARTAN_ROUND0_PRODUCTS");  // "0": round 1 of the outer sum-check evaluates straight from Az, Bz, Cz (A/B)
    if (N >= 2 && !(no_p0 && no_p0[0] == '0')) {
      ck(sp_table_zeros(ctx, N / 2, (size_t)-1, (size_t)-1, &ps->p0), "alloc round-0 products");
      ck(sp_table_zeros(ctx, N / 2, (size_t)-1, (size_t)-1, &ps->p1), "alloc round-0 products");
    }
    ck(sp_table_zeros(ctx, 2 * M, (size_t)-1, (size_t)-1, &ps->abc), "alloc poly_ABC");
    ck(sp_ctx_synchronize(ctx), "sync");
    phase(4);
    ps->prep_ms[6] = now_ms() - t_begin;
    if (getenv("SPARTAN_PREP_TRACE"))
      fprintf(stderr, "prep_prove: witness %.3f commit %.3f tables %.3f matvec %.3f scratch %.3f total %.3f ms\n", ps->prep_ms[0], ps->prep_ms[1], ps->prep_ms[2], ps->prep_ms[3],
              ps->prep_ms[4], ps->prep_ms[6]);
  } catch (...) {
    delete ps;
    throw;
  }
  return ps;
}


// SpartanSNARK::prove (src/spartan.rs:219-466)
// `synth`: circuit.synthesize(.., Some(&challenges)) for circuits with verifier challenges (bellpepper/r1cs.rs:443-461): receives the challenges and
// writes the num_rest_unpadded values of the rest segment (Montgomery limbs); non-zero return = SynthesisError
typedef int (*ss_rest_hook)(void* user, const uint64_t* challenges, size_t num_challenges, uint64_t* out_rest);
SpartanProofBuf prove_reference_order(const SpartanProverKey& pk, SpartanPrepSNARK& ps, const uint64_t* publics_u64, size_t npub, Tape& tape, PhaseTimes* pt, ss_rest_hook synth,
                                      void* synth_user);
SpartanProofBuf prove(const SpartanProverKey& pk, SpartanPrepSNARK& ps, const uint64_t* publics_u64, size_t npub, Tape& tape, PhaseTimes* pt, ss_rest_hook synth = nullptr,
                      void* synth_user = nullptr) {
  if (ps.flags & FLAG_REFERENCE_ORDER) return prove_reference_order(pk, ps, publics_u64, npub, tape, pt, synth, synth_user);
  const sp_dims& d = pk.dims;
  sp_ctx* ctx = pk.ctx;
  const size_t M = pk.num_vars, N = d.num_cons, W_ = DEFAULT_COMMITMENT_WIDTH;
  if (npub != d.num_public) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "public_values length");
  ck(sp_ctx_bind_thread(ctx), "device");  // the caller may be a thread other than the one that created the context
  const ActiveProve active_prove;
  const double t_start = now_ms();
  double t_lap = t_start;
  auto lap = [&](const char* name) {
    if (!pt) return;
    double t = now_ms();
    pt->laps.emplace_back(name, t - t_lap);
    t_lap = t;
  };
  // First thing: the commitment to the rest segment (bellpepper/r1cs.rs:463-491) when it is all padding, i.e. commit_zeros (hyrax_pc.rs:305-319) = h * blind
  // per row. Its kernel (~30 us of dependent point additions) + the host's batch normalisation + the absorb of the rows + the squeeze of tau is the chain
  // the outer sum-check's first evaluation waits for, so the blinds are drawn and the kernel launched before anything else on this thread.
  const size_t rows_pre = ps.comm_W_fixed.size(), rows_rest = (d.num_rest + W_ - 1) / W_;
  std::vector<fe_t> r_W_rest(rows_rest);
  for (auto& b : r_W_rest) b = tape.next();
  lap("rest_blinds");
  sp_fb_job* rest_job = nullptr;
  struct FbJobGuard {  // an error exit between begin and finish must not leave the context's one asynchronous fixed-base job outstanding
    sp_ctx* ctx;
    sp_fb_job*& job;
    size_t n;
    ~FbJobGuard() {
      if (job) {
        std::vector<uint64_t> sink(8 * n + 8);
        (void)sp_fixed_base_mul_h_finish(ctx, job, sink.data());
        job = nullptr;
      }
    }
  } rest_job_guard{ctx, rest_job, rows_rest};
  if (rows_rest && d.num_rest_unpadded == 0) ck(sp_fixed_base_mul_h_begin(ctx, pk.ck, u64p(r_W_rest.data()), rows_rest, &rest_job), "commit_zeros (begin)");
  lap("commit_zeros_begin");
  std::vector<fe_t> publics(npub);
  for (size_t i = 0; i < npub; ++i) publics[i] = fe_from_u64<S>(publics_u64[i]);
  lap("publics");

  // ONE job for the second helper, submitted before anything else, carries the two host chains nothing on this thread depends on until much later:
  //   (1) the transcript prefix (src/spartan.rs:226-236, bellpepper/r1cs.rs:422-427): new + vk + public_values + comm_W_shared + comm_W_precommitted -
  //       about 300 Keccak blocks at config 2, ~45 us. Re-hashed in every prove, as the reference does (FLAG_PREFIX_CACHE keeps the sponge state
  //       across proves instead). Needed when comm_W_rest is absorbed, i.e. after commit_zeros has come back from the device: `prefix_ready`.
  //   (2) the IPA mask d_vec (ipa.rs:139-145) and the blinds of delta and beta, which do not depend on the transcript: 2048 wide reductions from their
  //       tape position, ~70 us. Needed when delta's MSM is issued (a third of the way into the outer sum-check): `dvec_ready`.
  // The first helper stays free for the opening's job, which is posted the moment comm_W is complete.
  const size_t n_ipa = M < W_ ? M : W_;
  const std::unique_ptr<fe_t[]> dvec_store(new fe_t[n_ipa]);  // (not value-initialised: the helper's draw writes every element; a zeroed vector was 12 us here)
  fe_t* const dvec = dvec_store.get();
  fe_t r_delta_ahead, r_beta_ahead, eval_W_term_blind;
  aff_t eval_W_term, beta_term;  // h_s * blind_eval_W, h_s * r_beta (valid once eval_W_term_ready is set)
  std::atomic<int> prefix_ready{0}, dvec_ready{0}, eval_W_term_ready{0};
  const bool prefix_cached = (ps.flags & FLAG_PREFIX_CACHE) != 0;
  if (prefix_cached) {
    if (!ps.tr_prefix) {
      Tr t0(ctx, "SpartanSNARK");
      t0.absorb("vk", pk.vk_digest, 32);
      t0.absorb_scalars("public_values", publics.data(), npub);
      if (ps.rows_shared) t0.absorb("comm_W_shared", ps.comm_shared_bytes.data(), ps.comm_shared_bytes.size());
      if (ps.rows_precommitted) t0.absorb("comm_W_precommitted", ps.comm_pre_bytes.data(), ps.comm_pre_bytes.size());
      ps.tr_prefix = t0.t;
      t0.t = nullptr;
      ps.tr_publics = publics;
    } else if (ps.tr_publics.size() != publics.size() || memcmp(ps.tr_publics.data(), publics.data(), publics.size() * sizeof(fe_t)) != 0) {
      throw Error(SP_ERR_INTERNAL, "public values changed between proves on one prep state");
    }
  }
  {
    SpartanPrepSNARK* psp = &ps;
    const SpartanProverKey* pkp = &pk;
    const fe_t* pub = publics.data();
    // tape order: the rest-row blinds (drawn below), blind_eval_W, then d_vec, r_delta, r_beta
    Tape peek{tape.bytes, tape.blocks, tape.pos + 1};  // (the rest-row blinds have been drawn above)
    fe_t* dv = dvec;
    const size_t dn = n_ipa;
    fe_t *rd = &r_delta_ahead, *rb = &r_beta_ahead;
    std::atomic<int>*pr = &prefix_ready, *dr = &dvec_ready;
    // (3) the blind's term of comm_eval_W (src/spartan.rs:423-437): blind_eval_W is the next block of the tape, h_s * blind a host walk of 32 additions
    //     that otherwise stands between the inner sum-check's last round and the opening's transcript
    const fe_t bew = fe_from_uniform<S>(tape.bytes + 64 * (tape.pos < tape.blocks ? tape.pos : 0));
    aff_t* ewt = &eval_W_term;
    fe_t* ewb = &eval_W_term_blind;
    std::atomic<int>* ewr = &eval_W_term_ready;
    aff_t* btt = &beta_term;
    ps.bg2.submit([ctx, psp, pkp, pub, npub, prefix_cached, peek, dv, dn, rd, rb, pr, dr, bew, ewt, ewb, ewr, btt]() mutable {
      struct Flags {  // whatever happens in here, a waiter must not spin for ever (an exception resurfaces at the next wait())
        std::atomic<int>*a, *b;
        ~Flags() {
          a->store(1, std::memory_order_release);
          b->store(1, std::memory_order_release);
        }
      } flags{pr, dr};
      if (!prefix_cached) {
        sp_transcript_free(psp->tr_fresh);
        psp->tr_fresh = nullptr;
        Tr t0(ctx, "SpartanSNARK");
        t0.absorb("vk", pkp->vk_digest, 32);
        t0.absorb_scalars("public_values", pub, npub);
        if (psp->rows_shared) t0.absorb("comm_W_shared", psp->comm_shared_bytes.data(), psp->comm_shared_bytes.size());
        if (psp->rows_precommitted) t0.absorb("comm_W_precommitted", psp->comm_pre_bytes.data(), psp->comm_pre_bytes.size());
        psp->tr_fresh = t0.t;
        t0.t = nullptr;
      }
      pr->store(1, std::memory_order_release);
      for (size_t i = 0; i < dn; ++i) dv[i] = peek.next();
      *rd = peek.next();  // ipa.rs:146: the blind of delta follows d_vec on the tape
      *rb = peek.next();  // then beta's
      dr->store(1, std::memory_order_release);
      // (both blind terms of the opening's one-value commitments: comm_eval_W above and beta = <R, d> g_s + r_beta h_s, ipa.rs:148-149)
      aff_t two[2];
      const fe_t bl[2] = {bew, *rb};
      if (sp_fixed_base_mul_h(ctx, pkp->ck_s, u64p(bl), 2, u64p(&two[0].x)) == SP_OK) {
        *ewt = two[0];
        *ewb = bew;
        *btt = two[1];
        ewr->store(1, std::memory_order_release);
      }
    });
  }
  lap("helper_submit");
  // a flag the second helper's job raises; if the helper has not even begun the job 30 us after it was posted, the job runs here (Background::try_steal)
  auto await_flag = [&ps](std::atomic<int>& flag) {
    for (unsigned spins = 0; flag.load(std::memory_order_acquire) == 0; ++spins) {
      if ((spins & 63u) == 63u) ps.bg2.try_steal();
      sp_relax();
    }
  };
  struct PrefixJoin {  // publics must outlive the job on every exit path
    Background& b;
    ~PrefixJoin() { b.wait_nothrow(); }
  } prefix_join{ps.bg2};
  lap("transcript_prefix");
  Tr tr(nullptr, Tr::Adopt{});
  auto acquire_transcript = [&] {
    if (tr.t) return;
    if (prefix_cached) ck(sp_transcript_clone(ps.tr_prefix, &tr.t), "transcript_clone");
    else {
      await_flag(prefix_ready);
      tr.t = ps.tr_fresh;
      ps.tr_fresh = nullptr;
      if (!tr.t) {
        ps.bg2.wait();  // (rethrows what the job threw)
        throw Error(SP_ERR_INTERNAL, "transcript prefix was not prepared");
      }
    }
  };
  // verifier challenges (bellpepper/r1cs.rs:429-431) and the rest of the witness that depends on them (:443-461): squeezed right after the
  // precommitted commitment, before anything that needs z
  std::vector<fe_t> challenges(d.num_challenges);
  if (d.num_challenges) {
    if (!synth) throw Error(SP_ERR_INTERNAL, "a circuit with verifier challenges needs its synthesize callback");
    acquire_transcript();
    for (auto& c : challenges) c = tr.squeeze("challenge");
    std::vector<fe_t> rest(d.num_rest_unpadded + 1);
    if (synth(synth_user, u64p(challenges.data()), challenges.size(), u64p(rest.data())) != 0) throw Error(SP_ERR_INTERNAL, "SynthesisError: the circuit's synthesize callback failed");
    if (d.num_rest_unpadded) ck(sp_table_write(ctx, ps.W, d.num_shared + d.num_precommitted, u64p(rest.data()), d.num_rest_unpadded), "W rest");
  }

  // z = [W | 1 | public | challenges]   (src/spartan.rs:246-253); the table is 2M long so the inner sum-check can run in place
  // The matrix-vector product below reads only the rest / public / challenge columns of z (multiply_vec_incremental_into, src/r1cs/mod.rs:1170-1211: the
  // precommitted part is cached), so only those go through the main stream in front of it; the bulk - the shared and precommitted columns and the zeros
  // of the high half, 96 MB at config 2 - is assembled beside the main stream and joined in front of the outer sum-check (its first reader is the inner one).
  ck(sp_table_set_len(ps.z, 2 * M, (size_t)-1, (size_t)-1), "z len");
  // (a rest segment that is all padding has no matrix entries in its columns: it goes with the bulk)
  const size_t z_fixed = d.num_rest_unpadded ? d.num_shared + d.num_precommitted : M;
  if (M > z_fixed) ck(sp_table_copy(ctx, ps.z, z_fixed, ps.W, z_fixed, M - z_fixed), "z <- W rest");
  {
    std::vector<fe_t> tail(pk.num_extra);
    tail[0] = fe_one<S>();
    std::copy(publics.begin(), publics.end(), tail.begin() + 1);
    std::copy(challenges.begin(), challenges.end(), tail.begin() + 1 + npub);
    if (tail.size() <= 2048) ck(sp_table_write_async(ctx, ps.z, M, u64p(tail.data()), tail.size()), "z tail");  // no synchronisation in front of the matrix-vector product
    else ck(sp_table_write(ctx, ps.z, M, u64p(tail.data()), tail.size()), "z tail");
  }
  ck(sp_table_set_len(ps.z, pk.num_cols, (size_t)-1, (size_t)-1), "z len");
  lap("z_build");
  // Az, Bz, Cz (src/spartan.rs:271-279) depend only on z: issue the product now so that it runs under the commit_zeros job and the host work
  // below instead of after them (the reference computes it after the commitments; the values are the same)
  const double t_mv0 = now_ms();
  if (ps.p0) ck(sp_multiply_vec_incremental_round0(ctx, pk.S, ps.z, ps.caz, ps.cbz, ps.ccz, ps.az, ps.bz, ps.cz, ps.p0, ps.p1), "multiply_vec_incremental");
  else ck(sp_multiply_vec_incremental(ctx, pk.S, ps.z, ps.caz, ps.cbz, ps.ccz, ps.az, ps.bz, ps.cz), "multiply_vec_incremental");
  lap("spmv_issue");
  ck(sp_table_assemble_aside(ctx, ps.z, 0, ps.W, 0, z_fixed, M + pk.num_extra, M - pk.num_extra, 0), "z bulk");
  ck(sp_ctx_aside_join(ctx), "z bulk (join)");  // queued behind the product: by the time the main stream gets there the bulk of z has long been written
  const double t_mv_issue = now_ms() - t_mv0;
  // (the rest segment's commitment: commit_zeros was started at the top and is collected here, after everything the device can be given in the meantime has
  // been issued; a segment with values takes PCS::commit on the resident witness)
  std::vector<aff_t> comm_W(rows_pre + rows_rest);
  std::copy(ps.comm_W_fixed.begin(), ps.comm_W_fixed.end(), comm_W.begin());
  lap("z_bulk_issue");

  const bool rest_job_used = rest_job != nullptr;  // the rest rows are h * blind (commit_zeros)
  if (rest_job) {
    sp_fb_job* j = rest_job;
    rest_job = nullptr;  // finish() consumes the job whatever it returns
    ck(sp_fixed_base_mul_h_finish(ctx, j, u64p(&comm_W[rows_pre].x)), "commit_zeros (finish)");
  }
  else if (rows_rest)
    ck(sp_hyrax_commit(ctx, pk.ck, ps.W, d.num_shared + d.num_precommitted, d.num_rest, u64p(r_W_rest.data()), ps.is_small ? 1 : 0, u64p(&comm_W[rows_pre].x)),
       "commit rest");
  lap("commit_zeros_finish");
  acquire_transcript();
  lap("prefix_join");
  {
    std::vector<uint8_t> b = commitment_bytes(comm_W.data() + rows_pre, rows_rest);
    tr.absorb("comm_W_rest", b.data(), b.size());
  }
  lap("absorb_comm_W_rest");
  std::vector<fe_t> r_W = ps.r_W_fixed;  // combine_blinds (bellpepper/r1cs.rs:515-524)
  r_W.insert(r_W.end(), r_W_rest.begin(), r_W_rest.end());
  // comm_W is complete: its transcript encoding (64 B per row, Montgomery -> canonical) and the Keccak blocks of absorb("poly_com", ..), the
  // first absorb after the inner sum-check's last squeeze (hyrax_pc.rs:387-400), are computed on the helper thread from here on
  ps.bg.wait();  // (idle: nothing has been posted to the first helper in this prove yet)
  sp_absorb_state_free(ps.poly_com);
  ps.poly_com = nullptr;
  // The same helper then computes comm_LZ = sum_i L[i] comm_W[i] (= commit(L . W; <L, r_W>), hyrax_pc.rs:430-455, by the homomorphism of the
  // commitment) as soon as the inner sum-check has bound the row variables: the MSM runs under the remaining rounds instead of after them.
  const size_t lz_rows = (M + W_ - 1) / W_, lz_nvr = log2_ceil(lz_rows);
  const bool lz_direct = (ps.flags & FLAG_LZ_DIRECT) != 0;  // the reference's own order (bind W with L first, then the MSM over the key)
  struct LzAhead {
    std::atomic<int> state{0};  // 0: row challenges not drawn yet, 1: drawn, 2: abandoned
    std::atomic<int> rows_known{0};  // how many of them r[] holds so far (the helper expands eq(r, .) a level per challenge, as they arrive)
    size_t nvr = 0;
    fe_t r[24];
    sp_msm_job* job = nullptr;
    sp_vec_job* vec = nullptr;
    fe_t r_LZ;
    std::atomic<int> delta_state{0};  // 0: delta's MSM not issued yet, 1: issued, 2: abandoned
    sp_msm_job* delta_job = nullptr;
    fe_t r_delta;
    aff_t delta;
    bool delta_done = false;
    std::atomic<int> early_done{0};  // poly_com prepared and delta finished: what the PCS phase needs first (1 = done, 2 = gave up)
    aff_t comm_LZ;                   // the tables path (SpartanPrepSNARK::lz_tables): the helper delivers comm_LZ itself
    bool have_comm_LZ = false;
    // a helper that may not spin sleeps on this pair; the owner notifies after each state change (a notify without a sleeper is a user-space check)
    std::mutex mu;
    std::condition_variable cv;
    void publish(std::atomic<int>& flag, int v) {
      flag.store(v, std::memory_order_release);
      cv.notify_one();
    }
  } lz;
  // (lz.r_delta is set by the opening's job once d_vec and the blinds are drawn: dvec_ready)
  lz.nvr = lz_nvr;
  const bool lz_ahead = !lz_direct && lz_nvr > 0 && lz_nvr <= 20 && comm_W.size() == ((size_t)1 << lz_nvr) && r_W.size() == comm_W.size();
  const size_t lz_cols = (size_t)1 << (log2_ceil(M) - lz_nvr);
  // (tables whose build - queued by prep_prove a moment ago - has not ended are not waited for: this prove takes the MSM over the row commitments)
  const bool lz_tables_path = lz_ahead && ps.lz_tables && rest_job_used && ps.comm_W_fixed.size() + rows_rest == comm_W.size() && sp_fbtables_ready(ps.lz_tables, 0) == 1;
  {
    const aff_t* rows = comm_W.data();
    const size_t nrows = comm_W.size();
    SpartanPrepSNARK* psp = &ps;
    const fe_t* blinds = r_W.data();
    LzAhead* lzp = lz_ahead ? &lz : nullptr;
    const sp_ck* key = pk.ck;
    const size_t cols = lz_cols;
    // tables of the fixed rows + h: usable when every other row of comm_W is h * blind (commit_zeros)
    const sp_fbtables* tabs = lz_tables_path ? ps.lz_tables : nullptr;
    const size_t nfixed = ps.comm_W_fixed.size();
    const fe_t* dv = dvec;
    const size_t dn = n_ipa;
    std::atomic<int>* dvr = &dvec_ready;
    const fe_t* rda = &r_delta_ahead;
    ps.bg.submit([ctx, rows, nrows, psp, blinds, lzp, key, cols, tabs, nfixed, dv, dn, dvr, rda] {
      const std::vector<uint8_t> b = commitment_bytes(rows, nrows);
      ck(sp_transcript_preabsorb((const uint8_t*)"poly_com", 8, b.data(), b.size(), &psp->poly_com), "poly_com (prepare)");
      if (!lzp) return;
      ck(sp_ctx_bind_thread(ctx), "helper thread: device");  // a new thread starts on device 0; the context may live on another GPU
      ck(sp_points_upload(ctx, u64p(&rows[0].x), nrows, &psp->comm_pts), "comm_W (upload)");
      const auto t0 = std::chrono::steady_clock::now();
      // sleeps until the owner publishes the state (helper_may_spin above)
      auto wait_for = [&](std::atomic<int>& flag) {
        int st;
        while ((st = flag.load(std::memory_order_acquire)) == 0) {
          if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) return 2;  // the prover never got there
          if (!helper_may_spin()) {
            std::unique_lock<std::mutex> lk(lzp->mu);  // woken by publish(); the timeout covers a notify that raced the predicate
            lzp->cv.wait_for(lk, std::chrono::microseconds(100), [&] { return flag.load(std::memory_order_acquire) != 0; });
          } else {
            sp_relax();
          }
        }
        return st;
      };
      // delta (ipa.rs:147): issued from here once the outer sum-check has left its streaming rounds (the owner publishes delta_state from its
      // challenge observer): the MSM then runs under the resident tail kernel, which leaves the device idle, instead of beside the first rounds
      // of the inner sum-check (measured there: k_eval_quad_stream_lowhi 32 us against 18 us alone, profiles/r04_kernel_stats.md)
      if (wait_for(lzp->delta_state) != 1) return;
      while (dvr->load(std::memory_order_acquire) == 0) sp_relax();  // d_vec, r_delta (the second helper's job, posted first thing in the prove: long done)
      lzp->r_delta = *rda;
      ck(sp_msm_ck_begin(ctx, key, u64p(dv), dn, &lzp->delta_job), "delta (begin)");
      {
        sp_msm_job* j = lzp->delta_job;
        lzp->delta_job = nullptr;  // finish() consumes the job whatever it returns
        ck(sp_msm_ck_finish(ctx, key, j, u64p(&lzp->r_delta), u64p(&lzp->delta.x)), "delta (finish)");
      }
      lzp->delta_done = true;
      lzp->early_done.store(1, std::memory_order_release);
      if (tabs && lzp->nvr >= 1 && nrows == ((size_t)1 << lzp->nvr)) {
        // comm_LZ = sum_{fixed rows} L_i comm_W[i] + (sum_{zero rows} L_i blind_i) h, L = eq(r_rows, .): one multi_mul over the prepared tables. Its ~100 us
        // of dependent point additions are the critical path of the opening, so everything in front of the launch is kept off the last row challenge:
        // eq(r, .) is expanded a level per challenge as the inner sum-check draws them (the observer counts them in rows_known), and what the last
        // challenge r leaves to do is the last level (2^(nvr-1) products) and h's scalar, (1 - r) S0 + r S1 with S_c = sum over the zero rows of parity c
        // of P[i >> 1] blind_i taken one level up; <L, r_W> for z_delta follows the launch. (All of it at once behind the last challenge was ~40 us.)
        const size_t nvr = lzp->nvr;
        std::vector<fe_t> P(1, fe_one<S>());
        for (size_t known = 0; known + 1 < nvr; ++known) {
          const auto w0 = std::chrono::steady_clock::now();
          while ((size_t)lzp->rows_known.load(std::memory_order_acquire) <= known) {
            if (lzp->state.load(std::memory_order_acquire) == 2 || std::chrono::steady_clock::now() - w0 > std::chrono::seconds(20)) return;
            if (!helper_may_spin()) {
              std::unique_lock<std::mutex> lk(lzp->mu);
              lzp->cv.wait_for(lk, std::chrono::microseconds(50), [&] { return (size_t)lzp->rows_known.load(std::memory_order_acquire) > known; });
            } else {
              sp_relax();
            }
          }
          const fe_t rk = lzp->r[known];
          std::vector<fe_t> Q(2 * P.size());
          for (size_t i = 0; i < P.size(); ++i) {  // the new variable is the index LSB (eq.rs:66-76)
            const fe_t hi = fe_mul<S>(P[i], rk);
            Q[2 * i + 1] = hi;
            Q[2 * i] = fe_sub<S>(P[i], hi);
          }
          P.swap(Q);
        }
        fe_t S0 = fe_zero(), S1 = fe_zero();
        for (size_t i = nfixed; i < nrows; ++i) {
          const fe_t t = fe_mul<S>(P[i >> 1], blinds[i]);
          if (i & 1) S1 = fe_add<S>(S1, t);
          else S0 = fe_add<S>(S0, t);
        }
        if (wait_for(lzp->state) != 1) return;
        const fe_t rl = lzp->r[nvr - 1];
        bool launched = false;
        if (nfixed + 1 < 1024) {  // the last level on the device (k_multi_mul_coop EXPAND): the launch goes out before this thread forms it for r_LZ
          const fe_t s01[2] = {S0, S1};
          ck(sp_fbtables_multi_mul_begin_eq(ctx, tabs, u64p(P.data()), nfixed, u64p(s01), u64p(&rl)), "comm_LZ (begin)");
          ck(sp_rowmat_vec_eq_begin_with(ctx, psp->W, u64p(lzp->r), lzp->nvr, cols, dn == cols ? u64p(dv) : nullptr, &lzp->vec), "bind_with_delayed (begin)");
          launched = true;
        }
        std::vector<fe_t> sc(nfixed + 1);
        for (size_t h = 0; h < P.size(); ++h) {
          const fe_t hi = fe_mul<S>(P[h], rl), lo = fe_sub<S>(P[h], hi);
          if (2 * h < nfixed) sc[2 * h] = lo;
          if (2 * h + 1 < nfixed) sc[2 * h + 1] = hi;
        }
        const fe_t hs = fe_add<S>(S0, fe_mul<S>(rl, fe_sub<S>(S1, S0)));
        sc[nfixed] = hs;
        if (!launched) {
          ck(sp_fbtables_multi_mul_begin(ctx, tabs, u64p(sc.data()), sc.size()), "comm_LZ (begin)");
          ck(sp_rowmat_vec_eq_begin_with(ctx, psp->W, u64p(lzp->r), lzp->nvr, cols, dn == cols ? u64p(dv) : nullptr, &lzp->vec), "bind_with_delayed (begin)");
        }
        fe_t acc = hs;  // r_LZ = <L, r_W> (hyrax_pc.rs:446-455)
        for (size_t i = 0; i < nfixed; ++i) acc = fe_add<S>(acc, fe_mul<S>(sc[i], blinds[i]));
        lzp->r_LZ = acc;
        ck(sp_fbtables_multi_mul_finish(ctx, u64p(&lzp->comm_LZ.x)), "comm_LZ (finish)");
        lzp->have_comm_LZ = true;
        return;
      }
      if (wait_for(lzp->state) != 1) return;
      const size_t hb = lzp->nvr / 2, lb = lzp->nvr - hb;
      ck(sp_msm_eq_begin(ctx, psp->comm_pts, u64p(lzp->r), lzp->nvr, &lzp->job), "comm_LZ (begin)");
      ck(sp_rowmat_vec_eq_begin_with(ctx, psp->W, u64p(lzp->r), lzp->nvr, cols, dn == cols ? u64p(dv) : nullptr, &lzp->vec), "bind_with_delayed (begin)");
      // r_LZ = <L, r_W> with L = eq(r) = left (x) right: 2^nvr + 2^(nvr/2) products instead of 2 * 2^nvr
      const std::vector<fe_t> left = eq_evals_host(lzp->r, hb), right = eq_evals_host(lzp->r + hb, lb);
      fe_t acc = fe_zero();
      for (size_t a = 0; a < left.size(); ++a) {
        fe_t inner = fe_zero();
        for (size_t c2 = 0; c2 < right.size(); ++c2) inner = fe_add<S>(inner, fe_mul<S>(right[c2], blinds[a * right.size() + c2]));
        acc = fe_add<S>(acc, fe_mul<S>(left[a], inner));
      }
      lzp->r_LZ = acc;
    });
  }
  struct BgJoin {  // comm_W, r_W and lz must outlive the job on every exit path
    Background& b;
    std::atomic<int>&st, &st2;
    sp_ctx* ctx;
    sp_msm_job *&j1, *&j2;
    sp_vec_job*& v1;
    ~BgJoin() {
      int zero = 0;
      st2.compare_exchange_strong(zero, 2);
      zero = 0;
      st.compare_exchange_strong(zero, 2);
      b.wait_nothrow();
      // an abandoned prove (error exit) still owns whatever the helper had started: finish the jobs to release them
      uint64_t sink[8];
      if (j1) sp_msm_job_finish(ctx, j1, sink);
      if (j2) sp_msm_job_finish(ctx, j2, sink);
      if (v1) {
        std::vector<uint64_t> big(4 * 2048);
        sp_rowmat_vec_eq_finish(ctx, v1, big.data());
      }
      j1 = j2 = nullptr;
      v1 = nullptr;
    }
  } bg_join{ps.bg, lz.state, lz.delta_state, ctx, lz.job, lz.delta_job, lz.vec};
  const double t_wit = now_ms();

  const size_t num_rounds_x = log2_ceil(N), num_rounds_y = log2_ceil(M) + 1;
  std::vector<fe_t> tau(num_rounds_x);
  for (auto& t : tau) t = tr.squeeze("t");
  lap("tau");

  const double t_mv = now_ms();

  SpartanProofBuf proof;
  for (const aff_t& a : comm_W) proof.pp(a);
  for (const fe_t& f : publics) proof.pf(f);
  for (const fe_t& f : challenges) proof.pf(f);  // SplitR1CSInstance carries them (src/r1cs/mod.rs:1423-1437)
  // outer sum-check (src/spartan.rs:291-310)
  std::vector<fe_t> outer_polys(3 * num_rounds_x), r_x(num_rounds_x);
  fe_t claims_outer[3];
  const fe_t zero = fe_zero();
  // evals_rx started four rounds before r_x is complete (sp_eq_table_begin builds the half tables of the first ell - 4 coordinates on a stream of
  // its own; sp_eq_table_finish behind the last challenge is then one launch).
  struct EqObs {
    sp_ctx* ctx;
    size_t ell;
    fe_t r[24];
    LzAhead* lz;  // non-null: delta's MSM is the helper's to issue (see there)
    int rc = 0;
    bool begun = false;
    static void fn(void* u, size_t round, const uint64_t r[4]) {
      EqObs* o = (EqObs*)u;
      if (o->lz && round + delta_rounds_before_end() == o->ell) o->lz->publish(o->lz->delta_state, 1);
      if (round + 4 >= o->ell) return;  // the last four coordinates (the rounds the host runs itself after the hand-over) are applied by sp_eq_table_finish
      memcpy(&o->r[round], r, 32);
      if (round + 5 == o->ell) {
        o->rc = sp_eq_table_begin(o->ctx, u64p(o->r), o->ell - 4, o->ell);
        o->begun = o->rc == 0;
      }
    }
  } eq_obs{ctx, num_rounds_x, {}, lz_ahead ? &lz : nullptr};
  const bool use_eq_obs = num_rounds_x >= 12 && num_rounds_x <= 20;
  if (use_eq_obs)
    ck(sp_sumcheck_cubic3_observed(ctx, u64p(&zero), u64p(tau.data()), num_rounds_x, ps.az, ps.bz, ps.cz, ps.p0, ps.p0 ? ps.p1 : nullptr, tr.t, &EqObs::fn, &eq_obs,
                                   u64p(outer_polys.data()), u64p(r_x.data()), u64p(claims_outer)),
       "outer sum-check");
  else if (ps.p0)
    ck(sp_sumcheck_cubic3_round0(ctx, u64p(&zero), u64p(tau.data()), num_rounds_x, ps.az, ps.bz, ps.cz, ps.p0, ps.p1, tr.t, u64p(outer_polys.data()), u64p(r_x.data()),
                                 u64p(claims_outer)),
       "outer sum-check");
  else
    ck(sp_sumcheck_cubic3(ctx, u64p(&zero), u64p(tau.data()), num_rounds_x, ps.az, ps.bz, ps.cz, tr.t, u64p(outer_polys.data()), u64p(r_x.data()),
                          u64p(claims_outer)),
       "outer sum-check");
  if (eq_obs.rc) throw Error(eq_obs.rc, std::string("evals_rx (begin): ") + sp_last_error());
  if (lz_ahead) lz.publish(lz.delta_state, 1);  // (a sum-check of fewer than 15 rounds, or one without the observer, has not done it)
  tr.absorb_scalars("claims_outer", claims_outer, 3);
  for (const fe_t& f : outer_polys) proof.pf(f);
  for (int i = 0; i < 3; ++i) proof.pf(claims_outer[i]);
  const double t_outer = now_ms();

  const fe_t r = tr.squeeze("r");
  const fe_t claim_inner_joint = fe_add<S>(fe_add<S>(claims_outer[0], fe_mul<S>(r, claims_outer[1])), fe_mul<S>(fe_mul<S>(r, r), claims_outer[2]));
  // evals_rx + bind_and_prepare_poly_ABC (src/spartan.rs:316-322)
  ck(sp_eq_table_finish(ctx, u64p(r_x.data()), num_rounds_x, ps.rx), "evals_rx");  // (= sp_eq_table_into when nothing was begun)
  ck(sp_poly_abc(ctx, pk.S, ps.rx, u64p(&r), 2 * M, ps.abc), "poly_ABC");
  const double t_abc = now_ms();

  sp_msm_job* delta_job = nullptr;
  await_flag(dvec_ready);  // d_vec, r_delta, r_beta from here on
  if (!lz_ahead) ck(sp_msm_ck_begin(ctx, pk.ck, u64p(dvec), n_ipa, &delta_job), "delta (begin)");
  // inner sum-check. The reference runs round 0 by hand on the compact vectors (src/spartan.rs:323-384); that round is
  // value-identical to a generic prove_quad round on the 2M-long tables with (lo_eff, hi_eff) = (M, num_extra).
  ck(sp_table_set_len(ps.abc, 2 * M, M, pk.num_extra), "abc len");
  ck(sp_table_set_len(ps.z, 2 * M, M, pk.num_extra), "z len");
  std::vector<fe_t> inner_polys(2 * num_rounds_y), r_y(num_rounds_y);
  fe_t claims_inner[2];
  // r_y[0] selects W against (1, X); r_y[1 ..= nvr] are the row variables of W (MSB first): once they are drawn the helper can start comm_LZ
  // <R, d> (ipa.rs:148) with R = eq(point[nvr..]) = left (x) right: the partial sums T[b] = sum_a left[a] d[a |right| + b] need only `left`, whose
  // variables are bound hb rounds after the row variables - the second helper computes them under the remaining rounds and the prover finishes with
  // |right| products instead of |R| + |left| after the last round
  struct IpAhead {
    size_t first = 0, hb = 0, nright = 0;  // left = eq(r_y[first .. first + hb))
    fe_t left_r[16], right_r[16];
    std::vector<fe_t> T;
    bool submitted = false;
    // ... and once the last challenge is drawn the same job finishes <R, d> and beta = <R, d> ck_c + r_beta h_c (ipa.rs:148-149) beside the prover's
    // own work on eval_W; it waits for that challenge with a short bounded spin
    std::atomic<int> right_ready{0};
    size_t nright_vars = 0, last_round = 0;
    fe_t r_beta, ip;
    aff_t beta;
    bool beta_ready = false;
    sp_ctx* ctx = nullptr;
    const sp_ck* ck_s = nullptr;
    const aff_t* beta_term = nullptr;           // h_s * r_beta from the first helper job ...
    const std::atomic<int>* term_ready = nullptr;  // ... valid when this is set
  } ipa;
  ipa.beta_term = &beta_term;
  ipa.term_ready = &eval_W_term_ready;
  ipa.r_beta = r_beta_ahead;
  ipa.ctx = ctx;
  ipa.ck_s = pk.ck_s;
  if (lz_ahead) {
    const size_t k = (num_rounds_y - 1) - lz_nvr;
    ipa.first = 1 + lz_nvr;
    ipa.hb = k / 2;
    ipa.nright = (size_t)1 << (k - ipa.hb);
    ipa.nright_vars = k - ipa.hb;
    ipa.last_round = num_rounds_y - 1;
    if (ipa.nright_vars > 16) ipa.hb = 0;
    if (ipa.hb == 0 || ipa.hb > 16 || (((size_t)1 << ipa.hb) * ipa.nright) != n_ipa) ipa.hb = 0;  // (not this shape: the prover computes <R, d> itself)
  }
  struct Obs {
    decltype(lz)* lz;
    bool on;
    IpAhead* ipa;
    Background* bg2;
    const fe_t* dvec;
    static void fn(void* u, size_t round, const uint64_t r[4]) {
      Obs* o = (Obs*)u;
      if (!o->on || round == 0) return;
      if (round <= o->lz->nvr) {
        memcpy(&o->lz->r[round - 1], r, 32);
        o->lz->rows_known.store((int)round, std::memory_order_release);
        if (round == o->lz->nvr) o->lz->publish(o->lz->state, 1);
        return;
      }
      IpAhead* ip = o->ipa;
      if (!ip->hb || round < ip->first) return;
      if (round >= ip->first + ip->hb) {
        memcpy(&ip->right_r[round - ip->first - ip->hb], r, 32);
        if (round == ip->last_round) ip->right_ready.store(1, std::memory_order_release);
        return;
      }
      memcpy(&ip->left_r[round - ip->first], r, 32);
      if (round + 1 == ip->first + ip->hb) {
        const fe_t* dv = o->dvec;
        o->bg2->submit([ip, dv] {
          const std::vector<fe_t> left = eq_evals_host(ip->left_r, ip->hb);
          ip->T.assign(ip->nright, fe_zero());
          for (size_t a = 0; a < left.size(); ++a)
            for (size_t b = 0; b < ip->nright; ++b) ip->T[b] = fe_add<S>(ip->T[b], fe_mul<S>(left[a], dv[a * ip->nright + b]));
          const auto t0 = std::chrono::steady_clock::now();
          while (ip->right_ready.load(std::memory_order_acquire) == 0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(150)) return;  // (the prover finishes <R, d> and beta itself; a helper does not sit on a CPU of the quota)
            sp_relax();
          }
          const std::vector<fe_t> right = eq_evals_host(ip->right_r, ip->nright_vars);
          fe_t acc = fe_zero();
          for (size_t b = 0; b < right.size(); ++b) acc = fe_add<S>(acc, fe_mul<S>(right[b], ip->T[b]));
          ip->ip = acc;
          if (ip->term_ready->load(std::memory_order_acquire)) ck(sp_hyrax_commit_small_with_term(ip->ctx, ip->ck_s, u64p(&ip->ip), 1, u64p(&ip->beta_term->x), u64p(&ip->beta.x)), "beta");
          else ck(sp_hyrax_commit_small(ip->ctx, ip->ck_s, u64p(&ip->ip), 1, u64p(&ip->r_beta), u64p(&ip->beta.x)), "beta");
          ip->beta_ready = true;
        });
        ip->submitted = true;
      }
    }
  } obs{&lz, lz_ahead, &ipa, &ps.bg2, dvec};
  struct IpJoin {  // ipa and dvec outlive the job on every exit path
    Background& b;
    ~IpJoin() { b.wait_nothrow(); }
  } ip_join{ps.bg2};
  ck(sp_sumcheck_quad_observed(ctx, u64p(&claim_inner_joint), num_rounds_y, ps.abc, ps.z, tr.t, &Obs::fn, &obs, u64p(inner_polys.data()), u64p(r_y.data()),
                               u64p(claims_inner)),
     "inner sum-check");
  for (const fe_t& f : inner_polys) proof.pf(f);
  const fe_t eval_Z = claims_inner[1];
  // eval_W = (eval_Z - r_y0 eval_X) / (1 - r_y0)   (src/spartan.rs:411-421)
  std::vector<fe_t> X;
  X.push_back(fe_one<S>());
  X.insert(X.end(), publics.begin(), publics.end());
  X.insert(X.end(), challenges.begin(), challenges.end());  // to_regular_instance: X = public_values ++ challenges (src/r1cs/mod.rs:1546-1549)
  const fe_t eval_X = sparse_poly_evaluate(num_rounds_y - 1, X, r_y.data() + 1);
  const fe_t denom = fe_sub<S>(fe_one<S>(), r_y[0]);
  if (fe_is_zero(denom)) throw Error(SP_ERR_DIVISION_BY_ZERO, "DivisionByZero");
  const fe_t eval_W = fe_mul<S>(fe_sub<S>(eval_Z, fe_mul<S>(r_y[0], eval_X)), fe_inv_vartime<S>(denom));
  lap("inner+eval_W");
  const double t_inner = now_ms();

  // pcs (src/spartan.rs:423-437) -> HyraxPCS::prove (hyrax_pc.rs:387-478) -> InnerProductArgumentLinear::prove (ipa.rs:125-170).
  // Device work is issued first (row-matrix product, comm_LZ's MSM on the auxiliary stream); the host-side pieces that do not
  // depend on it (commitment to eval_W, hashing comm_W, <R, d>, beta) run underneath. Transcript ORDER is the reference's.
  const fe_t blind_eval_W = tape.next();
  proof.pf(eval_W);
  proof.pf(blind_eval_W);
  const fe_t* point = r_y.data() + 1;
  const size_t npoint = num_rounds_y - 1;
  const size_t num_rows = (M + W_ - 1) / W_, nvr = log2_ceil(num_rows);
  aff_t comm_LZ;
  std::vector<fe_t> R, LZ;
  fe_t r_LZ;
  sp_msm_job* lz_job = nullptr;
  if (nvr == 0) {
    comm_LZ = comm_W[0];
    R = eq_evals_host(point, npoint);
    LZ.resize(M);
    ck(sp_table_read(ctx, ps.W, 0, M, u64p(LZ.data())), "read W");
    r_LZ = r_W[0];
  } else if (lz_ahead) {
    // comm_LZ's MSM and the row-matrix product have been running since round nvr of the inner sum-check
    LZ.resize(lz_cols);
    if (lz_tables_path) {
      // the helper delivers comm_LZ itself (one multi_mul over the prepared tables): join it only where comm_LZ is absorbed; what is needed before
      // that (poly_com, delta) was finished under the inner sum-check
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 0; lz.early_done.load(std::memory_order_acquire) == 0; ++spins) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) throw Error(SP_ERR_INTERNAL, "the PCS helper did not finish delta");
        if ((spins & 63u) == 63u) ps.bg.try_steal();  // the helper thread never woke for this prove: its whole job runs here (every state it waits for is published)
        sp_relax();
      }
    } else {
      ps.bg.wait();
      if (!lz.job || !lz.vec || !lz.delta_done) throw Error(SP_ERR_INTERNAL, "comm_LZ was not started");
      lz_job = lz.job;
      lz.job = nullptr;
      r_LZ = lz.r_LZ;
    }
    lap("helper_join");
  } else {
    std::vector<fe_t> L = eq_evals_host(point, nvr);
    LZ.resize((size_t)1 << (npoint - nvr));
    ck(sp_rowmat_vec(ctx, ps.W, L.size(), LZ.size(), u64p(L.data()), u64p(LZ.data())), "bind_with_delayed");
    ck(sp_msm_ck_begin(ctx, pk.ck, u64p(LZ.data()), LZ.size(), &lz_job), "comm_LZ (begin)");
    lap("eq_L+rowmat_vec+msm_begin");
    r_LZ = fe_zero();
    for (size_t i = 0; i < L.size(); ++i) r_LZ = fe_add<S>(r_LZ, fe_mul<S>(L[i], r_W[i]));
  }
  aff_t comm_eval_W;
  if (eval_W_term_ready.load(std::memory_order_acquire) && memcmp(&eval_W_term_blind, &blind_eval_W, sizeof(fe_t)) == 0)
    ck(sp_hyrax_commit_small_with_term(ctx, pk.ck_s, u64p(&eval_W), 1, u64p(&eval_W_term.x), u64p(&comm_eval_W.x)), "commit eval_W");
  else
    ck(sp_hyrax_commit_small(ctx, pk.ck_s, u64p(&eval_W), 1, u64p(&blind_eval_W), u64p(&comm_eval_W.x)), "commit eval_W");
  lap("comm_eval_W");
  if (!lz_tables_path) ps.bg.wait();
  ck(sp_transcript_absorb_prepared(tr.t, ps.poly_com), "poly_com");
  lap("poly_com");
  tr.dom_sep("inner product argument (linear)");
  const size_t n = (size_t)1 << (nvr == 0 ? npoint : npoint - nvr);  // |R|
  if (n != n_ipa) throw Error(SP_ERR_INTERNAL, "IPA width mismatch");
  tape.skip(n);  // the d_vec blocks that were peeked at the start
  const fe_t r_delta = tape.next(), r_beta = tape.next();
  // <R, d> (ipa.rs:148) with R = eq(point[nvr..]) = left (x) right: n + sqrt(n) products, R itself is never needed
  fe_t ip = fe_zero();
  aff_t delta, beta;
  bool have_beta = false;
  if (R.empty() && ipa.submitted) {
    ps.bg2.wait();
    if (ipa.beta_ready && memcmp(&ipa.r_beta, &r_beta, sizeof(fe_t)) == 0) {
      ip = ipa.ip;
      beta = ipa.beta;
      have_beta = true;
    }
  }
  lap("ipa_join");
  if (have_beta) {
  } else if (R.empty() && ipa.submitted) {
    const size_t k = npoint - nvr;
    const std::vector<fe_t> right = eq_evals_host(point + nvr + ipa.hb, k - ipa.hb);
    if (right.size() != ipa.T.size()) throw Error(SP_ERR_INTERNAL, "<R, d>: partial sums of the wrong width");
    for (size_t b = 0; b < right.size(); ++b) ip = fe_add<S>(ip, fe_mul<S>(right[b], ipa.T[b]));
  } else if (R.empty()) {
    const size_t k = npoint - nvr, hb = k / 2;
    const std::vector<fe_t> left = eq_evals_host(point + nvr, hb), right = eq_evals_host(point + nvr + hb, k - hb);
    for (size_t a = 0; a < left.size(); ++a) {
      fe_t inner = fe_zero();
      for (size_t b = 0; b < right.size(); ++b) inner = fe_add<S>(inner, fe_mul<S>(right[b], dvec[a * right.size() + b]));
      ip = fe_add<S>(ip, fe_mul<S>(left[a], inner));
    }
  } else {
    for (size_t i = 0; i < n; ++i) ip = fe_add<S>(ip, fe_mul<S>(R[i], dvec[i]));
  }
  if (!have_beta) ck(sp_hyrax_commit_small(ctx, pk.ck_s, u64p(&ip), 1, u64p(&r_beta), u64p(&beta.x)), "beta");
  if (lz_ahead) {
    delta = lz.delta;  // finished by the helper with the blind peeked from the same tape position
    if (memcmp(&r_delta, &lz.r_delta, sizeof(fe_t)) != 0) throw Error(SP_ERR_INTERNAL, "tape positions of r_delta disagree");
  } else {
    // delta's device part finished long ago (it was issued before the inner sum-check): its window Horner runs on the host while the device
    // still works on comm_LZ
    ck(sp_msm_ck_finish(ctx, pk.ck, delta_job, u64p(&r_delta), u64p(&delta.x)), "delta (finish)");
  }
  lap("host_side_under_msm");
  if (lz_tables_path) {
    ps.bg.wait();
    if (!lz.have_comm_LZ || !lz.vec) throw Error(SP_ERR_INTERNAL, "comm_LZ was not delivered");
    comm_LZ = lz.comm_LZ;
    r_LZ = lz.r_LZ;
  } else if (lz_job && lz_ahead) ck(sp_msm_job_finish(ctx, lz_job, u64p(&comm_LZ.x)), "comm_LZ (finish)");  // the blinds are inside the row commitments
  else if (lz_job) ck(sp_msm_ck_finish(ctx, pk.ck, lz_job, u64p(&r_LZ), u64p(&comm_LZ.x)), "comm_LZ (finish)");
  lap("comm_LZ_finish");
  {
    uint8_t b[128];
    point_bytes(comm_LZ, b);
    point_bytes(comm_eval_W, b + 64);
    tr.absorb("U", b, 128);
  }
  {
    uint8_t b[64];
    point_bytes(delta, b);
    tr.absorb("delta", b, 64);
    point_bytes(beta, b);
    tr.absorb("beta", b, 64);
  }
  const fe_t rr = tr.squeeze("r");
  proof.pp(delta);
  proof.pp(beta);
  const bool zvec_on_device = lz_ahead && n == lz_cols && n == n_ipa;  // the helper's job gave d_vec to the product's job (begin_with)
  if (zvec_on_device) {
    // z_vec = r * LZ + d (ipa.rs:160-163), the last step of the prove: one launch behind the product the device already holds, delivered through mapped
    // memory straight into the proof (2048 products were 33 us on this thread and the two helpers)
    sp_vec_job* vj = lz.vec;
    lz.vec = nullptr;
    const size_t at = proof.words.size();
    proof.words.resize(at + 4 * n);
    ck(sp_rowmat_vec_eq_finish_scaled(ctx, vj, u64p(&rr), proof.words.data() + at), "z_vec");
  } else {
    if (lz_ahead) {
      sp_vec_job* vj = lz.vec;
      lz.vec = nullptr;
      ck(sp_rowmat_vec_eq_finish(ctx, vj, u64p(LZ.data())), "bind_with_delayed (finish)");
    }
    // z_vec = r * LZ + d: split three ways with the two helper threads when it is wide enough to pay
    std::vector<fe_t> zv(n);
    auto part = [&zv, &LZ, dvec, rr](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) zv[i] = fe_add<S>(fe_mul<S>(rr, LZ[i]), dvec[i]);
    };
    if (n >= 1024 && lz_ahead) {
      const size_t a = n * 46 / 100, b = a + (n - a) / 2;  // the owner starts at once, the helpers have to wake up first
      ps.bg.submit([&part, a, b] { part(a, b); });
      ps.bg2.submit([&part, b, n] { part(b, n); });
      part(0, a);
      ps.bg.wait();
      ps.bg2.wait();
    } else {
      part(0, n);
    }
    proof.words.insert(proof.words.end(), u64p(zv.data()), u64p(zv.data()) + 4 * n);
  }
  proof.pf(fe_add<S>(fe_mul<S>(rr, r_LZ), r_delta));
  proof.pf(fe_add<S>(fe_mul<S>(rr, blind_eval_W), r_beta));
  lap("z_vec");
  const double t_end = now_ms();
  if (pt) {
    pt->ms[0] = t_wit - t_start - t_mv_issue;  // the matrix-vector product is issued inside the witness phase and runs under it
    pt->ms[1] = t_mv - t_wit + t_mv_issue;
    pt->ms[2] = t_outer - t_mv;
    pt->ms[3] = t_abc - t_outer;
    pt->ms[4] = t_inner - t_abc;
    pt->ms[5] = t_end - t_inner;
    pt->ms[6] = t_end - t_start;
  }
  ps.queue_lz_tables(ctx);
  return proof;
}

// SpartanSNARK::prove exactly as src/spartan.rs:219-466 states it — one statement of the reference per call of the ABI, in its order, on the calling
// thread alone: no helper threads, nothing issued ahead of where the reference computes it; of prep_prove's products only the FixedBaseMul tables of the rows it
// committed are used (the shim keeps them in its PrepSNARK and names them when it announces the opening). Whatever overlap there is
// happens BELOW the ABI (the round loops' launch-ahead and resident tails, sp_hyrax_prove's two walks beside its hashing). This is the time an unchanged
// spartan.rs gets from a shim that binds include/spartan_hip.h; the proof is the same bytes as prove()'s.
SpartanProofBuf prove_reference_order(const SpartanProverKey& pk, SpartanPrepSNARK& ps, const uint64_t* publics_u64, size_t npub, Tape& tape, PhaseTimes* pt, ss_rest_hook synth,
                                      void* synth_user) {
  const sp_dims& d = pk.dims;
  sp_ctx* ctx = pk.ctx;
  const size_t M = pk.num_vars, N = d.num_cons, W_ = DEFAULT_COMMITMENT_WIDTH;
  if (npub != d.num_public) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "public_values length");
  ck(sp_ctx_bind_thread(ctx), "device");
  const double t_start = now_ms();
  static const bool laps_on = getenv("SPARTAN_HOST_LAPS") != nullptr;
  double t_lap = t_start;
  auto lap = [&](const char* name) {
    if (!laps_on) return;
    const double t = now_ms();
    fprintf(stderr, "ref lap %-28s %.3f ms\n", name, t - t_lap);
    t_lap = t;
  };
  std::vector<fe_t> publics(npub);
  for (size_t i = 0; i < npub; ++i) publics[i] = fe_from_u64<S>(publics_u64[i]);
  // :226-236 transcript, vk, public values
  Tr tr(ctx, "SpartanSNARK");
  ck(sp_transcript_set_async(tr.t, 1), "transcript");  // (the shim's TranscriptEngine: long absorbs hashed beside the caller's next calls)
  tr.absorb("vk", pk.vk_digest, 32);
  tr.absorb_scalars("public_values", publics.data(), npub);
  // r1cs_instance_and_witness (src/bellpepper/r1cs.rs:411-540)
  if (ps.rows_shared) tr.absorb("comm_W_shared", ps.comm_shared_bytes.data(), ps.comm_shared_bytes.size());
  if (ps.rows_precommitted) tr.absorb("comm_W_precommitted", ps.comm_pre_bytes.data(), ps.comm_pre_bytes.size());
  std::vector<fe_t> challenges(d.num_challenges);
  if (d.num_challenges) {
    if (!synth) throw Error(SP_ERR_INTERNAL, "a circuit with verifier challenges needs its synthesize callback");
    for (auto& c : challenges) c = tr.squeeze("challenge");
    std::vector<fe_t> rest(d.num_rest_unpadded + 1);
    if (synth(synth_user, u64p(challenges.data()), challenges.size(), u64p(rest.data())) != 0) throw Error(SP_ERR_INTERNAL, "SynthesisError: the circuit's synthesize callback failed");
    if (d.num_rest_unpadded) ck(sp_table_write(ctx, ps.W, d.num_shared + d.num_precommitted, u64p(rest.data()), d.num_rest_unpadded), "W rest");
  }
  lap("transcript prefix");
  const size_t rows_pre = ps.comm_W_fixed.size(), rows_rest = (d.num_rest + W_ - 1) / W_;
  std::vector<fe_t> r_W_rest(rows_rest);
  for (auto& b : r_W_rest) b = tape.next();  // PCS::blind (r1cs.rs:466)
  std::vector<aff_t> comm_W(rows_pre + rows_rest);
  std::copy(ps.comm_W_fixed.begin(), ps.comm_W_fixed.end(), comm_W.begin());
  if (rows_rest && d.num_rest_unpadded == 0) ck(sp_fixed_base_mul_h(ctx, pk.ck, u64p(r_W_rest.data()), rows_rest, u64p(&comm_W[rows_pre].x)), "commit_zeros");  // :467-469
  else if (rows_rest)
    ck(sp_hyrax_commit(ctx, pk.ck, ps.W, d.num_shared + d.num_precommitted, d.num_rest, u64p(r_W_rest.data()), ps.is_small ? 1 : 0, u64p(&comm_W[rows_pre].x)), "commit rest");
  lap("commit rest rows");
  {
    std::vector<uint8_t> b = commitment_bytes(comm_W.data() + rows_pre, rows_rest);
    tr.absorb("comm_W_rest", b.data(), b.size());
  }
  lap("absorb comm_W_rest");
  std::vector<fe_t> r_W = ps.r_W_fixed;  // combine_blinds (r1cs.rs:515-524)
  r_W.insert(r_W.end(), r_W_rest.begin(), r_W_rest.end());
  // The shim's r1cs_instance_and_witness wrapper ends by announcing the opening PCS::prove will be asked for (sp_hyrax_prove_announce): commitment, blinds
  // and the IPA's randomness exist here — the reference draws that randomness inside InnerProductArgumentLinear::prove from OsRng; with the injected
  // tape it is the cols + 2 blocks behind blind_eval_W's one — and the library starts the opening's transcript-independent parts under the sum-checks.
  struct Announced {  // an error exit must not leave the announcement (it points at ps.W) on the context
    sp_ctx* ctx;
    bool done = false;
    ~Announced() {
      if (!done) (void)sp_hyrax_prove_retract(ctx);
    }
  } announced{ctx};
  if (tape.pos + 1 < tape.blocks) {
    // (with the tables of the committed rows the shim's prep_prove built - ps.lz_tables: the precommitted rows and h - when the remaining rows are commit_zeros)
    if (ps.lz_tables && rows_rest && d.num_rest_unpadded == 0)
      ck(sp_hyrax_prove_announce_tables(ctx, pk.ck, u64p(&comm_W[0].x), comm_W.size(), ps.W, M, u64p(r_W.data()), tape.bytes + 64 * (tape.pos + 1), tape.blocks - tape.pos - 1,
                                        ps.lz_tables, rows_pre),
         "PCS::prove (announce)");
    else
      ck(sp_hyrax_prove_announce(ctx, pk.ck, u64p(&comm_W[0].x), comm_W.size(), ps.W, M, u64p(r_W.data()), tape.bytes + 64 * (tape.pos + 1), tape.blocks - tape.pos - 1),
         "PCS::prove (announce)");
  }
  lap("announce");
  const double t_wit = now_ms();
  // :246-253 z = [W | 1 | public | challenges]
  ck(sp_table_set_len(ps.z, 2 * M, (size_t)-1, (size_t)-1), "z len");
  ck(sp_table_copy(ctx, ps.z, 0, ps.W, 0, M), "z <- W");
  {
    ck(sp_table_zero(ctx, ps.z, M, M), "clear z high half");
    std::vector<fe_t> tail(pk.num_extra);
    tail[0] = fe_one<S>();
    std::copy(publics.begin(), publics.end(), tail.begin() + 1);
    std::copy(challenges.begin(), challenges.end(), tail.begin() + 1 + npub);
    if (tail.size() <= 2048) ck(sp_table_write_async(ctx, ps.z, M, u64p(tail.data()), tail.size()), "z tail");
    else ck(sp_table_write(ctx, ps.z, M, u64p(tail.data()), tail.size()), "z tail");
  }
  ck(sp_table_set_len(ps.z, pk.num_cols, (size_t)-1, (size_t)-1), "z len");
  lap("z");
  // :262-264 tau
  const size_t num_rounds_x = log2_ceil(N), num_rounds_y = log2_ceil(M) + 1;
  std::vector<fe_t> tau(num_rounds_x);
  for (auto& t : tau) t = tr.squeeze("t");
  lap("tau");
  // :267-283 multiply_vec_incremental_into
  ck(sp_multiply_vec_incremental(ctx, pk.S, ps.z, ps.caz, ps.cbz, ps.ccz, ps.az, ps.bz, ps.cz), "multiply_vec_incremental");
  lap("multiply_vec_incremental");
  const double t_mv = now_ms();
  SpartanProofBuf proof;
  for (const aff_t& a : comm_W) proof.pp(a);
  for (const fe_t& f : publics) proof.pf(f);
  for (const fe_t& f : challenges) proof.pf(f);
  // :291-310 outer sum-check
  std::vector<fe_t> outer_polys(3 * num_rounds_x), r_x(num_rounds_x);
  fe_t claims_outer[3];
  const fe_t zero = fe_zero();
  ck(sp_sumcheck_cubic3(ctx, u64p(&zero), u64p(tau.data()), num_rounds_x, ps.az, ps.bz, ps.cz, tr.t, u64p(outer_polys.data()), u64p(r_x.data()), u64p(claims_outer)),
     "outer sum-check");
  tr.absorb_scalars("claims_outer", claims_outer, 3);
  for (const fe_t& f : outer_polys) proof.pf(f);
  for (int i = 0; i < 3; ++i) proof.pf(claims_outer[i]);
  const double t_outer = now_ms();
  // :311-322 r, evals_rx, bind_and_prepare_poly_ABC
  const fe_t r = tr.squeeze("r");
  const fe_t claim_inner_joint = fe_add<S>(fe_add<S>(claims_outer[0], fe_mul<S>(r, claims_outer[1])), fe_mul<S>(fe_mul<S>(r, r), claims_outer[2]));
  ck(sp_eq_table_into(ctx, u64p(r_x.data()), num_rounds_x, ps.rx), "evals_rx");
  ck(sp_poly_abc(ctx, pk.S, ps.rx, u64p(&r), 2 * M, ps.abc), "poly_ABC");
  const double t_abc = now_ms();
  // :323-404 inner sum-check (manual round 0 == a generic round on (lo_eff, hi_eff) = (M, num_extra) tables)
  ck(sp_table_set_len(ps.abc, 2 * M, M, pk.num_extra), "abc len");
  ck(sp_table_set_len(ps.z, 2 * M, M, pk.num_extra), "z len");
  std::vector<fe_t> inner_polys(2 * num_rounds_y), r_y(num_rounds_y);
  fe_t claims_inner[2];
  ck(sp_sumcheck_quad(ctx, u64p(&claim_inner_joint), num_rounds_y, ps.abc, ps.z, tr.t, u64p(inner_polys.data()), u64p(r_y.data()), u64p(claims_inner)), "inner sum-check");
  for (const fe_t& f : inner_polys) proof.pf(f);
  // :405-421 eval_W
  const fe_t eval_Z = claims_inner[1];
  std::vector<fe_t> X;
  X.push_back(fe_one<S>());
  X.insert(X.end(), publics.begin(), publics.end());
  X.insert(X.end(), challenges.begin(), challenges.end());
  const fe_t eval_X = sparse_poly_evaluate(num_rounds_y - 1, X, r_y.data() + 1);
  const fe_t denom = fe_sub<S>(fe_one<S>(), r_y[0]);
  if (fe_is_zero(denom)) throw Error(SP_ERR_DIVISION_BY_ZERO, "DivisionByZero");
  const fe_t eval_W = fe_mul<S>(fe_sub<S>(eval_Z, fe_mul<S>(r_y[0], eval_X)), fe_inv_vartime<S>(denom));
  lap("eval_W");
  const double t_inner = now_ms();
  // :423-436 blind, commit to eval_W, PCS::prove
  const fe_t blind_eval_W = tape.next();
  aff_t comm_eval_W;
  ck(sp_hyrax_commit_small(ctx, pk.ck_s, u64p(&eval_W), 1, u64p(&blind_eval_W), u64p(&comm_eval_W.x)), "commit eval_W");
  lap("commit eval_W");
  proof.pf(eval_W);
  proof.pf(blind_eval_W);
  const size_t num_rows = (M + W_ - 1) / W_, cols = M / num_rows;
  // (the draws of ipa.rs:139-149 happen inside the call, from the randomness stream at its current position: cols + 2 blocks)
  if (tape.pos + cols + 2 > tape.blocks) throw Error(SP_ERR_INTERNAL, "random tape exhausted");
  std::vector<uint64_t> arg(16 + 4 * cols + 8);
  ck(sp_hyrax_prove(ctx, pk.ck, pk.ck_s, tr.t, u64p(&comm_W[0].x), comm_W.size(), ps.W, M, u64p(r_W.data()), u64p(r_y.data() + 1), num_rounds_y - 1, u64p(&comm_eval_W.x),
                    u64p(&blind_eval_W), tape.bytes + 64 * tape.pos, tape.blocks - tape.pos, arg.data()),
     "PCS::prove");
  announced.done = true;  // consumed
  tape.skip(cols + 2);
  proof.words.insert(proof.words.end(), arg.begin(), arg.end());
  const double t_end = now_ms();
  if (pt) {
    pt->ms[0] = t_wit - t_start;
    pt->ms[1] = t_mv - t_wit;
    pt->ms[2] = t_outer - t_mv;
    pt->ms[3] = t_abc - t_outer;
    pt->ms[4] = t_inner - t_abc;
    pt->ms[5] = t_end - t_inner;
    pt->ms[6] = t_end - t_start;
  }
  ps.queue_lz_tables(ctx);
  return proof;
}

// ---- SpartanSNARK::verify (src/spartan.rs:469-578) — SURVEY 8(f) rank 3 ---------------------------------------------------------------
// The work that scales with the instance runs on the device through the same ABI: the three matrix evaluations A(rx,ry), B, C
// (`evaluate_with_tables_fast`, src/r1cs/mod.rs:1216-1226) as ONE sp_multiply_vec against the table T_y = eq(r_y) followed by three dot
// products with T_x = eq(r_x); comm_LZ = <L, comm rows> (hyrax_pc.rs:480-531) and the IPA check's <z_vec, ck> (ipa.rs:173-221) as device MSMs.
// The O(log N) parts (transcript, round-polynomial checks, single scalar multiplications) stay on the host. Returns 0 = accept, else the index
// of the failed check (1 shape, 2 outer sum-check, 3 outer claim, 4 inner sum-check, 5 inner claim, 6 opening) — the oracle's codes.
int verify(const SpartanProverKey& pk, const uint64_t* words, size_t nwords, uint64_t* out_publics) {
  sp_ctx* ctx = pk.ctx;
  const sp_dims& d = pk.dims;

  const size_t W_ = DEFAULT_COMMITMENT_WIDTH, N = d.num_cons, M = pk.num_vars;
  const size_t rows_sh = d.num_shared_unpadded ? (d.num_shared + W_ - 1) / W_ : 0, rows_pre = d.num_precommitted_unpadded ? (d.num_precommitted + W_ - 1) / W_ : 0;
  const size_t rows_rest = (d.num_rest + W_ - 1) / W_, rows = rows_sh + rows_pre + rows_rest;
  const size_t lx = log2_ceil(N), ly = log2_ceil(M) + 1, nz = M < W_ ? M : W_;
  if (nwords != 8 * rows + 4 * (d.num_public + d.num_challenges) + 12 * lx + 12 + 8 * ly + 8 + 16 + 4 * nz + 8) return 1;
  const fe_t* w = reinterpret_cast<const fe_t*>(words);
  const aff_t* comm_W = reinterpret_cast<const aff_t*>(w);
  w += 2 * rows;
  const fe_t* publics = w;
  w += d.num_public;
  const fe_t* challenges = w;
  w += d.num_challenges;
  const fe_t* outer = w;
  w += 3 * lx;
  const fe_t* claims = w;
  w += 3;
  const fe_t* inner = w;
  w += 2 * ly;
  const fe_t eval_W = w[0], blind_eval_W = w[1];
  w += 2;
  const aff_t delta = *reinterpret_cast<const aff_t*>(w), beta = *reinterpret_cast<const aff_t*>(w + 2);
  w += 4;
  const fe_t* z_vec = w;
  w += nz;
  const fe_t z_delta = w[0], z_beta = w[1];
  {
    const fe_t* all = reinterpret_cast<const fe_t*>(words);
    const size_t n_el = nwords / 4, p0 = 2 * rows, p1 = p0 + d.num_public + d.num_challenges + 3 * lx + 3 + 2 * ly + 2;  // [0, p0): comm_W coordinates; [p1, p1 + 4): delta, beta
    for (size_t i = 0; i < n_el; ++i) {
      const bool coord = i < p0 || (i >= p1 && i < p1 + 4);
      if (!(coord ? limbs_canonical<B>(all[i]) : limbs_canonical<S>(all[i]))) return 1;
    }
  }
  for (size_t i = 0; i < rows; ++i)
    if (!aff_on_curve(comm_W[i])) return 1;
  if (!aff_on_curve(delta) || !aff_on_curve(beta)) return 1;

  // <z_vec, ck> (ipa.rs:196-203) depends on nothing but the proof: its device part runs under everything that follows
  struct ZJob {
    sp_ctx* ctx;
    const sp_ck* key;
    sp_msm_job* job = nullptr;
    ~ZJob() {
      uint64_t sink[8];
      if (job) sp_msm_ck_finish(ctx, key, job, nullptr, sink);  // an early return still owns the job
    }
  } zjob{ctx, pk.ck};
  ck(sp_msm_ck_begin(ctx, pk.ck, u64p(z_vec), nz, &zjob.job), "<z, ck> (begin)");

  Tr tr(ctx, "SpartanSNARK");
  tr.absorb("vk", pk.vk_digest, 32);
  tr.absorb_scalars("public_values", publics, d.num_public);
  auto absorb_rows = [&](const char* label, size_t lo, size_t cnt) {
    std::vector<uint8_t> b = commitment_bytes(comm_W + lo, cnt);
    tr.absorb(label, b.data(), b.size());
  };
  if (rows_sh) absorb_rows("comm_W_shared", 0, rows_sh);
  if (rows_pre) absorb_rows("comm_W_precommitted", rows_sh, rows_pre);
  for (size_t i = 0; i < d.num_challenges; ++i)  // validate (src/r1cs/mod.rs:1516-1526): the instance's challenges must be the transcript's
    if (!fe_eq(tr.squeeze("challenge"), challenges[i])) return 1;
  absorb_rows("comm_W_rest", rows_sh + rows_pre, rows_rest);
  std::vector<fe_t> tau(lx);
  for (auto& t : tau) t = tr.squeeze("t");
  fe_t claim_outer_final;
  std::vector<fe_t> r_x, r_y;
  if (!sumcheck_verify(tr, fe_zero(), lx, 3, outer, &claim_outer_final, &r_x)) return 2;
  const fe_t one = fe_one<S>();
  fe_t taus_bound_rx = one;  // EqPolynomial::evaluate (src/polys/eq.rs:45-57)
  for (size_t i = 0; i < lx; ++i)
    taus_bound_rx = fe_mul<S>(taus_bound_rx, fe_add<S>(fe_mul<S>(tau[i], r_x[i]), fe_mul<S>(fe_sub<S>(one, tau[i]), fe_sub<S>(one, r_x[i]))));
  if (!fe_eq(claim_outer_final, fe_mul<S>(taus_bound_rx, fe_sub<S>(fe_mul<S>(claims[0], claims[1]), claims[2])))) return 3;
  tr.absorb_scalars("claims_outer", claims, 3);
  const fe_t r = tr.squeeze("r"), r2 = fe_mul<S>(r, r);
  const fe_t claim_inner_joint = fe_add<S>(fe_add<S>(claims[0], fe_mul<S>(r, claims[1])), fe_mul<S>(r2, claims[2]));
  fe_t claim_inner_final;
  if (!sumcheck_verify(tr, claim_inner_joint, ly, 2, inner, &claim_inner_final, &r_y)) return 4;
  std::vector<fe_t> X(1 + d.num_public + d.num_challenges);
  X[0] = one;
  std::copy(publics, publics + d.num_public + d.num_challenges, X.begin() + 1);  // challenges follow the public values in the buffer and in X
  const fe_t eval_X = sparse_poly_evaluate(ly - 1, X, r_y.data() + 1);
  const fe_t eval_Z = fe_add<S>(fe_mul<S>(fe_sub<S>(one, r_y[0]), eval_W), fe_mul<S>(r_y[0], eval_X));
  // A(rx,ry), B(rx,ry), C(rx,ry) = T_x^T (M T_y): one SpMV against T_y, three dot products with T_x
  fe_t eabc[3];
  {
    if (!pk.v_Tx) {
      ck(sp_table_zeros(ctx, (size_t)1 << lx, (size_t)-1, (size_t)-1, &pk.v_Tx), "T_x alloc");
      ck(sp_table_zeros(ctx, (size_t)1 << ly, (size_t)-1, (size_t)-1, &pk.v_Ty), "T_y alloc");
      for (int i = 0; i < 3; ++i) ck(sp_table_zeros(ctx, N, (size_t)-1, (size_t)-1, &pk.v_mv[i]), "M T_y alloc");
    }
    sp_table *Tx = pk.v_Tx, *Ty = pk.v_Ty, **mv = pk.v_mv;
    ck(sp_table_set_len(Tx, (size_t)1 << lx, (size_t)-1, (size_t)-1), "T_x len");
    ck(sp_table_set_len(Ty, (size_t)1 << ly, (size_t)-1, (size_t)-1), "T_y len");
    ck(sp_eq_table_into(ctx, u64p(r_x.data()), lx, Tx), "T_x");
    ck(sp_eq_table_into(ctx, u64p(r_y.data()), ly, Ty), "T_y");
    ck(sp_table_set_len(Ty, pk.num_cols, (size_t)-1, (size_t)-1), "T_y as z");
    ck(sp_multiply_vec(ctx, pk.S, Ty, mv[0], mv[1], mv[2]), "M T_y");
    for (int i = 0; i < 3; ++i) ck(sp_table_dot(ctx, Tx, mv[i], N, u64p(&eabc[i])), "T_x . (M T_y)");
  }
  if (!fe_eq(claim_inner_final, fe_mul<S>(fe_add<S>(fe_add<S>(eabc[0], fe_mul<S>(r, eabc[1])), fe_mul<S>(r2, eabc[2])), eval_Z))) return 5;
  // HyraxPCS::verify (hyrax_pc.rs:480-531) + InnerProductArgumentLinear::verify (ipa.rs:173-221)
  aff_t comm_eval_W;
  ck(sp_hyrax_commit_small(ctx, pk.ck_s, u64p(&eval_W), 1, u64p(&blind_eval_W), u64p(&comm_eval_W.x)), "commit eval_W");
  {
    std::vector<uint8_t> b = commitment_bytes(comm_W, rows);
    tr.absorb("poly_com", b.data(), b.size());
  }
  const fe_t* point = r_y.data() + 1;
  const size_t npoint = ly - 1, num_rows = ((size_t)1 << npoint) / nz, nvr = log2_ceil(num_rows);
  aff_t comm_LZ;
  std::vector<fe_t> R;
  if (nvr == 0) {
    comm_LZ = comm_W[0];
    R = eq_evals_host(point, npoint);
  } else {
    std::vector<fe_t> L = eq_evals_host(point, nvr);
    R = eq_evals_host(point + nvr, npoint - nvr);
    if (rows < L.size()) return 6;
    ck(sp_msm(ctx, u64p(L.data()), reinterpret_cast<const uint64_t*>(comm_W), L.size(), u64p(&comm_LZ.x)), "comm_LZ");
  }
  tr.dom_sep("inner product argument (linear)");
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
  if (R.size() != nz) return 6;
  // r * comm_LZ and r * comm_eval_W: one wNAF scalar per pair of points (vartime_scalar_mul, msm.rs:779-867); h * z_delta and
  // <z_vec, R> * ck_c + z_beta * h_c through the fixed-base tables of the keys (hyrax_pc.rs:81-96)
  aff_t pr2[2] = {comm_LZ, comm_eval_W}, rp[2], hzd, rhs2;
  ck(sp_vartime_scalar_mul(ctx, u64p(&pr2[0].x), 2, u64p(&rr), u64p(&rp[0].x)), "r * (comm_LZ, comm_eval_W)");
  ck(sp_fixed_base_mul_h(ctx, pk.ck, u64p(&z_delta), 1, u64p(&hzd.x)), "h * z_delta");
  fe_t ip = fe_zero();
  for (size_t i = 0; i < nz; ++i) ip = fe_add<S>(ip, fe_mul<S>(z_vec[i], R[i]));
  ck(sp_hyrax_commit_small(ctx, pk.ck_s, u64p(&ip), 1, u64p(&z_beta), u64p(&rhs2.x)), "<z, R> * ck_c + z_beta * h_c");
  aff_t zc;  // <z_vec, ck>: started before the transcript work
  {
    sp_msm_job* j = zjob.job;
    zjob.job = nullptr;
    ck(sp_msm_ck_finish(ctx, pk.ck, j, nullptr, u64p(&zc.x)), "<z, ck> (finish)");
  }
  const jac_t lhs1 = jac_add_mixed(jac_from_affine(rp[0]), delta), rhs1 = jac_add_mixed(jac_from_affine(zc), hzd);
  if (!same_point(lhs1, rhs1)) return 6;
  const jac_t lhs2 = jac_add_mixed(jac_from_affine(rp[1]), beta);
  if (!same_point(lhs2, jac_from_affine(rhs2))) return 6;
  if (out_publics) memcpy(out_publics, publics, d.num_public * sizeof(fe_t));  // verify() returns the public values it accepted (src/spartan.rs:577)
  return 0;
}

}  // namespace spartan2

// ---- C surface for the harness (tests, bench.py) -----------------------------------------------------------------------------
using namespace spartan2;
static thread_local std::string g_err;
static int catch_all() {
  try {
    throw;
  } catch (const Error& e) {
    g_err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_err = e.what();
    return SP_ERR_INTERNAL;
  }
}

extern "C" {
const char* ss_last_error() { return g_err.c_str(); }
void ss_set_error(const char* msg) { g_err = msg; }

// SplitR1CSShape::new for callers that drive the sparse kernels directly. Returns an opaque PaddedShape.
int ss_pad_shape(size_t num_cons, size_t num_shared, size_t num_precommitted, size_t num_rest, size_t num_public, size_t num_challenges, const int64_t* Ad,
                 const uint32_t* Ai, const uint64_t* Ap, const int64_t* Bd, const uint32_t* Bi, const uint64_t* Bp, const int64_t* Cd, const uint32_t* Ci,
                 const uint64_t* Cp, void** out) {
  try {
    *out = new PaddedShape(pad_shape(make_view(num_cons, num_shared, num_precommitted, num_rest, num_public, num_challenges, Ad, Ai, Ap, Bd, Bi, Bp, Cd, Ci, Cp)));
    return 0;
  } catch (...) {
    return catch_all();
  }
}
void ss_padded_free(void* p) { delete (PaddedShape*)p; }
void ss_padded_equalize(void* a, void* b) { equalize(*(PaddedShape*)a, *(PaddedShape*)b); }  // SplitR1CSShape::equalize (src/r1cs/mod.rs:913-971)
void ss_padded_dims(void* p, uint64_t out[10]) { memcpy(out, &((PaddedShape*)p)->dims, sizeof(sp_dims)); }
void ss_padded_csr(void* p, int which, const uint64_t** data, const uint32_t** idx, const uint64_t** ptr, uint64_t* nnz) {
  auto* P = (PaddedShape*)p;
  *data = u64p(P->data[which].data());
  *idx = P->idx[which].data();
  *ptr = P->ptr[which].data();
  *nnz = P->data[which].size();
}
int ss_from_label(const char* label, size_t n, uint64_t* out) {
  try {
    std::vector<aff_t> g = from_label(label, n);
    memcpy(out, g.data(), n * sizeof(aff_t));
    return 0;
  } catch (...) {
    return catch_all();
  }
}

int ss_setup(sp_ctx* ctx, size_t num_cons, size_t num_shared, size_t num_precommitted, size_t num_rest, size_t num_public, size_t num_challenges,
             const int64_t* Ad, const uint32_t* Ai, const uint64_t* Ap, const int64_t* Bd, const uint32_t* Bi, const uint64_t* Bp, const int64_t* Cd,
             const uint32_t* Ci, const uint64_t* Cp, void** out_pk) {
  try {
    *out_pk = setup(ctx, make_view(num_cons, num_shared, num_precommitted, num_rest, num_public, num_challenges, Ad, Ai, Ap, Bd, Bi, Bp, Cd, Ci, Cp));
    return 0;
  } catch (...) {
    return catch_all();
  }
}
void ss_pk_free(void* pk) { delete (SpartanProverKey*)pk; }
int ss_pk_shape_info(void* pk_, uint64_t out[8]) { return sp_shape_info(((SpartanProverKey*)pk_)->S, out); }
void ss_pk_info(void* pk_, uint64_t dims_out[10], uint8_t digest[32]) {
  auto* pk = (SpartanProverKey*)pk_;
  memcpy(dims_out, &pk->dims, sizeof(sp_dims));
  memcpy(digest, pk->vk_digest, 32);
}
int ss_prep_prove(void* pk, const uint64_t* witness_u64, size_t n, int is_small, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, void** out_ps) {
  try {
    Tape t{tape, tape_blocks};
    *out_ps = prep_prove(*(SpartanProverKey*)pk, witness_u64, n, is_small != 0, t);
    if (tape_used) *tape_used = t.pos;
    return 0;
  } catch (...) {
    return catch_all();
  }
}
void ss_prep_free(void* ps) { delete (SpartanPrepSNARK*)ps; }
// host wall-clock of the last prep_prove's phases, ms: witness, commit, tables, matvec, scratch, (spare), total, (spare)
// the FixedBaseMul tables of the committed rows (queued by prep_prove): 1 = built, 0 = still building (wait != 0: blocks), -1 = this state has none
int ss_prep_tables_ready(void* ps, int wait) {
  auto* p = (SpartanPrepSNARK*)ps;
  return p->lz_tables ? sp_fbtables_ready(p->lz_tables, wait) : -1;
}
void ss_prep_phases(void* ps, double out[8]) { memcpy(out, ((SpartanPrepSNARK*)ps)->prep_ms, sizeof(double) * 8); }
// driver options of one prep state: bit 0 = cache the transcript prefix across proves, bit 1 = opening in the reference's order (see FLAG_*)
void ss_prep_set_flags(void* ps, unsigned flags) { ((SpartanPrepSNARK*)ps)->flags = flags; }
unsigned ss_prep_get_flags(void* ps) { return ((SpartanPrepSNARK*)ps)->flags; }
// comm_W_precommitted rows (affine) and the cached Az/Bz/Cz for parity checks
int ss_prep_export(void* pk_, void* ps_, uint64_t* comm_rows, uint64_t* caz, uint64_t* cbz, uint64_t* ccz) {
  try {
    auto* pk = (SpartanProverKey*)pk_;
    auto* ps = (SpartanPrepSNARK*)ps_;
    if (comm_rows && !ps->comm_W_fixed.empty()) memcpy(comm_rows, ps->comm_W_fixed.data(), ps->comm_W_fixed.size() * sizeof(aff_t));
    const size_t N = pk->dims.num_cons;
    if (caz) ck(sp_table_read(pk->ctx, ps->caz, 0, N, caz), "read caz");
    if (cbz) ck(sp_table_read(pk->ctx, ps->cbz, 0, N, cbz), "read cbz");
    if (ccz) ck(sp_table_read(pk->ctx, ps->ccz, 0, N, ccz), "read ccz");
    return 0;
  } catch (...) {
    return catch_all();
  }
}
// SpartanSNARK::verify on the device-backed path: 0 = accept, 1..6 = the failed check (see spartan2::verify); < 0 = library error
// out_publics (may be NULL): num_public scalars (Montgomery limbs) of the statement that was accepted
int ss_verify(void* pk, const uint64_t* words, size_t nwords, uint64_t* out_publics) {
  try {
    return verify(*(SpartanProverKey*)pk, words, nwords, out_publics);
  } catch (...) {
    return catch_all();
  }
}
// SpartanSNARK's shape-dependent lengths (what the bincode length prefixes of a proof for this key must say)
static sp_spartan_layout proof_layout(const SpartanProverKey& pk) {
  const sp_dims& d = pk.dims;
  sp_spartan_layout L;
  L.rows_shared = d.num_shared_unpadded ? (d.num_shared + 2047) / 2048 : 0;
  L.rows_precommitted = d.num_precommitted_unpadded ? (d.num_precommitted + 2047) / 2048 : 0;
  L.rows_rest = (d.num_rest + 2047) / 2048;
  L.num_public = d.num_public;
  L.num_challenges = d.num_challenges;
  L.rounds_x = log2_ceil(d.num_cons);
  L.rounds_y = log2_ceil(pk.num_vars) + 1;
  L.z_len = pk.num_vars < 2048 ? pk.num_vars : 2048;
  return L;
}
void ss_proof_layout(void* pk, sp_spartan_layout* out) { *out = proof_layout(*(SpartanProverKey*)pk); }
// the proof as bincode bytes of SpartanSNARK (src/spartan.rs:125-137); out may be NULL to learn *len
int ss_proof_to_bytes(void* pk, const uint64_t* words, size_t nwords, uint8_t* out, size_t cap, size_t* len) {
  try {
    const sp_spartan_layout L = proof_layout(*(SpartanProverKey*)pk);
    ck(sp_proof_serialize(&L, words, nwords, out, cap, len), "proof_serialize");
    return 0;
  } catch (...) {
    return catch_all();
  }
}
// verify on the serialised proof (what a verifier that received bytes runs): 0 = accept, 1..6 = the failed check — bytes that do not decode, or decode
// to a proof of another shape, fail check 1 like a malformed flat proof —, < 0 = error
int ss_verify_bytes(void* pk_, const uint8_t* bytes, size_t n, uint64_t* out_publics) {
  try {
    auto* pk = (SpartanProverKey*)pk_;
    sp_spartan_layout L;
    size_t nwords = 0;
    if (sp_proof_deserialize(bytes, n, &L, nullptr, 0, &nwords) != SP_OK) return 1;
    const sp_spartan_layout want = proof_layout(*pk);
    if (memcmp(&L, &want, sizeof L) != 0) return 1;
    std::vector<uint64_t> words(nwords);
    ck(sp_proof_deserialize(bytes, n, &L, words.data(), words.size(), &nwords), "proof_deserialize");
    return verify(*pk, words.data(), nwords, out_publics);
  } catch (...) {
    return catch_all();
  }
}
size_t ss_proof_words(void* pk_) {
  auto* pk = (SpartanProverKey*)pk_;
  const sp_dims& d = pk->dims;
  size_t rows = (d.num_shared_unpadded ? (d.num_shared + 2047) / 2048 : 0) + (d.num_precommitted_unpadded ? (d.num_precommitted + 2047) / 2048 : 0) + (d.num_rest + 2047) / 2048;
  size_t lx = log2_ceil(d.num_cons), ly = log2_ceil(pk->num_vars) + 1, nz = pk->num_vars < 2048 ? pk->num_vars : 2048;
  return 8 * rows + 4 * (d.num_public + d.num_challenges) + 12 * lx + 12 + 8 * ly + 8 + 16 + 4 * nz + 8;
}
// prove() — writes the proof in the canonical layout; phase_ms[7]: witness_commit, matrix_vector_multiply, outer_sumcheck,
// prepare_poly_ABC, inner_sumcheck, pcs_prove, total (the reference's span names, src/spartan.rs:267-437)
int ss_prove_hook(void* pk, void* ps, const uint64_t* publics_u64, size_t npub, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, uint64_t* out_words,
                  size_t out_cap, double* phase_ms, ss_rest_hook synth, void* synth_user);
int ss_prove(void* pk, void* ps, const uint64_t* publics_u64, size_t npub, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, uint64_t* out_words,
             size_t out_cap, double* phase_ms) {
  return ss_prove_hook(pk, ps, publics_u64, npub, tape, tape_blocks, tape_used, out_words, out_cap, phase_ms, nullptr, nullptr);
}
// the same for circuits with verifier challenges: `synth` is the circuit's synthesize callback (see prove)
int ss_prove_hook(void* pk, void* ps, const uint64_t* publics_u64, size_t npub, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, uint64_t* out_words,
                  size_t out_cap, double* phase_ms, ss_rest_hook synth, void* synth_user) {
  try {
    Tape t{tape, tape_blocks};
    PhaseTimes pt;
    SpartanProofBuf pf = prove(*(SpartanProverKey*)pk, *(SpartanPrepSNARK*)ps, publics_u64, npub, t, &pt, synth, synth_user);
    if (pf.words.size() > out_cap) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "proof buffer too small");
    memcpy(out_words, pf.words.data(), pf.words.size() * 8);
    if (tape_used) *tape_used = t.pos;
    if (phase_ms) memcpy(phase_ms, pt.ms, 7 * sizeof(double));
    if (getenv("SPARTAN_HOST_LAPS")) {
      for (auto& l : pt.laps) fprintf(stderr, "lap %-28s %.3f ms\n", l.first.c_str(), l.second);
    }
    // SPARTAN_SLOW_PROVE_MS=<t>: a prove that took longer than t ms leaves its phases and laps on stderr (diagnostics for rare stalls)
    static const double slow_ms = [] {
      const char* e = getenv("SPARTAN_SLOW_PROVE_MS");
      return e ? atof(e) : 0.0;
    }();
    if (slow_ms > 0 && pt.ms[6] > slow_ms) {
      fprintf(stderr, "[slow prove] %.3f ms: witness %.3f mv %.3f outer %.3f abc %.3f inner %.3f pcs %.3f\n", pt.ms[6], pt.ms[0], pt.ms[1], pt.ms[2], pt.ms[3], pt.ms[4], pt.ms[5]);
      for (auto& l : pt.laps) fprintf(stderr, "[slow prove]   lap %-28s %.3f ms\n", l.first.c_str(), l.second);
    }
    return 0;
  } catch (...) {
    return catch_all();
  }
}
// ---- several proofs in flight on ONE thread ------------------------------------------------------------------------------------------------------
// A prove is a chain of ~41 host <-> device round trips whose host side is a few microseconds of hashing and two kernel launches: one CPU spinning
// per proof in flight wastes the CPU and, under a CPU quota, caps the number of proofs in flight. Here every proof runs on a stack of its own
// (ucontext) and the library's wait hook (sp_set_wait_hook) hands the thread to the next proof at every poll: `threads` OS threads keep
// n_ctx proofs in flight, the GPU sees the same launches. The prove code itself is untouched — it blocks, its stack waits.
}  // extern "C"
#include <ucontext.h>
namespace {
struct Fiber {
  ucontext_t uc;
  std::unique_ptr<char[]> stack;
  std::function<void()> body;
  bool done = false;
  std::exception_ptr err;
};
struct Scheduler {
  ucontext_t main;
  std::vector<Fiber*> fibers;
  Fiber* cur = nullptr;
  uint64_t switches = 0;
  double wait_s = 0;  // wall time the thread's proofs spent in polls (summed over the proofs: what is left of the batch's time is host work)
  static void hook(void* u) {
    Scheduler* s = (Scheduler*)u;
    ++s->switches;
    const auto t0 = std::chrono::steady_clock::now();
    swapcontext(&s->cur->uc, &s->main);
    s->wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  static void entry(unsigned lo, unsigned hi) {
    Fiber* f = (Fiber*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
    try {
      f->body();
    } catch (...) {
      f->err = std::current_exception();
    }
    f->done = true;  // returning resumes uc_link = the scheduler
  }
  void add(Fiber* f, size_t stack_bytes) {
    f->stack.reset(new char[stack_bytes]);
    getcontext(&f->uc);
    f->uc.uc_stack.ss_sp = f->stack.get();
    f->uc.uc_stack.ss_size = stack_bytes;
    f->uc.uc_link = &main;
    const uintptr_t p = (uintptr_t)f;
    makecontext(&f->uc, (void (*)())entry, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
    fibers.push_back(f);
  }
  void run() {
    sp_set_wait_hook(&Scheduler::hook, this);
    size_t live = fibers.size();
    while (live) {
      for (Fiber* f : fibers) {
        if (f->done) continue;
        cur = f;
        swapcontext(&main, &f->uc);
        if (f->done) --live;
      }
    }
    sp_set_wait_hook(nullptr, nullptr);
  }
};
}  // namespace
extern "C" {
// n_ctx prepared states (pks[i], pss[i]: one context each, all on one device) prove `proofs_each` times each, multiplexed over `threads` OS threads.
// out_words: n_ctx x words_cap, the LAST proof of every state; out_stats: seconds of the whole batch, context switches, proofs, seconds spent inside polls summed over the proofs.
int ss_prove_multiplexed(void** pks, void** pss, size_t n_ctx, const uint64_t* publics_u64, size_t npub, const uint8_t* tape, size_t tape_blocks, size_t proofs_each,
                         size_t threads, uint64_t* out_words, size_t words_cap, double out_stats[4]) {
  try {
    if (n_ctx == 0 || threads == 0 || proofs_each == 0) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "ss_prove_multiplexed: nothing to do");
    if (threads > n_ctx) threads = n_ctx;
    std::vector<Fiber> fibers(n_ctx);
    std::vector<Scheduler> sched(threads);
    for (size_t i = 0; i < n_ctx; ++i) {
      auto* pk = (SpartanProverKey*)pks[i];
      auto* ps = (SpartanPrepSNARK*)pss[i];
      uint64_t* dst = out_words + i * words_cap;
      fibers[i].body = [=] {
        for (size_t k = 0; k < proofs_each; ++k) {
          Tape t{tape, tape_blocks};
          SpartanProofBuf pf = prove(*pk, *ps, publics_u64, npub, t, nullptr, nullptr, nullptr);
          if (pf.words.size() > words_cap) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "proof buffer too small");
          if (k + 1 == proofs_each) memcpy(dst, pf.words.data(), pf.words.size() * 8);
        }
      };
      sched[i % threads].add(&fibers[i], (size_t)1 << 20);
    }
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> th;
    std::vector<int> rc(threads, 0);
    for (size_t t = 0; t < threads; ++t)
      th.emplace_back([&, t] {
        rc[t] = sp_ctx_bind_thread(((SpartanProverKey*)pks[t])->ctx);
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        if (rc[t] == 0) sched[t].run();
      });
    while (ready.load() < (int)threads) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto& x : th) x.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t sw = 0;
    double waited = 0;
    for (auto& s : sched) sw += s.switches, waited += s.wait_s;
    if (out_stats) {
      out_stats[0] = secs;
      out_stats[1] = (double)sw;
      out_stats[2] = (double)(n_ctx * proofs_each);
      out_stats[3] = waited;
    }
    for (size_t t = 0; t < threads; ++t)
      if (rc[t]) throw Error(rc[t], "ss_prove_multiplexed: could not bind a poller thread to the device");
    for (auto& f : fibers)
      if (f.err) std::rethrow_exception(f.err);
    return 0;
  } catch (...) {
    return catch_all();
  }
}
}
