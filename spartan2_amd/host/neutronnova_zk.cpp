// NeutronNovaZkSNARK::{setup, prep_prove, prove, verify} (src/neutronnova_zk.rs:1394-2343) above the C ABI — SURVEY.md 8(f) rank 1, BASELINE config 3 as a real
// prove(): rerandomization, the step / core instances, NeutronNovaNIFS::prove with the verifier circuit's `process_round` as its round hook, the batched
// outer and inner sum-checks (process_round again), the verifier-circuit instance folded with a fresh random relaxed instance (NovaNIFS::prove,
// src/nifs.rs:34-61 + commit_T, src/r1cs/folds.rs:28-88), RelaxedR1CSSpartanProof::prove (src/spartan_relaxed.rs:98-213) and the folded Hyrax opening.
// Everything that scales with the step instances runs on the device through include/spartan_hip.h; the verifier-circuit instance (a few hundred
// constraints, ~1.3 k variables) is host-side algebra except its commitments (sp_hyrax_commit_small / sp_hyrax_commit on the width-32 key) and its
// two sum-checks (sp_sumcheck_cubic3 / sp_sumcheck_quad).
// Scope = the bench circuits' class: step and core circuits without rest variables and without verifier challenges (`can_cache_matvec`, :1520).
// The proof layout is the oracle's NNProof::serialize (oracle/neutronnova_zk.hpp); parity = word-for-word equality on the same inputs and tape.
#include <deque>

#include "neutronnova_nifs.hpp"
#include "snark_common.hpp"
#include "verifier_circuit.hpp"

namespace spartan2 {

// One helper thread taking jobs in order (FIFO). submit() never blocks; a job's failure is kept for the next wait() and does not stop the jobs behind it.
class Worker {
  std::thread th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
  bool stop_ = false;
  size_t submitted_ = 0;  // written by the submitting thread only
  std::atomic<size_t> done_{0};
  std::exception_ptr err_;
  void loop() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        f = std::move(q_.front());
        q_.pop_front();
      }
      try {
        f();
      } catch (...) {
        std::lock_guard<std::mutex> l(m_);
        if (!err_) err_ = std::current_exception();
      }
      done_.fetch_add(1, std::memory_order_release);
    }
  }

 public:
  Worker() = default;
  Worker(const Worker&) = delete;
  Worker& operator=(const Worker&) = delete;
  ~Worker() {
    if (!th_.joinable()) return;
    drain();
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_one();
    th_.join();
  }
  size_t submit(std::function<void()> f) {  // returns the ticket to wait for
    size_t t;
    {
      std::lock_guard<std::mutex> l(m_);
      q_.push_back(std::move(f));
      t = ++submitted_;
    }
    if (!th_.joinable()) th_ = std::thread([this] { loop(); });
    cv_.notify_one();
    return t;
  }
  void drain() {  // every job submitted so far has run; drops what they threw (exit paths)
    while (done_.load(std::memory_order_acquire) < submitted_) sp_relax();
    std::lock_guard<std::mutex> l(m_);
    err_ = nullptr;
  }
  void wait(size_t ticket) {  // jobs up to `ticket` have run; rethrows the first failure
    while (done_.load(std::memory_order_acquire) < ticket) sp_relax();
    std::exception_ptr e;
    {
      std::lock_guard<std::mutex> l(m_);
      e = err_;
      err_ = nullptr;
    }
    if (e) std::rethrow_exception(e);
  }
};

struct NNZkKey {
  sp_ctx* ctx = nullptr;
  sp_shape *S_step = nullptr, *S_core = nullptr;
  sp_dims dims, dims_core;
  sp_ck *ck = nullptr, *vc_ck = nullptr;
  std::vector<aff_t> gens;  // "ck": 2048 bases + h; the width-32 key is its first 32 bases with gens[32] as h (PCS::setup(b"ck", ., 32), src/r1cs/mod.rs:1690-1693)
  vcirc::Shape vc;
  size_t nb = 0, nx = 0, ny = 0, num_steps = 0, num_vars = 0;
  uint8_t vk_digest[32];
  // verify(): eq tables and the three M * T_y products of the matrix evaluations, allocated on first use
  mutable sp_table *v_Tx = nullptr, *v_Ty = nullptr, *v_mv[3] = {nullptr, nullptr, nullptr};
  // r_b, r_x and r_y are fields of the proof, so the commitment fold and the matrix evaluations can start before the transcript has been replayed:
  // they run as jobs on a second context of the same GPU beside the replay, NovaNIFS::verify and the relaxed Spartan check
  mutable sp_ctx* v_ctx2 = nullptr;
  mutable Worker v_wk;
  ~NNZkKey() {
    v_wk.drain();
    sp_ctx_destroy(v_ctx2);
    sp_table_free(v_Tx);
    sp_table_free(v_Ty);
    for (sp_table* t : v_mv) sp_table_free(t);
    sp_shape_free(S_step);
    sp_shape_free(S_core);
    sp_ck_free(ck);
    sp_ck_free(vc_ck);
  }
};

struct NNPre {  // PrecommittedState, values only
  sp_table* W = nullptr;
  std::vector<aff_t> comm_pre;
  std::vector<fe_t> r_pre, publics;
};
struct NNZkPrep {
  std::vector<NNPre> steps;
  NNPre core;
  std::vector<aff_t> comm_shared;
  std::vector<fe_t> r_shared;
  bool is_small = true;
  // cached_step_matvec / cached_step_i64 (:1520-1590): the step instances' (Az, Bz, Cz) layers and their i64 mirrors; the rounds only read them
  sp_nifs* nifs_cached = nullptr;
  // small device tables reused by every prove (the verifier-circuit instance is a few thousand elements: its commits and two sum-checks would
  // otherwise pay eight hipMalloc / hipFree pairs per prove)
  sp_table* vws[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t vws_cap[5] = {0, 0, 0, 0, 0};
  // Two pieces of prove() read nothing the transcript produces before they are needed: the random relaxed instance of the verifier circuit with its
  // two commitments (values from the randomness tape only, src/r1cs/mod.rs:474-531) and the fold of the step instances' commitments (read first by the
  // opening, :2019-2065). They run as jobs of one helper thread on a SECOND context of the same GPU (own streams and workspaces), under the NIFS rounds
  // and the two batched sum-checks, instead of 0.8 ms each on the critical path.
  // the sixteen working tables of a prove (layers out of the NIFS, core products, pow / eq tables, both poly_ABC and z pairs, the folded witnesses):
  // their sizes are the key's, so they are allocated once (a prove used to pay sixteen hipMalloc / hipFree pairs, the frees after its last phase)
  sp_table* work[16] = {};
  size_t work_cap[16] = {};
  sp_table* rest_stage = nullptr;  // rest segments of several instances side by side (one commitment call for all of them)
  size_t rest_stage_cap = 0;
  sp_ctx* ctx2 = nullptr;
  Worker wk;
  sp_table* bws[2] = {nullptr, nullptr};
  size_t bws_cap[2] = {0, 0};
  struct RandomInstance {
    std::vector<fe_t> Z, rW, rE, E;
    std::vector<aff_t> comm_W, comm_E;
    size_t tape_from = 0, tape_count = 0;
    bool valid = false;
  } rnd;
  // two more jobs of the same kind for the opening (hyrax_pc.rs:387-478, ipa.rs:125-170): `delta`, the commitment of the IPA's mask (tape values only), and
  // - once r_y is known - P_f = <L, folded rows>, P_c = <L, core rows> and beta: comm_LZ = commit(L^T W; <L, blinds>) is the same group element as
  // P_f + c_eval * P_c (W = folded W + c_eval * core W row by row), so the 2048-point MSM behind the last challenge becomes one scalar multiplication
  struct OpeningAhead {
    std::vector<fe_t> dv;
    fe_t r_delta, r_beta;
    aff_t delta, beta, P_f, P_c;
    size_t tape_from = 0, tape_count = 0;
    bool delta_valid = false, points_valid = false;
    sp_fold2_job* lz_fold = nullptr;  // comm_LZ = P_f + c_eval P_c: P_c's doubling ladder, begun by the helper when it has the point

  } open;
  ~NNZkPrep() {
    wk.drain();
    sp_fold_commitments2_drop(open.lz_fold);
    for (sp_table* t : bws) sp_table_free(t);
    for (sp_table* t : work) sp_table_free(t);
    sp_table_free(rest_stage);
    sp_ctx_destroy(ctx2);
    for (auto& s : steps) sp_table_free(s.W);
    sp_table_free(core.W);
    sp_nifs_free(nifs_cached);
    for (sp_table* t : vws) sp_table_free(t);
  }
};

static NNZkKey* nn_setup(sp_ctx* ctx, const R1CSIntView& Rs, const R1CSIntView& Rc, size_t num_steps) {
  auto* pk = new NNZkKey();
  try {
    pk->ctx = ctx;
    pk->num_steps = num_steps;
    // one step means zero NIFS rounds, and the reference's verifier circuit then reads prior_round_vars[round_index - 1] at round 0 (src/zk.rs:637-641):
    // NeutronNovaZkSNARK::setup panics there, this driver refuses
    if (num_steps < 2) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "NeutronNova: at least two step circuits (the verifier circuit has no NIFS round to take its claim from)");
    PaddedShape Ps = pad_shape(Rs), Pc = pad_shape(Rc);
    equalize(Ps, Pc);  // src/neutronnova_zk.rs:1413
    // equalize leaves the shared and precommitted segments alone. Step and core may split their variables into precommitted | rest differently (the folds and
    // the opening work on the combined rows, whose number equalize has made equal); the SHARED segment is one commitment for every circuit (comm_W_shared of
    // the proof is checked against S_step and S_core alike, src/neutronnova_zk.rs:2112-2158), so its padded size has to agree
    if (Ps.dims.num_shared != Pc.dims.num_shared)
      throw Error(SP_ERR_INVALID_INPUT_LENGTH, "NeutronNova: step and core circuits with different padded shared segments (one shared commitment serves both)");
    if (Ps.dims.num_challenges || Pc.dims.num_challenges) throw Error(SP_ERR_INTERNAL, "NeutronNova: step / core circuits with verifier challenges are not driven by this layer");
    // Rest variables (SpartanCircuit::synthesize; the reference's own test circuit is nothing else, src/neutronnova_zk.rs:2357-2418) come with the witness:
    // without challenges, what prove() re-synthesizes (bellpepper/r1cs.rs:443-461) is a function of the circuit alone. BESIDE shared / precommitted variables
    // they cannot be driven: NeutronNovaNIFS::prove folds only that prefix of the step witnesses when it is non-empty and takes the folded rest rows from the
    // blinds ("the rest portion is all zero for step circuits", :1215-1261), so the reference's own proof of such a shape does not verify.
    if (Ps.dims.num_rest_unpadded && Ps.dims.num_shared + Ps.dims.num_precommitted)
      throw Error(SP_ERR_INVALID_INPUT_LENGTH, "NeutronNova: step circuits with rest variables beside shared / precommitted ones (the reference's fold drops the rest segment)");
    pk->dims = Ps.dims;
    pk->dims_core = Pc.dims;
    pk->num_vars = Ps.num_vars();
    auto mk = [&](const PaddedShape& P, sp_shape** out) {
      sp_csr cs[3];
      for (int m = 0; m < 3; ++m) cs[m] = sp_csr{u64p(P.data[m].data()), P.idx[m].data(), P.ptr[m].data()};
      ck(sp_shape_from_csr(ctx, &cs[0], &cs[1], &cs[2], &P.dims, out), "shape_from_csr");
    };
    mk(Ps, &pk->S_step);
    mk(Pc, &pk->S_core);
    pk->gens = from_label("ck", DEFAULT_COMMITMENT_WIDTH + 1);
    ck(sp_ck_create(ctx, u64p(&pk->gens[0].x), DEFAULT_COMMITMENT_WIDTH, u64p(&pk->gens[DEFAULT_COMMITMENT_WIDTH].x), &pk->ck), "ck_create");
    ck(sp_ck_create(ctx, u64p(&pk->gens[0].x), 32, u64p(&pk->gens[32].x), &pk->vc_ck), "vc_ck_create");
    size_t np = 1;
    while (np < num_steps) np <<= 1;
    pk->nb = log2_ceil(np);
    pk->nx = log2_ceil(Ps.dims.num_cons);
    pk->ny = log2_ceil(pk->num_vars) + 1;
    pk->vc = vcirc::Shape::from_circuit(vcirc::Circuit(pk->nb, pk->nx, pk->ny, 32));
    {  // NeutronNovaVerifierKey::write_bytes (src/neutronnova_zk.rs:1305-1333) -> SHA-256 (src/digest.rs:62-76)
      struct Sink {
        sp_wire* w = nullptr;
        ~Sink() { sp_wire_free(w); }
      } sink;
      ck(sp_wire_new(1, &sink.w), "wire sink");
      const uint64_t *g = u64p(&pk->gens[0].x), *h = u64p(&pk->gens[DEFAULT_COMMITMENT_WIDTH].x), *vh = u64p(&pk->gens[32].x);
      ck(sp_wire_hyrax_key(sink.w, g, DEFAULT_COMMITMENT_WIDTH, h), "vk: ck");
      ck(sp_wire_hyrax_key(sink.w, g, DEFAULT_COMMITMENT_WIDTH, h), "vk: vk_ee");  // the same generators (SplitR1CSShape::commitment_key)
      sp_csr cs[3];
      padded_csr(Ps, cs);
      ck(sp_wire_shape(sink.w, &Ps.dims, &cs[0], &cs[1], &cs[2], 1), "vk: S_step");
      padded_csr(Pc, cs);
      ck(sp_wire_shape(sink.w, &Pc.dims, &cs[0], &cs[1], &cs[2], 1), "vk: S_core");
      pk->vc.write_bincode(sink.w, false);  // vc_shape: SplitMultiRoundR1CSShape
      pk->vc.write_bincode(sink.w, true);   // vc_shape_regular: R1CSShape (to_regular_shape)
      ck(sp_wire_hyrax_key(sink.w, g, 32, vh), "vk: vc_ck");
      ck(sp_wire_hyrax_key(sink.w, g, 32, vh), "vk: vc_vk");
      ck(sp_wire_digest(sink.w, pk->vk_digest), "vk digest");
    }
  } catch (...) {
    delete pk;
    throw;
  }
  return pk;
}

// The witness of one circuit as a device table [shared | precommitted | rest] at the padded offsets (bellpepper/r1cs.rs:306-409): the machine words cross the bus
// as they are and become Montgomery-form elements on the device (sp_table_write_u64). `shared_words`: step 0's shared segment - every circuit of a batch
// carries it (src/neutronnova_zk.rs:1485-1488).
static void upload_witness(sp_ctx* ctx, const sp_dims& d, const uint64_t* w, const uint64_t* shared_words, size_t shared_count, sp_table** out) {
  ck(sp_table_zeros(ctx, d.num_shared + d.num_precommitted + d.num_rest, (size_t)-1, (size_t)-1, out), "alloc W");
  ck(sp_table_write_u64(ctx, *out, 0, shared_words, shared_count), "upload W (shared)");
  ck(sp_table_write_u64(ctx, *out, d.num_shared, w + d.num_shared_unpadded, d.num_precommitted_unpadded), "upload W (precommitted)");
  ck(sp_table_write_u64(ctx, *out, d.num_shared + d.num_precommitted, w + d.num_shared_unpadded + d.num_precommitted_unpadded, d.num_rest_unpadded), "upload W (rest)");
}

// prep_prove (:1477-1603): shared commitment from step 0's witness, one precommitted commitment per step and for the core
static NNZkPrep* nn_prep_prove(const NNZkKey& pk, size_t n, const uint64_t* step_wit, size_t wit_len, const uint64_t* step_pub, size_t npub, const uint64_t* core_wit,
                               const uint64_t* core_pub, bool is_small, Tape& tape) {
  const sp_dims& d = pk.dims;
  if (n != pk.num_steps || wit_len != d.num_shared_unpadded + d.num_precommitted_unpadded + d.num_rest_unpadded || npub != d.num_public)
    throw Error(SP_ERR_INVALID_WITNESS_LENGTH, "InvalidWitnessLength");
  auto* ps = new NNZkPrep();
  try {
    sp_ctx* ctx = pk.ctx;
    const size_t CW = DEFAULT_COMMITMENT_WIDTH, rows_sh = d.num_shared / CW;
    ps->is_small = is_small;
    ps->steps.resize(n);
    if (d.num_shared_unpadded) {
      ps->r_shared.resize(rows_sh);
      for (auto& b : ps->r_shared) b = tape.next();
      sp_table* t = nullptr;
      ck(sp_table_zeros(ctx, d.num_shared, (size_t)-1, (size_t)-1, &t), "alloc shared");
      int rc = sp_table_write_u64(ctx, t, 0, step_wit, d.num_shared_unpadded);
      ps->comm_shared.resize(rows_sh);
      if (!rc) rc = sp_hyrax_commit(ctx, pk.ck, t, 0, d.num_shared, u64p(ps->r_shared.data()), is_small ? 1 : 0, u64p(&ps->comm_shared[0].x));
      sp_table_free(t);
      ck(rc, "commit shared");
    }
    auto precommit = [&](const sp_dims& dd, const uint64_t* wit, const uint64_t* pub, NNPre* p) {
      upload_witness(ctx, dd, wit, step_wit, d.num_shared_unpadded, &p->W);  // every circuit shares step 0's shared witness (:1485-1488)
      p->publics.resize(dd.num_public);
      for (size_t i = 0; i < dd.num_public; ++i) p->publics[i] = fe_from_u64<S>(pub[i]);
      if (dd.num_precommitted_unpadded) {
        const size_t rows_pre = dd.num_precommitted / CW;
        p->r_pre.resize(rows_pre);
        for (auto& b : p->r_pre) b = tape.next();
        p->comm_pre.resize(rows_pre);
        ck(sp_hyrax_commit(ctx, pk.ck, p->W, dd.num_shared, dd.num_precommitted, u64p(p->r_pre.data()), is_small ? 1 : 0, u64p(&p->comm_pre[0].x)), "commit precommitted");
      }
    };
    for (size_t i = 0; i < n; ++i) precommit(d, step_wit + i * wit_len, step_pub + i * npub, &ps->steps[i]);
    precommit(pk.dims_core, core_wit, core_pub, &ps->core);
    {  // can_cache_matvec (:1523) always holds here: nn_setup rejects rest variables and challenges, so z = [W | 1 | X] is fully known
      std::vector<fe_t> X(n * d.num_public);
      std::vector<const sp_table*> Ws(n);
      for (size_t i = 0; i < n; ++i) {
        std::copy(ps->steps[i].publics.begin(), ps->steps[i].publics.end(), X.begin() + i * d.num_public);
        Ws[i] = ps->steps[i].W;
      }
      ps->nifs_cached = nifs_prepare(ctx, pk.S_step, d, n, X.data(), Ws.data(), true);
    }
  } catch (...) {
    delete ps;
    throw;
  }
  return ps;
}

struct ProofBuf {
  std::vector<uint64_t> words;
  void pf(const fe_t& f) { words.insert(words.end(), u64p(&f), u64p(&f) + 4); }
  void pp(const aff_t& a) {
    pf(a.x);
    pf(a.y);
  }
  void pc(const std::vector<aff_t>& c) {
    for (const aff_t& a : c) pp(a);
  }
};

static void absorb_instance(Tr& tr, const char* label, const std::vector<aff_t>& comm, const std::vector<fe_t>& X) {  // R1CSInstance bytes (src/r1cs/mod.rs:728-736)
  std::vector<uint8_t> b = commitment_bytes(comm.data(), comm.size());
  const size_t off = b.size();
  b.resize(off + 32 * X.size());
  for (size_t j = 0; j < X.size(); ++j) sp::fe_to_be_bytes<S>(X[j], b.data() + off + 32 * j);
  tr.absorb(label, b.data(), b.size());
}
static std::vector<fe_t> eq_evals(const fe_t* r, size_t ell) {  // EqPolynomial::evals_from_points (src/polys/eq.rs:59-92), host side for O(sqrt) tables
  std::vector<fe_t> ev((size_t)1 << ell, fe_zero());
  ev[0] = fe_one<S>();
  size_t size = 1;
  for (size_t k = ell; k-- > 0;) {
    for (size_t i = 0; i < size; ++i) {
      const fe_t y = fe_mul<S>(ev[i], r[k]);
      ev[size + i] = y;
      ev[i] = fe_sub<S>(ev[i], y);
    }
    size *= 2;
  }
  return ev;
}
// rows of `n` host scalars committed with the width-32 key in one device call (many rows: T, the random instance)
// slot `i` of the prep state's scratch tables holding `n` elements of host data
static sp_table* stage(sp_ctx* ctx, NNZkPrep& ps, int i, const fe_t* data, size_t n) {
  if (n > ps.vws_cap[i]) {  // grow-only, per slot: other slots may be in use by the caller
    sp_table_free(ps.vws[i]);
    ps.vws[i] = nullptr;
    ps.vws_cap[i] = n < 4096 ? 4096 : 2 * n;
  }
  if (!ps.vws[i]) ck(sp_table_zeros(ctx, ps.vws_cap[i], (size_t)-1, (size_t)-1, &ps.vws[i]), "scratch table");
  ck(sp_table_set_len(ps.vws[i], n, (size_t)-1, (size_t)-1), "scratch len");
  ck(sp_table_write(ctx, ps.vws[i], 0, u64p(data), n), "upload");
  return ps.vws[i];
}
// working table `i` of the prep state with `len` elements (grow-only; contents are whatever the last prove left: every user writes what it reads)
static sp_table* work_table(sp_ctx* ctx, NNZkPrep& ps, int i, size_t len) {
  if (len > ps.work_cap[i]) {
    sp_table_free(ps.work[i]);
    ps.work[i] = nullptr;
    ps.work_cap[i] = len;
  }
  if (!ps.work[i]) ck(sp_table_zeros(ctx, ps.work_cap[i], (size_t)-1, (size_t)-1, &ps.work[i]), "working table");
  ck(sp_table_set_len(ps.work[i], len, (size_t)-1, (size_t)-1), "working table len");
  return ps.work[i];
}
// the same for the helper thread's jobs: tables of their own, on the second context
static std::vector<aff_t> commit_rows32_side(NNZkPrep& ps, int slot, const sp_ck* vc_ck, const std::vector<fe_t>& v, const std::vector<fe_t>& blinds) {
  sp_ctx* ctx = ps.ctx2;
  if (v.size() > ps.bws_cap[slot]) {
    sp_table_free(ps.bws[slot]);
    ps.bws[slot] = nullptr;
    ps.bws_cap[slot] = v.size() < 4096 ? 4096 : 2 * v.size();
  }
  if (!ps.bws[slot]) ck(sp_table_zeros(ctx, ps.bws_cap[slot], (size_t)-1, (size_t)-1, &ps.bws[slot]), "scratch table");
  ck(sp_table_set_len(ps.bws[slot], v.size(), (size_t)-1, (size_t)-1), "scratch len");
  ck(sp_table_write(ctx, ps.bws[slot], 0, u64p(v.data()), v.size()), "upload");
  std::vector<aff_t> out(blinds.size());
  ck(sp_hyrax_commit(ctx, vc_ck, ps.bws[slot], 0, v.size(), u64p(blinds.data()), 0, u64p(&out[0].x)), "commit (width 32)");
  return out;
}
static std::vector<aff_t> commit_rows32(sp_ctx* ctx, NNZkPrep& ps, const sp_ck* vc_ck, const std::vector<fe_t>& v, const std::vector<fe_t>& blinds, bool latency = false) {
  std::vector<aff_t> out(blinds.size());
  // the latency form (one launch through mapped memory, the rows added by the library's polling threads) where the commitment sits in the transcript chain
  if (latency && sp_walkers() > 0 && v.size() <= 32 * blinds.size() && blinds.size() * 33 <= 640 &&
      sp_hyrax_commit_rows_host(ctx, vc_ck, u64p(v.data()), v.size(), u64p(blinds.data()), u64p(&out[0].x)) == SP_OK)
    return out;
  sp_table* t = stage(ctx, ps, 0, v.data(), v.size());
  ck(sp_hyrax_commit(ctx, vc_ck, t, 0, v.size(), u64p(blinds.data()), 0, u64p(&out[0].x)), "commit (width 32)");
  return out;
}
// prove_direct (hyrax_pc.rs:609-652) on host vectors
static void prove_direct(size_t num_cols, const std::vector<fe_t>& poly, const std::vector<fe_t>& blind, const fe_t* point, size_t npoint, std::vector<fe_t>* v, fe_t* cb) {
  const size_t n = (size_t)1 << npoint, rows = (n + num_cols - 1) / num_cols;
  if (rows == 1) {
    *v = poly;
    v->resize(num_cols, fe_zero());
    *cb = blind[0];
    return;
  }
  const size_t nvr = log2_ceil(rows);
  const std::vector<fe_t> L = eq_evals(point, nvr);
  v->assign(num_cols, fe_zero());
  for (size_t j = 0; j < L.size(); ++j)
    for (size_t i = 0; i < num_cols; ++i) {
      const size_t k = j * num_cols + i;
      if (k < poly.size()) (*v)[i] = fe_add<S>((*v)[i], fe_mul<S>(L[j], poly[k]));
    }
  *cb = fe_zero();
  for (size_t i = 0; i < blind.size() && i < L.size(); ++i) *cb = fe_add<S>(*cb, fe_mul<S>(L[i], blind[i]));
}

// prove (:1609-2093)
// reference_order: ONE thread, the statements of src/neutronnova_zk.rs:1609-2093 in their order, ABI calls only — no jobs on a second context, the folded
// opening as the single PCS::prove call (sp_hyrax_prove) — what an unchanged neutronnova_zk.rs over the shim of integration/ gets. Same proof.
static ProofBuf nn_prove(const NNZkKey& pk, NNZkPrep& ps, Tape& tape, double* phase_ms, bool reference_order = false) {
  sp_ctx* ctx = pk.ctx;
  const sp_dims& d = pk.dims;
  const size_t CW = DEFAULT_COMMITMENT_WIDTH, n = ps.steps.size(), nv = pk.num_vars, N = d.num_cons;
  const size_t rows_sh = d.num_shared_unpadded ? d.num_shared / CW : 0, rows_pre = d.num_precommitted_unpadded ? d.num_precommitted / CW : 0, rows_rest = d.num_rest / CW;
  const size_t rows = rows_sh + rows_pre + rows_rest, dpub = d.num_public;
  // the core circuit may split the same number of rows differently (nn_setup: equal shared segment, equal total after equalize)
  const sp_dims& dc = pk.dims_core;
  const size_t rows_pre_c = dc.num_precommitted_unpadded ? dc.num_precommitted / CW : 0, rows_rest_c = dc.num_rest / CW;
  if (rows_sh + rows_pre_c + rows_rest_c != rows) throw Error(SP_ERR_INTERNAL, "NeutronNova: step and core rows differ after equalize");
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  static const bool laps = [] {
    const char* e = getenv("SPARTAN_HOST_LAPS");
    return e && e[0] == '1';
  }();
  double t_lap = t_start;
  auto lap = [&](const char* what) {
    if (!laps) return;
    const double t = now();
    fprintf(stderr, "nn_prove lap %-28s %8.3f ms\n", what, t - t_lap);
    t_lap = t;
  };
  ck(sp_ctx_bind_thread(ctx), "device");
  ck(sp_walkers_keep_hot(20000), "walkers");  // the round commitments and the host loops of the verifier-circuit instance are spread over them
  // process_round is host work when its commitments go through the walkers: the batched sum-checks may then queue a round's launch ahead of it
  struct HookPromise {
    sp_ctx* c;
    ~HookPromise() { sp_ctx_round_hooks_host_only(c, 0); }
  } hook_promise{ctx};
  ck(sp_ctx_round_hooks_host_only(ctx, vcirc::split_commitments_on(pk.vc_ck) ? 1 : 0), "round hooks");
  const bool side = !reference_order;  // (false = everything inline on the caller's context: the order the phase comments describe)
  struct SideGuard {  // no exit path leaves a job running on state this call owns
    NNZkPrep& ps;
    ~SideGuard() { ps.wk.drain(); }
  } side_guard{ps};
  ps.rnd.valid = false;
  ps.open.delta_valid = ps.open.points_valid = false;
  size_t t_rnd = 0, t_fold = 0, t_open = 0;  // tickets of the helper's jobs
  if (side) {
    if (!ps.ctx2) ck(sp_ctx_create(sp_ctx_device(ctx), &ps.ctx2), "second context");
    // the random relaxed instance: its values sit at a position of the tape that only the shapes determine (the draws in front of it are the
    // rerandomisation blinds, the rest-row blinds and one blind per committed row of the verifier circuit's rounds); checked again where it is used
    size_t before = ps.comm_shared.size() + ps.core.comm_pre.size() + n * rows_rest + rows_rest_c + pk.vc.total_vars / 32;
    for (const auto& st : ps.steps) before += st.comm_pre.size();
    Tape ahead = tape;
    ahead.pos = tape.pos + before;
    t_rnd = ps.wk.submit([&pk, &ps, ahead]() mutable {
      const vcirc::Shape& vs = pk.vc;
      const size_t vnv = vs.total_vars, vcons = vs.num_cons, vio = vs.num_io();
      NNZkPrep::RandomInstance& R = ps.rnd;
      try {
        ck(sp_ctx_bind_thread(ps.ctx2), "device");
        R.tape_from = ahead.pos;
        R.Z.resize(vnv + vio + 1);
        for (auto& z : R.Z) z = ahead.next();
        R.rW.resize(vnv / 32);
        R.rE.resize(vcons / 32);
        for (auto& b : R.rW) b = ahead.next();
        for (auto& b : R.rE) b = ahead.next();
        R.tape_count = ahead.pos - R.tape_from;
        std::vector<fe_t> mv[3];
        vs.multiply_vec(R.Z, mv);
        const fe_t u = R.Z[vnv];
        R.E.resize(vcons);
        for (size_t i = 0; i < vcons; ++i) R.E[i] = fe_sub<S>(fe_mul<S>(mv[0][i], mv[1][i]), fe_mul<S>(u, mv[2][i]));
        R.comm_W = commit_rows32_side(ps, 0, pk.vc_ck, std::vector<fe_t>(R.Z.begin(), R.Z.begin() + vnv), R.rW);
        R.comm_E = commit_rows32_side(ps, 1, pk.vc_ck, R.E, R.rE);
        R.valid = true;
      } catch (...) {
        R.valid = false;  // the prover computes it inline when it gets there (and reports what fails, if it fails again)
      }
    });
  }
  // rerandomize (:1619-1627, hyrax_pc.rs:321-344): core (shared, precommitted), then every step's precommitted commitment. The new blinds are
  // drawn in the reference's order; the row updates are independent, so all of them go through ONE sp_hyrax_rerandomize call.
  {
    std::vector<std::pair<std::vector<aff_t>*, std::vector<fe_t>*>> parts;
    parts.push_back({&ps.comm_shared, &ps.r_shared});
    parts.push_back({&ps.core.comm_pre, &ps.core.r_pre});
    for (auto& st : ps.steps) parts.push_back({&st.comm_pre, &st.r_pre});
    std::vector<aff_t> all_c;
    std::vector<fe_t> all_old, all_new;
    for (auto& pr_ : parts)
      for (size_t i = 0; i < pr_.first->size(); ++i) {
        all_c.push_back((*pr_.first)[i]);
        all_old.push_back((*pr_.second)[i]);
        all_new.push_back(tape.next());
      }
    if (!all_c.empty()) {
      std::vector<aff_t> out(all_c.size());
      ck(sp_hyrax_rerandomize(ctx, pk.ck, u64p(&all_c[0].x), all_c.size(), u64p(all_old.data()), u64p(all_new.data()), u64p(&out[0].x)), "rerandomize_commitment");
      size_t o = 0;
      for (auto& pr_ : parts)
        for (size_t i = 0; i < pr_.first->size(); ++i, ++o) {
          (*pr_.first)[i] = out[o];
          (*pr_.second)[i] = all_new[o];
        }
    }
  }
  // instances and witnesses (:1662-1719): rest rows = commit_zeros (h * blind) — one sp_fixed_base_mul_h call for the rest rows of all instances — or, for a
  // circuit with rest variables, the commitment of its rest segment under the same blinds (bellpepper/r1cs.rs:463-500); no challenges for these circuits
  ProofBuf proof;
  proof.pc(ps.comm_shared);
  std::vector<aff_t> comms(n * rows);
  std::vector<fe_t> X(n * dpub), r_W(n * rows);
  std::vector<const sp_table*> Ws(n);
  std::vector<fe_t> all_rest(n * rows_rest + rows_rest_c);
  for (auto& b : all_rest) b = tape.next();  // steps in order, then the core: the reference's call order
  std::vector<aff_t> all_c_rest(all_rest.size());
  if (!all_rest.empty() && !(d.num_rest_unpadded && pk.dims_core.num_rest_unpadded))
    ck(sp_fixed_base_mul_h(ctx, pk.ck, u64p(all_rest.data()), all_rest.size(), u64p(&all_c_rest[0].x)), "commit_zeros");
  {
    // the rest segments of several instances side by side in one table are ONE row-wise commitment (a Hyrax commitment is one MSM per 2048-entry row): a call
    // per instance costs 0.25 ms of fixed work each (64 instances of the reference's two-block test circuit: 15.9 ms of a 23 ms prove)
    struct RestJob {
      const NNPre* p;
      size_t off, len, blind_at, nrows;  // the segment inside p->W, its blinds / rows inside all_rest / all_c_rest
    };
    std::vector<RestJob> todo;
    if (d.num_rest_unpadded)
      for (size_t i = 0; i < n; ++i) todo.push_back({&ps.steps[i], d.num_shared + d.num_precommitted, d.num_rest, i * rows_rest, rows_rest});
    if (dc.num_rest_unpadded) todo.push_back({&ps.core, dc.num_shared + dc.num_precommitted, dc.num_rest, n * rows_rest, rows_rest_c});
    const size_t cap = (size_t)1 << 25;  // <= 1 GiB of staged elements per call
    for (size_t at = 0; at < todo.size();) {
      size_t cnt = 1, elems = todo[at].len;
      while (at + cnt < todo.size() && elems + todo[at + cnt].len <= cap) elems += todo[at + cnt++].len;
      if (cnt == 1) {
        const RestJob& j = todo[at];
        ck(sp_hyrax_commit(ctx, pk.ck, j.p->W, j.off, j.len, u64p(all_rest.data() + j.blind_at), ps.is_small ? 1 : 0, u64p(&all_c_rest[j.blind_at].x)), "commit rest");
        at += 1;
        continue;
      }
      if (elems > ps.rest_stage_cap) {
        sp_table_free(ps.rest_stage);
        ps.rest_stage = nullptr;
        ps.rest_stage_cap = elems;
        ck(sp_table_zeros(ctx, ps.rest_stage_cap, (size_t)-1, (size_t)-1, &ps.rest_stage), "rest staging table");
      }
      ck(sp_table_set_len(ps.rest_stage, elems, (size_t)-1, (size_t)-1), "rest staging len");
      std::vector<fe_t> bl;
      size_t pos = 0;
      for (size_t k = 0; k < cnt; ++k) {
        const RestJob& j = todo[at + k];
        ck(sp_table_copy(ctx, ps.rest_stage, pos, j.p->W, j.off, j.len), "stage rest segment");
        pos += j.len;
        bl.insert(bl.end(), all_rest.begin() + j.blind_at, all_rest.begin() + j.blind_at + j.nrows);
      }
      std::vector<aff_t> out(bl.size());
      ck(sp_hyrax_commit(ctx, pk.ck, ps.rest_stage, 0, elems, u64p(bl.data()), ps.is_small ? 1 : 0, u64p(&out[0].x)), "commit rest (batched)");
      size_t o = 0;
      for (size_t k = 0; k < cnt; ++k) {
        const RestJob& j = todo[at + k];
        std::copy(out.begin() + o, out.begin() + o + j.nrows, all_c_rest.begin() + j.blind_at);
        o += j.nrows;
      }
      at += cnt;
    }
  }
  auto instance = [&](NNPre& p, size_t which, aff_t* comm_out, fe_t* r_out) {
    const size_t my_pre = which == n ? rows_pre_c : rows_pre, my_rest = which == n ? rows_rest_c : rows_rest;
    const fe_t* r_rest = all_rest.data() + which * rows_rest;  // (the core's blinds follow the n steps')
    const aff_t* c_rest = all_c_rest.data() + which * rows_rest;
    if (p.comm_pre.size() != my_pre) throw Error(SP_ERR_INTERNAL, "NeutronNova: precommitted rows of an instance");
    std::copy(ps.comm_shared.begin(), ps.comm_shared.end(), comm_out);
    std::copy(p.comm_pre.begin(), p.comm_pre.end(), comm_out + rows_sh);
    std::copy(c_rest, c_rest + my_rest, comm_out + rows_sh + my_pre);
    std::copy(ps.r_shared.begin(), ps.r_shared.end(), r_out);
    std::copy(p.r_pre.begin(), p.r_pre.end(), r_out + rows_sh);
    std::copy(r_rest, r_rest + my_rest, r_out + rows_sh + my_pre);
    proof.pc(p.comm_pre);
    for (size_t i = 0; i < my_rest; ++i) proof.pp(c_rest[i]);
    for (const fe_t& f : p.publics) proof.pf(f);
  };
  for (size_t i = 0; i < n; ++i) {
    instance(ps.steps[i], i, &comms[i * rows], &r_W[i * rows]);
    std::copy(ps.steps[i].publics.begin(), ps.steps[i].publics.end(), X.begin() + i * dpub);
    Ws[i] = ps.steps[i].W;
  }
  std::vector<aff_t> core_comm(rows);
  std::vector<fe_t> core_rW(rows);
  instance(ps.core, n, core_comm.data(), core_rW.data());
  // the opening folds comm = folded + c_eval * core rows and comm_eval = eW_step + c_eval * eW_core with a weight drawn at the very end: the doubling
  // ladders of the core rows (now) and of the core evaluation's commitment (behind its round) are built on the library's polling threads meanwhile
  struct Fold2Guard {
    sp_fold2_job *rows = nullptr, *eval = nullptr;
    ~Fold2Guard() {
      sp_fold_commitments2_drop(rows);
      sp_fold_commitments2_drop(eval);
    }
  } fold2;
  if (side) ck(sp_fold_commitments2_begin(ctx, u64p(&core_comm[0].x), rows, &fold2.rows), "fold_commitments (ladders)");
  const double t_inst = now();

  Tr tr(ctx, "neutronnova_prove");
  tr.absorb("vk", pk.vk_digest, 32);
  absorb_instance(tr, "core_instance", core_comm, ps.core.publics);
  vcirc::Circuit vc(pk.nb, pk.nx, pk.ny, 32);
  vcirc::State vst(pk.vc);
  struct HookCtx {
    const NNZkKey* pk;
    vcirc::Circuit* vc;
    vcirc::State* st;
    Tr* tr;
    Tape* tape;
    size_t outer_start, inner_start;
    std::exception_ptr err;
  } hc{&pk, &vc, &vst, &tr, &tape, pk.nb + 1, pk.nb + 1 + pk.nx + 1, nullptr};
  // NIFS (:1770-1783): finish_round! = vc.nifs_polys[t] <- coefficients, then process_round (:703-735); once more after the rounds (:1207-1210)
  auto nifs_hook = [](void* u, size_t t, const uint64_t* co, uint64_t* r_b) {
    HookCtx* h = (HookCtx*)u;
    if (h->err) return;
    try {
      if (t < h->pk->nb) {
        for (int q = 0; q < 4; ++q) memcpy(&h->vc->nifs_polys[t][q], co + 4 * q, 32);
        const fe_t r = vcirc::process_round(h->pk->ctx, *h->st, h->pk->vc, h->pk->vc_ck, *h->vc, t, *h->tr, *h->tape)[0];
        memcpy(r_b, &r, 32);
      } else {
        memcpy(&h->vc->t_out_step, co, 32);
        memcpy(&h->vc->eq_rho_at_rb, co + 4, 32);
        vcirc::process_round(h->pk->ctx, *h->st, h->pk->vc, h->pk->vc_ck, *h->vc, t, *h->tr, *h->tape);
      }
    } catch (...) {
      h->err = std::current_exception();
    }
  };
  size_t ell, left, right;
  compute_tensor_decomp(N, &ell, &left, &right);
  const size_t nb = pk.nb;
  std::vector<uint64_t> polys(16 * std::max<size_t>(nb, 1)), r_bs(4 * std::max<size_t>(nb, 1)), E_eq(4 * (left + right)), tail(8), f_rW(4 * rows), f_X(4 * std::max<size_t>(dpub, 1));
  std::vector<aff_t> f_comm(rows);
  SideGuard side_guard_rows{ps};  // f_comm and core_comm are read and written by the helper's jobs: drained before they go out of scope
  sp_table *A = work_table(ctx, ps, 0, N), *B = work_table(ctx, ps, 1, N), *C = work_table(ctx, ps, 2, N), *fW = work_table(ctx, ps, 3, nv);
  sp_table *core_abc[3] = {work_table(ctx, ps, 4, N), work_table(ctx, ps, 5, N), work_table(ctx, ps, 6, N)}, *zc = work_table(ctx, ps, 7, nv + 1 + dpub);
  sp_table *pl = work_table(ctx, ps, 8, left), *pr = work_table(ctx, ps, 9, right), *rx = work_table(ctx, ps, 10, N);
  sp_table *abc_s = work_table(ctx, ps, 11, 2 * nv), *abc_c = work_table(ctx, ps, 12, 2 * nv), *zs = work_table(ctx, ps, 13, 2 * nv), *zcc = work_table(ctx, ps, 14, 2 * nv);
  sp_table* Wf = work_table(ctx, ps, 15, nv);
  NifsOutputs no{polys.data(), r_bs.data(), E_eq.data(), tail.data(), f_rW.data(), f_X.data(), (uint64_t*)f_comm.data(), A, B, C, fW};
  std::function<void(sp_ctx*)> fold_job;
  if (side) no.deferred_fold_commitments = &fold_job;
  nifs_prove(ctx, pk.S_step, d, pk.ck, n, rows, comms.data(), X.data(), Ws.data(), r_W.data(), true, ps.nifs_cached, tr.t, nifs_hook, &hc, no);
  if (hc.err) std::rethrow_exception(hc.err);
  if (fold_job) {
    t_fold = ps.wk.submit([&ps, job = std::move(fold_job)] {
      ck(sp_ctx_bind_thread(ps.ctx2), "device");
      job(ps.ctx2);
    });
    // delta (ipa.rs:139-147): the mask and its blinds follow the random instance and NovaNIFS's r_T on the tape
    Tape ahead = tape;
    {
      const vcirc::Shape& vs = pk.vc;
      ahead.pos = tape.pos + (pk.vc.total_vars / 32 - vst.commits) + (vs.total_vars + vs.num_io() + 1) + vs.total_vars / 32 + 2 * (vs.num_cons / 32);
    }
    ps.wk.submit([&pk, &ps, ahead]() mutable {
      NNZkPrep::OpeningAhead& O = ps.open;
      try {
        O.tape_from = ahead.pos;
        O.dv.resize(DEFAULT_COMMITMENT_WIDTH);
        for (auto& x : O.dv) x = ahead.next();
        O.r_delta = ahead.next();
        O.r_beta = ahead.next();
        O.tape_count = ahead.pos - O.tape_from;
        ck(sp_msm_ck(ps.ctx2, pk.ck, u64p(O.dv.data()), O.dv.size(), u64p(&O.r_delta), u64p(&O.delta.x)), "delta");
        O.delta_valid = true;
      } catch (...) {
        O.delta_valid = false;
      }
    });
  }
  const double t_nifs = now();

  // core products, batched outer sum-check (:1786-1850)
  const fe_t one = fe_one<S>();
  {
    ck(sp_table_copy(ctx, zc, 0, ps.core.W, 0, nv), "z <- W");
    std::vector<fe_t> tl(1 + dpub);
    tl[0] = one;
    std::copy(ps.core.publics.begin(), ps.core.publics.end(), tl.begin() + 1);
    ck(sp_table_write(ctx, zc, nv, u64p(tl.data()), tl.size()), "z tail");
    ck(sp_multiply_vec(ctx, pk.S_core, zc, core_abc[0], core_abc[1], core_abc[2]), "multiply_vec (core)");
  }
  ck(sp_table_write(ctx, pl, 0, E_eq.data(), left), "pow left");
  ck(sp_table_write(ctx, pr, 0, E_eq.data() + 4 * left, right), "pow right");
  auto batched_hook = [](void* u, size_t round, const uint64_t* cs_, const uint64_t* cc_, size_t ncoeffs, uint64_t r_out[4]) -> int {
    HookCtx* h = (HookCtx*)u;
    try {
      if (ncoeffs == 4) {
        const size_t i = round - h->outer_start;
        for (int q = 0; q < 4; ++q) {
          memcpy(&h->vc->outer_step[i][q], cs_ + 4 * q, 32);
          memcpy(&h->vc->outer_core[i][q], cc_ + 4 * q, 32);
        }
      } else {
        const size_t j = round - h->inner_start;
        for (int q = 0; q < 3; ++q) {
          memcpy(&h->vc->inner_step[j][q], cs_ + 4 * q, 32);
          memcpy(&h->vc->inner_core[j][q], cc_ + 4 * q, 32);
        }
      }
      const fe_t r = vcirc::process_round(h->pk->ctx, *h->st, h->pk->vc, h->pk->vc_ck, *h->vc, round, *h->tr, *h->tape)[0];
      memcpy(r_out, &r, 32);
      return 0;
    } catch (...) {
      h->err = std::current_exception();
      return SP_ERR_INTERNAL;
    }
  };
  std::vector<fe_t> r_x(pk.nx);
  {
    int rc = sp_sumcheck_cubic_outer_pow_batched(ctx, pk.nx, pl, pr, A, B, C, core_abc[0], core_abc[1], core_abc[2], tail.data(), hc.outer_start, batched_hook, &hc, u64p(r_x.data()));
    if (hc.err) std::rethrow_exception(hc.err);
    ck(rc, "outer sum-check (batched)");
  }
  {
    sp_table* cl[6] = {A, B, C, core_abc[0], core_abc[1], core_abc[2]};
    fe_t v[6];
    for (int q = 0; q < 6; ++q) ck(sp_table_read(ctx, cl[q], 0, 1, u64p(&v[q])), "claims");
    for (int q = 0; q < 3; ++q) {
      vc.claim_step[q] = v[q];
      vc.claim_core[q] = v[3 + q];
    }
    ck(sp_table_read(ctx, pl, 0, 1, u64p(&vc.tau_at_rx)), "tau_at_rx");
  }
  const fe_t r = vcirc::process_round(ctx, vst, pk.vc, pk.vc_ck, vc, hc.outer_start + pk.nx, tr, tape)[0];
  const double t_outer = now();
  const fe_t r2 = fe_mul<S>(r, r);
  fe_t claims[2] = {fe_add<S>(fe_add<S>(vc.claim_step[0], fe_mul<S>(r, vc.claim_step[1])), fe_mul<S>(r2, vc.claim_step[2])),
                    fe_add<S>(fe_add<S>(vc.claim_core[0], fe_mul<S>(r, vc.claim_core[1])), fe_mul<S>(r2, vc.claim_core[2]))};
  // evals_rx, both poly_ABC, the z tables, batched inner sum-check (:1852-1945)
  ck(sp_eq_table_into(ctx, u64p(r_x.data()), pk.nx, rx), "evals_rx");
  ck(sp_poly_abc(ctx, pk.S_step, rx, u64p(&r), 2 * nv, abc_s), "poly_ABC (step)");
  ck(sp_poly_abc(ctx, pk.S_core, rx, u64p(&r), 2 * nv, abc_c), "poly_ABC (core)");
  std::vector<fe_t> folded_X(dpub);
  memcpy(folded_X.data(), f_X.data(), dpub * sizeof(fe_t));
  auto fill_z = [&](sp_table* z, const sp_table* W, const std::vector<fe_t>& Xv) {
    ck(sp_table_copy(ctx, z, 0, W, 0, nv), "z <- W");
    std::vector<fe_t> tl(1 + Xv.size());
    tl[0] = one;
    std::copy(Xv.begin(), Xv.end(), tl.begin() + 1);
    ck(sp_table_write(ctx, z, nv, u64p(tl.data()), tl.size()), "z tail");
    ck(sp_table_zero(ctx, z, nv + tl.size(), nv - tl.size()), "z: zero high half");  // a working table: the last prove's rounds were bound in place
  };
  fill_z(zs, fW, folded_X);
  fill_z(zcc, ps.core.W, ps.core.publics);
  for (sp_table* t : {abc_s, abc_c, zs, zcc}) ck(sp_table_set_len(t, 2 * nv, nv, 1 + dpub), "halves");
  std::vector<fe_t> r_y(pk.ny);
  fe_t fin[4];
  {
    int rc = sp_sumcheck_quad_batched(ctx, u64p(claims), pk.ny, abc_s, abc_c, zs, zcc, hc.inner_start, batched_hook, &hc, u64p(r_y.data()), u64p(fin));
    if (hc.err) std::rethrow_exception(hc.err);
    ck(rc, "inner sum-check (batched)");
  }
  if (t_fold) {  // r_y is known: the halves of comm_LZ and beta, under the verifier-circuit phase
    const size_t nvr = log2_ceil(rows);
    t_open = ps.wk.submit([&pk, &ps, &f_comm, &core_comm, rows, L = eq_evals(r_y.data() + 1, nvr), Rv = eq_evals(r_y.data() + 1 + nvr, pk.ny - 1 - nvr)] {
      NNZkPrep::OpeningAhead& O = ps.open;
      try {
        if (!O.delta_valid || Rv.size() != O.dv.size() || L.size() > rows) return;
        ck(sp_ctx_bind_thread(ps.ctx2), "device");
        ck(sp_msm(ps.ctx2, u64p(L.data()), u64p(&f_comm[0].x), L.size(), u64p(&O.P_f.x)), "<L, folded rows>");
        ck(sp_msm(ps.ctx2, u64p(L.data()), u64p(&core_comm[0].x), L.size(), u64p(&O.P_c.x)), "<L, core rows>");
        fe_t ip = fe_zero();
        for (size_t i = 0; i < Rv.size(); ++i) ip = fe_add<S>(ip, fe_mul<S>(Rv[i], O.dv[i]));
        ck(sp_hyrax_commit_small(ps.ctx2, pk.vc_ck, u64p(&ip), 1, u64p(&O.r_beta), u64p(&O.beta.x)), "beta");
        sp_fold_commitments2_drop(O.lz_fold);
        O.lz_fold = nullptr;
        if (sp_fold_commitments2_begin(ps.ctx2, u64p(&O.P_c.x), 1, &O.lz_fold) != SP_OK) O.lz_fold = nullptr;
        O.points_valid = true;

      } catch (...) {
        O.points_valid = false;
      }
    });
  }
  // The opening's W = folded_W + c_eval core_W exists only once c_eval is drawn, at the very end - but L^T W (bind_with_delayed, hyrax_pc.rs:38-54) is linear in
  // the table: LZ = L^T folded_W + c_eval L^T core_W, and L = eq(r_y[1..]) is known now. The two products run as jobs of the library's vector stream under the
  // verifier-circuit phase (one job at a time: the second is begun where the first is collected).
  struct LzAhead {
    sp_ctx* ctx;
    sp_vec_job* job = nullptr;
    size_t cols = 0;
    std::vector<fe_t> f, c;
    int have = 0;  // 1: f, 2: f and c
    ~LzAhead() {
      if (job) {
        std::vector<fe_t> sink(cols);
        (void)sp_rowmat_vec_eq_finish(ctx, job, u64p(sink.data()));
      }
    }
  } lz{ctx};
  if (side) {
    const size_t nvr = log2_ceil(rows);
    lz.cols = (size_t)1 << (pk.ny - 1 - nvr);
    if (sp_rowmat_vec_eq_begin(ctx, fW, u64p(r_y.data() + 1), nvr, lz.cols, &lz.job) != SP_OK) lz.job = nullptr;
  }
  auto lz_step = [&] {  // collect the product in flight, begin the next
    if (!lz.job) return;
    const size_t nvr = log2_ceil(rows);
    std::vector<fe_t>& dst = lz.have == 0 ? lz.f : lz.c;
    dst.resize(lz.cols);
    sp_vec_job* j = lz.job;
    lz.job = nullptr;
    if (sp_rowmat_vec_eq_finish(ctx, j, u64p(dst.data())) != SP_OK) return;
    if (++lz.have == 1 && sp_rowmat_vec_eq_begin(ctx, ps.core.W, u64p(r_y.data() + 1), nvr, lz.cols, &lz.job) != SP_OK) lz.job = nullptr;
  };
  auto eval_X = [&](const std::vector<fe_t>& Xv) {
    std::vector<fe_t> v{one};
    v.insert(v.end(), Xv.begin(), Xv.end());
    return sparse_poly_evaluate(pk.ny - 1, v, r_y.data() + 1);
  };
  vc.eval_X_step = eval_X(folded_X);
  vc.eval_X_core = eval_X(ps.core.publics);
  const fe_t den = fe_sub<S>(one, r_y[0]);
  if (fe_is_zero(den)) throw Error(SP_ERR_DIVISION_BY_ZERO, "DivisionByZero");
  const fe_t inv = fe_inv_vartime<S>(den);
  vc.eval_W_step = fe_mul<S>(fe_sub<S>(fin[2], fe_mul<S>(r_y[0], vc.eval_X_step)), inv);
  vc.eval_W_core = fe_mul<S>(fe_sub<S>(fin[3], fe_mul<S>(r_y[0], vc.eval_X_core)), inv);
  const size_t inner_final = hc.inner_start + pk.ny;
  for (size_t k = 0; k < 3; ++k) vcirc::process_round(ctx, vst, pk.vc, pk.vc_ck, vc, inner_final + k, tr, tape);
  if (side) ck(sp_fold_commitments2_begin(ctx, u64p(&vst.comm_per_round[inner_final + 2][0].x), 1, &fold2.eval), "fold eval commitments (ladder)");
  const double t_inner = now();

  // finalize_multiround_witness (:1948-1952): U_verifier, its regular form, W_verifier
  const vcirc::Shape& vs = pk.vc;
  std::vector<fe_t> vpub(vst.cs.inputs.begin() + 1 + vs.total_challenges, vst.cs.inputs.end());
  std::vector<aff_t> Uv_comm;
  std::vector<fe_t> Uv_X, Wv_r;
  for (const auto& c : vst.comm_per_round) Uv_comm.insert(Uv_comm.end(), c.begin(), c.end());
  for (const auto& c : vst.challenges) Uv_X.insert(Uv_X.end(), c.begin(), c.end());
  Uv_X.insert(Uv_X.end(), vpub.begin(), vpub.end());
  for (const auto& b : vst.blind_per_round) Wv_r.insert(Wv_r.end(), b.begin(), b.end());
  lap("(up to the vc instance)");
  if (laps) fprintf(stderr, "nn_prove: process_round totals: synthesis %.3f ms, commitments %.3f ms, transcript %.3f ms over %zu rounds\n", vst.synth_ms, vst.commit_ms, vst.hash_ms, vst.current);
  // sample_random_instance_witness (src/r1cs/mod.rs:474-531) on the verifier-circuit shape
  const size_t vnv = vs.total_vars, vcons = vs.num_cons, vio = vs.num_io();
  if (t_rnd) ps.wk.wait(t_rnd);
  const bool ahead_ok = side && ps.rnd.valid && ps.rnd.tape_from == tape.pos;
  std::vector<fe_t> Z, rnd_rW, rnd_rE, rnd_E, mv[3];
  std::vector<aff_t> rnd_comm_W, rnd_comm_E;
  if (ahead_ok) {
    Z.swap(ps.rnd.Z);
    rnd_rW.swap(ps.rnd.rW);
    rnd_rE.swap(ps.rnd.rE);
    rnd_E.swap(ps.rnd.E);
    rnd_comm_W.swap(ps.rnd.comm_W);
    rnd_comm_E.swap(ps.rnd.comm_E);
    tape.skip(ps.rnd.tape_count);
  } else {
    if (laps && side) fprintf(stderr, "nn_prove: random instance computed inline (job valid %d, tape %zu vs %zu)\n", (int)ps.rnd.valid, ps.rnd.tape_from, tape.pos);
    Z.resize(vnv + vio + 1);
    for (auto& z : Z) z = tape.next();
    rnd_rW.resize(vnv / 32);
    rnd_rE.resize(vcons / 32);
    for (auto& b : rnd_rW) b = tape.next();
    for (auto& b : rnd_rE) b = tape.next();
    vs.multiply_vec(Z, mv);
    rnd_E.resize(vcons);
    for (size_t i = 0; i < vcons; ++i) rnd_E[i] = fe_sub<S>(fe_mul<S>(mv[0][i], mv[1][i]), fe_mul<S>(Z[vnv], mv[2][i]));
    rnd_comm_W = commit_rows32(ctx, ps, pk.vc_ck, std::vector<fe_t>(Z.begin(), Z.begin() + vnv), rnd_rW);
    rnd_comm_E = commit_rows32(ctx, ps, pk.vc_ck, rnd_E, rnd_rE);
  }
  const fe_t rnd_u = Z[vnv];
  const std::vector<fe_t> rnd_W(Z.begin(), Z.begin() + vnv), rnd_X(Z.begin() + vnv + 1, Z.end());
  lap("random instance + 2 commits");
  // NovaNIFS::prove (src/nifs.rs:34-61)
  {
    std::vector<uint8_t> b = commitment_bytes(rnd_comm_W.data(), rnd_comm_W.size()), e = commitment_bytes(rnd_comm_E.data(), rnd_comm_E.size());
    b.insert(b.end(), e.begin(), e.end());
    const size_t off = b.size();
    b.resize(off + 32 * (1 + rnd_X.size()));
    sp::fe_to_be_bytes<S>(rnd_u, b.data() + off);
    for (size_t j = 0; j < rnd_X.size(); ++j) sp::fe_to_be_bytes<S>(rnd_X[j], b.data() + off + 32 * (1 + j));
    tr.absorb("U1", b.data(), b.size());
  }
  absorb_instance(tr, "U2", Uv_comm, Uv_X);
  std::vector<fe_t> r_T(vcons / 32);
  for (auto& b : r_T) b = tape.next();
  std::vector<fe_t> Zs(vnv + 1 + vio);
  par_for(vnv, 256, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) Zs[i] = fe_add<S>(rnd_W[i], vst.w[i]);
  });
  const fe_t u1 = fe_add<S>(rnd_u, one);
  Zs[vnv] = u1;
  for (size_t i = 0; i < vio; ++i) Zs[vnv + 1 + i] = fe_add<S>(rnd_X[i], Uv_X[i]);
  vs.multiply_vec(Zs, mv);
  std::vector<fe_t> T(vcons);
  par_for(vcons, 48, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) T[i] = fe_sub<S>(fe_sub<S>(fe_mul<S>(mv[0][i], mv[1][i]), fe_mul<S>(u1, mv[2][i])), rnd_E[i]);
  });
  lap("NovaNIFS: absorbs, Z, multiply_vec, T");
  const std::vector<aff_t> comm_T = commit_rows32(ctx, ps, pk.vc_ck, T, r_T, side);
  lap("NovaNIFS: commit_T");
  {
    const std::vector<uint8_t> b = commitment_bytes(comm_T.data(), comm_T.size());
    tr.absorb("comm_T", b.data(), b.size());
  }
  lz_step();  // L^T folded_W is in; L^T core_W goes out
  lap("NovaNIFS: T + commit_T");
  const fe_t rf = tr.squeeze("r");
  std::vector<fe_t> Wfold(vnv), Efold(vcons), rWfold(rnd_rW.size()), rEfold(rnd_rE.size()), Xfold(vio);
  par_for(vnv + vcons, 96, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      if (i < vnv) Wfold[i] = fe_add<S>(rnd_W[i], fe_mul<S>(rf, vst.w[i]));
      else Efold[i - vnv] = fe_add<S>(rnd_E[i - vnv], fe_mul<S>(rf, T[i - vnv]));
    }
  });
  for (size_t i = 0; i < rWfold.size(); ++i) rWfold[i] = fe_add<S>(rnd_rW[i], fe_mul<S>(rf, Wv_r[i]));
  for (size_t i = 0; i < rEfold.size(); ++i) rEfold[i] = fe_add<S>(rnd_rE[i], fe_mul<S>(rf, r_T[i]));
  for (size_t i = 0; i < vio; ++i) Xfold[i] = fe_add<S>(rnd_X[i], fe_mul<S>(rf, Uv_X[i]));
  const fe_t ufold = fe_add<S>(rnd_u, rf);
  // RelaxedR1CSSpartanProof::prove (src/spartan_relaxed.rs:98-213)
  tr.absorb_scalars("u_relaxed", &ufold, 1);
  tr.absorb_scalars("X_relaxed", Xfold.data(), Xfold.size());
  const size_t vlx = log2_ceil(vcons), vnvp = next_pow2(vnv), vly = log2_ceil(vnvp) + 1, vz_len = 2 * vnvp;
  std::vector<fe_t> zr = Wfold;
  zr.push_back(ufold);
  zr.insert(zr.end(), Xfold.begin(), Xfold.end());
  vs.multiply_vec(zr, mv);
  std::vector<fe_t> vtau(vlx);
  for (auto& t : vtau) t = tr.squeeze("t");
  std::vector<fe_t> uczE(vcons);
  par_for(vcons, 96, [&](size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) uczE[i] = fe_add<S>(fe_mul<S>(ufold, mv[2][i]), Efold[i]);
  });
  std::vector<fe_t> v_outer(3 * vlx), v_rx(vlx), v_inner(2 * vly), v_ry(vly);
  fe_t v_claims[3];
  // the two relaxed-Spartan sum-checks run on tables this driver holds on the host (2^9 and 2^12 elements): with polling host threads in the process
  // every round is run there (sp_sumcheck_*_host); staged to the device they are a trip over the bus a round
  static const bool host_sc_off = [] {
    const char* e = getenv("SPARTAN_HOST_SC");  // "0": the device provers on staged tables (A/B runs)
    return e && e[0] == '0';
  }();
  const bool host_sc = side && sp_walkers() > 0 && !host_sc_off;
  if (host_sc) {
    const fe_t zero = fe_zero();
    ck(sp_sumcheck_cubic3_host(ctx, u64p(&zero), u64p(vtau.data()), vlx, u64p(mv[0].data()), u64p(mv[1].data()), u64p(uczE.data()), tr.t, u64p(v_outer.data()),
                               u64p(v_rx.data()), u64p(v_claims)),
       "relaxed outer sum-check (host tables)");
  } else {
    sp_table *ta = stage(ctx, ps, 0, mv[0].data(), vcons), *tb = stage(ctx, ps, 1, mv[1].data(), vcons), *tc = stage(ctx, ps, 2, uczE.data(), vcons);
    const fe_t zero = fe_zero();
    ck(sp_sumcheck_cubic3(ctx, u64p(&zero), u64p(vtau.data()), vlx, ta, tb, tc, tr.t, u64p(v_outer.data()), u64p(v_rx.data()), u64p(v_claims)), "relaxed outer sum-check");
  }
  lap("folds + relaxed outer sc");
  tr.absorb_scalars("claims_outer", v_claims, 3);
  const fe_t vr = tr.squeeze("r"), vr2 = fe_mul<S>(vr, vr);
  const std::vector<fe_t> v_evals_rx = eq_evals(v_rx.data(), vlx);
  fe_t claim_E = fe_zero();
  {
    fe_t partial[32];
    for (auto& x : partial) x = fe_zero();
    std::atomic<unsigned> slot{0};
    par_for(vcons, 64, [&](size_t lo, size_t hi) {
      fe_t acc = fe_zero();
      for (size_t i = lo; i < hi; ++i) acc = fe_add<S>(acc, fe_mul<S>(Efold[i], v_evals_rx[i]));
      partial[slot.fetch_add(1) & 31u] = acc;
    });
    for (const auto& x : partial) claim_E = fe_add<S>(claim_E, x);
  }
  const fe_t v_claim_inner = fe_add<S>(fe_add<S>(v_claims[0], fe_mul<S>(vr, v_claims[1])), fe_mul<S>(vr2, fe_sub<S>(v_claims[2], claim_E)));
  const size_t vcols = vs.num_cols();
  std::vector<fe_t> vabc(vz_len, fe_zero());
  {
    std::vector<fe_t> ev[3];
    // bind_matrix_row_vars (:22-41): a scatter into the columns of one matrix - the three matrices side by side
    par_for(3, 1, [&](size_t lo, size_t hi) {
      for (size_t m = lo; m < hi; ++m) {
        ev[m].assign(vcols, fe_zero());
        for (size_t row = 0; row < vcons; ++row) {
          if (fe_is_zero(v_evals_rx[row])) continue;
          for (uint64_t k = vs.M[m].ptr[row]; k < vs.M[m].ptr[row + 1]; ++k)
            ev[m][vs.M[m].idx[k]] = fe_add<S>(ev[m][vs.M[m].idx[k]], fe_mul<S>(v_evals_rx[row], vs.M[m].data[k]));
        }
      }
    });
    const fe_t r2u = fe_mul<S>(vr2, ufold);
    par_for(vcols, 128, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) vabc[i] = fe_add<S>(fe_add<S>(ev[0][i], fe_mul<S>(vr, ev[1][i])), fe_mul<S>(r2u, ev[2][i]));
    });
  }
  lap("claim_E + bind_matrix_rows");
  zr.resize(vz_len, fe_zero());
  if (host_sc) {
    fe_t ci[2];
    ck(sp_sumcheck_quad_host(ctx, u64p(&v_claim_inner), vly, u64p(vabc.data()), u64p(zr.data()), tr.t, u64p(v_inner.data()), u64p(v_ry.data()), u64p(ci)),
       "relaxed inner sum-check (host tables)");
  } else {
    sp_table *ta = stage(ctx, ps, 3, vabc.data(), vz_len), *tb = stage(ctx, ps, 4, zr.data(), vz_len);
    fe_t ci[2];
    ck(sp_sumcheck_quad(ctx, u64p(&v_claim_inner), vly, ta, tb, tr.t, u64p(v_inner.data()), u64p(v_ry.data()), u64p(ci)), "relaxed inner sum-check");
  }
  lap("relaxed inner sc");
  std::vector<fe_t> v_W, v_E;
  fe_t blind_vW, blind_vE;
  prove_direct(32, Wfold, rWfold, v_ry.data() + 1, vly - 1, &v_W, &blind_vW);
  prove_direct(32, Efold, rEfold, v_rx.data(), vlx, &v_E, &blind_vE);
  tr.absorb_scalars("v_W", v_W.data(), v_W.size());
  tr.absorb_scalars("v_E", v_E.data(), v_E.size());
  lap("prove_direct x2");
  const double t_vc = now();

  // fold the two evaluation claims (:2019-2051) and open (:2054-2065)
  const std::vector<aff_t>& comm_eW_s = vst.comm_per_round[inner_final + 1];
  const std::vector<aff_t>& comm_eW_c = vst.comm_per_round[inner_final + 2];
  const fe_t c_eval = tr.squeeze("c_eval");
  std::vector<aff_t> comm(rows);
  if (t_fold) ps.wk.wait(t_fold);  // the folded commitment of the step instances: the helper's second job, started behind the NIFS rounds
  lap("pcs: wait for the folded commitment");
  if (fold2.rows) {
    sp_fold2_job* j = fold2.rows;
    fold2.rows = nullptr;
    ck(sp_fold_commitments2_finish(ctx, j, u64p(&f_comm[0].x), u64p(&c_eval), u64p(&comm[0].x)), "fold_commitments");
  } else {
    ck(sp_fold_commitments2(ctx, u64p(&f_comm[0].x), u64p(&core_comm[0].x), rows, u64p(&c_eval), u64p(&comm[0].x)), "fold_commitments");
  }
  lap("pcs: fold_commitments2 (rows)");
  std::vector<fe_t> blind(rows);
  for (size_t i = 0; i < rows; ++i) {
    fe_t a;
    memcpy(&a, f_rW.data() + 4 * i, 32);
    blind[i] = fe_add<S>(a, fe_mul<S>(c_eval, core_rW[i]));
  }
  auto fold_W = [&] {  // W = folded_W + c_eval * core_W as a table: only where the opening's halves were not computed ahead
    const sp_table* two[2] = {fW, ps.core.W};
    const fe_t wts[2] = {one, c_eval};
    ck(sp_fold_tables(ctx, two, 2, u64p(wts), nv, Wf), "W = folded_W + c_eval * core_W");
  };
  aff_t comm_eval;
  if (fold2.eval) {
    sp_fold2_job* j = fold2.eval;
    fold2.eval = nullptr;
    ck(sp_fold_commitments2_finish(ctx, j, u64p(&comm_eW_s[0].x), u64p(&c_eval), u64p(&comm_eval.x)), "fold eval commitments");
  } else {
    ck(sp_fold_commitments2(ctx, u64p(&comm_eW_s[0].x), u64p(&comm_eW_c[0].x), 1, u64p(&c_eval), u64p(&comm_eval.x)), "fold eval commitments");
  }
  const fe_t blind_eval = fe_add<S>(vst.blind_per_round[inner_final + 1][0], fe_mul<S>(c_eval, vst.blind_per_round[inner_final + 2][0]));
  lap("pcs: W fold + eval commitment fold");
  // HyraxPCS::prove (hyrax_pc.rs:387-478) + InnerProductArgumentLinear::prove (ipa.rs:125-170); ck_eval = the width-32 key (ck_c = its first base, its h)
  aff_t delta, beta;
  std::vector<fe_t> z_vec(CW);
  fe_t z_delta, z_beta;
  if (reference_order) {  // PCS::prove(ck, ck_eval = the width-32 key, transcript, comm, W, blind, r_y[1..], comm_eval, blind_eval) (:2067-2080) as one call
    fold_W();
    std::vector<uint64_t> arg(16 + 4 * CW + 8);
    ck(sp_hyrax_prove(ctx, pk.ck, pk.vc_ck, tr.t, u64p(&comm[0].x), comm.size(), Wf, nv, u64p(blind.data()), u64p(r_y.data() + 1), pk.ny - 1, u64p(&comm_eval.x),
                      u64p(&blind_eval), tape.bytes + 64 * tape.pos, tape.blocks - tape.pos, arg.data()),
       "PCS::prove");
    tape.skip(CW + 2);
    memcpy(&delta, arg.data(), 64);
    memcpy(&beta, arg.data() + 8, 64);
    memcpy(z_vec.data(), arg.data() + 16, 32 * CW);
    memcpy(&z_delta, arg.data() + 16 + 4 * CW, 32);
    memcpy(&z_beta, arg.data() + 16 + 4 * CW + 4, 32);
  } else {
    const std::vector<uint8_t> b = commitment_bytes(comm.data(), comm.size());
    tr.absorb("poly_com", b.data(), b.size());
    lap("pcs: absorb poly_com");
    const fe_t* point = r_y.data() + 1;
    const size_t npoint = pk.ny - 1, nvr = log2_ceil(rows);
    const std::vector<fe_t> L = eq_evals(point, nvr);
    const size_t ncols = (size_t)1 << (npoint - nvr);
    if (t_open) ps.wk.wait(t_open);
    lap("pcs: wait for the opening's points");
    const bool open_ahead = t_open && ps.open.delta_valid && ps.open.points_valid && ps.open.tape_from == tape.pos && ncols == ps.open.dv.size();
    std::vector<fe_t> LZ(ncols), Rv;
    lz_step();
    if (lz.have == 2 && lz.f.size() == ncols && lz.c.size() == ncols) {
      const fe_t *lf = lz.f.data(), *lc = lz.c.data();
      par_for(ncols, 256, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) LZ[i] = fe_add<S>(lf[i], fe_mul<S>(c_eval, lc[i]));
      });
    } else {
      fold_W();
      ck(sp_rowmat_vec(ctx, Wf, L.size(), ncols, u64p(L.data()), u64p(LZ.data())), "bind_with_delayed");
    }
    lap("pcs: L^T W");
    fe_t r_LZ = fe_zero();
    for (size_t i = 0; i < L.size(); ++i) r_LZ = fe_add<S>(r_LZ, fe_mul<S>(L[i], blind[i]));
    std::vector<fe_t> dv;
    fe_t r_delta, r_beta;
    aff_t comm_LZ;
    if (open_ahead) {
      // the helper has delta, beta and the two halves of comm_LZ = <L, rows of (folded + c_eval core)> = P_f + c_eval P_c
      dv.swap(ps.open.dv);
      r_delta = ps.open.r_delta;
      r_beta = ps.open.r_beta;
      tape.skip(ps.open.tape_count);
      delta = ps.open.delta;
      beta = ps.open.beta;
      if (ps.open.lz_fold) {
        sp_fold2_job* j = ps.open.lz_fold;
        ps.open.lz_fold = nullptr;
        ck(sp_fold_commitments2_finish(ctx, j, u64p(&ps.open.P_f.x), u64p(&c_eval), u64p(&comm_LZ.x)), "comm_LZ = P_f + c_eval P_c");
      } else {
        ck(sp_fold_commitments2(ctx, u64p(&ps.open.P_f.x), u64p(&ps.open.P_c.x), 1, u64p(&c_eval), u64p(&comm_LZ.x)), "comm_LZ = P_f + c_eval P_c");
      }
    } else {
      if (laps && side) fprintf(stderr, "nn_prove: opening inputs computed inline (delta %d, points %d, tape %zu vs %zu)\n", (int)ps.open.delta_valid, (int)ps.open.points_valid, ps.open.tape_from, tape.pos);
      // the mask d and its blinds do not depend on the transcript (ipa.rs:139-147): delta's MSM runs on the auxiliary stream beside comm_LZ's
      Rv = eq_evals(point + nvr, npoint - nvr);
      dv.resize(Rv.size());
      for (auto& x : dv) x = tape.next();
      r_delta = tape.next();
      r_beta = tape.next();
      sp_msm_job* dj = nullptr;
      ck(sp_msm_ck_begin(ctx, pk.ck, u64p(dv.data()), dv.size(), &dj), "delta (begin)");
      int rc_lz = sp_msm_ck(ctx, pk.ck, u64p(LZ.data()), LZ.size(), u64p(&r_LZ), u64p(&comm_LZ.x));
      int rc_d = sp_msm_ck_finish(ctx, pk.ck, dj, u64p(&r_delta), u64p(&delta.x));  // always collected: the job owns device work
      ck(rc_lz, "comm_LZ");
      ck(rc_d, "delta (finish)");
      fe_t ip = fe_zero();
      for (size_t i = 0; i < Rv.size(); ++i) ip = fe_add<S>(ip, fe_mul<S>(Rv[i], dv[i]));
      ck(sp_hyrax_commit_small(ctx, pk.vc_ck, u64p(&ip), 1, u64p(&r_beta), u64p(&beta.x)), "beta");
    }
    tr.dom_sep("inner product argument (linear)");
    uint8_t pb[128];
    point_bytes(comm_LZ, pb);
    point_bytes(comm_eval, pb + 64);
    tr.absorb("U", pb, 128);
    point_bytes(delta, pb);
    tr.absorb("delta", pb, 64);
    point_bytes(beta, pb);
    tr.absorb("beta", pb, 64);
    const fe_t rr = tr.squeeze("r");
    lap("pcs: comm_LZ + transcript");
    par_for(ncols, 256, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) z_vec[i] = fe_add<S>(fe_mul<S>(rr, LZ[i]), dv[i]);
    });
    z_delta = fe_add<S>(fe_mul<S>(rr, r_LZ), r_delta);
    z_beta = fe_add<S>(fe_mul<S>(rr, blind_eval), r_beta);
    lap("pcs: z_vec");
  }
  const double t_end = now();
  // the rest of the proof in the canonical layout
  proof.pp(delta);
  proof.pp(beta);
  for (const fe_t& f : z_vec) proof.pf(f);
  proof.pf(z_delta);
  proof.pf(z_beta);
  for (const auto& c : vst.comm_per_round) proof.pc(c);
  for (const fe_t& f : vpub) proof.pf(f);
  for (const auto& cr : vst.challenges)
    for (const fe_t& f : cr) proof.pf(f);
  proof.pc(comm_T);
  proof.pc(rnd_comm_W);
  proof.pc(rnd_comm_E);
  proof.pf(rnd_u);
  for (const fe_t& f : rnd_X) proof.pf(f);
  for (const fe_t& f : v_outer) proof.pf(f);
  for (int i = 0; i < 3; ++i) proof.pf(v_claims[i]);
  for (const fe_t& f : v_inner) proof.pf(f);
  for (const fe_t& f : v_W) proof.pf(f);
  proof.pf(blind_vW);
  for (const fe_t& f : v_E) proof.pf(f);
  proof.pf(blind_vE);
  if (phase_ms) {
    phase_ms[0] = t_inst - t_start;   // rerandomize + instances
    phase_ms[1] = t_nifs - t_inst;    // NIFS (layers, rounds, folds, process_round per round)
    phase_ms[2] = t_outer - t_nifs;   // core products + batched outer sum-check
    phase_ms[3] = t_inner - t_outer;  // poly_ABC x 2 + batched inner sum-check
    phase_ms[4] = t_vc - t_inner;     // verifier-circuit instance: random instance, NovaNIFS, relaxed Spartan
    phase_ms[5] = t_end - t_vc;       // folded opening
    phase_ms[6] = t_end - t_start;
    phase_ms[7] = vst.commit_ms;  // inside the phases above: the per-round verifier-circuit commitments (process_round)
  }
  return proof;
}


// number of 64-bit words of a proof in the canonical layout (NNProof::serialize)
static size_t proof_words(const NNZkKey& pk) {
  const sp_dims& d = pk.dims;
  const size_t CW = DEFAULT_COMMITMENT_WIDTH, rows_sh = d.num_shared_unpadded ? d.num_shared / CW : 0, rows_pre = d.num_precommitted_unpadded ? d.num_precommitted / CW : 0,
               rows_rest = d.num_rest / CW;
  const vcirc::Shape& vs = pk.vc;
  const size_t vlx = log2_ceil(vs.num_cons), vly = log2_ceil(next_pow2(vs.total_vars)) + 1;
  size_t w = 8 * rows_sh + pk.num_steps * (8 * (rows_pre + rows_rest) + 4 * d.num_public) + 8 * (rows_pre + rows_rest) + 4 * pk.dims_core.num_public + 16 + 4 * CW + 8;
  w += 8 * (vs.total_vars / 32) + 4 * vs.num_public + 4 * vs.total_challenges;
  w += 8 * (vs.num_cons / 32) + 8 * (vs.total_vars / 32) + 8 * (vs.num_cons / 32) + 4 + 4 * vs.num_io();
  w += 12 * vlx + 12 + 8 * vly + 4 * 32 + 4 + 4 * 32 + 4;
  return w;
}

// ---- NeutronNovaZkSNARK on the wire (src/neutronnova_zk.rs:1373-1385; bincode framing through the library's sink / source) ------------------------
// flat layout (NNProof::serialize) <-> { comm_W_shared: Option, step_instances: Vec<SplitR1CSInstance>, core_instance, eval_arg, U_verifier:
// SplitMultiRoundR1CSInstance, nifs: NovaNIFS { comm_T }, random_U: RelaxedR1CSInstance { comm_W, comm_E, X, u }, relaxed_snark }. The instances carry
// comm_W_shared = None (:2069-2078) and no challenges.
struct WireSinkGuard {
  sp_wire* w = nullptr;
  ~WireSinkGuard() { sp_wire_free(w); }
};
static std::vector<uint8_t> nn_proof_to_bytes(const NNZkKey& pk, const uint64_t* words, size_t nwords) {
  if (nwords != proof_words(pk)) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "nn_proof_to_bytes: the word count does not match the key");
  const sp_dims& d = pk.dims;
  const vcirc::Shape& vs = pk.vc;
  const size_t CW = DEFAULT_COMMITMENT_WIDTH, rows_sh = d.num_shared_unpadded ? d.num_shared / CW : 0, rows_pre = d.num_precommitted_unpadded ? d.num_precommitted / CW : 0,
               rows_rest = d.num_rest / CW, VW = vs.width;
  WireSinkGuard g;
  ck(sp_wire_new(0, &g.w), "wire sink");
  sp_wire* w = g.w;
  const uint64_t* p = words;
  auto u64v = [&](uint64_t v) { ck(sp_wire_u64s(w, &v, 1, 0), "wire"); };
  auto commitment = [&](size_t rows) {
    ck(sp_wire_points(w, p, rows, 1), "wire");
    p += 8 * rows;
  };
  auto option_commitment = [&](size_t rows) {
    ck(sp_wire_u8(w, rows ? 1 : 0), "wire");
    if (rows) commitment(rows);
  };
  auto scalars = [&](size_t n, int with_len) {
    ck(sp_wire_scalars(w, p, n, with_len), "wire");
    p += 4 * n;
  };
  auto instance = [&](size_t npub, size_t my_pre, size_t my_rest) {
    ck(sp_wire_u8(w, 0), "wire");
    option_commitment(my_pre);
    commitment(my_rest);
    scalars(npub, 1);
    u64v(0);  // challenges: an empty Vec
  };
  // the core circuit may split the same rows into precommitted | rest differently (nn_setup)
  const sp_dims& dc = pk.dims_core;
  const size_t rows_pre_c = dc.num_precommitted_unpadded ? dc.num_precommitted / CW : 0, rows_rest_c = dc.num_rest / CW;
  auto sumcheck = [&](size_t rounds, size_t per) {
    u64v(rounds);
    for (size_t i = 0; i < rounds; ++i) scalars(per, 1);
  };
  option_commitment(rows_sh);
  u64v(pk.num_steps);
  for (size_t i = 0; i < pk.num_steps; ++i) instance(d.num_public, rows_pre, rows_rest);
  instance(pk.dims_core.num_public, rows_pre_c, rows_rest_c);
  ck(sp_wire_points(w, p, 2, 0), "wire");  // ipa.delta, ipa.beta
  p += 16;
  scalars(CW, 1);
  scalars(2, 0);
  u64v(vs.num_rounds);  // U_verifier.comm_w_per_round
  for (size_t r = 0; r < vs.num_rounds; ++r) commitment(vs.vars_padded[r] / VW);
  scalars(vs.num_public, 1);
  u64v(vs.num_rounds);  // challenges_per_round
  for (size_t r = 0; r < vs.num_rounds; ++r) scalars(vs.chals_per_round[r], 1);
  commitment(vs.num_cons / VW);    // nifs.comm_T
  commitment(vs.total_vars / VW);  // random_U.comm_W
  commitment(vs.num_cons / VW);    // random_U.comm_E
  const uint64_t* u = p;           // flat: u, then X; on the wire X, then u (src/r1cs/mod.rs:213-218)
  p += 4;
  scalars(vs.num_io(), 1);
  ck(sp_wire_scalars(w, u, 1, 0), "wire");
  sumcheck(log2_ceil(vs.num_cons), 3);
  scalars(3, 0);
  sumcheck(log2_ceil(next_pow2(vs.total_vars)) + 1, 2);
  scalars(VW, 1);
  scalars(1, 0);
  scalars(VW, 1);
  scalars(1, 0);
  if ((size_t)(p - words) != nwords) throw Error(SP_ERR_INTERNAL, "nn_proof_to_bytes: layout walk out of step");
  std::vector<uint8_t> out(sp_wire_len(w));
  ck(sp_wire_bytes(w, out.data(), out.size()), "wire bytes");
  return out;
}
// the inverse; every length prefix must be the one the key's shape dictates (the flat layout has no room for anything else)
static std::vector<uint64_t> nn_proof_from_bytes(const NNZkKey& pk, const uint8_t* bytes, size_t n) {
  const sp_dims& d = pk.dims;
  const vcirc::Shape& vs = pk.vc;
  const size_t CW = DEFAULT_COMMITMENT_WIDTH, rows_sh = d.num_shared_unpadded ? d.num_shared / CW : 0, rows_pre = d.num_precommitted_unpadded ? d.num_precommitted / CW : 0,
               rows_rest = d.num_rest / CW, VW = vs.width;
  struct Src {
    sp_unwire* r = nullptr;
    ~Src() { sp_unwire_free(r); }
  } src;
  ck(sp_unwire_new(bytes, n, &src.r), "wire source");
  sp_unwire* r = src.r;
  std::vector<uint64_t> out;
  out.reserve(proof_words(pk));
  auto expect_len = [&](size_t want, size_t elem) {
    size_t got;
    ck(sp_unwire_len(r, elem, &got), "wire length");
    if (got != want) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "wire: a length prefix does not match the key's shape");
  };
  auto points = [&](size_t cnt) {
    out.resize(out.size() + 8 * cnt);
    ck(sp_unwire_points(r, cnt, out.data() + out.size() - 8 * cnt), "wire points");
  };
  auto scalars = [&](size_t cnt) {
    out.resize(out.size() + 4 * cnt);
    ck(sp_unwire_scalars(r, cnt, out.data() + out.size() - 4 * cnt), "wire scalars");
  };
  auto tag = [&](uint8_t want) {
    uint8_t t;
    ck(sp_unwire_u8(r, &t), "wire tag");
    if (t != want) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "wire: an Option tag does not match the key's shape");
  };
  auto commitment = [&](size_t rows) {
    expect_len(rows, 96);
    points(rows);
  };
  auto option_commitment = [&](size_t rows) {
    tag(rows ? 1 : 0);
    if (rows) commitment(rows);
  };
  auto vec_scalars = [&](size_t cnt) {
    expect_len(cnt, 32);
    scalars(cnt);
  };
  auto instance = [&](size_t npub, size_t my_pre, size_t my_rest) {
    tag(0);
    option_commitment(my_pre);
    commitment(my_rest);
    vec_scalars(npub);
    expect_len(0, 32);
  };
  const sp_dims& dc = pk.dims_core;
  const size_t rows_pre_c = dc.num_precommitted_unpadded ? dc.num_precommitted / CW : 0, rows_rest_c = dc.num_rest / CW;
  auto sumcheck = [&](size_t rounds, size_t per) {
    expect_len(rounds, 8 + 32 * per);
    for (size_t i = 0; i < rounds; ++i) vec_scalars(per);
  };
  option_commitment(rows_sh);
  expect_len(pk.num_steps, 1 + 1 + 8 + 8 + 8);
  for (size_t i = 0; i < pk.num_steps; ++i) instance(d.num_public, rows_pre, rows_rest);
  instance(pk.dims_core.num_public, rows_pre_c, rows_rest_c);
  points(2);
  vec_scalars(CW);
  scalars(2);
  expect_len(vs.num_rounds, 8);
  for (size_t k = 0; k < vs.num_rounds; ++k) commitment(vs.vars_padded[k] / VW);
  vec_scalars(vs.num_public);
  expect_len(vs.num_rounds, 8);
  for (size_t k = 0; k < vs.num_rounds; ++k) vec_scalars(vs.chals_per_round[k]);
  commitment(vs.num_cons / VW);
  commitment(vs.total_vars / VW);
  commitment(vs.num_cons / VW);
  const size_t u_at = out.size();
  out.resize(u_at + 4);  // flat: u in front of X
  vec_scalars(vs.num_io());
  ck(sp_unwire_scalars(r, 1, out.data() + u_at), "wire scalars");
  sumcheck(log2_ceil(vs.num_cons), 3);
  scalars(3);
  sumcheck(log2_ceil(next_pow2(vs.total_vars)) + 1, 2);
  vec_scalars(VW);
  scalars(1);
  vec_scalars(VW);
  scalars(1);
  ck(sp_unwire_done(r), "wire end");
  if (out.size() != proof_words(pk)) throw Error(SP_ERR_INTERNAL, "nn_proof_from_bytes: layout walk out of step");
  return out;
}

// ---- NeutronNovaZkSNARK::verify (src/neutronnova_zk.rs:2096-2343) — SURVEY 8(f) rank 3 for caller #2 -----------------------------------------------
// What scales with the step circuits runs on the device through the same ABI: the fold of the step instances' commitments (one shared-weights MSM per
// Hyrax row, hyrax_pc.rs:737-793), the six matrix evaluations A, B, C (rx, ry) of the step and core shapes (ONE sp_multiply_vec against T_y = eq(r_y)
// per shape + three dot products with T_x, `evaluate_with_tables_fast` src/r1cs/mod.rs:1216-1226), comm_LZ = <L, comm rows> and the IPA's <z_vec, ck>
// (hyrax_pc.rs:480-531, ipa.rs:173-221). The verifier-circuit instance (replayed transcript of its per-round commitments, NovaNIFS::verify
// src/nifs.rs:65-77, RelaxedR1CSSpartanProof::verify src/spartan_relaxed.rs:216-316 with its two direct openings hyrax_pc.rs:654-711) is host algebra
// over a few hundred constraints plus small MSMs / two-term folds through the ABI. Returns 0 = accept, else the index of the failed check — the
// oracle's codes (oracle/neutronnova_zk.hpp nn_verify): 1 shape / encoding, 2 verifier-circuit instance does not replay, 4 relaxed Spartan proof,
// 5 public values of the verifier circuit, 6 folded opening.
static int nn_verify(const NNZkKey& pk, const uint64_t* words, size_t nwords) {
  sp_ctx* ctx = pk.ctx;
  const sp_dims& d = pk.dims;
  const vcirc::Shape& vs = pk.vc;
  const size_t CW = DEFAULT_COMMITMENT_WIDTH, n = pk.num_steps, nv = pk.num_vars, N = d.num_cons, dpub = d.num_public, cpub = pk.dims_core.num_public;
  const size_t rows_sh = d.num_shared_unpadded ? d.num_shared / CW : 0, rows_pre = d.num_precommitted_unpadded ? d.num_precommitted / CW : 0, rows_rest = d.num_rest / CW;
  const size_t rows = rows_sh + rows_pre + rows_rest;
  const size_t vnv = vs.total_vars, vcons = vs.num_cons, vio = vs.num_io(), vlx = log2_ceil(vcons), vnvp = next_pow2(vnv), vly = log2_ceil(vnvp) + 1, VW = vs.width;
  if (n == 0 || nwords != proof_words(pk)) return 1;
  ck(sp_ctx_bind_thread(ctx), "device");
  static const bool laps = [] {
    const char* e = getenv("SPARTAN_HOST_LAPS");
    return e && e[0] == '1';
  }();
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_lap = now();
  auto lap = [&](const char* what) {
    if (!laps) return;
    const double t = now();
    fprintf(stderr, "nn_verify lap %-28s %8.3f ms\n", what, t - t_lap);
    t_lap = t;
  };
  // ---- the proof in its canonical layout; every coordinate and scalar of an untrusted proof must be a canonical residue and every point on the curve
  const fe_t* w = reinterpret_cast<const fe_t*>(words);
  size_t o = 0;
  bool well_formed = true;
  auto pts = [&](size_t cnt) {
    const aff_t* p = reinterpret_cast<const aff_t*>(w + o);
    for (size_t i = 0; i < cnt; ++i)
      if (!limbs_canonical<B>(p[i].x) || !limbs_canonical<B>(p[i].y) || !aff_on_curve(p[i])) well_formed = false;
    o += 2 * cnt;
    return p;
  };
  auto fes = [&](size_t cnt) {
    const fe_t* p = w + o;
    for (size_t i = 0; i < cnt; ++i)
      if (!limbs_canonical<S>(p[i])) well_formed = false;
    o += cnt;
    return p;
  };
  struct Inst {
    const aff_t *pre, *rest;
    const fe_t* pub;
  };
  const aff_t* comm_shared = pts(rows_sh);
  std::vector<Inst> steps(n);
  for (auto& u : steps) {
    u.pre = pts(rows_pre);
    u.rest = pts(rows_rest);
    u.pub = fes(dpub);
  }
  Inst core;
  core.pre = pts(rows_pre);
  core.rest = pts(rows_rest);
  core.pub = fes(cpub);
  const aff_t delta = *pts(1), beta = *pts(1);
  const fe_t* z_vec = fes(CW);
  const fe_t z_delta = *fes(1), z_beta = *fes(1);
  std::vector<const aff_t*> vcomm(vs.num_rounds);
  for (size_t r = 0; r < vs.num_rounds; ++r) vcomm[r] = pts(vs.vars_padded[r] / VW);
  const fe_t* vpub = fes(vs.num_public);
  std::vector<const fe_t*> vchal(vs.num_rounds);
  for (size_t r = 0; r < vs.num_rounds; ++r) vchal[r] = fes(vs.chals_per_round[r]);
  const aff_t* comm_T = pts(vcons / VW);
  const aff_t* rnd_comm_W = pts(vnv / VW);
  const aff_t* rnd_comm_E = pts(vcons / VW);
  const fe_t rnd_u = *fes(1);
  const fe_t* rnd_X = fes(vio);
  const fe_t* v_outer = fes(3 * vlx);
  const fe_t* v_claims = fes(3);
  const fe_t* v_inner = fes(2 * vly);
  const fe_t* v_W = fes(VW);
  const fe_t blind_vW = *fes(1);
  const fe_t* v_E = fes(VW);
  const fe_t blind_vE = *fes(1);
  if (4 * o != nwords) throw Error(SP_ERR_INTERNAL, "nn_verify: layout / proof_words disagree");
  if (!well_formed) return 1;

  lap("parse + encoding checks");
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
  ck(sp_msm_ck_begin(ctx, pk.ck, u64p(z_vec), CW, &zjob.job), "<z, ck> (begin)");

  // regular instances: comm_W = shared | precommitted | rest rows, X = the public values (src/r1cs/mod.rs:1723-1760)
  auto regular = [&](const Inst& u) {
    std::vector<aff_t> c(rows);
    std::copy(comm_shared, comm_shared + rows_sh, c.begin());
    std::copy(u.pre, u.pre + rows_pre, c.begin() + rows_sh);
    std::copy(u.rest, u.rest + rows_rest, c.begin() + rows_sh + rows_pre);
    return c;
  };
  size_t np = 2;
  while (np < n) np <<= 1;
  const size_t nb = pk.nb, nx = pk.nx, ny = pk.ny;
  if (((size_t)1 << nb) != np) throw Error(SP_ERR_INTERNAL, "nn_verify: a single step instance is not driven by this layer");
  auto inst = [&](size_t i) { return i < n ? i : 0; };  // padding clones instance 0 (:549-552)
  std::vector<std::vector<aff_t>> Ucomm(n);
  for (size_t i = 0; i < n; ++i) Ucomm[i] = regular(steps[i]);
  const std::vector<aff_t> core_comm = regular(core);
  const std::vector<fe_t> core_X(core.pub, core.pub + cpub);
  // the verifier-circuit instance in its regular form: comm_W = the rounds' rows, X = challenges | public values; the challenges are checked against the
  // transcript below, what is computed from them before that is only used after the check
  const std::vector<aff_t> Uv_comm(vcomm[0], vcomm[0] + vnv / VW);  // the rounds are consecutive in the layout
  std::vector<fe_t> Uv_X;
  for (size_t r = 0; r < vs.num_rounds; ++r) Uv_X.insert(Uv_X.end(), vchal[r], vchal[r] + vs.chals_per_round[r]);
  Uv_X.insert(Uv_X.end(), vpub, vpub + vs.num_public);
  const size_t num_chal = nb + nx + 1 + ny;
  if (vs.total_challenges != num_chal || vs.num_public != 6) return 2;
  const fe_t *r_b = Uv_X.data(), *r_x = r_b + nb, *r_y = r_x + nx + 1, *pub = Uv_X.data() + num_chal;
  const fe_t r = Uv_X[nb + nx], r2 = fe_mul<S>(r, r), one = fe_one<S>();
  // fold_multiple of the step instances (src/r1cs/mod.rs:695-722): X on the host, the commitment rows as shared-weights MSMs
  std::vector<fe_t> wts(np);
  ck(sp_weights_from_r(u64p(r_b), nb, np, u64p(wts.data())), "weights_from_r");
  std::vector<fe_t> Xf(dpub, fe_zero());
  for (size_t i = 0; i < np; ++i)
    for (size_t j = 0; j < dpub; ++j) Xf[j] = fe_add<S>(Xf[j], fe_mul<S>(wts[i], steps[inst(i)].pub[j]));
  std::vector<aff_t> folded_comm(rows), fold_bases(rows * np);
  for (size_t rr = 0; rr < rows; ++rr)
    for (size_t i = 0; i < np; ++i) fold_bases[rr * np + i] = Ucomm[inst(i)][rr];
  fe_t eabc[2][3];
  const bool side = true;  // (false = everything inline on the caller's context: the order the phase comments describe)
  auto fold_commitments = [&](sp_ctx* on) {
    ck(sp_msm_shared_weights(on, u64p(wts.data()), np, reinterpret_cast<const uint64_t*>(fold_bases.data()), rows, reinterpret_cast<uint64_t*>(folded_comm.data())), "fold_commitments");
  };
  // the matrix evaluations A, B, C (r_x, r_y) of the step and core shapes: T_x^T (M T_y)
  auto matrix_evaluations = [&](sp_ctx* on) {
    if (!pk.v_Tx) {
      ck(sp_table_zeros(on, (size_t)1 << nx, (size_t)-1, (size_t)-1, &pk.v_Tx), "T_x alloc");
      ck(sp_table_zeros(on, (size_t)1 << ny, (size_t)-1, (size_t)-1, &pk.v_Ty), "T_y alloc");
      for (int i = 0; i < 3; ++i) ck(sp_table_zeros(on, N, (size_t)-1, (size_t)-1, &pk.v_mv[i]), "M T_y alloc");
    }
    sp_table *Tx = pk.v_Tx, *Ty = pk.v_Ty, **mv = pk.v_mv;
    ck(sp_table_set_len(Tx, (size_t)1 << nx, (size_t)-1, (size_t)-1), "T_x len");
    ck(sp_table_set_len(Ty, (size_t)1 << ny, (size_t)-1, (size_t)-1), "T_y len");
    ck(sp_eq_table_into(on, u64p(r_x), nx, Tx), "T_x");
    ck(sp_eq_table_into(on, u64p(r_y), ny, Ty), "T_y");
    const sp_shape* shapes[2] = {pk.S_step, pk.S_core};
    const size_t cols[2] = {nv + 1 + dpub, nv + 1 + cpub};
    for (int s2 = 0; s2 < 2; ++s2) {
      ck(sp_table_set_len(Ty, cols[s2], (size_t)-1, (size_t)-1), "T_y as z");
      ck(sp_multiply_vec(on, shapes[s2], Ty, mv[0], mv[1], mv[2]), "M T_y");
      for (int i = 0; i < 3; ++i) ck(sp_table_dot(on, Tx, mv[i], N, u64p(&eabc[s2][i])), "T_x . (M T_y)");
    }
  };
  struct Drain {  // the jobs read and write locals of this call
    Worker& w;
    ~Drain() { w.drain(); }
  } drain{pk.v_wk};
  size_t t_fold = 0, t_mat = 0;
  if (side) {
    if (!pk.v_ctx2) ck(sp_ctx_create(sp_ctx_device(ctx), &pk.v_ctx2), "second context");
    t_fold = pk.v_wk.submit([&] {
      ck(sp_ctx_bind_thread(pk.v_ctx2), "device");
      fold_commitments(pk.v_ctx2);
    });
    t_mat = pk.v_wk.submit([&] { matrix_evaluations(pk.v_ctx2); });
  }
  Tr tr(ctx, "neutronnova_prove");
  tr.absorb("vk", pk.vk_digest, 32);
  absorb_instance(tr, "core_instance", core_comm, core_X);
  for (size_t i = 0; i < np; ++i) absorb_instance(tr, "U", Ucomm[inst(i)], std::vector<fe_t>(steps[inst(i)].pub, steps[inst(i)].pub + dpub));
  {
    uint8_t zero_be[32] = {0};  // T = 0 (:556-557)
    tr.absorb("T", zero_be, 32);
  }
  const fe_t tau = tr.squeeze("tau");
  std::vector<fe_t> rhos(nb);
  for (auto& x : rhos) x = tr.squeeze("rho");
  lap("instances absorbed");
  // U_verifier.validate (src/r1cs/mod.rs:1808-1834): the per-round commitments reproduce the challenges the instance carries
  for (size_t round = 0; round < vs.num_rounds; ++round) {
    const std::vector<uint8_t> b = commitment_bytes(vcomm[round], vs.vars_padded[round] / VW);
    tr.absorb("comm_w_round", b.data(), b.size());
    for (size_t i = 0; i < vs.chals_per_round[round]; ++i)
      if (!fe_eq(tr.squeeze("challenge"), vchal[round][i])) return 2;
  }
  lap("vc instance replayed");
  // NovaNIFS::verify (src/nifs.rs:65-77): the random relaxed instance folded with the verifier-circuit instance
  {
    std::vector<uint8_t> b = commitment_bytes(rnd_comm_W, vnv / VW), e = commitment_bytes(rnd_comm_E, vcons / VW);
    b.insert(b.end(), e.begin(), e.end());
    const size_t off = b.size();
    b.resize(off + 32 * (1 + vio));
    sp::fe_to_be_bytes<S>(rnd_u, b.data() + off);
    for (size_t j = 0; j < vio; ++j) sp::fe_to_be_bytes<S>(rnd_X[j], b.data() + off + 32 * (1 + j));
    tr.absorb("U1", b.data(), b.size());
  }
  absorb_instance(tr, "U2", Uv_comm, Uv_X);
  {
    const std::vector<uint8_t> b = commitment_bytes(comm_T, vcons / VW);
    tr.absorb("comm_T", b.data(), b.size());
  }
  const fe_t rf = tr.squeeze("r");
  std::vector<aff_t> fU_W(vnv / VW), fU_E(vcons / VW);
  ck(sp_fold_commitments2(ctx, u64p(&rnd_comm_W[0].x), u64p(&Uv_comm[0].x), fU_W.size(), u64p(&rf), u64p(&fU_W[0].x)), "fold comm_W");
  ck(sp_fold_commitments2(ctx, u64p(&rnd_comm_E[0].x), u64p(&comm_T[0].x), fU_E.size(), u64p(&rf), u64p(&fU_E[0].x)), "fold comm_E");
  const fe_t fU_u = fe_add<S>(rnd_u, rf);
  std::vector<fe_t> fU_X(vio);
  for (size_t i = 0; i < vio; ++i) fU_X[i] = fe_add<S>(rnd_X[i], fe_mul<S>(rf, Uv_X[i]));
  lap("NovaNIFS::verify");
  // RelaxedR1CSSpartanProof::verify (src/spartan_relaxed.rs:216-316)
  {
    // verify_direct (hyrax_pc.rs:654-711) under the width-32 key: <L, comm rows> must equal <v, ck> + cb * h; returns the evaluation <v, R>
    auto verify_direct = [&](const std::vector<aff_t>& comm, const fe_t* v, const fe_t& cb, const fe_t* point, size_t npoint, fe_t* eval) {
      const size_t nn = (size_t)1 << npoint, drows = (nn + VW - 1) / VW, nvr = log2_ceil(drows);
      aff_t comm_LZ;
      if (nvr == 0) {
        comm_LZ = comm[0];
      } else {
        const std::vector<fe_t> L = eq_evals(point, nvr);
        if (comm.size() > L.size()) return false;
        ck(sp_msm(ctx, u64p(L.data()), u64p(&comm[0].x), comm.size(), u64p(&comm_LZ.x)), "direct opening: <L, rows>");
      }
      aff_t expected;
      ck(sp_hyrax_commit_small(ctx, pk.vc_ck, u64p(v), VW, u64p(&cb), u64p(&expected.x)), "direct opening: <v, ck> + cb h");
      if (!fe_eq(comm_LZ.x, expected.x) || !fe_eq(comm_LZ.y, expected.y)) return false;
      const std::vector<fe_t> R = eq_evals(point + nvr, npoint - nvr);
      fe_t e = fe_zero();
      for (size_t i = 0; i < VW; ++i) e = fe_add<S>(e, fe_mul<S>(v[i], R[i]));
      *eval = e;
      return true;
    };
    tr.absorb_scalars("u_relaxed", &fU_u, 1);
    tr.absorb_scalars("X_relaxed", fU_X.data(), fU_X.size());
    std::vector<fe_t> vtau(vlx);
    for (auto& t : vtau) t = tr.squeeze("t");
    fe_t claim_outer_final, claim_inner_final;
    std::vector<fe_t> vrx, vry;
    if (!sumcheck_verify(tr, fe_zero(), vlx, 3, v_outer, &claim_outer_final, &vrx)) return 4;
    fe_t eq_tau = one;  // EqPolynomial::evaluate (src/polys/eq.rs:45-57)
    for (size_t i = 0; i < vlx; ++i) eq_tau = fe_mul<S>(eq_tau, fe_add<S>(fe_mul<S>(vtau[i], vrx[i]), fe_mul<S>(fe_sub<S>(one, vtau[i]), fe_sub<S>(one, vrx[i]))));
    if (!fe_eq(claim_outer_final, fe_mul<S>(eq_tau, fe_sub<S>(fe_mul<S>(v_claims[0], v_claims[1]), v_claims[2])))) return 4;
    tr.absorb_scalars("claims_outer", v_claims, 3);
    const fe_t vr = tr.squeeze("r"), vr2 = fe_mul<S>(vr, vr);
    fe_t eval_E, eval_W;
    if (!verify_direct(fU_E, v_E, blind_vE, vrx.data(), vlx, &eval_E)) return 4;
    const fe_t claim_inner = fe_add<S>(fe_add<S>(v_claims[0], fe_mul<S>(vr, v_claims[1])), fe_mul<S>(vr2, fe_sub<S>(v_claims[2], eval_E)));
    if (!sumcheck_verify(tr, claim_inner, vly, 2, v_inner, &claim_inner_final, &vry)) return 4;
    if (!verify_direct(fU_W, v_W, blind_vW, vry.data() + 1, vly - 1, &eval_W)) return 4;
    const std::vector<fe_t> Tx = eq_evals(vrx.data(), vlx), Ty = eq_evals(vry.data(), vly);
    fe_t eval_Z = fe_add<S>(fe_mul<S>(fe_sub<S>(one, vry[0]), eval_W), fe_mul<S>(fU_u, Ty[vnv]));
    for (size_t j = 0; j < vio; ++j) eval_Z = fe_add<S>(eval_Z, fe_mul<S>(fU_X[j], Ty[vnv + 1 + j]));
    fe_t em[3];  // evaluate_with_tables on the verifier-circuit matrices (spartan_relaxed.rs:44-67)
    for (int m = 0; m < 3; ++m) {
      fe_t acc = fe_zero();
      for (size_t row = 0; row < vcons; ++row) {
        if (fe_is_zero(Tx[row])) continue;
        fe_t rs = fe_zero();
        for (uint64_t k = vs.M[m].ptr[row]; k < vs.M[m].ptr[row + 1]; ++k) rs = fe_add<S>(rs, fe_mul<S>(Ty[vs.M[m].idx[k]], vs.M[m].data[k]));
        acc = fe_add<S>(acc, fe_mul<S>(Tx[row], rs));
      }
      em[m] = acc;
    }
    const fe_t eval_ABC = fe_add<S>(fe_add<S>(em[0], fe_mul<S>(vr, em[1])), fe_mul<S>(fe_mul<S>(vr2, fU_u), em[2]));
    if (!fe_eq(claim_inner_final, fe_mul<S>(eval_ABC, eval_Z))) return 4;
    tr.absorb_scalars("v_W", v_W, VW);
    tr.absorb_scalars("v_E", v_E, VW);
  }
  lap("relaxed Spartan verify");
  // the six public values of the verifier circuit (:2280-2330): tau(r_x), the X evaluations, eq(r_b, rho) and the matrix evaluations at (r_x, r_y)
  if (t_mat)
    pk.v_wk.wait(t_mat);
  else
    matrix_evaluations(ctx);
  {
    auto eval_X = [&](const fe_t* Xv, size_t cnt) {
      std::vector<fe_t> v{one};
      v.insert(v.end(), Xv, Xv + cnt);
      return sparse_poly_evaluate(ny - 1, v, r_y + 1);
    };
    const fe_t q_step = fe_add<S>(fe_add<S>(eabc[0][0], fe_mul<S>(r, eabc[0][1])), fe_mul<S>(r2, eabc[0][2]));
    const fe_t q_core = fe_add<S>(fe_add<S>(eabc[1][0], fe_mul<S>(r, eabc[1][1])), fe_mul<S>(r2, eabc[1][2]));
    fe_t tau_at_rx = one, p = tau;  // PowPolynomial::evaluate (src/polys/power.rs:33-50): tau^(2^i) pairs with the i-th variable from the end
    for (size_t i = 0; i < nx; ++i) {
      tau_at_rx = fe_mul<S>(tau_at_rx, fe_add<S>(one, fe_mul<S>(fe_sub<S>(p, one), r_x[nx - 1 - i])));
      p = fe_mul<S>(p, p);
    }
    fe_t eq_rho = one;
    for (size_t i = 0; i < nb; ++i) eq_rho = fe_mul<S>(eq_rho, fe_add<S>(fe_mul<S>(rhos[i], r_b[i]), fe_mul<S>(fe_sub<S>(one, rhos[i]), fe_sub<S>(one, r_b[i]))));
    if (!fe_eq(pub[0], tau_at_rx) || !fe_eq(pub[1], eval_X(Xf.data(), dpub)) || !fe_eq(pub[2], eval_X(core.pub, cpub)) || !fe_eq(pub[3], eq_rho) || !fe_eq(pub[4], q_step) ||
        !fe_eq(pub[5], q_core))
      return 5;
  }
  lap("matrix evaluations + publics");
  // the folded opening (:2332-2343): HyraxPCS::verify (hyrax_pc.rs:480-531) + InnerProductArgumentLinear::verify (ipa.rs:173-221)
  const fe_t c_eval = tr.squeeze("c_eval");
  const size_t commit_round = nb + 1 + nx + 1 + ny + 1;
  if (commit_round + 1 >= vs.num_rounds) throw Error(SP_ERR_INTERNAL, "nn_verify: verifier-circuit round layout");
  std::vector<aff_t> comm(rows);
  if (t_fold)
    pk.v_wk.wait(t_fold);
  else
    fold_commitments(ctx);
  ck(sp_fold_commitments2(ctx, u64p(&folded_comm[0].x), u64p(&core_comm[0].x), rows, u64p(&c_eval), u64p(&comm[0].x)), "fold_commitments");
  aff_t comm_eval;
  ck(sp_fold_commitments2(ctx, u64p(&vcomm[commit_round][0].x), u64p(&vcomm[commit_round + 1][0].x), 1, u64p(&c_eval), u64p(&comm_eval.x)), "fold eval commitments");
  {
    const std::vector<uint8_t> b = commitment_bytes(comm.data(), rows);
    tr.absorb("poly_com", b.data(), b.size());
  }
  const fe_t* point = r_y + 1;
  const size_t npoint = ny - 1, num_rows = (((size_t)1 << npoint) + CW - 1) / CW, nvr = log2_ceil(num_rows);
  aff_t comm_LZ;
  std::vector<fe_t> R;
  if (nvr == 0) {
    comm_LZ = comm[0];
    R = eq_evals(point, npoint);
  } else {
    const std::vector<fe_t> L = eq_evals(point, nvr);
    R = eq_evals(point + nvr, npoint - nvr);
    if (rows < L.size()) return 6;
    ck(sp_msm(ctx, u64p(L.data()), u64p(&comm[0].x), L.size(), u64p(&comm_LZ.x)), "comm_LZ");
  }
  if (R.size() != CW) return 6;
  tr.dom_sep("inner product argument (linear)");
  {
    uint8_t b[128];
    point_bytes(comm_LZ, b);
    point_bytes(comm_eval, b + 64);
    tr.absorb("U", b, 128);
    point_bytes(delta, b);
    tr.absorb("delta", b, 64);
    point_bytes(beta, b);
    tr.absorb("beta", b, 64);
  }
  const fe_t rr = tr.squeeze("r");
  aff_t pr2[2] = {comm_LZ, comm_eval}, rp[2], hzd, rhs2;
  ck(sp_vartime_scalar_mul(ctx, u64p(&pr2[0].x), 2, u64p(&rr), u64p(&rp[0].x)), "r * (comm_LZ, comm_eval)");
  ck(sp_fixed_base_mul_h(ctx, pk.ck, u64p(&z_delta), 1, u64p(&hzd.x)), "h * z_delta");
  fe_t ip = fe_zero();
  for (size_t i = 0; i < CW; ++i) ip = fe_add<S>(ip, fe_mul<S>(z_vec[i], R[i]));
  ck(sp_hyrax_commit_small(ctx, pk.vc_ck, u64p(&ip), 1, u64p(&z_beta), u64p(&rhs2.x)), "<z, R> * ck_c + z_beta * h_c");
  aff_t zc;
  {
    sp_msm_job* j = zjob.job;
    zjob.job = nullptr;
    ck(sp_msm_ck_finish(ctx, pk.ck, j, nullptr, u64p(&zc.x)), "<z, ck> (finish)");
  }
  if (!same_point(jac_add_mixed(jac_from_affine(rp[0]), delta), jac_add_mixed(jac_from_affine(zc), hzd))) return 6;
  if (!same_point(jac_add_mixed(jac_from_affine(rp[1]), beta), jac_from_affine(rhs2))) return 6;
  lap("folded opening");
  return 0;
}

}  // namespace spartan2

using namespace spartan2;
extern "C" void ss_set_error(const char* msg);
static int catch_all_nn() {
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
// args: the integer R1CS of the step circuit and of the core circuit (same padded dimensions), as for ss_setup
int nnz_setup(sp_ctx* ctx, size_t num_steps, size_t num_cons, size_t num_shared, size_t num_precommitted, size_t num_rest, size_t num_public, size_t num_challenges,
              const int64_t* Ad, const uint32_t* Ai, const uint64_t* Ap, const int64_t* Bd, const uint32_t* Bi, const uint64_t* Bp, const int64_t* Cd, const uint32_t* Ci,
              const uint64_t* Cp, size_t c_num_cons, size_t c_num_shared, size_t c_num_precommitted, size_t c_num_rest, size_t c_num_public, size_t c_num_challenges,
              const int64_t* cAd, const uint32_t* cAi, const uint64_t* cAp, const int64_t* cBd, const uint32_t* cBi, const uint64_t* cBp, const int64_t* cCd,
              const uint32_t* cCi, const uint64_t* cCp, void** out_pk) {
  try {
    *out_pk = nn_setup(ctx, make_view(num_cons, num_shared, num_precommitted, num_rest, num_public, num_challenges, Ad, Ai, Ap, Bd, Bi, Bp, Cd, Ci, Cp),
                       make_view(c_num_cons, c_num_shared, c_num_precommitted, c_num_rest, c_num_public, c_num_challenges, cAd, cAi, cAp, cBd, cBi, cBp, cCd, cCi, cCp),
                       num_steps);
    return 0;
  } catch (...) {
    return catch_all_nn();
  }
}
void nnz_pk_free(void* pk) { delete (NNZkKey*)pk; }
// out: nb, nx, ny, vc rounds, vc total vars, vc num_cons, vc num_cons_unpadded, vc num_public; digest = the vk digest (SHA-256 over NeutronNovaVerifierKey::write_bytes)
void nnz_pk_info(void* pk_, uint64_t out[8], uint8_t digest[32]) {
  auto* pk = (NNZkKey*)pk_;
  uint64_t v[8] = {pk->nb, pk->nx, pk->ny, pk->vc.num_rounds, pk->vc.total_vars, pk->vc.num_cons, pk->vc.num_cons_unpadded, pk->vc.num_public};
  memcpy(out, v, sizeof v);
  memcpy(digest, pk->vk_digest, 32);
}
size_t nnz_proof_words(void* pk_) { return proof_words(*(NNZkKey*)pk_); }
int nnz_prep_prove(void* pk, size_t n, const uint64_t* step_wit, size_t wit_len, const uint64_t* step_pub, size_t npub, const uint64_t* core_wit, const uint64_t* core_pub,
                   int is_small, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, void** out_ps) {
  try {
    Tape t{tape, tape_blocks};
    *out_ps = nn_prep_prove(*(NNZkKey*)pk, n, step_wit, wit_len, step_pub, npub, core_wit, core_pub, is_small != 0, t);
    if (tape_used) *tape_used = t.pos;
    return 0;
  } catch (...) {
    return catch_all_nn();
  }
}
void nnz_prep_free(void* ps) { delete (NNZkPrep*)ps; }
// 0 = accept, 1..6 = the failed check (nn_verify above), < 0 = error
int nnz_verify(void* pk, const uint64_t* words, size_t nwords) {
  try {
    return nn_verify(*(NNZkKey*)pk, words, nwords);
  } catch (...) {
    return catch_all_nn();
  }
}
// phase_ms[8]: instances, nifs, outer, inner, verifier-circuit instance, opening, total, (of which) per-round vc commitments
static int nnz_prove_impl(void* pk, void* ps, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, uint64_t* out_words, size_t out_cap, double* phase_ms,
                          bool reference_order) {
  try {
    Tape t{tape, tape_blocks};
    ProofBuf pf = nn_prove(*(NNZkKey*)pk, *(NNZkPrep*)ps, t, phase_ms, reference_order);
    if (pf.words.size() > out_cap) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "proof buffer too small");
    memcpy(out_words, pf.words.data(), pf.words.size() * 8);
    if (tape_used) *tape_used = t.pos;
    return 0;
  } catch (...) {
    return catch_all_nn();
  }
}
int nnz_prove(void* pk, void* ps, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, uint64_t* out_words, size_t out_cap, double* phase_ms) {
  return nnz_prove_impl(pk, ps, tape, tape_blocks, tape_used, out_words, out_cap, phase_ms, false);
}
// the reference-order driver (see nn_prove): the time of an unchanged neutronnova_zk.rs over the ABI; the proof is the same
int nnz_prove_reference_order(void* pk, void* ps, const uint8_t* tape, size_t tape_blocks, size_t* tape_used, uint64_t* out_words, size_t out_cap, double* phase_ms) {
  return nnz_prove_impl(pk, ps, tape, tape_blocks, tape_used, out_words, out_cap, phase_ms, true);
}
// NeutronNovaZkSNARK as bincode bytes (the reference's serde framing; include/spartan_hip.h "wire formats"). `_to_bytes`: out may be NULL to learn *len.
int nnz_proof_to_bytes(void* pk, const uint64_t* words, size_t nwords, uint8_t* out, size_t cap, size_t* len) {
  try {
    std::vector<uint8_t> b = nn_proof_to_bytes(*(NNZkKey*)pk, words, nwords);
    *len = b.size();
    if (out) {
      if (cap < b.size()) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "byte buffer too small");
      memcpy(out, b.data(), b.size());
    }
    return 0;
  } catch (...) {
    return catch_all_nn();
  }
}
int nnz_proof_from_bytes(void* pk, const uint8_t* bytes, size_t n, uint64_t* out_words, size_t cap_words) {
  try {
    std::vector<uint64_t> w = nn_proof_from_bytes(*(NNZkKey*)pk, bytes, n);
    if (cap_words < w.size()) throw Error(SP_ERR_INVALID_INPUT_LENGTH, "word buffer too small");
    memcpy(out_words, w.data(), 8 * w.size());
    return 0;
  } catch (...) {
    return catch_all_nn();
  }
}
// verify on the serialised proof: 0 = accept, 1..6 = the failed check (a proof that does not decode fails check 1), < 0 = error
int nnz_verify_bytes(void* pk, const uint8_t* bytes, size_t n) {
  try {
    std::vector<uint64_t> w;
    try {
      w = nn_proof_from_bytes(*(NNZkKey*)pk, bytes, n);
    } catch (const Error& e) {
      if (e.code != SP_ERR_INVALID_INPUT_LENGTH) throw;
      ss_set_error(e.what());
      return 1;
    }
    return nn_verify(*(NNZkKey*)pk, w.data(), w.size());
  } catch (...) {
    return catch_all_nn();
  }
}
}
