// NeutronNovaVerifierCircuit (src/zk.rs:473-943) with its gadgets (:18-236), the multi-round R1CS shape built from it
// (multiround_r1cs_shape, src/bellpepper/r1cs.rs:603-693; SplitMultiRoundR1CSShape::new, src/r1cs/mod.rs:1556-1661) and the per-round witness state
// (initialize / process_round / finalize, bellpepper/r1cs.rs:695-848) — host side ABOVE the C ABI: the instance has a few hundred constraints, its
// only heavy step is the commitment of each round's 32 variables, which goes through sp_hyrax_commit_small (the width-32 fixed-base path,
// hyrax_pc.rs:81-96,221-260).
// Synthesis ORDER is the reference's, statement by statement: it fixes the matrices and the layout of the round witnesses.
#pragma once
#include <algorithm>
#include <array>
#include <chrono>

#include "host_common.hpp"

namespace spartan2 {
namespace vcirc {

struct Var {  // aux j -> j; input i -> INPUT | i (input 0 is the constant ONE)
  static constexpr uint32_t INPUT = 0x80000000u;
  uint32_t id;
};
typedef std::vector<std::pair<Var, fe_t>> LC;
struct Num {
  Var var;
  fe_t val;
};
struct CS {
  std::vector<fe_t> aux, inputs;
  std::vector<std::array<LC, 3>> cons;
  CS() { inputs.push_back(fe_one<S>()); }
  static Var one() { return Var{Var::INPUT}; }
  Num alloc(const fe_t& v) {
    aux.push_back(v);
    return Num{Var{(uint32_t)(aux.size() - 1)}, v};
  }
  Num alloc_input(const fe_t& v) {
    inputs.push_back(v);
    return Num{Var{Var::INPUT | (uint32_t)(inputs.size() - 1)}, v};
  }
  void enforce(LC a, LC b, LC c) { cons.push_back({std::move(a), std::move(b), std::move(c)}); }
};
static inline fe_t P1() { return fe_one<S>(); }
static inline fe_t M1() { return fe_neg<S>(fe_one<S>()); }
static inline LC lc1(const Var& v) { return LC{{v, P1()}}; }

static inline Num horner(CS& cs, const Num* co, size_t n, const Num& x) {  // eval_poly_horner (:18-45)
  Num acc = co[n - 1];
  for (size_t k = n - 1; k-- > 0;) {
    Num na = cs.alloc(fe_add<S>(fe_mul<S>(acc.val, x.val), co[k].val));
    cs.enforce(lc1(acc.var), lc1(x.var), LC{{na.var, P1()}, {co[k].var, M1()}});
    acc = na;
  }
  return acc;
}
static inline Num alloc_zero(CS& cs) {  // :48-61
  Num z = cs.alloc(fe_zero());
  cs.enforce(lc1(z.var), lc1(CS::one()), LC());
  return z;
}
static inline std::vector<Num> alloc_coeffs(CS& cs, const fe_t* c, size_t n) {  // :64-73
  std::vector<Num> v;
  for (size_t i = 0; i < n; ++i) v.push_back(cs.alloc(c[i]));
  return v;
}
static inline void enforce_sc_claim(CS& cs, const std::vector<Num>& poly, const Num& claim) {  // :85-105
  LC a;
  for (const Num& p : poly) a.push_back({p.var, P1()});
  a.push_back({poly[0].var, P1()});
  cs.enforce(a, lc1(CS::one()), lc1(claim.var));
}
static inline Num mul(CS& cs, const Num& a, const Num& b) {
  Num p = cs.alloc(fe_mul<S>(a.val, b.val));
  cs.enforce(lc1(a.var), lc1(b.var), lc1(p.var));
  return p;
}
static inline void inputize(CS& cs, const Num& a) {
  Num in = cs.alloc_input(a.val);
  cs.enforce(lc1(in.var), lc1(CS::one()), lc1(a.var));
}
static inline void enforce_outer_final(CS& cs, const Num& Az, const Num& Bz, const Num& Cz, const Num& tau, const Num& prev) {  // :109-131
  Num prod = mul(cs, Az, Bz);
  cs.enforce(lc1(tau.var), LC{{prod.var, P1()}, {Cz.var, M1()}}, lc1(prev.var));
}
static inline Num joint_claim(CS& cs, const Num& Az, const Num& Bz, const Num& Cz, const Num& r, const Num& r_sq) {  // :134-165
  Num rB = mul(cs, r, Bz);
  Num joint = cs.alloc(fe_add<S>(fe_add<S>(Az.val, rB.val), fe_mul<S>(r_sq.val, Cz.val)));
  cs.enforce(lc1(Cz.var), lc1(r_sq.var), LC{{joint.var, P1()}, {Az.var, M1()}, {rB.var, M1()}});
  return joint;
}
static inline void enforce_inner_final(CS& cs, const Num& r_y0, const Num& eW, const Num& eX, const Num& prev) {  // :171-236
  Num tmp_w = cs.alloc(fe_mul<S>(eW.val, fe_sub<S>(fe_one<S>(), r_y0.val)));
  cs.enforce(lc1(eW.var), LC{{CS::one(), P1()}, {r_y0.var, M1()}}, lc1(tmp_w.var));
  Num sum_z = cs.alloc(fe_add<S>(tmp_w.val, fe_mul<S>(eX.val, r_y0.val)));
  cs.enforce(lc1(eX.var), lc1(r_y0.var), LC{{sum_z.var, P1()}, {tmp_w.var, M1()}});
  Num q = cs.alloc_input(fe_is_zero(sum_z.val) ? fe_zero() : fe_mul<S>(prev.val, fe_inv<S>(sum_z.val)));
  cs.enforce(lc1(q.var), lc1(sum_z.var), lc1(prev.var));
}

struct Circuit {
  std::vector<std::array<fe_t, 4>> nifs_polys, outer_step, outer_core;
  std::vector<std::array<fe_t, 3>> inner_step, inner_core;
  fe_t eq_rho_at_rb, t_out_step, claim_step[3], claim_core[3], tau_at_rx, eval_W_step, eval_W_core, eval_X_step, eval_X_core;
  size_t width = 32;
  Circuit(size_t nb, size_t nx, size_t ny, size_t w) : width(w) {
    const fe_t z = fe_zero();
    nifs_polys.assign(nb, {z, z, z, z});
    outer_step.assign(nx, {z, z, z, z});
    outer_core.assign(nx, {z, z, z, z});
    inner_step.assign(ny, {z, z, z});
    inner_core.assign(ny, {z, z, z});
    eq_rho_at_rb = t_out_step = tau_at_rx = eval_W_step = eval_W_core = eval_X_step = eval_X_core = z;
    for (int i = 0; i < 3; ++i) claim_step[i] = claim_core[i] = z;
  }
  size_t nb() const { return nifs_polys.size(); }
  size_t idx_outer_start() const { return nb() + 1; }
  size_t idx_outer_final() const { return idx_outer_start() + outer_step.size(); }
  size_t idx_inner_start() const { return idx_outer_final() + 1; }
  size_t idx_inner_final() const { return idx_inner_start() + inner_step.size(); }
  size_t num_rounds() const { return idx_inner_final() + 3; }
  size_t num_challenges(size_t round) const {  // :631-649
    if (round < nb()) return 1;
    if (round == nb()) return 0;
    return round < idx_inner_final() ? 1 : 0;
  }
  // The variables of a sum-check round in allocation order: first the round's own prover message (the coefficients of its polynomials: `fresh_vars`),
  // then values that are functions of EARLIER rounds' messages and of the previous challenge alone (`early_values`: the Horner steps of the previous
  // polynomials at that challenge, the joint claims in front of the inner sum-check). The second group's part of the round commitment can be walked while
  // the device still computes the first (process_round below). 0 = not a sum-check round (everything counts as fresh).
  size_t fresh_vars(size_t round) const {
    if (round < nb()) return 4;
    if (round > nb() && round < idx_outer_final()) return 8;
    if (round >= idx_inner_start() && round < idx_inner_final()) return 6;
    return 0;
  }
  static void horner_values(const fe_t* co, size_t n, const fe_t& x, std::vector<fe_t>* out) {  // the allocations of horner() above
    fe_t acc = co[n - 1];
    for (size_t k = n - 1; k-- > 0;) {
      acc = fe_add<S>(fe_mul<S>(acc, x), co[k]);
      out->push_back(acc);
    }
  }
  // mirrors rounds() for the rounds with fresh_vars != 0; process_round compares the result with what rounds() allocated before it relies on it
  void early_values(size_t round, const fe_t& c0, std::vector<fe_t>* out) const {
    out->clear();
    if (round < nb()) {
      if (round == 0) out->push_back(fe_zero());
      else horner_values(nifs_polys[round - 1].data(), 4, c0, out);
    } else if (round > nb() && round < idx_outer_final()) {
      const size_t i = round - idx_outer_start();
      if (i == 0) out->push_back(fe_zero());
      else {
        horner_values(outer_step[i - 1].data(), 4, c0, out);
        horner_values(outer_core[i - 1].data(), 4, c0, out);
      }
    } else if (round >= idx_inner_start() && round < idx_inner_final()) {
      const size_t idx = round - idx_inner_start();
      if (idx == 0) {
        const fe_t r_sq = fe_mul<S>(c0, c0);
        out->push_back(r_sq);
        for (const fe_t* cl : {claim_step, claim_core}) {
          const fe_t rB = fe_mul<S>(c0, cl[1]);
          out->push_back(rB);
          out->push_back(fe_add<S>(fe_add<S>(cl[0], rB), fe_mul<S>(r_sq, cl[2])));
        }
      } else {
        horner_values(inner_step[idx - 1].data(), 3, c0, out);
        horner_values(inner_core[idx - 1].data(), 3, c0, out);
      }
    }
  }
  void rounds(CS& cs, size_t round, const std::vector<std::vector<Num>>& prior, const std::vector<std::vector<Num>>& prev_chals, const fe_t* chal, std::vector<Num>* vars,
              std::vector<Num>* chals) const {
    const fe_t c0 = chal ? *chal : fe_zero();
    vars->clear();
    chals->clear();
    if (round < nb()) {
      std::vector<Num> poly = alloc_coeffs(cs, nifs_polys[round].data(), 4);
      Num claim;
      if (round == 0) claim = alloc_zero(cs);
      else {
        Num r = cs.alloc_input(c0);
        claim = horner(cs, prior[round - 1].data(), prior[round - 1].size(), r);
      }
      enforce_sc_claim(cs, poly, claim);
      *vars = poly;
    } else if (round == nb()) {
      Num r = cs.alloc_input(c0);
      Num claim = horner(cs, prior[round - 1].data(), prior[round - 1].size(), r);
      Num t_out = cs.alloc(t_out_step), eq_rho = cs.alloc(eq_rho_at_rb);
      cs.enforce(lc1(eq_rho.var), lc1(t_out.var), lc1(claim.var));
      *vars = {eq_rho, t_out};
    } else if (round < idx_outer_final()) {
      const size_t i = round - idx_outer_start();
      std::vector<Num> ps = alloc_coeffs(cs, outer_step[i].data(), 4), pc = alloc_coeffs(cs, outer_core[i].data(), 4);
      Num cs_, cc_;
      if (i == 0) {
        cs_ = prior[round - 1][1];
        cc_ = alloc_zero(cs);
      } else {
        Num r = cs.alloc_input(c0);
        cs_ = horner(cs, prior[round - 1].data(), 4, r);
        cc_ = horner(cs, prior[round - 1].data() + 4, 4, r);
      }
      enforce_sc_claim(cs, ps, cs_);
      enforce_sc_claim(cs, pc, cc_);
      *vars = ps;
      vars->insert(vars->end(), pc.begin(), pc.end());
    } else if (round == idx_outer_final()) {
      Num r = cs.alloc_input(c0);
      Num cs_ = horner(cs, prior[round - 1].data(), 4, r), cc_ = horner(cs, prior[round - 1].data() + 4, 4, r);
      Num s[3], c[3];
      for (int q = 0; q < 3; ++q) s[q] = cs.alloc(claim_step[q]);
      for (int q = 0; q < 3; ++q) c[q] = cs.alloc(claim_core[q]);
      Num tau = cs.alloc(tau_at_rx);
      enforce_outer_final(cs, s[0], s[1], s[2], tau, cs_);
      enforce_outer_final(cs, c[0], c[1], c[2], tau, cc_);
      *vars = {s[0], s[1], s[2], c[0], c[1], c[2], tau};
    } else if (round < idx_inner_final()) {
      const size_t idx = round - idx_inner_start();
      std::vector<Num> ps = alloc_coeffs(cs, inner_step[idx].data(), 3), pc = alloc_coeffs(cs, inner_core[idx].data(), 3);
      Num r = cs.alloc_input(c0);
      Num cs_, cc_;
      if (idx == 0) {
        Num r_sq = mul(cs, r, r);
        const std::vector<Num>& co = prior[idx_outer_final()];
        cs_ = joint_claim(cs, co[0], co[1], co[2], r, r_sq);
        cc_ = joint_claim(cs, co[3], co[4], co[5], r, r_sq);
      } else {
        cs_ = horner(cs, prior[round - 1].data(), 3, r);
        cc_ = horner(cs, prior[round - 1].data() + 3, 3, r);
      }
      enforce_sc_claim(cs, ps, cs_);
      enforce_sc_claim(cs, pc, cc_);
      *vars = ps;
      vars->insert(vars->end(), pc.begin(), pc.end());
      *chals = {r};
    } else if (round == idx_inner_final()) {
      Num r = cs.alloc_input(c0);
      Num cs_ = horner(cs, prior[round - 1].data(), 3, r), cc_ = horner(cs, prior[round - 1].data() + 3, 3, r);
      inputize(cs, prior[idx_outer_final()][6]);
      Num eXs = cs.alloc_input(eval_X_step), eXc = cs.alloc_input(eval_X_core);
      inputize(cs, prior[nb()][0]);
      Num eWs = cs.alloc(eval_W_step), eWc = cs.alloc(eval_W_core);
      const Num& r_y0 = prev_chals[idx_inner_start() + 1][0];
      enforce_inner_final(cs, r_y0, eWs, eXs, cs_);
      enforce_inner_final(cs, r_y0, eWc, eXc, cc_);
      *vars = {eWs, eWc};
    } else if (round < num_rounds()) {
      const bool step = round == idx_inner_final() + 1;
      Num e = cs.alloc(step ? eval_W_step : eval_W_core);
      const Num& prev = step ? prior[round - 1][0] : prior[round - 2][1];
      cs.enforce(lc1(e.var), lc1(CS::one()), lc1(prev.var));
      for (size_t j = 0; j + 1 < width; ++j) alloc_zero(cs);
    } else {
      throw Error(SP_ERR_INTERNAL, "verifier circuit: round out of range");
    }
  }
};

struct Csr {
  std::vector<fe_t> data;
  std::vector<uint32_t> idx;
  std::vector<uint64_t> ptr;
};
struct Shape {  // SplitMultiRoundR1CSShape + to_regular_shape
  size_t num_cons = 0, num_cons_unpadded = 0, num_rounds = 0, num_public = 0, width = 32, total_vars = 0, total_challenges = 0;
  std::vector<size_t> vars_unpadded, vars_padded, chals_per_round;
  Csr M[3];
  size_t num_io() const { return total_challenges + num_public; }
  size_t num_cols() const { return total_vars + 1 + num_io(); }
  void multiply_vec(const std::vector<fe_t>& z, std::vector<fe_t> out[3]) const {
    if (z.size() != num_cols()) throw Error(SP_ERR_INVALID_WITNESS_LENGTH, "InvalidWitnessLength");
    for (int m = 0; m < 3; ++m) out[m].assign(num_cons, fe_zero());
    par_for(3 * num_cons, 96, [&](size_t lo, size_t hi) {  // rows of the three matrices as one range
      for (size_t i = lo; i < hi; ++i) {
        const int m = (int)(i / num_cons);
        const size_t r = i % num_cons;
        fe_t acc = fe_zero();
        for (uint64_t k = M[m].ptr[r]; k < M[m].ptr[r + 1]; ++k) acc = fe_add<S>(acc, fe_mul<S>(M[m].data[k], z[M[m].idx[k]]));
        out[m][r] = acc;
      }
    });
  }
  static Shape from_circuit(const Circuit& vc) {
    Shape sh;
    sh.width = vc.width;
    sh.num_rounds = vc.num_rounds();
    CS cs;
    std::vector<std::vector<Num>> vars, chals;
    for (size_t round = 0; round < sh.num_rounds; ++round) {
      sh.chals_per_round.push_back(vc.num_challenges(round));
      sh.total_challenges += sh.chals_per_round.back();
      const size_t prev = cs.aux.size();
      std::vector<Num> v, c;
      vc.rounds(cs, round, vars, chals, nullptr, &v, &c);
      sh.vars_unpadded.push_back(cs.aux.size() - prev);
      vars.push_back(v);
      chals.push_back(c);
    }
    const size_t num_inputs = cs.inputs.size();
    sh.num_public = num_inputs - 1 - sh.total_challenges;
    sh.num_cons_unpadded = cs.cons.size();
    sh.num_cons = 1;
    while (sh.num_cons < sh.num_cons_unpadded) sh.num_cons <<= 1;
    std::vector<size_t> off_u(sh.num_rounds + 1, 0), off_p(sh.num_rounds + 1, 0);
    for (size_t r = 0; r < sh.num_rounds; ++r) {
      sh.vars_padded.push_back((sh.vars_unpadded[r] + sh.width - 1) / sh.width * sh.width);
      off_u[r + 1] = off_u[r] + sh.vars_unpadded[r];
      off_p[r + 1] = off_p[r] + sh.vars_padded[r];
    }
    sh.total_vars = off_p[sh.num_rounds];
    auto col_of = [&](const Var& v) -> uint32_t {
      if (v.id & Var::INPUT) return (uint32_t)(sh.total_vars + (v.id & ~Var::INPUT));
      size_t r = 0;
      while (!(v.id >= off_u[r] && v.id < off_u[r + 1])) ++r;
      return (uint32_t)(off_p[r] + (v.id - off_u[r]));
    };
    for (int m = 0; m < 3; ++m) {
      sh.M[m].ptr.push_back(0);
      for (const auto& con : cs.cons) {
        std::vector<std::pair<uint32_t, fe_t>> row;
        for (const auto& t : con[m]) {
          const uint32_t c = col_of(t.first);
          bool merged = false;
          for (auto& e : row)
            if (e.first == c) {
              e.second = fe_add<S>(e.second, t.second);
              merged = true;
            }
          if (!merged) row.push_back({c, t.second});
        }
        // the order add_constraint (bellpepper/r1cs.rs:234-287) sees: bellpepper-core's LinearCombination::iter() yields the input terms first, then
        // the aux terms, each list sorted by variable index (third-party type: the assumption is stated in oracle/neutronnova_zk.hpp)
        std::stable_sort(row.begin(), row.end(), [&](const std::pair<uint32_t, fe_t>& x, const std::pair<uint32_t, fe_t>& y) {
          const bool xi = x.first >= sh.total_vars, yi = y.first >= sh.total_vars;
          return xi != yi ? xi : x.first < y.first;
        });
        for (const auto& e : row) {
          if (fe_is_zero(e.second)) continue;
          sh.M[m].idx.push_back(e.first);
          sh.M[m].data.push_back(e.second);
        }
        sh.M[m].ptr.push_back(sh.M[m].idx.size());
      }
      while (sh.M[m].ptr.size() < sh.num_cons + 1) sh.M[m].ptr.push_back(sh.M[m].idx.size());
    }
    return sh;
  }
  // derived Serialize of SplitMultiRoundR1CSShape (src/r1cs/mod.rs:1401-1419), or of the R1CSShape that to_regular_shape (:1659-1672) makes of it,
  // into a wire sink of the library (bincode framing; the matrices as data / indices / indptr Vecs + cols, src/r1cs/sparse.rs:383-394)
  void write_bincode(sp_wire* w, bool regular) const {
    auto u64v = [&](uint64_t v) { ck(sp_wire_u64s(w, &v, 1, 0), "wire"); };
    auto usizes = [&](const std::vector<size_t>& v) {
      std::vector<uint64_t> t(v.begin(), v.end());
      ck(sp_wire_u64s(w, t.data(), t.size(), 1), "wire");
    };
    const size_t num_io = total_challenges + num_public;
    if (regular) {
      u64v(num_cons), u64v(total_vars), u64v(num_io);
    } else {
      u64v(num_cons), u64v(num_cons_unpadded), u64v(num_rounds);
      usizes(vars_unpadded), usizes(vars_padded), usizes(chals_per_round);
      u64v(num_public), u64v(width);
    }
    for (int m = 0; m < 3; ++m) {
      sp_csr cs{u64p(M[m].data.data()), M[m].idx.data(), M[m].ptr.data()};
      ck(sp_wire_matrix(w, &cs, num_cons, total_vars + 1 + num_io, 0), "wire matrix");
    }
  }
};

struct State {  // MultiRoundState (bellpepper/r1cs.rs:695-707)
  CS cs;
  std::vector<std::vector<Num>> vars_per_round, chal_vars_per_round;
  std::vector<std::vector<fe_t>> challenges;
  std::vector<std::vector<aff_t>> comm_per_round;  // rows of each round's commitment
  std::vector<std::vector<fe_t>> blind_per_round;
  std::vector<fe_t> w;
  size_t current = 0;
  double commit_ms = 0;  // wall time inside the per-round commitments (reported as a phase of its own)
  double synth_ms = 0, hash_ms = 0;  // diagnostics (SPARTAN_HOST_LAPS): synthesis of the rounds, transcript work of process_round
  size_t commits = 0, split_commits = 0, device_commits = 0;
  // the NEXT round's commitment, begun when this round's challenge was drawn (sp_hyrax_commit_split_begin: the blind's term and the early values)
  struct Ahead {
    sp_split_commit* job = nullptr;
    size_t round = 0, tape_pos = 0, fresh = 0;
    std::vector<fe_t> vals;
  } ahead;
  explicit State(const Shape& s) : w(s.total_vars, fe_zero()) {}
  State(const State&) = delete;
  State& operator=(const State&) = delete;
  ~State() { sp_hyrax_commit_split_drop(ahead.job); }
};
// Round `next`'s commitment, as far as it is known once round `next - 1` has drawn its challenge `c0`: h * blind (the tape's next value: only process_round
// draws between the rounds - it is PEEKED here and checked against the tape position when the round comes) and the early values of a sum-check round.
// The terms go to the library's table walkers (sp_hyrax_commit_split_begin) and are added under the device's work on the round's polynomials.
static const size_t SPLIT_COLS = 16;
// are the round commitments walked on the host (sp_hyrax_commit_split_*: the library keeps walkers and host tables of the key's leading columns)?
static inline bool split_commitments_on(const sp_ck* vc_ck) {
  static const bool off = [] {
    const char* e = getenv("SPARTAN_VC_SPLIT");  // "0": every round commitment through sp_hyrax_commit_small (the device walk), for A/B runs and tests
    return e && e[0] == '0';
  }();
  return !off && sp_hyrax_commit_split_available(vc_ck, SPLIT_COLS) != 0;
}
static inline void begin_next_commitment(sp_ctx* ctx, State& st, const Shape& s, const sp_ck* vc_ck, const Circuit& vc, size_t next, const fe_t& c0, const Tape& tape) {
  if (next >= s.num_rounds || s.vars_padded[next] != s.width || tape.pos >= tape.blocks || !split_commitments_on(vc_ck)) return;
  State::Ahead& a = st.ahead;
  sp_hyrax_commit_split_drop(a.job);
  a.job = nullptr;
  a.round = next;
  a.tape_pos = tape.pos;
  a.fresh = vc.fresh_vars(next);
  if (a.fresh) vc.early_values(next, c0, &a.vals);
  else {
    a.vals.clear();
    a.fresh = s.vars_unpadded[next];  // not a sum-check round: only the blind's term is early
  }
  const fe_t blind = fe_from_uniform<S>(tape.bytes + 64 * tape.pos);
  uint32_t cols[64];
  for (size_t k = 0; k < a.vals.size() && k < 64; ++k) cols[k] = (uint32_t)(a.fresh + k);
  if (a.fresh + a.vals.size() > 64 || sp_hyrax_commit_split_begin(ctx, vc_ck, cols, u64p(a.vals.data()), a.vals.size(), u64p(&blind), &a.job) != SP_OK) a.job = nullptr;
}
// process_round (bellpepper/r1cs.rs:734-816): synthesize the round, commit its (padded) variables with the width-32 key, absorb, squeeze
static inline std::vector<fe_t> process_round(sp_ctx* ctx, State& st, const Shape& s, const sp_ck* vc_ck, const Circuit& vc, size_t round, Tr& tr, Tape& tape) {
  if (round != st.current) throw Error(SP_ERR_INTERNAL, "process_round: rounds out of order");
  const fe_t* chal = (round > 0 && !st.challenges[round - 1].empty()) ? &st.challenges[round - 1][0] : nullptr;
  std::vector<Num> v, c;
  const auto ts0 = std::chrono::steady_clock::now();
  vc.rounds(st.cs, round, st.vars_per_round, st.chal_vars_per_round, chal, &v, &c);
  size_t su = 0, sp_ = 0;
  for (size_t r = 0; r < round; ++r) {
    su += s.vars_unpadded[r];
    sp_ += s.vars_padded[r];
  }
  for (size_t k = 0; k < s.vars_unpadded[round]; ++k) st.w[sp_ + k] = st.cs.aux[su + k];
  const size_t rows = s.vars_padded[round] / s.width;
  std::vector<fe_t> blinds(rows);
  for (auto& b : blinds) b = tape.next();
  std::vector<aff_t> comm(rows);
  const auto tc0 = std::chrono::steady_clock::now();
  st.synth_ms += std::chrono::duration<double, std::milli>(tc0 - ts0).count();
  // the commitment begun behind the previous challenge, if it is this round's and the tape and the synthesis agree with what it assumed
  bool split = false;
  if (st.ahead.job) {
    State::Ahead& a = st.ahead;
    const size_t nu = s.vars_unpadded[round];
    bool ok = a.round == round && rows == 1 && a.tape_pos + 1 == tape.pos && a.fresh + a.vals.size() == nu;
    for (size_t k = 0; ok && k < a.vals.size(); ++k) ok = fe_eq(a.vals[k], st.cs.aux[su + a.fresh + k]);
    for (size_t k = SPLIT_COLS; ok && k < nu; ++k) ok = fe_is_zero(st.cs.aux[su + k]);  // the library keeps host tables of the leading columns only
    sp_split_commit* job = a.job;
    a.job = nullptr;
    if (ok) {
      uint32_t cols[64];
      for (size_t k = 0; k < a.fresh && k < 64; ++k) cols[k] = (uint32_t)k;
      ck(sp_hyrax_commit_split_finish(ctx, job, cols, u64p(st.w.data() + sp_), a.fresh, u64p(&comm[0].x)), "commit round witness (split)");
      split = true;
      st.split_commits++;
    } else {
      sp_hyrax_commit_split_drop(job);
    }
  }
  if (!split && rows == 1 && split_commitments_on(vc_ck)) {
    // no commitment was begun for this round (the first round, or one whose early values did not match): the whole row through the table walkers all
    // the same - the hook stays host-only, which is what nn_prove promises the batched sum-checks (sp_ctx_round_hooks_host_only)
    bool low = true;
    for (size_t k = SPLIT_COLS; low && k < s.vars_unpadded[round]; ++k) low = fe_is_zero(st.cs.aux[su + k]);
    if (low) {
      const size_t nl = std::min(s.vars_unpadded[round], SPLIT_COLS);
      uint32_t cols[SPLIT_COLS];
      for (size_t k = 0; k < nl; ++k) cols[k] = (uint32_t)k;
      sp_split_commit* job = nullptr;
      ck(sp_hyrax_commit_split_begin(ctx, vc_ck, cols, nullptr, 0, u64p(&blinds[0]), &job), "commit round witness (walkers)");
      ck(sp_hyrax_commit_split_finish(ctx, job, cols, u64p(st.w.data() + sp_), nl, u64p(&comm[0].x)), "commit round witness (walkers)");
      split = true;
      st.split_commits++;
    }
  }
  if (!split) {
    st.device_commits++;
    for (size_t r = 0; r < rows; ++r)
      ck(sp_hyrax_commit_small(ctx, vc_ck, u64p(st.w.data() + sp_ + r * s.width), s.width, u64p(&blinds[r]), u64p(&comm[r].x)), "commit round witness");
  }
  st.commit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();
  st.commits += rows;
  const auto th0 = std::chrono::steady_clock::now();
  const std::vector<uint8_t> b = commitment_bytes(comm.data(), rows);
  tr.absorb("comm_w_round", b.data(), b.size());
  std::vector<fe_t> out(s.chals_per_round[round]);
  for (auto& x : out) x = tr.squeeze("challenge");
  st.vars_per_round.push_back(v);
  st.chal_vars_per_round.push_back(c);
  st.comm_per_round.push_back(comm);
  st.blind_per_round.push_back(blinds);
  st.challenges.push_back(out);
  st.current++;
  st.hash_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
  begin_next_commitment(ctx, st, s, vc_ck, vc, round + 1, out.empty() ? fe_zero() : out[0], tape);
  return out;
}

}  // namespace vcirc
}  // namespace spartan2
