"""A Python synthesis of the NeutronNova verifier circuit, written from the reference alone (no code shared with oracle/neutronnova_zk.hpp or
spartan2_amd/host/verifier_circuit.hpp):

  NeutronNovaVerifierCircuit::{num_challenges, rounds, num_rounds}      src/zk.rs:473-943
  its gadgets                                                          src/zk.rs:18-226
  ShapeCS (variable numbering: input 0 is ONE)                          src/bellpepper/shape_cs.rs:60-200
  multiround_r1cs_shape, add_constraint                                 src/bellpepper/r1cs.rs:606-693, :234-287
  SplitMultiRoundR1CSShape::new, to_regular_shape                       src/r1cs/mod.rs:1555-1672
  the serde layout of the two shapes and of SparseMatrix                src/r1cs/mod.rs:1401-1419, :169-179, src/r1cs/sparse.rs:383-394
  NeutronNovaVerifierKey::write_bytes                                   src/neutronnova_zk.rs:1305-1333

Third-party behaviour this relies on (bellpepper-core 0.4.0, absent from /root/reference; the same assumptions are stated in the oracle):
AllocatedNum::alloc / alloc_input allocate one aux / one input variable; mul and square allocate the product and enforce a * b = c; inputize
allocates an input and enforces input * 1 = a; LinearCombination merges equal variables and iterates its input terms first, then its aux terms,
each in increasing variable index."""
import hashlib
import struct

import pywire
from pywire import P_SCALAR as Q

ONE = ("in", 0)


class LC:
    """bellpepper_core::LinearCombination as add_constraint sees it"""

    def __init__(self):
        self.inputs, self.aux = {}, {}

    def add(self, var, coeff=1):
        d = self.inputs if var[0] == "in" else self.aux
        d[var[1]] = (d.get(var[1], 0) + coeff) % Q
        return self

    def sub(self, var):
        return self.add(var, Q - 1)

    def terms(self):  # iter(): inputs, then aux, each by index
        return [(("in", k), self.inputs[k]) for k in sorted(self.inputs)] + [(("aux", k), self.aux[k]) for k in sorted(self.aux)]


def lc(*vars_):
    out = LC()
    for v in vars_:
        out.add(v)
    return out


class ShapeCS:
    def __init__(self):
        self.num_aux, self.num_inputs, self.constraints = 0, 1, []  # input 0 = ONE

    def alloc(self):
        self.num_aux += 1
        return ("aux", self.num_aux - 1)

    def alloc_input(self):
        self.num_inputs += 1
        return ("in", self.num_inputs - 1)

    def enforce(self, a, b, c):
        self.constraints.append((a, b, c))


# ---- gadgets (src/zk.rs:18-226) ----------------------------------------------------------------------------------------------------------------------
def eval_poly_horner(cs, coeffs, x):
    acc = coeffs[-1]
    for c_i in reversed(coeffs[:-1]):
        new_acc = cs.alloc()
        cs.enforce(lc(acc), lc(x), lc(new_acc).sub(c_i))
        acc = new_acc
    return acc


def alloc_zero(cs):
    z = cs.alloc()
    cs.enforce(lc(z), lc(ONE), LC())
    return z


def alloc_coeffs(cs, n):
    return [cs.alloc() for _ in range(n)]


def enforce_sc_claim(cs, poly, claim):
    cs.enforce(lc(*poly).add(poly[0]), lc(ONE), lc(claim))


def num_mul(cs, a, b):
    p_ = cs.alloc()
    cs.enforce(lc(a), lc(b), lc(p_))
    return p_


def inputize(cs, a):
    inp = cs.alloc_input()
    cs.enforce(lc(inp), lc(ONE), lc(a))


def enforce_outer_sc_final_check(cs, Az, Bz, Cz, tau_at_rx, prev_claim):
    prod = num_mul(cs, Az, Bz)
    cs.enforce(lc(tau_at_rx), lc(prod).sub(Cz), lc(prev_claim))


def compute_joint_claim(cs, Az, Bz, Cz, r, r_sq):
    r_times_Bz = num_mul(cs, r, Bz)
    joint = cs.alloc()
    cs.enforce(lc(Cz), lc(r_sq), lc(joint).sub(Az).sub(r_times_Bz))
    return joint


def enforce_inner_sc_final_check(cs, r_y0, eval_W, eval_X, prev_claim):
    tmp_w = cs.alloc()
    cs.enforce(lc(eval_W), lc(ONE).sub(r_y0), lc(tmp_w))
    sum_z_expected = cs.alloc()
    cs.enforce(lc(eval_X), lc(r_y0), lc(sum_z_expected).sub(tmp_w))
    quotient = cs.alloc_input()
    cs.enforce(lc(quotient), lc(sum_z_expected), lc(prev_claim))


# ---- NeutronNovaVerifierCircuit (src/zk.rs:473-943) -------------------------------------------------------------------------------------------------
class VerifierCircuit:
    def __init__(self, num_rounds_z, num_rounds_x, num_rounds_y, width):
        self.nz, self.nx, self.ny, self.width = num_rounds_z, num_rounds_x, num_rounds_y, width
        self.idx_nifs_final = self.nz
        self.idx_outer_start = self.idx_nifs_final + 1
        self.idx_outer_final = self.idx_outer_start + self.nx
        self.idx_inner_start = self.idx_outer_final + 1
        self.idx_inner_final = self.idx_inner_start + self.ny
        self.idx_commit_w_step = self.idx_inner_final + 1
        self.idx_commit_w_core = self.idx_commit_w_step + 1
        self.num_rounds = self.idx_commit_w_core + 1

    def num_challenges(self, i):
        if i < self.nz:
            return 1
        if i == self.idx_nifs_final:
            return 0
        if i < self.idx_inner_final:
            return 1
        if i in (self.idx_inner_final, self.idx_commit_w_step, self.idx_commit_w_core):
            return 0
        raise ValueError("Unsatisfiable")

    def rounds(self, cs, i, prior, prev_challenges):
        if i < self.nz:
            poly = alloc_coeffs(cs, 4)
            if i == 0:
                claim = alloc_zero(cs)
            else:
                r = cs.alloc_input()
                claim = eval_poly_horner(cs, prior[i - 1], r)
            enforce_sc_claim(cs, poly, claim)
            return poly, []
        if i == self.idx_nifs_final:
            r = cs.alloc_input()
            claim = eval_poly_horner(cs, prior[i - 1], r)
            t_out_step = cs.alloc()
            eq_rho_at_rb = cs.alloc()
            cs.enforce(lc(eq_rho_at_rb), lc(t_out_step), lc(claim))
            return [eq_rho_at_rb, t_out_step], []
        if self.idx_nifs_final < i < self.idx_outer_final:
            k = i - self.idx_outer_start
            poly_step, poly_core = alloc_coeffs(cs, 4), alloc_coeffs(cs, 4)
            if k == 0:
                claim_step = prior[i - 1][1]
                claim_core = alloc_zero(cs)
            else:
                r = cs.alloc_input()
                claim_step = eval_poly_horner(cs, prior[i - 1][0:4], r)
                claim_core = eval_poly_horner(cs, prior[i - 1][4:8], r)
            enforce_sc_claim(cs, poly_step, claim_step)
            enforce_sc_claim(cs, poly_core, claim_core)
            return poly_step + poly_core, []
        if i == self.idx_outer_final:
            r = cs.alloc_input()
            claim_step = eval_poly_horner(cs, prior[i - 1][0:4], r)
            claim_core = eval_poly_horner(cs, prior[i - 1][4:8], r)
            Az_s, Bz_s, Cz_s, Az_c, Bz_c, Cz_c, tau = (cs.alloc() for _ in range(7))
            enforce_outer_sc_final_check(cs, Az_s, Bz_s, Cz_s, tau, claim_step)
            enforce_outer_sc_final_check(cs, Az_c, Bz_c, Cz_c, tau, claim_core)
            return [Az_s, Bz_s, Cz_s, Az_c, Bz_c, Cz_c, tau], []
        if self.idx_inner_start <= i < self.idx_inner_final:
            idx = i - self.idx_inner_start
            poly_step, poly_core = alloc_coeffs(cs, 3), alloc_coeffs(cs, 3)
            r = cs.alloc_input()
            if idx == 0:
                r_sq = num_mul(cs, r, r)  # AllocatedNum::square
                co = prior[self.idx_outer_final]
                claim_step = compute_joint_claim(cs, co[0], co[1], co[2], r, r_sq)
                claim_core = compute_joint_claim(cs, co[3], co[4], co[5], r, r_sq)
            else:
                claim_step = eval_poly_horner(cs, prior[i - 1][0:3], r)
                claim_core = eval_poly_horner(cs, prior[i - 1][3:6], r)
            enforce_sc_claim(cs, poly_step, claim_step)
            enforce_sc_claim(cs, poly_core, claim_core)
            return poly_step + poly_core, [r]
        if i == self.idx_inner_final:
            r = cs.alloc_input()
            claim_step = eval_poly_horner(cs, prior[i - 1][0:3], r)
            claim_core = eval_poly_horner(cs, prior[i - 1][3:6], r)
            inputize(cs, prior[self.idx_outer_final][6])  # tau_at_rx
            eval_X_step, eval_X_core = cs.alloc_input(), cs.alloc_input()
            inputize(cs, prior[self.idx_nifs_final][0])  # eq_rho_at_rb
            eval_W_step, eval_W_core = cs.alloc(), cs.alloc()
            r_y0 = prev_challenges[self.idx_inner_start + 1][0]
            enforce_inner_sc_final_check(cs, r_y0, eval_W_step, eval_X_step, claim_step)
            enforce_inner_sc_final_check(cs, r_y0, eval_W_core, eval_X_core, claim_core)
            return [eval_W_step, eval_W_core], []
        if i in (self.idx_commit_w_step, self.idx_commit_w_core):
            e = cs.alloc()
            prev = prior[i - 1][0] if i == self.idx_commit_w_step else prior[i - 2][1]
            cs.enforce(lc(e), lc(ONE), lc(prev))
            for _ in range(self.width - 1):
                alloc_zero(cs)
            return [], []
        raise ValueError("Unsatisfiable")


# ---- multiround_r1cs_shape + SplitMultiRoundR1CSShape::new ---------------------------------------------------------------------------------------------
def multiround_shape(circuit):
    cs = ShapeCS()
    prior, chals, vars_per_round, chals_per_round = [], [], [], []
    for rnd in range(circuit.num_rounds):
        chals_per_round.append(circuit.num_challenges(rnd))
        before = cs.num_aux
        v, c = circuit.rounds(cs, rnd, prior, chals)
        vars_per_round.append(cs.num_aux - before)
        prior.append(v)
        chals.append(c)
    total_vars, num_inputs, num_cons = cs.num_aux, cs.num_inputs, len(cs.constraints)
    mats = []
    for m in range(3):  # add_constraint: zero coefficients are not stored; inputs sit at num_vars + index
        data, idx, ptr = [], [], [0]
        for con in cs.constraints:
            for (kind, k), coeff in con[m].terms():
                if coeff != 0:
                    data.append(coeff)
                    idx.append(k + total_vars if kind == "in" else k)
            ptr.append(len(idx))
        mats.append([data, idx, ptr, total_vars + num_inputs])
    num_public = num_inputs - 1 - sum(chals_per_round)
    # SplitMultiRoundR1CSShape::new (r1cs/mod.rs:1555-1657)
    width = circuit.width
    padded = [-(-n // width) * width for n in vars_per_round]
    total_padded, cons_padded = sum(padded), 1 << max(0, (num_cons - 1).bit_length())
    assert mats[0][3] == total_vars + 1 + num_public + sum(chals_per_round)
    starts, pstarts = [0], [0]
    for n, p_ in zip(vars_per_round, padded):
        starts.append(starts[-1] + n)
        pstarts.append(pstarts[-1] + p_)

    def remap(c):
        for r in range(len(vars_per_round)):
            if starts[r] <= c < starts[r + 1]:
                return pstarts[r] + (c - starts[r])
        return c + total_padded - total_vars

    for mt in mats:
        mt[1] = [remap(c) for c in mt[1]]
        mt[3] += total_padded - total_vars
        mt[2] += [mt[2][-1]] * (cons_padded - num_cons)
    return dict(num_cons=cons_padded, num_cons_unpadded=num_cons, num_rounds=circuit.num_rounds, vars_unpadded=vars_per_round, vars_padded=padded,
                chals_per_round=chals_per_round, num_public=num_public, width=width, mats=mats)


# ---- serde (bincode, little-endian, fixint) -------------------------------------------------------------------------------------------------------
def _u64(v):
    return struct.pack("<Q", v)


def _usizes(v):
    return _u64(len(v)) + b"".join(_u64(x) for x in v)


def _matrix(mt):  # SparseMatrix { data, indices, indptr, cols }
    data, idx, ptr, cols = mt
    return _u64(len(data)) + b"".join(int(c).to_bytes(32, "little") for c in data) + _usizes(idx) + _usizes(ptr) + _u64(cols)


def multiround_shape_bytes(sh):
    return (_u64(sh["num_cons"]) + _u64(sh["num_cons_unpadded"]) + _u64(sh["num_rounds"]) + _usizes(sh["vars_unpadded"]) + _usizes(sh["vars_padded"])
            + _usizes(sh["chals_per_round"]) + _u64(sh["num_public"]) + _u64(sh["width"]) + b"".join(_matrix(m) for m in sh["mats"]))


def regular_shape_bytes(sh):  # to_regular_shape (:1659-1672): R1CSShape { num_cons, num_vars, num_io, A, B, C }
    return _u64(sh["num_cons"]) + _u64(sum(sh["vars_padded"])) + _u64(sum(sh["chals_per_round"]) + sh["num_public"]) + b"".join(_matrix(m) for m in sh["mats"])


def nn_vk_digest(step_inst, core_inst, num_steps, gens):
    """SHA-256 over NeutronNovaVerifierKey::write_bytes (src/neutronnova_zk.rs:1305-1333). gens: the (>= 2049, 8) generator limbs of label "ck"
    (PCS::setup(b"ck", ., 2048) for ck / vk_ee, PCS::setup(b"ck", ., 32) for vc_ck / vc_vk: the same label, a shorter prefix, h right after it)."""
    S_step, S_core = pywire.equalize(pywire.pad_shape(step_inst), pywire.pad_shape(core_inst))
    nv = S_step[0]["num_shared"] + S_step[0]["num_precommitted"] + S_step[0]["num_rest"]
    nb = max(0, (num_steps - 1).bit_length())
    sh = multiround_shape(VerifierCircuit(nb, S_step[0]["num_cons"].bit_length() - 1, nv.bit_length(), 32))
    h = hashlib.sha256()
    w = pywire.Writer()
    w.hyrax_key(gens[:2048], gens[2048])
    w.hyrax_key(gens[:2048], gens[2048])
    h.update(w.bytes())
    pywire.shape_write_bytes(h, *S_step)
    pywire.shape_write_bytes(h, *S_core)
    h.update(multiround_shape_bytes(sh))
    h.update(regular_shape_bytes(sh))
    w = pywire.Writer()
    w.hyrax_key(gens[:32], gens[32])
    w.hyrax_key(gens[:32], gens[32])
    h.update(w.bytes())
    return h.digest(), sh
