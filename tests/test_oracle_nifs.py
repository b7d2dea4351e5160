"""CPU: the NIFS oracle (oracle/nifs.hpp) against the reference's own property tests (src/big_num/small_value.rs:254-403,
src/polys/power.rs:93-160) and against independent Python-integer arithmetic; the cached-i64 branch of NeutronNovaNIFS::prove
(src/neutronnova_zk.rs:511-1273) must agree with the field branch, and the folded layers must satisfy the folded claim."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol

P = ol.MODULI[0]
SMALL_MAX = (1 << 62) - 1


def test_to_small_vec_or_zero_reference_cases():
    # small_value.rs:257-296
    small, large = ol.to_small_vec_or_zero(ol.mont_array(list(range(10))))
    assert large.size == 0 and small.tolist() == list(range(10))
    rng = np.random.default_rng(11111)
    big = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(2)]
    small, large = ol.to_small_vec_or_zero(ol.mont_array([5, big[0], P - 3, big[1], 100]))
    assert small.tolist() == [5, 0, -3, 0, 100] and large.tolist() == [1, 3]
    small, large = ol.to_small_vec_or_zero(ol.mont_array([SMALL_MAX, P - SMALL_MAX]))
    assert large.size == 0 and small.tolist() == [SMALL_MAX, -SMALL_MAX]
    small, large = ol.to_small_vec_or_zero(ol.mont_array([SMALL_MAX + 1, P - SMALL_MAX - 1, 0]))
    assert large.tolist() == [0, 1] and small.tolist() == [0, 0, 0]


def _small_dot(f_ints, a, b):
    L = ol.lib()
    f = ol.mont_array(f_ints)
    a = np.asarray(a, dtype=np.int64)
    b = np.asarray(b, dtype=np.int64)
    out = np.zeros(4, dtype=np.uint64)
    rc = L.orc_small_acc_dot(ol.p64(f), a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(f_ints)), ol.p64(out))
    assert rc == 0
    return ol.from_mont(out)


def test_small_accumulator_property_tests():
    rng = np.random.default_rng(12345)
    n = 1000
    f = [int.from_bytes(rng.bytes(32), "little") % P for _ in range(n)]
    s = [(i % 201) - 100 for i in range(n)]  # small_value.rs:304-306
    assert _small_dot(f, s, [1] * n) == sum(fi * si for fi, si in zip(f, s)) % P
    a = [(i + 1) * 1_000_000 for i in range(100)]  # :341-342
    b = [(i + 1) * 2_000_000 for i in range(100)]
    assert _small_dot(f[:100], a, b) == sum(fi * ai * bi for fi, ai, bi in zip(f, a, b)) % P
    for k in range(1, 4):  # few products: the <= 4-limb fast path of reduce_7_to_field (:372-395)
        assert _small_dot(f[:k], [1] * k, [1] * k) == sum(f[:k]) % P
    # extremes: differences of +-(2^62-1) values multiply to ~2^126, both signs
    ext = [2 * SMALL_MAX, -2 * SMALL_MAX, 2 * SMALL_MAX, -1]
    ext2 = [2 * SMALL_MAX, 2 * SMALL_MAX, -2 * SMALL_MAX, 1]
    assert _small_dot(f[:4], ext, ext2) == sum(fi * x * y for fi, x, y in zip(f, ext, ext2)) % P


def test_pow_split_evals_outer_product():
    tau = 0x1234567890ABCDEF1122334455667788 % P
    for n in (16, 32, 1 << 9):
        ell, left, right = ol.tensor_decomp(n)
        assert left * right == 1 << ell and left >= right
        e = ol.ints_of(ol.pow_split_evals(ol.to_mont(tau), ell, left, right))
        for k in (0, 1, left - 1, left, n - 1, n // 3):
            assert e[k % left] * e[left + k // left] % P == pow(tau, k, P)  # power.rs:93-160


def _instances(rng, n_padded, total, n_large):
    """Satisfying layers Az*Bz = Cz with mostly small entries (bits / small signed) and a few full-size ones."""
    a = rng.integers(-3, 4, size=(n_padded, total)).astype(object)
    b = rng.integers(0, 2, size=(n_padded, total)).astype(object)
    for _ in range(n_large):
        i, k = int(rng.integers(n_padded)), int(rng.integers(total))
        a[i, k] = int.from_bytes(rng.bytes(32), "little") % P
        b[i, k] = int.from_bytes(rng.bytes(32), "little") % P
    c = (a * b) % P
    arr = lambda m: np.stack([ol.mont_array([int(v) % P for v in row]) for row in m])
    return arr(a), arr(b), arr(c), a, b, c


@pytest.mark.parametrize("n_padded,num_cons,n_large", [(2, 16, 0), (4, 32, 3), (8, 64, 5), (16, 32, 0)])
def test_nifs_i64_branch_equals_field_branch_and_claim_holds(n_padded, num_cons, n_large):
    rng = np.random.default_rng(1000 + n_padded * num_cons)
    ell, left, right = ol.tensor_decomp(num_cons)
    total = left * right
    A, B, C, ai, bi, ci = _instances(rng, n_padded, total, n_large)
    tau = int.from_bytes(rng.bytes(32), "little") % P
    E = ol.pow_split_evals(ol.to_mont(tau), ell, left, right)
    ell_b = n_padded.bit_length() - 1
    rhos = ol.mont_array([int.from_bytes(rng.bytes(32), "little") % P for _ in range(ell_b)])
    outs = []
    for use_i64 in (False, True):
        tr = ol.Transcript(b"nifs-test")
        outs.append(ol.nifs_prove_core(left, right, E, rhos, A, B, C, use_i64, ol.transcript_round_hook(tr)))
    f, s = outs
    for key in ("polys", "r_bs", "A", "B", "C", "T_out", "eq_rho_at_rb"):
        assert (f[key] == s[key]).all(), key
    # folded layers = sum_b w_b * layer_b with w = weights_from_r(r_bs) (src/r1cs/mod.rs:153-166), by Python integers
    r = ol.ints_of(f["r_bs"])
    w = []
    for i in range(n_padded):
        wi, k = 1, i
        for rj in r:
            wi = wi * (rj if k & 1 else (1 - rj)) % P
            k >>= 1
        w.append(wi)
    fa, fb, fc = ol.ints_of(f["A"]), ol.ints_of(f["B"]), ol.ints_of(f["C"])
    for k in (0, 1, total // 2, total - 1):
        assert fa[k] == sum(w[i] * int(ai[i, k]) for i in range(n_padded)) % P
        assert fc[k] == sum(w[i] * int(ci[i, k]) for i in range(n_padded)) % P
    # the folded claim: T_out = sum_k E[k] (A_f B_f - C_f)[k]  (what the verifier circuit later checks)
    e = ol.ints_of(E)
    lhs = sum(e[k % left] * e[left + k // left] % P * ((fa[k] * fb[k] - fc[k]) % P) for k in range(total)) % P
    assert lhs == ol.from_mont(f["T_out"])
    # every round polynomial satisfies the sum-check consistency used by finish_round!: p(0) + p(1) = T_prev (p carries the eq(X, rho_t) factor)  — checked through
    # T chaining: p_t evaluated at r_t is the next target; the last one over eq(r_b, rho) is T_out
    rho = ol.ints_of(rhos)
    T, acc = 0, 1
    for t in range(ell_b):
        co = ol.ints_of(f["polys"][t])
        p0, p1 = co[0], sum(co) % P
        assert (p0 + p1) % P == T
        T = sum(co[i] * pow(r[t], i, P) for i in range(4)) % P
        acc = acc * ((1 - r[t]) * (1 - rho[t]) + r[t] * rho[t]) % P
    assert T * pow(acc, -1, P) % P == ol.from_mont(f["T_out"]) and acc == ol.from_mont(f["eq_rho_at_rb"])


def test_nifs_prove_whole_both_branches_agree():
    """oracle nifs_prove (preamble + rounds + witness / instance folds) on 3 synthetic instances: both branches give the same transcript-driven
    outputs; instance 3 is the padding clone of instance 0."""
    from spartan2_amd import frontend

    L = ol.lib()
    okey = ctypes.c_void_p(L.orc_hyrax_setup(b"ck", ctypes.c_size_t(2048)))
    insts = [frontend.synthetic_circuit(20, 5, num_public=2, shared_permille=0, precommitted_permille=1000, witness_seed=s) for s in (1, 2, 3)]
    osh = ol.OracleShape(insts[0])
    rows = osh.num_vars // 2048
    rng = np.random.default_rng(3)
    Ws = np.zeros((3, osh.num_vars, 4), dtype=np.uint64)
    for k, inst in enumerate(insts):
        Ws[k, : len(inst.witness)] = ol.mont_array([int(x) for x in inst.witness])
    X = np.stack([ol.mont_array([int(x) for x in i.publics]) for i in insts])
    r_W = np.stack([ol.random_field_array(rng, rows) for _ in insts])
    comms = np.zeros((3, rows, 8), dtype=np.uint64)
    for k in range(3):
        assert L.orc_hyrax_commit(okey, ol.p64(Ws[k]), ctypes.c_size_t(osh.num_vars), ol.p64(r_W[k]), 1, ol.p64(comms[k])) == 0
    outs = [ol.nifs_prove(osh, okey, comms, X, Ws, r_W, use, ol.Transcript(b"nn"), ol.transcript_round_hook(ol.Transcript(b"vc"))) for use in (False, True)]
    for key in outs[0]:
        assert (outs[0][key] == outs[1][key]).all(), key
    o = outs[0]
    assert o["polys"].shape[0] == 2
    # folded commitment opens to the folded witness under the folded blind (no rest segment here)
    recommit = np.zeros_like(o["folded_comm"])
    assert L.orc_hyrax_commit(okey, ol.p64(o["folded_W"]), ctypes.c_size_t(osh.num_vars), ol.p64(o["folded_rW"]), 0, ol.p64(recommit)) == 0
    assert (recommit == o["folded_comm"]).all()
    L.orc_hyrax_free(okey)


def test_oracle_reproduces_the_frozen_nifs_vectors():
    """tests/golden/nifs_small.json regenerates bit-for-bit from the oracle (both branches)."""
    import hashlib
    import json
    import os

    import make_golden_impl

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nifs_small.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        left, right, E, rhos, A, B, C = make_golden_impl.nifs_inputs(case["n_inst"], case["num_cons"])
        for use_i64 in (True, False):
            o = ol.nifs_prove_core(left, right, E, rhos, A, B, C, use_i64, ol.transcript_round_hook(ol.Transcript(b"golden-nifs")))
            assert o["polys"].tobytes().hex() == case["polys"] and o["T_out"].tobytes().hex() == case["T_out"]
            assert hashlib.sha256(o["C"].tobytes()).hexdigest() == case["C_sha256"]
