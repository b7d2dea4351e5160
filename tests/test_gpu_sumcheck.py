"""GPU parity: multilinear tables, eq tables and both sum-check provers through the C ABI vs the CPU oracle,
bit-exact on canonical Montgomery limbs (integer field arithmetic). Mirrors the reference's own tests:
bind-sequence == evaluate (src/polys/multilinear.rs:346-379), sum-check prove -> verify on seeded tables
(src/sumcheck.rs:1443-1572, seed 0xDEADBEEF)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import lib as olib, p64
from spartan2_amd import hip

pytestmark = pytest.mark.gpu

SEED = 0xDEADBEEF


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def rand_table(rng, n):
    return ol.random_field_array(rng, n)


def oracle_bind(Z, lo, hi, r):
    cur = Z.copy()
    lo_c, hi_c = ctypes.c_size_t(lo), ctypes.c_size_t(hi)
    assert olib().orc_bind_top(p64(cur), ctypes.c_size_t(len(Z)), ctypes.byref(lo_c), ctypes.byref(hi_c), p64(r)) == 0
    return cur[: len(Z) // 2], lo_c.value, hi_c.value


@pytest.mark.parametrize("logn", [1, 2, 7, 12, 15])
def test_bind_top_dense(ctx, logn):
    rng = np.random.default_rng(SEED + logn)
    n = 1 << logn
    Z = rand_table(rng, n)
    r = rand_table(rng, 1)[0]
    t = hip.Table.from_host(ctx, Z)
    t.bind_top(r)
    want, lo, hi = oracle_bind(Z, hip.SIZE_MAX, hip.SIZE_MAX, r)
    assert t.info() == (n // 2, lo, hi)
    assert (t.read() == want).all()


@pytest.mark.parametrize("lo_eff,hi_eff", [(300, 0), (300, 77), (77, 300), (512, 512), (0, 0), (1, 1), (0, 5), (511, 512)])
def test_bind_top_zero_structure_branches(ctx, lo_eff, hi_eff):
    # src/polys/multilinear.rs:101-163
    rng = np.random.default_rng(SEED)
    n2, n = 1024, 512
    Z = rand_table(rng, n2)
    Z[lo_eff:n] = 0
    Z[n + hi_eff :] = 0
    r = rand_table(rng, 1)[0]
    t = hip.Table.from_host(ctx, Z, lo_eff, hi_eff)
    t.bind_top(r)
    want, lo, hi = oracle_bind(Z, lo_eff, hi_eff, r)
    assert t.info() == (n, lo, hi)
    assert (t.read() == want).all()


def test_bind_sequence_equals_evaluate(ctx):
    rng = np.random.default_rng(SEED + 1)
    for ell in (3, 9, 13):
        Z = rand_table(rng, 1 << ell)
        r = rand_table(rng, ell)
        want = np.zeros(4, dtype=np.uint64)
        olib().orc_multilinear_evaluate(p64(Z), ctypes.c_size_t(1 << ell), p64(r), ctypes.c_size_t(ell), p64(want))
        t = hip.Table.from_host(ctx, Z)
        for k in range(ell):
            t.bind_top(r[k])
        assert (t.read(0, 1)[0] == want).all()


@pytest.mark.parametrize("ell", [0, 1, 5, 10, 11, 16, 20, 21])
def test_eq_table(ctx, ell):
    rng = np.random.default_rng(SEED + 2 + ell)
    r = rand_table(rng, ell) if ell else np.zeros((0, 4), dtype=np.uint64)
    t = hip.Table.eq(ctx, r)
    want = np.zeros((1 << ell, 4), dtype=np.uint64)
    olib().orc_eq_evals(p64(r) if ell else None, ctypes.c_size_t(ell), p64(want))
    assert (t.read() == want).all()


def test_table_dot(ctx):
    rng = np.random.default_rng(SEED + 3)
    for n in (1, 1000, 5000):
        a, b = rand_table(rng, n), rand_table(rng, n)
        want = np.zeros(4, dtype=np.uint64)
        olib().orc_field_dot(0, p64(a), p64(b), ctypes.c_size_t(n), p64(want))
        got = hip.table_dot(ctx, hip.Table.from_host(ctx, a), hip.Table.from_host(ctx, b), n)
        assert (got == want).all()


def oracle_cubic(claim, taus, A, B, C, label=b"sc"):
    ell = len(taus)
    tr = ol.Transcript(label)
    polys = np.zeros((ell, 3, 4), dtype=np.uint64)
    r = np.zeros((ell, 4), dtype=np.uint64)
    fin = np.zeros((3, 4), dtype=np.uint64)
    a, b, c = A.copy(), B.copy(), C.copy()
    assert olib().orc_sumcheck_cubic3(p64(claim), p64(taus), ctypes.c_size_t(ell), p64(a), p64(b), p64(c), tr.h, p64(polys), p64(r), p64(fin)) == 0
    return polys, r, fin, tr


def satisfying_tables(rng, n):
    A, B = rand_table(rng, n), rand_table(rng, n)
    C = np.zeros_like(A)
    # C = A o B (an R1CS-satisfying triple, so the zero-check claim 0 is honest)
    for i in range(n):
        olib().orc_field_binop(0, 2, p64(A[i]), p64(B[i]), p64(C[i]))
    return A, B, C


@pytest.mark.parametrize("ell", [1, 2, 3, 4, 7, 11, 12, 14])
def test_sumcheck_cubic_matches_oracle(ctx, ell):
    rng = np.random.default_rng(SEED + 10 + ell)
    n = 1 << ell
    A, B, C = satisfying_tables(rng, n)
    taus = rand_table(rng, ell)
    claim = np.zeros(4, dtype=np.uint64)
    want_polys, want_r, want_fin, otr = oracle_cubic(claim, taus, A, B, C)
    tr = hip.Transcript(ctx, b"sc")
    tA, tB, tC = (hip.Table.from_host(ctx, x) for x in (A, B, C))
    polys, r, fin = hip.sumcheck_cubic3(ctx, claim, taus, tA, tB, tC, tr)
    assert (r == want_r).all()
    assert (polys == want_polys).all()
    assert (fin == want_fin).all()
    # transcripts stay in lock-step afterwards
    assert (tr.squeeze(b"after") == otr.squeeze(b"after")).all()
    # and the oracle's restated verifier accepts the GPU proof (src/sumcheck.rs:67-114)
    vtr = ol.Transcript(b"sc")
    e = np.zeros(4, dtype=np.uint64)
    rr = np.zeros((ell, 4), dtype=np.uint64)
    flat = np.ascontiguousarray(polys.reshape(-1))
    assert olib().orc_sumcheck_verify(p64(claim), ctypes.c_size_t(ell), ctypes.c_size_t(3), p64(flat), vtr.h, p64(e), p64(rr)) == 0
    assert (rr == r).all()


def test_sumcheck_large_tables_take_the_streaming_kernels(ctx):
    """2^19-entry tables: the first fused launches run k_bind_eval_*_stream (wave-level lazy partials + k_sum_partials_lazy)."""
    rng = np.random.default_rng(SEED + 77)
    ell = 19
    n = 1 << ell
    A, B = rand_table(rng, n), rand_table(rng, n)
    C = rand_table(rng, n)  # not a satisfying triple: the claim below is simply whatever the honest sum is NOT; parity still must hold
    taus = rand_table(rng, ell)
    claim = rand_table(rng, 1)[0]
    want_polys, want_r, want_fin, _ = oracle_cubic(claim, taus, A, B, C)
    tr = hip.Transcript(ctx, b"sc")
    polys, r, fin = hip.sumcheck_cubic3(ctx, claim, taus, *(hip.Table.from_host(ctx, x) for x in (A, B, C)), tr)
    assert (polys == want_polys).all() and (r == want_r).all() and (fin == want_fin).all()
    qclaim = np.zeros(4, dtype=np.uint64)
    olib().orc_field_dot(0, p64(A), p64(B), ctypes.c_size_t(n), p64(qclaim))
    full = (hip.SIZE_MAX, hip.SIZE_MAX)
    want = oracle_quad(qclaim, ell, A, full, B, full)
    trq = hip.Transcript(ctx, b"sq")
    got = hip.sumcheck_quad(ctx, qclaim, ell, hip.Table.from_host(ctx, A), hip.Table.from_host(ctx, B), trq)
    for g, w in zip(got, want):
        assert (g == w).all()


def test_sumcheck_cubic_2_pow_21_rows(ctx):
    """One size up from the bench shape (a 4 KiB SHA-256 message pads to 2^21 constraints): eq pyramids of 10 and 11 variables, every kernel
    regime from the streaming ones down to the single-block tail."""
    rng = np.random.default_rng(SEED + 2121)
    ell = 21
    n = 1 << ell
    A, B, C = rand_table(rng, n), rand_table(rng, n), rand_table(rng, n)
    taus = rand_table(rng, ell)
    claim = rand_table(rng, 1)[0]
    want_polys, want_r, want_fin, _ = oracle_cubic(claim, taus, A, B, C)
    tr = hip.Transcript(ctx, b"sc")
    polys, r, fin = hip.sumcheck_cubic3(ctx, claim, taus, *(hip.Table.from_host(ctx, x) for x in (A, B, C)), tr)
    assert (polys == want_polys).all() and (r == want_r).all() and (fin == want_fin).all()


def test_sumcheck_cubic_tau_zero_fallback(ctx):
    # derive_from_claim returns None when tau_i == 0 (src/sumcheck.rs:1289-1291) -> third sum computed directly
    rng = np.random.default_rng(SEED + 30)
    ell = 6
    n = 1 << ell
    A, B, C = satisfying_tables(rng, n)
    taus = rand_table(rng, ell)
    taus[0] = 0
    taus[4] = 0
    claim = np.zeros(4, dtype=np.uint64)
    want_polys, want_r, want_fin, _ = oracle_cubic(claim, taus, A, B, C)
    tr = hip.Transcript(ctx, b"sc")
    polys, r, fin = hip.sumcheck_cubic3(ctx, claim, taus, *(hip.Table.from_host(ctx, x) for x in (A, B, C)), tr)
    assert (polys == want_polys).all() and (r == want_r).all() and (fin == want_fin).all()


@pytest.mark.parametrize("ell", [3, 4, 9, 10, 13])
def test_two_round_trips_of_the_resident_tail(ctx, ell):
    """Once the resident tail is down to one block it sends two rounds per trip (kernels_poly.hpp TAIL_WIDE_VALS: the next round's sums as
    polynomials in this round's challenge). Both parities of the remaining round count, a dishonest claim (derive_from_claim uses it), and
    tau = 0 at the first / second round of a pair (fallback_three_inputs from the coefficient sums of t(-1)); the quadratic form too."""
    rng = np.random.default_rng(SEED + 600 + ell)
    n = 1 << ell
    A, B, C = rand_table(rng, n), rand_table(rng, n), rand_table(rng, n)
    for zero_at in (None, ell - 1, ell - 2, ell - 3 if ell > 3 else 0):
        taus = rand_table(rng, ell)
        if zero_at is not None:
            taus[zero_at] = 0
        claim = rand_table(rng, 1)[0]
        want_polys, want_r, want_fin, _ = oracle_cubic(claim, taus, A, B, C)
        tr = hip.Transcript(ctx, b"sc")
        polys, r, fin = hip.sumcheck_cubic3(ctx, claim, taus, *(hip.Table.from_host(ctx, x) for x in (A, B, C)), tr)
        assert (polys == want_polys).all() and (r == want_r).all() and (fin == want_fin).all(), zero_at
    full = (hip.SIZE_MAX, hip.SIZE_MAX)
    qclaim = rand_table(rng, 1)[0]
    want = oracle_quad(qclaim, ell, A, full, B, full)
    got = hip.sumcheck_quad(ctx, qclaim, ell, hip.Table.from_host(ctx, A), hip.Table.from_host(ctx, B), hip.Transcript(ctx, b"sq"))
    for g, w in zip(got, want):
        assert (g == w).all()


def oracle_quad(claim, rounds, A, effA, B, effB):
    tr = ol.Transcript(b"sq")
    polys = np.zeros((rounds, 2, 4), dtype=np.uint64)
    r = np.zeros((rounds, 4), dtype=np.uint64)
    fin = np.zeros((2, 4), dtype=np.uint64)
    a, b = A.copy(), B.copy()
    assert olib().orc_sumcheck_quad(p64(claim), ctypes.c_size_t(rounds), p64(a), ctypes.c_size_t(effA[0]), ctypes.c_size_t(effA[1]), p64(b),
                                    ctypes.c_size_t(effB[0]), ctypes.c_size_t(effB[1]), tr.h, p64(polys), p64(r), p64(fin)) == 0
    return polys, r, fin


@pytest.mark.parametrize("rounds", [1, 2, 5, 11, 13])
def test_sumcheck_quad_matches_oracle(ctx, rounds):
    rng = np.random.default_rng(SEED + 40 + rounds)
    n = 1 << rounds
    A, B = rand_table(rng, n), rand_table(rng, n)
    claim = np.zeros(4, dtype=np.uint64)
    olib().orc_field_dot(0, p64(A), p64(B), ctypes.c_size_t(n), p64(claim))
    full = (hip.SIZE_MAX, hip.SIZE_MAX)
    want = oracle_quad(claim, rounds, A, full, B, full)
    tr = hip.Transcript(ctx, b"sq")
    got = hip.sumcheck_quad(ctx, claim, rounds, hip.Table.from_host(ctx, A), hip.Table.from_host(ctx, B), tr)
    for g, w in zip(got, want):
        assert (g == w).all()


def test_sumcheck_quad_with_zero_structure_like_spartan_inner(ctx):
    """The inner sum-check of src/spartan.rs:323-394: tables of length 2M whose high halves are zero except the first
    num_extra entries. The manual round 0 there is value-identical to a generic round with (lo_eff, hi_eff) = (M, num_extra)."""
    rng = np.random.default_rng(SEED + 50)
    rounds, M, extra = 11, 1024, 5
    A, B = rand_table(rng, 2 * M), rand_table(rng, 2 * M)
    A[M + extra :] = 0
    B[M + extra :] = 0
    claim = np.zeros(4, dtype=np.uint64)
    olib().orc_field_dot(0, p64(A), p64(B), ctypes.c_size_t(2 * M), p64(claim))
    want_dense = oracle_quad(claim, rounds, A, (hip.SIZE_MAX, hip.SIZE_MAX), B, (hip.SIZE_MAX, hip.SIZE_MAX))
    want = oracle_quad(claim, rounds, A, (M, extra), B, (M, extra))
    for a, b in zip(want, want_dense):
        assert (a == b).all()
    tr = hip.Transcript(ctx, b"sq")
    got = hip.sumcheck_quad(ctx, claim, rounds, hip.Table.from_host(ctx, A, M, extra), hip.Table.from_host(ctx, B, M, extra), tr)
    for g, w in zip(got, want):
        assert (g == w).all()


def test_sumcheck_quad_zero_structure_at_streaming_size(ctx):
    """The same shape at 2M = 2^20: round 0 runs k_eval_quad_stream (lazy sums, the zero high halves are not read) and the first bind
    k_bind_eval_quad_stream_sparse — the kernels of the sha256_spartan 2 KiB inner sum-check, here against the oracle."""
    rng = np.random.default_rng(SEED + 51)
    rounds, M, extra = 20, 1 << 19, 257
    A, B = rand_table(rng, 2 * M), rand_table(rng, 2 * M)
    A[M + extra :] = 0
    B[M + extra :] = 0
    claim = np.zeros(4, dtype=np.uint64)
    olib().orc_field_dot(0, p64(A), p64(B), ctypes.c_size_t(2 * M), p64(claim))
    want = oracle_quad(claim, rounds, A, (M, extra), B, (M, extra))
    tr = hip.Transcript(ctx, b"sq")
    got = hip.sumcheck_quad(ctx, claim, rounds, hip.Table.from_host(ctx, A, M, extra), hip.Table.from_host(ctx, B, M, extra), tr)
    for g, w in zip(got, want):
        assert (g == w).all()


def test_bind_kernel_reports_algorithmic_bytes(ctx):
    rng = np.random.default_rng(SEED + 60)
    n = 1 << 14
    t = hip.Table.from_host(ctx, rand_table(rng, n))
    ctx.reset_stats(True)
    t.bind_top(rand_table(rng, 1)[0])
    ms, launches, nbytes = ctx.kernel_stats("bind")
    ctx.reset_stats(False)
    assert launches == 1 and nbytes == 48 * n and ms > 0


def test_transcript_prepared_absorb_equals_plain_absorb(ctx):
    """sp_transcript_preabsorb + sp_transcript_absorb_prepared (the 64 KiB comm_W absorb hashed off the critical path) give the very same
    challenges as absorb(); installing a prepared state into a transcript that has absorbed something since its last squeeze is refused."""
    rng = np.random.default_rng(SEED + 9)
    data = rng.integers(0, 256, size=70001, dtype=np.uint8).tobytes()
    a, b = hip.Transcript(ctx, b"t"), hip.Transcript(ctx, b"t")
    a.absorb(b"poly_com", data)
    b.absorb_prepared(b"poly_com", data)
    assert (a.squeeze(b"r") == b.squeeze(b"r")).all()
    # after a squeeze the running hasher is fresh again
    a.absorb(b"x", b"")
    b.absorb_prepared(b"x", b"")
    a.dom_sep(b"sep")
    b.dom_sep(b"sep")
    assert (a.squeeze(b"r") == b.squeeze(b"r")).all()
    b.absorb(b"y", b"123")
    with pytest.raises(hip.SpartanHipError):
        b.absorb_prepared(b"poly_com", data)
    a.absorb(b"y", b"123")
    assert (a.squeeze(b"r") == b.squeeze(b"r")).all()  # the refused call changed nothing


def test_sumchecks_randomised_shapes_stay_bit_exact(ctx):
    """Seeded sweep over the regimes of the round loop — plain launches, launches issued ahead of their challenge, per-block host slots, the
    multi-block and the single-block resident tail, the tau = 0 fallback, zero structure — with dishonest claims (parity, not soundness,
    is what is checked): every (ell, zero structure, tau pattern) must reproduce the oracle's polynomials, challenges and final claims, and
    reuse of one context across all of them must not leak sequence numbers or mailbox state from one sum-check into the next."""
    rng = np.random.default_rng(SEED + 4242)
    for case in range(24):
        ell = int(rng.integers(1, 18))
        n = 1 << ell
        A, B, C = rand_table(rng, n), rand_table(rng, n), rand_table(rng, n)
        taus = rand_table(rng, ell)
        for j in range(ell):
            if rng.random() < 0.08:
                taus[j] = 0  # three-sum fallback round in the middle of rounds launched ahead / inside the tail
        claim = rand_table(rng, 1)[0]
        want_polys, want_r, want_fin, _ = oracle_cubic(claim, taus, A, B, C)
        tr = hip.Transcript(ctx, b"sc")
        polys, r, fin = hip.sumcheck_cubic3(ctx, claim, taus, *(hip.Table.from_host(ctx, x) for x in (A, B, C)), tr)
        assert (polys == want_polys).all() and (r == want_r).all() and (fin == want_fin).all(), f"cubic case {case}: ell={ell}"
        # quadratic, with a random zero structure of the kind bind_poly_var_top tracks (lo_eff / hi_eff)
        half = n // 2
        lo = int(rng.integers(0, half + 1)) if rng.random() < 0.6 else half
        hi = int(rng.integers(0, half + 1)) if rng.random() < 0.6 else half
        A2, B2 = A.copy(), B.copy()
        for t in (A2, B2):
            t[lo:half] = 0
            t[half + hi :] = 0
        qclaim = rand_table(rng, 1)[0]
        want = oracle_quad(qclaim, ell, A2, (lo, hi), B2, (lo, hi))
        trq = hip.Transcript(ctx, b"sq")
        got = hip.sumcheck_quad(ctx, qclaim, ell, hip.Table.from_host(ctx, A2, lo, hi), hip.Table.from_host(ctx, B2, lo, hi), trq)
        for g, w in zip(got, want):
            assert (g == w).all(), f"quad case {case}: ell={ell} lo={lo} hi={hi}"
