"""GPU parity for the group side: MSM variants, Hyrax row commitments, fixed-base multiples, the row-matrix product —
through the C ABI vs the CPU oracle, bit-exact on canonical affine coordinates. Mirrors src/provider/msm.rs:878-934
(naive vs msm, msm vs msm_small at nine bit widths)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import lib as olib, p64, to_mont
from spartan2_amd import hip

pytestmark = pytest.mark.gpu
SEED = 0xDEADBEEF


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def gens():
    g = np.zeros((2049, 8), dtype=np.uint64)
    olib().orc_from_label(b"ck", ctypes.c_size_t(2049), p64(g))
    return g


def oracle_msm(scalars, bases):
    out = np.zeros(8, dtype=np.uint64)
    olib().orc_msm(p64(scalars), p64(bases), ctypes.c_size_t(len(scalars)), ctypes.c_size_t(1), p64(out))
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 8, 31, 33, 100, 1000, 2048])
def test_msm_matches_oracle(ctx, gens, n):
    rng = np.random.default_rng(SEED + n)
    scalars = ol.random_field_array(rng, n)
    if n > 8:
        scalars[1] = to_mont(1)  # scalar == 1 peel (msm.rs:93-95)
        scalars[2] = 0
        scalars[3] = to_mont(ol.MODULI[0] - 1)  # all-ones-ish digits exercise the carry window (msm.rs:137-145)
    bases = np.ascontiguousarray(gens[:n])
    assert (hip.msm(ctx, scalars, bases) == oracle_msm(scalars, bases)).all()


def test_msm_naive_sum_small_n(ctx, gens):
    # naive sum_i s_i * P_i, n = 8 (msm.rs:878-901)
    rng = np.random.default_rng(SEED)
    scalars = ol.random_field_array(rng, 8)
    bases = np.ascontiguousarray(gens[:8])
    naive = np.zeros(8, dtype=np.uint64)
    olib().orc_msm_naive(p64(scalars), p64(bases), ctypes.c_size_t(8), p64(naive))
    assert (hip.msm(ctx, scalars, bases) == naive).all()


def test_msm_with_repeated_and_opposite_bases(ctx, gens):
    """P + P (doubling inside a bucket) and P + (-P) (identity) must be handled exactly (vartime add special cases)."""
    rng = np.random.default_rng(SEED + 5)
    n = 64
    bases = np.ascontiguousarray(gens[:n]).copy()
    bases[1] = bases[0]
    neg = bases[2].copy()
    y = ol.from_mont(neg[4:], 1)
    neg[4:] = ol.to_mont((-y) % ol.MODULI[1], 1)
    bases[3] = neg
    scalars = ol.random_field_array(rng, n)
    scalars[1] = scalars[0]
    scalars[3] = scalars[2]
    assert (hip.msm(ctx, scalars, bases) == oracle_msm(scalars, bases)).all()


@pytest.mark.parametrize("bits", [1, 4, 8, 10, 16, 20, 32, 40, 64])
def test_msm_small_matches_full_msm(ctx, gens, bits):
    # msm.rs:903-934
    rng = np.random.default_rng(SEED + bits)
    n = 300
    small = rng.integers(0, 2**bits if bits < 64 else 2**63, size=n, dtype=np.uint64)
    if bits == 64:
        small = small * np.uint64(2) + np.uint64(1)
    bases = np.ascontiguousarray(gens[:n])
    want = np.zeros(8, dtype=np.uint64)
    olib().orc_msm_small(p64(small), p64(bases), ctypes.c_size_t(n), p64(want))
    assert (hip.msm_small(ctx, small, bases) == want).all()
    sc = ol.mont_array([int(v) for v in small])
    assert (hip.msm(ctx, sc, bases) == want).all()


@pytest.fixture(scope="module")
def key(ctx, gens):
    return hip.CommitmentKey(ctx, gens[:2048], gens[2048])


def oracle_key():
    return ctypes.c_void_p(olib().orc_hyrax_setup(b"ck", ctypes.c_size_t(2048)))


def test_fixed_base_mul_h(ctx, key, gens):
    rng = np.random.default_rng(SEED + 100)
    ks = ol.random_field_array(rng, 70)
    ks[0] = 0
    ks[1] = to_mont(1)
    ks[2] = to_mont(255)
    ks[3] = to_mont(256)
    want = np.zeros((70, 8), dtype=np.uint64)
    olib().orc_fixed_base_mul(p64(np.ascontiguousarray(gens[2048])), p64(ks), ctypes.c_size_t(70), p64(want))
    assert (key.fixed_base_mul_h(ks) == want).all()


@pytest.mark.parametrize("n", [70, 600])  # the mapped-page form (<= 128 scalars) and the copy form
def test_fixed_base_mul_h_async_one_job_at_a_time(ctx, key, gens, n):
    """sp_fixed_base_mul_h_begin / _finish: the result is the synchronous call's, and a second begin before the first finish is refused (the jobs share
    the context's landing area: ADVICE r2 — the first finish used to return the second job's points)."""
    rng = np.random.default_rng(SEED + 101 + n)
    k1, k2 = ol.random_field_array(rng, n), ol.random_field_array(rng, n)
    want1, want2 = key.fixed_base_mul_h(k1), key.fixed_base_mul_h(k2)
    job = key.fixed_base_mul_h_begin(k1)
    with pytest.raises(hip.SpartanHipError) as e:
        key.fixed_base_mul_h_begin(k2)
    assert "rc=-1" in str(e.value)
    assert (key.fixed_base_mul_h_finish(job) == want1).all()
    job = key.fixed_base_mul_h_begin(k2)  # the lane is free again
    assert (key.fixed_base_mul_h_finish(job) == want2).all()


@pytest.mark.parametrize("kind", ["bits", "small", "full", "mixed_rows"])
def test_hyrax_commit_matches_oracle(ctx, key, kind):
    # PCS::commit (hyrax_pc.rs:207-303): zero rows, trailing zeros, binary / small / full scalar rows
    rng = np.random.default_rng(SEED + 200)
    n = 2048 * 3 + 100  # ragged last row
    v = np.zeros((n, 4), dtype=np.uint64)
    one = to_mont(1)
    if kind == "bits":
        bits = rng.integers(0, 2, size=n)
        v[bits == 1] = one
        v[2048:4096] = 0  # an all-zero row
        v[5000:6144] = 0  # trailing zeros in row 2
    elif kind == "small":
        vals = rng.integers(0, 1 << 20, size=n)
        v[:] = ol.mont_array([int(x) for x in vals])
    elif kind == "full":
        v[:] = ol.random_field_array(rng, n)
    else:
        bits = rng.integers(0, 2, size=2048)
        v[:2048][bits == 1] = one
        v[2048:4096] = ol.mont_array([int(x) for x in rng.integers(0, 1 << 9, size=2048)])
        v[4096:6144] = ol.random_field_array(rng, 2048)
    rows = (n + 2047) // 2048
    blinds = ol.random_field_array(rng, rows)
    ok = oracle_key()
    want = np.zeros((rows, 8), dtype=np.uint64)
    assert olib().orc_hyrax_commit(ok, p64(v), ctypes.c_size_t(n), p64(blinds), 1 if kind in ("bits", "small") else 0, p64(want)) == 0
    olib().orc_hyrax_free(ok)
    t = hip.Table.from_host(ctx, np.concatenate([np.zeros((7, 4), dtype=np.uint64), v]))  # commit at an offset inside a table
    got = key.commit(t, 7, n, blinds)
    assert (got == want).all()


def test_hyrax_commit_many_full_rows_take_the_batched_path(ctx, key):
    """20 full-scalar rows + a ragged narrow tail: the row-batched digit path (one lane per (row, window, bucket)) against the oracle."""
    rng = np.random.default_rng(SEED + 300)
    n = 2048 * 21 + 5
    v = ol.random_field_array(rng, n)
    v[2048 * 20 :] = ol.mont_array([int(x) for x in rng.integers(0, 1 << 40, size=n - 2048 * 20)])  # last two rows: 64-bit values
    v[2048 * 3 : 2048 * 4] = 0  # a zero row in the middle
    rows = (n + 2047) // 2048
    blinds = ol.random_field_array(rng, rows)
    ok = oracle_key()
    want = np.zeros((rows, 8), dtype=np.uint64)
    assert olib().orc_hyrax_commit(ok, p64(v), ctypes.c_size_t(n), p64(blinds), 0, p64(want)) == 0
    olib().orc_hyrax_free(ok)
    got = key.commit(hip.Table.from_host(ctx, v), 0, n, blinds)
    assert (got == want).all()


def test_rowmat_vec(ctx):
    # bind_with_delayed (hyrax_pc.rs:38-54)
    rng = np.random.default_rng(SEED + 300)
    for rows, cols in ((1, 64), (8, 2048), (37, 256), (128, 64), (512, 2048), (300, 72), (129, 8)):  # rows >= 128: the one-launch streaming kernel
        poly = ol.random_field_array(rng, rows * cols)
        L = ol.random_field_array(rng, rows)
        want = np.zeros((cols, 4), dtype=np.uint64)
        olib().orc_rowmat_vec(p64(poly), p64(L), ctypes.c_size_t(rows), ctypes.c_size_t(cols), p64(want))
        got = hip.rowmat_vec(ctx, hip.Table.from_host(ctx, poly), rows, cols, L)
        assert (got == want).all()


def test_rowmat_vec_eq_job_and_its_scaled_finish(ctx):
    """sp_rowmat_vec_eq_begin[_with] / _finish[_scaled]: LZ = eq(r, .)^T poly against the oracle's bind_with_delayed over the oracle's eq table, and
    z_vec = scale * LZ + addend (ipa.rs:160-163) against Python integers; twice on one context (the arrival flags carry a sequence number), and a job
    begun without an addend refuses the scaled finish."""
    rng = np.random.default_rng(SEED + 310)
    for ell, cols in ((9, 2048), (4, 256), (9, 2048), (11, 300)):
        rows = 1 << ell
        poly = ol.random_field_array(rng, rows * cols)
        r = ol.random_field_array(rng, ell)
        L = np.zeros((rows, 4), dtype=np.uint64)
        assert olib().orc_eq_evals(p64(r), ctypes.c_size_t(ell), p64(L)) == 0
        want = np.zeros((cols, 4), dtype=np.uint64)
        olib().orc_rowmat_vec(p64(poly), p64(L), ctypes.c_size_t(rows), ctypes.c_size_t(cols), p64(want))
        t = hip.Table.from_host(ctx, poly)
        assert (hip.rowmat_vec_eq(ctx, t, r, cols) == want).all()
        d = ol.random_field_array(rng, cols)
        scale = ol.random_field_array(rng, 1)
        got = hip.rowmat_vec_eq(ctx, t, r, cols, addend=d, scale=scale)
        lz, dv, sc = ol.ints_of(want), ol.ints_of(d), ol.ints_of(scale)[0]
        mod = ol.MODULI[0]
        assert ol.ints_of(got) == [(sc * a + b) % mod for a, b in zip(lz, dv)]
        # the plain finish of a job that carried an addend still returns LZ
        assert (hip.rowmat_vec_eq(ctx, t, r, cols, addend=d) == want).all()
    with pytest.raises(hip.SpartanHipError):
        hip.rowmat_vec_eq(ctx, t, r, cols, scale=scale)


@pytest.mark.parametrize("npt,key_tables", [(11, "1"), (13, "1"), (16, "1"), (16, "0"), (9, "1"), (20, "1")])
def test_hyrax_prove_is_the_oracles_pcs_prove(ctx, key, gens, npt, key_tables, monkeypatch):
    """sp_hyrax_prove == HyraxPCS::prove + InnerProductArgumentLinear::prove (hyrax_pc.rs:387-478, ipa.rs:125-170) of the oracle on the same commitment,
    polynomial, point, blinds and randomness: every output word and the transcript state afterwards. npt = 11: one row; 9: a polynomial narrower than
    the key; 16 with SPARTAN_KEY_TABLES=0: the bucket-MSM fallback; 20 = 512 rows, BASELINE config 2's opening (one-launch rowmat kernel)."""
    monkeypatch.setenv("SPARTAN_KEY_TABLES", key_tables)
    rng = np.random.default_rng(SEED + 900 + npt)
    n = 1 << npt
    rows = max(1, n // 2048)
    cols = n // rows
    poly = ol.random_field_array(rng, n)
    blinds = ol.random_field_array(rng, rows)
    point = ol.random_field_array(rng, npt)
    okey = oracle_key()
    okey_s = ctypes.c_void_p(olib().orc_hyrax_setup(b"ck_s", ctypes.c_size_t(1)))
    g_s = np.zeros((2, 8), dtype=np.uint64)
    olib().orc_from_label(b"ck_s", ctypes.c_size_t(2), p64(g_s))
    key_s = hip.CommitmentKey(ctx, g_s[:1], g_s[1])
    comm = np.zeros((rows, 8), dtype=np.uint64)
    assert olib().orc_hyrax_commit(okey, p64(poly), ctypes.c_size_t(n), p64(blinds), 0, p64(comm)) == 0
    ev, b_ev = ol.random_field_array(rng, 1), ol.random_field_array(rng, 1)  # (the prover does not check the claimed evaluation)
    comm_eval = np.zeros((1, 8), dtype=np.uint64)
    assert olib().orc_hyrax_commit(okey_s, p64(ev), ctypes.c_size_t(1), p64(b_ev), 0, p64(comm_eval)) == 0
    tape = ol.make_tape(SEED + npt, cols + 2)
    want = np.zeros(16 + 4 * cols + 8, dtype=np.uint64)
    otr = ctypes.c_void_p(olib().orc_transcript_new(b"pcs"))
    assert olib().orc_transcript_absorb(otr, b"x", b"warm", ctypes.c_size_t(4)) == 0
    assert olib().orc_hyrax_prove(okey, okey_s, otr, p64(comm), ctypes.c_size_t(rows), p64(poly), ctypes.c_size_t(n), p64(blinds), p64(point), ctypes.c_size_t(npt),
                                  p64(comm_eval), p64(b_ev), tape.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_size_t(tape.shape[0]), p64(want)) == 0
    tr = hip.Transcript(ctx, b"pcs")
    tr.absorb(b"x", b"warm")  # (not a fresh hasher: the general path of the helper's hashing)
    got = key.prove(key_s, tr, comm, hip.Table.from_host(ctx, poly), n, blinds, point, comm_eval, b_ev, tape)
    assert (got == want).all()
    o_next = np.zeros(4, dtype=np.uint64)
    assert olib().orc_transcript_squeeze(otr, b"n", 0, p64(o_next)) == 0
    assert (tr.squeeze(b"n") == o_next).all()
    olib().orc_transcript_free(otr)
    olib().orc_hyrax_free(okey)
    olib().orc_hyrax_free(okey_s)


@pytest.mark.parametrize("case", ["match_fresh", "match_warm", "other_blinds", "other_rng", "other_table", "retracted", "replaced", "no_key_tables"])
def test_hyrax_prove_announced_ahead_is_the_same_opening(ctx, key, gens, case, monkeypatch):
    """sp_hyrax_prove_announce changes WHEN the opening's first half is computed, never a word of it: an announcement that matches is consumed (fresh
    transcript: the commitment's hashing too; warm one: delta and the mask vector only), one that differs in blinds / randomness / table is dropped,
    a retracted or replaced one leaves nothing behind, and a key without window tables ignores it. Every case equals the unannounced sp_hyrax_prove,
    which test_hyrax_prove_is_the_oracles_pcs_prove pins to the oracle."""
    if case == "no_key_tables":
        monkeypatch.setenv("SPARTAN_KEY_TABLES", "0")
    npt = 14
    rng = np.random.default_rng(SEED + 950)
    n = 1 << npt
    rows = n // 2048
    cols = n // rows
    poly = ol.random_field_array(rng, n)
    blinds = ol.random_field_array(rng, rows)
    point = ol.random_field_array(rng, npt)
    g_s = np.zeros((2, 8), dtype=np.uint64)
    olib().orc_from_label(b"ck_s", ctypes.c_size_t(2), p64(g_s))
    key_s = hip.CommitmentKey(ctx, g_s[:1], g_s[1])
    table = hip.Table.from_host(ctx, poly)
    comm = key.commit(table, 0, n, blinds, is_small=False)
    ev, b_ev = ol.random_field_array(rng, 1), ol.random_field_array(rng, 1)
    comm_eval = key_s.msm(ev, b_ev[0])
    tape = ol.make_tape(SEED + 77, cols + 2)

    def prove(warm):
        tr = hip.Transcript(ctx, b"pcs")
        if warm:
            tr.absorb(b"x", b"warm")
        out = key.prove(key_s, tr, comm, table, n, blinds, point, comm_eval, b_ev, tape)
        return out, tr.squeeze(b"n")

    warm = case == "match_warm"
    want, want_next = prove(warm)
    other = hip.Table.from_host(ctx, ol.random_field_array(rng, n))
    if case == "other_blinds":
        key.prove_announce(comm, table, n, ol.random_field_array(rng, rows), tape)
    elif case == "other_rng":
        key.prove_announce(comm, table, n, blinds, ol.make_tape(SEED + 78, cols + 2))
    elif case == "other_table":
        key.prove_announce(comm, other, n, blinds, tape)
    elif case == "replaced":
        key.prove_announce(comm, other, n, blinds, tape)
        key.prove_announce(comm, table, n, blinds, tape)
    else:
        key.prove_announce(comm, table, n, blinds, tape)
    if case == "retracted":
        key.prove_retract()
        key.prove_retract()  # nothing announced: a no-op
    got, got_next = prove(warm)
    assert (got == want).all() and (got_next == want_next).all()
    # the announcement was consumed or dropped: the next opening on the context is again the plain one
    again, again_next = prove(warm)
    assert (again == want).all() and (again_next == want_next).all()


def test_hyrax_prove_announced_then_two_sumchecks_of_the_openings_length(ctx, key, gens):
    """An announced opening listens to the challenges of every sp_sumcheck_quad whose round count is the opening's point length + 1. Two such
    sum-checks before sp_hyrax_prove: the row half of the point is frozen when comm_LZ's walk starts, and the partial sums of <R, d> are tied to the
    column challenges they were built from — whichever sum-check's point the caller then opens at, the opening is the unannounced one."""
    npt = 14
    rng = np.random.default_rng(SEED + 951)
    n = 1 << npt
    rows = n // 2048
    cols = n // rows
    poly = ol.random_field_array(rng, n)
    blinds = ol.random_field_array(rng, rows)
    g_s = np.zeros((2, 8), dtype=np.uint64)
    olib().orc_from_label(b"ck_s", ctypes.c_size_t(2), p64(g_s))
    key_s = hip.CommitmentKey(ctx, g_s[:1], g_s[1])
    table = hip.Table.from_host(ctx, poly)
    comm = key.commit(table, 0, n, blinds, is_small=False)
    ev, b_ev = ol.random_field_array(rng, 1), ol.random_field_array(rng, 1)
    comm_eval = key_s.msm(ev, b_ev[0])
    tape = ol.make_tape(SEED + 79, cols + 2)
    claim = ol.random_field_array(rng, 1)[0]

    def sumcheck(seed):  # a quadratic sum-check of npt + 1 rounds on tables of its own: only its challenges matter here
        r2 = np.random.default_rng(seed)
        A = hip.Table.from_host(ctx, ol.random_field_array(r2, 2 * n))
        Bt = hip.Table.from_host(ctx, ol.random_field_array(r2, 2 * n))
        tr = hip.Transcript(ctx, b"sc")
        tr.absorb(b"s", bytes([seed & 255]))
        _, r, _ = hip.sumcheck_quad(ctx, claim, npt + 1, A, Bt, tr)
        A.free()
        Bt.free()
        return np.ascontiguousarray(r[1:])  # round 0 binds the variable in front of the opening's point

    def prove(point):
        tr = hip.Transcript(ctx, b"pcs")
        out = key.prove(key_s, tr, comm, table, n, blinds, point, comm_eval, b_ev, tape)
        return out, tr.squeeze(b"n")

    # the two points, drawn without an announcement on the context (the transcripts make them reproducible)
    p1, p2 = sumcheck(1), sumcheck(2)
    assert not (p1 == p2).all()
    want1, want2 = prove(p1), prove(p2)
    for opened, want in ((p1, want1), (p2, want2)):
        key.prove_announce(comm, table, n, blinds, tape)
        assert (sumcheck(1) == p1).all() and (sumcheck(2) == p2).all()
        got = prove(opened)
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all()


@pytest.mark.parametrize("nfixed", [5, 8])
def test_hyrax_prove_announced_with_row_tables(ctx, key, gens, nfixed):
    """sp_hyrax_prove_announce_tables: with FixedBaseMul tables of the first nfixed commitment rows and of h, and the other rows commitments of zero rows
    (blind * h), comm_LZ is one walk over those tables behind the last row challenge - the same opening, word for word, as the unannounced call's
    <L^T W, ck> + r_LZ h. nfixed = rows: no zero row (h's scalar is zero)."""
    npt = 14
    rng = np.random.default_rng(SEED + 952 + nfixed)
    n = 1 << npt
    rows = n // 2048
    cols = n // rows
    poly = ol.random_field_array(rng, n)
    poly[nfixed * cols :] = 0
    blinds = ol.random_field_array(rng, rows)
    g_s = np.zeros((2, 8), dtype=np.uint64)
    olib().orc_from_label(b"ck_s", ctypes.c_size_t(2), p64(g_s))
    key_s = hip.CommitmentKey(ctx, g_s[:1], g_s[1])
    table = hip.Table.from_host(ctx, poly)
    comm = key.commit(table, 0, n, blinds, is_small=False)
    if nfixed < rows:  # the zero rows' commitments are blind * h
        assert (comm[nfixed:] == key.fixed_base_mul_h(blinds[nfixed:])).all()
    tabs = hip.FixedBaseTables(ctx, np.concatenate([comm[:nfixed], np.ascontiguousarray(gens[2048:2049])]))
    ev, b_ev = ol.random_field_array(rng, 1), ol.random_field_array(rng, 1)
    comm_eval = key_s.msm(ev, b_ev[0])
    tape = ol.make_tape(SEED + 80, cols + 2)
    claim = ol.random_field_array(rng, 1)[0]

    def sumcheck(seed):
        r2 = np.random.default_rng(seed)
        A = hip.Table.from_host(ctx, ol.random_field_array(r2, 2 * n))
        Bt = hip.Table.from_host(ctx, ol.random_field_array(r2, 2 * n))
        tr = hip.Transcript(ctx, b"sc")
        tr.absorb(b"s", bytes([seed & 255]))
        _, r, _ = hip.sumcheck_quad(ctx, claim, npt + 1, A, Bt, tr)
        A.free()
        Bt.free()
        return np.ascontiguousarray(r[1:])

    def prove(point):
        tr = hip.Transcript(ctx, b"pcs")
        out = key.prove(key_s, tr, comm, table, n, blinds, point, comm_eval, b_ev, tape)
        return out, tr.squeeze(b"n")

    p1 = sumcheck(3)
    want = prove(p1)
    for _ in range(2):
        key.prove_announce_tables(comm, table, n, blinds, tape, tabs, nfixed)
        assert (sumcheck(3) == p1).all()
        got = prove(p1)
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all()
    # opened at another point than the sum-check drew: the announcement's walk is dropped, the opening is the plain one
    p2 = sumcheck(4)
    want2 = prove(p2)
    key.prove_announce_tables(comm, table, n, blinds, tape, tabs, nfixed)
    assert (sumcheck(3) == p1).all()
    got2 = prove(p2)
    assert (got2[0] == want2[0]).all() and (got2[1] == want2[1]).all()
    with pytest.raises(hip.SpartanHipError):
        key.prove_announce_tables(comm, table, n, blinds, tape, tabs, nfixed - 1)  # one table per fixed row and one of h
    tabs.close()


def test_msm_ck_with_blind_and_commit_small(ctx, key, gens):
    rng = np.random.default_rng(SEED + 400)
    sc = ol.random_field_array(rng, 2048)
    blind = ol.random_field_array(rng, 1)[0]
    msm_part = oracle_msm(sc, np.ascontiguousarray(gens[:2048]))
    hb = np.zeros((1, 8), dtype=np.uint64)
    olib().orc_fixed_base_mul(p64(np.ascontiguousarray(gens[2048])), p64(blind.reshape(1, 4)), ctypes.c_size_t(1), p64(hb))
    want = np.zeros(8, dtype=np.uint64)
    olib().orc_point_add(p64(msm_part), p64(hb[0]), p64(want))
    assert (key.msm(sc, blind) == want).all()
    assert (key.msm(sc) == msm_part).all()
    # a rank's point range of the same MSM (sp_msm_ck_range_begin): the four quarters add up to the whole; a range outside the key is refused
    parts = np.stack([key.msm_range(sc[512 * q : 512 * (q + 1)], 512 * q) for q in range(4)])
    assert (parts[1] == oracle_msm(sc[512:1024], np.ascontiguousarray(gens[512:1024]))).all()
    assert (hip.point_sum(parts) == msm_part).all()
    with pytest.raises(hip.SpartanHipError):
        key.msm_range(sc[:8], 2045)
    # width-1 key (ck_s of src/spartan.rs:151): value * g + blind * h
    gs = np.zeros((2, 8), dtype=np.uint64)
    olib().orc_from_label(b"ck_s", ctypes.c_size_t(2), p64(gs))
    ks = hip.CommitmentKey(ctx, gs[:1], gs[1])
    val = ol.random_field_array(rng, 1)
    a = np.zeros(8, dtype=np.uint64)
    b = np.zeros(8, dtype=np.uint64)
    olib().orc_point_mul(p64(np.ascontiguousarray(gs[0])), p64(val[0]), p64(a))
    olib().orc_point_mul(p64(np.ascontiguousarray(gs[1])), p64(blind), p64(b))
    olib().orc_point_add(p64(a), p64(b), p64(want))
    assert (ks.commit_small(val, blind) == want).all()
    # the blind's term handed in (sp_hyrax_commit_small_with_term: h * blind computed beforehand), the identity term, more scalars than the host walk takes
    assert (ks.commit_small_with_term(val, ks.fixed_base_mul_h(blind.reshape(1, 4))[0]) == want).all()
    assert (ks.commit_small_with_term(val, np.zeros(8, dtype=np.uint64)) == a).all()
    with pytest.raises(hip.SpartanHipError):
        ks.commit_small_with_term(ol.random_field_array(rng, 7), b)


def test_row_range_commits_and_point_sum_compose(ctx, key, gens):
    """Multi-GPU building blocks on one GPU: committing row ranges separately equals the full commitment, and the
    sum of point-range partial MSMs (sp_point_sum) equals the full MSM."""
    rng = np.random.default_rng(SEED + 500)
    rows = 6
    v = np.zeros((rows * 2048, 4), dtype=np.uint64)
    v[rng.integers(0, 2, size=rows * 2048) == 1] = to_mont(1)
    blinds = ol.random_field_array(rng, rows)
    t = hip.Table.from_host(ctx, v)
    full = key.commit(t, 0, rows * 2048, blinds)
    parts = [key.commit(t, lo * 2048, (hi - lo) * 2048, blinds[lo:hi]) for lo, hi in ((0, 2), (2, 3), (3, 6))]
    assert (np.concatenate(parts) == full).all()
    sc = ol.random_field_array(rng, 900)
    bases = np.ascontiguousarray(gens[:900])
    whole = hip.msm(ctx, sc, bases)
    partials = np.stack([hip.msm(ctx, sc[lo:hi], bases[lo:hi]) for lo, hi in ((0, 113), (113, 450), (450, 900))])
    assert (hip.point_sum(partials) == whole).all()
    assert (hip.point_sum(np.zeros((0, 8), dtype=np.uint64)) == 0).all()


@pytest.mark.parametrize("ell", [0, 1, 3, 7, 10, 11])
def test_msm_eq_weights_matches_oracle(ctx, gens, ell):
    """sp_points_upload + sp_msm_eq_begin + sp_msm_job_finish: sum_i eq(r, i) P_i (comm_LZ as an MSM over the row commitments) equals the
    oracle's MSM with the oracle's eq table as scalars; ell = 11 takes the uploaded-scalars branch, the others the two-half-tables kernel."""
    rng = np.random.default_rng(SEED + 77 + ell)
    n = 1 << ell
    r = ol.random_field_array(rng, max(ell, 1))[:ell]
    w = np.zeros((n, 4), dtype=np.uint64)
    olib().orc_eq_evals(p64(r) if ell else None, ctypes.c_size_t(ell), p64(w))
    pts = np.ascontiguousarray(gens[:n]).copy()
    if n >= 8:
        pts[5] = 0  # the identity among the row commitments
        pts[6] = pts[2]  # a repeated row
    assert (hip.msm_eq(ctx, pts, r) == oracle_msm(w, pts)).all()


def test_msm_eq_rejects_length_mismatch(ctx, gens):
    rng = np.random.default_rng(SEED)
    r = ol.random_field_array(rng, 3)
    with pytest.raises(hip.SpartanHipError):
        hip.msm_eq(ctx, np.ascontiguousarray(gens[:7]), r)


@pytest.mark.parametrize("width", [8, 32, 33, 64])
def test_commit_small_device_form_matches_oracle(ctx, width):
    """sp_hyrax_commit_small on keys of <= 64 bases with more than six non-zero scalars (the device form: per-base table walks in one launch, the
    points added on the host) against the oracle's MSM + blind (hyrax_pc.rs:221-260, msm.rs:727-773).
    Cases: dense, short (zero-padded) vectors, a zero blind, repeated scalars on one call after the other (sequence numbers), all-equal digits."""
    rng = np.random.default_rng(SEED + 4100 + width)
    gs = np.zeros((width + 1, 8), dtype=np.uint64)
    olib().orc_from_label(b"narrow_key_test", ctypes.c_size_t(width + 1), p64(gs))
    k = hip.CommitmentKey(ctx, gs[:width], gs[width])

    def want(sc, blind):
        full = np.zeros((width + 1, 4), dtype=np.uint64)
        full[: len(sc)] = sc
        full[width] = blind
        return oracle_msm(full, np.ascontiguousarray(gs))

    for n in (width, width - 1, 7):
        sc = ol.random_field_array(rng, n)
        blind = ol.random_field_array(rng, 1)[0]
        assert (k.commit_small(sc, blind) == want(sc, blind)).all()
    sc = ol.random_field_array(rng, width)
    sc[::3] = 0
    zero = np.zeros(4, dtype=np.uint64)
    assert (k.commit_small(sc, zero) == want(sc, zero)).all()
    ones = np.tile(to_mont(int.from_bytes(b"\x01" * 31, "little")), (width, 1))  # every window of every scalar hits entry 1 of its table
    blind = ol.random_field_array(rng, 1)[0]
    for _ in range(3):
        assert (k.commit_small(ones, blind) == want(ones, blind)).all()


@pytest.mark.parametrize("width,n", [(32, 512), (32, 32), (32, 100), (8, 70), (64, 576)])
def test_commit_rows_host_equals_the_staged_commit_and_the_oracle(ctx, width, n):
    """sp_hyrax_commit_rows_host (a host vector on a narrow key through ONE launch of the cooperative table walk in mapped memory, the rows' points added by
    the polling host threads) against sp_hyrax_commit on the staged table and the oracle's MSM + blind, row by row: full rows, a ragged last row, zero
    scalars, an all-zero row, repeated calls (sequence numbers), and a vector too long for the mapped form (refused: the caller stages it)."""
    rng = np.random.default_rng(SEED + 4700 + width + n)
    gs = np.zeros((width + 1, 8), dtype=np.uint64)
    olib().orc_from_label(b"narrow_key_test", ctypes.c_size_t(width + 1), p64(gs))
    k = hip.CommitmentKey(ctx, gs[:width], gs[width])
    rows = (n + width - 1) // width
    for trial in range(3):
        v = ol.random_field_array(rng, n)
        v[rng.integers(0, n, size=max(1, n // 7))] = 0
        if rows > 2:
            v[width : 2 * width] = 0  # an all-zero row: its commitment is h * blind
        blinds = ol.random_field_array(rng, rows)
        got = k.commit_rows_host(v, blinds)
        t = hip.Table.from_host(ctx, v)
        assert (got == k.commit(t, 0, n, blinds)).all()
        for r in range(rows):
            full = np.zeros((width + 1, 4), dtype=np.uint64)
            seg = v[r * width : (r + 1) * width]
            full[: len(seg)] = seg
            full[width] = blinds[r]
            assert (got[r] == oracle_msm(full, np.ascontiguousarray(gs))).all(), (trial, r)
    too_many = (640 // (width + 1) + 1) * width
    with pytest.raises(hip.SpartanHipError):
        k.commit_rows_host(ol.random_field_array(rng, too_many), ol.random_field_array(rng, too_many // width))


@pytest.mark.parametrize("width", [32, 8])
def test_commit_split_equals_commit_small_and_the_oracle(ctx, width):
    """sp_hyrax_commit_split_begin / _finish (the round commitments of the ZK verifier circuit: the terms known early and the blind are posted to the host
    table walkers, the round's own scalars follow) against sp_hyrax_commit_small on the assembled row and the oracle's MSM + blind, for every way of
    cutting the row: nothing early but the blind, everything early, a blind handed over as NULL (no blind's term), zero scalars on either side, a dropped
    job, many jobs one after the other (slot reuse), and a column that has no host table (refused, not mis-computed)."""
    rng = np.random.default_rng(SEED + 4300 + width)
    gs = np.zeros((width + 1, 8), dtype=np.uint64)
    olib().orc_from_label(b"narrow_key_test", ctypes.c_size_t(width + 1), p64(gs))
    k = hip.CommitmentKey(ctx, gs[:width], gs[width])
    if hip.lib().sp_walkers() == 0:
        assert not k.commit_split_available()
        pytest.skip("SPARTAN_WALKERS=0: the split form is not offered")
    ncols = min(width, 16)
    assert k.commit_split_available(ncols) and not k.commit_split_available(width + 1)

    def want(row, blind):
        full = np.zeros((width + 1, 4), dtype=np.uint64)
        full[: len(row)] = row
        if blind is not None:
            full[width] = blind
        return oracle_msm(full, np.ascontiguousarray(gs))

    zero = np.zeros(4, dtype=np.uint64)
    for trial in range(24):
        row = ol.random_field_array(rng, ncols)
        if trial % 3 == 1:
            row[rng.integers(0, ncols, size=3)] = 0
        if trial == 5:
            row[:] = 0
        blind = None if trial % 4 == 3 else ol.random_field_array(rng, 1)[0]
        cut = [0, ncols, ncols // 2, 3][trial % 4] if trial < 8 else int(rng.integers(0, ncols + 1))
        cols = np.arange(ncols, dtype=np.uint32)
        if trial % 5 == 4:  # an arbitrary subset early, not a prefix
            perm = rng.permutation(ncols).astype(np.uint32)
            late, early = perm[:cut], perm[cut:]
        else:
            late, early = cols[:cut], cols[cut:]
        got = k.commit_split(early, row[early], blind, late, row[late])
        w = want(row, blind)
        assert (got == w).all(), (trial, cut)
        assert (k.commit_small(row, zero if blind is None else blind) == w).all()
    # a dropped job leaves nothing behind: the next one is served
    row = ol.random_field_array(rng, ncols)
    blind = ol.random_field_array(rng, 1)[0]
    cols = np.arange(ncols, dtype=np.uint32)
    assert k.commit_split(cols[2:], row[2:], blind, cols[:2], row[:2], drop=True) is None
    assert (k.commit_split(cols[2:], row[2:], blind, cols[:2], row[:2]) == want(row, blind)).all()
    if width > 16:  # column 16 has no host table: a non-zero scalar there is refused, a zero one is skipped
        with pytest.raises(hip.SpartanHipError):
            k.commit_split(np.array([16], dtype=np.uint32), row[:1], blind, cols[:1], row[:1])
        z = np.zeros((1, 4), dtype=np.uint64)
        one_col = np.zeros((ncols, 4), dtype=np.uint64)
        one_col[0] = row[0]
        assert (k.commit_split(np.array([16], dtype=np.uint32), z, blind, cols[:1], row[:1]) == want(one_col, blind)).all()


@pytest.mark.parametrize("k,nfixed", [(1, 2), (3, 5), (3, 8), (9, 425), (9, 512), (10, 1022)])
def test_fixed_base_tables_multi_mul_with_the_last_eq_level_on_the_device(ctx, k, nfixed):
    """sp_fbtables_multi_mul_begin_eq: the walk's scalars handed over one level short of eq(r_1..r_k, .) - the kernel's own last level gives the same point
    as sp_fbtables_multi_mul over the full table (odd and even counts of fixed rows, a partial last pair, the largest size below the wide kernel)."""
    rng = np.random.default_rng(SEED + 7800 + nfixed)
    n = nfixed + 1
    pts = np.zeros((n, 8), dtype=np.uint64)
    olib().orc_from_label(b"fbtables_eq_test", ctypes.c_size_t(n), p64(pts))
    t = hip.FixedBaseTables(ctx, pts)
    r = ol.random_field_array(rng, k)
    full = hip.Table.eq(ctx, r).read()  # eq(r_1..r_k, .): 2^k elements, r_1 on the index MSB
    prev = hip.Table.eq(ctx, r[: k - 1]).read() if k > 1 else ol.mont_array([1])
    s01 = ol.random_field_array(rng, 2)
    p = ol.MODULI[0]
    hs = (ol.from_mont(s01[0]) + ol.from_mont(r[k - 1]) * (ol.from_mont(s01[1]) - ol.from_mont(s01[0]))) % p
    sc = np.concatenate([full[:nfixed], ol.mont_array([hs])])
    want = t.multi_mul(sc)
    assert (want == oracle_msm(sc, np.ascontiguousarray(pts))).all()
    for _ in range(2):
        assert (t.multi_mul_eq(prev[: (nfixed + 1) // 2], nfixed, s01, r[k - 1]) == want).all()
    with pytest.raises(hip.SpartanHipError):
        t.multi_mul_eq(prev[: (nfixed + 1) // 2], nfixed - 1, s01, r[k - 1])  # one table per fixed row and one of h
    t.close()


@pytest.mark.parametrize("n", [1, 3, 4, 33, 130, 429, 512, 700, 2048, 2049, 2600])
def test_fixed_base_tables_multi_mul_matches_oracle_msm(ctx, n):
    """sp_fbtables_create + sp_fbtables_multi_mul (FixedBaseMul::precompute / multi_mul over arbitrary points, msm.rs:637-773; the one-launch form comm_LZ
    of the opening uses on the row commitments of a prepared witness) against the oracle's MSM: dense scalars, zeros, repeated calls (sequence numbers),
    eq-table weights as in the call site."""
    rng = np.random.default_rng(SEED + 7700 + n)
    pts = np.zeros((n, 8), dtype=np.uint64)
    olib().orc_from_label(b"fbtables_test", ctypes.c_size_t(n), p64(pts))
    t = hip.FixedBaseTables(ctx, pts)
    for rep in range(3):
        sc = ol.random_field_array(rng, n)
        if rep == 1 and n > 2:
            sc[::2] = 0
        assert (t.multi_mul(sc) == oracle_msm(sc, np.ascontiguousarray(pts))).all()
    with pytest.raises(hip.SpartanHipError):
        t.multi_mul(ol.random_field_array(rng, n + 1))
    t.close()
    if n == 4:
        with pytest.raises(hip.SpartanHipError):
            hip.FixedBaseTables(ctx, np.zeros((4097, 8), dtype=np.uint64))  # more than 1024 blocks of four scalars


@pytest.mark.parametrize("queued", [False, True])
def test_fbtables_every_entry(ctx, queued):
    """FixedBaseMul::precompute (msm.rs:653-689) as the three-launch build of round 6 (doubling ladder, batch-normalised ladder, 32-entry parts normalised
    with one inversion each): EVERY entry (i, j, d - 1) of the tables of a generator-derived point, of the identity and of a second point is the canonical
    affine d * 2^(8j) * P_i - checked against the oracle's scalar multiplication entry by entry for one point and at sampled entries for the others."""
    rng = np.random.default_rng(SEED + 9100)
    pts = np.zeros((3, 8), dtype=np.uint64)
    olib().orc_from_label(b"fbtables_entries", ctypes.c_size_t(3), p64(pts))
    pts[1] = 0  # the identity: every multiple is the identity, written as (0, 0)
    t = hip.FixedBaseTables(ctx, pts, queued=queued)
    assert t.ready(wait=True)
    per = 32 * 255
    got = t.read(0, 3 * per)
    assert (got[per : 2 * per] == 0).all()
    p = ol.MODULI[0]

    def want(i, j, d):
        c = (d << (8 * j)) % p  # (the group order is p: multiples of the top window beyond it wrap)
        return oracle_msm(ol.mont_array([c]), np.ascontiguousarray(pts[i : i + 1]))

    for j in range(32):
        for d in range(1, 256):
            assert (got[j * 255 + d - 1] == want(0, j, d)).all(), (j, d)
    for _ in range(200):
        j, d = int(rng.integers(0, 32)), int(rng.integers(1, 256))
        assert (got[2 * per + j * 255 + d - 1] == want(2, j, d)).all(), (j, d)
    t.close()


def test_table_write_u64_and_bits(ctx):
    """sp_table_write_u64 / sp_table_write_bits (the is_small witness upload, src/bellpepper/r1cs.rs:303-409 + hyrax_pc.rs:266-292): machine words and packed
    bits become the same Montgomery-form elements sp_table_write would have been given, at an offset, leaving the rest of the table alone."""
    rng = np.random.default_rng(SEED + 9200)
    n = 5000
    vals = rng.integers(0, 2, size=n, dtype=np.uint64)
    vals[::7] = rng.integers(0, 2**63, size=len(vals[::7]), dtype=np.uint64) * 2 + 1
    vals[3] = 2**64 - 1
    vals[4] = 0
    t = hip.Table.zeros(ctx, 8192)
    filler = ol.random_field_array(rng, 8192)
    t.write(0, filler)
    t.write_u64(100, vals)
    got = t.read()
    assert (got[:100] == filler[:100]).all() and (got[100 + n :] == filler[100 + n :]).all()
    assert (got[100 : 100 + n] == ol.mont_array([int(v) for v in vals])).all()
    bits = rng.integers(0, 2, size=3001, dtype=np.uint8)
    packed = np.packbits(bits, bitorder="little")
    t.write_bits(7, packed, len(bits))
    got = t.read()
    assert (got[7 : 7 + len(bits)] == ol.mont_array([int(b) for b in bits])).all()
    assert (got[:7] == filler[:7]).all() and (got[7 + len(bits) : 100] == filler[7 + len(bits) : 100]).all()
    with pytest.raises(hip.SpartanHipError):
        t.write_u64(8000, vals)  # range exceeds the table
