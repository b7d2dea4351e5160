"""The sum-check provers on HOST tables (sp_sumcheck_cubic3_host, sp_sumcheck_quad_host: every round on the calling thread and the library's polling host
threads - the relaxed-Spartan sum-checks over the ZK verifier circuit's instance) against the CPU oracle, without a GPU: bit-exact round polynomials,
challenges, final claims and transcript state, for honest and dishonest claims, tau = 0 rounds (the three-sum fallback) and sizes on both sides of the
point where the rounds are spread over the walkers."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import lib as olib, p64
from spartan2_amd import hip

SEED = 0xDEADBEEF


def rand_table(rng, n):
    return ol.random_field_array(rng, n)


def oracle_cubic(claim, taus, A, B, C):
    ell = len(taus)
    tr = ol.Transcript(b"sc")
    polys = np.zeros((ell, 3, 4), dtype=np.uint64)
    r = np.zeros((ell, 4), dtype=np.uint64)
    fin = np.zeros((3, 4), dtype=np.uint64)
    a, b, c = A.copy(), B.copy(), C.copy()
    assert olib().orc_sumcheck_cubic3(p64(claim), p64(taus), ctypes.c_size_t(ell), p64(a), p64(b), p64(c), tr.h, p64(polys), p64(r), p64(fin)) == 0
    return polys, r, fin, tr


@pytest.mark.parametrize("ell", [1, 2, 3, 5, 8, 9, 11])
@pytest.mark.parametrize("kind", ["honest", "dishonest", "tau_zero"])
def test_cubic_on_host_tables_matches_the_oracle(ell, kind):
    rng = np.random.default_rng(SEED + 9100 + 7 * ell + len(kind))
    n = 1 << ell
    A, B = rand_table(rng, n), rand_table(rng, n)
    C = np.zeros_like(A)
    for i in range(n):
        olib().orc_field_binop(0, 2, p64(A[i]), p64(B[i]), p64(C[i]))
    taus = rand_table(rng, ell)
    claim = np.zeros(4, dtype=np.uint64)
    if kind == "dishonest":
        C[rng.integers(0, n)] = rand_table(rng, 1)[0]  # the zero-check no longer holds: derive_from_claim follows the claim, not the sums
        claim = rand_table(rng, 1)[0]
    if kind == "tau_zero":
        taus[0] = 0
        if ell > 2:
            taus[ell - 2] = 0
    want_polys, want_r, want_fin, otr = oracle_cubic(claim, taus, A, B, C)
    tr = hip.Transcript(None, b"sc")
    a, b, c = A.copy(), B.copy(), C.copy()
    polys = np.zeros((ell, 3, 4), dtype=np.uint64)
    r = np.zeros((ell, 4), dtype=np.uint64)
    fin = np.zeros((3, 4), dtype=np.uint64)
    rc = hip.lib().sp_sumcheck_cubic3_host(None, hip.p64(np.ascontiguousarray(claim)), hip.p64(np.ascontiguousarray(taus)), ctypes.c_size_t(ell), hip.p64(a), hip.p64(b),
                                           hip.p64(c), tr.h, hip.p64(polys), hip.p64(r), hip.p64(fin))
    assert rc == 0, hip.lib().sp_last_error()
    assert (r == want_r).all() and (polys == want_polys).all() and (fin == want_fin).all()
    assert (a[0] == want_fin[0]).all() and (b[0] == want_fin[1]).all() and (c[0] == want_fin[2]).all()  # bound in place
    assert (tr.squeeze(b"after") == otr.squeeze(b"after", fid=0)).all()


@pytest.mark.parametrize("rounds", [1, 2, 5, 7, 12])
def test_quad_on_host_tables_matches_the_oracle(rounds):
    rng = np.random.default_rng(SEED + 9300 + rounds)
    n = 1 << rounds
    A, B = rand_table(rng, n), rand_table(rng, n)
    for honest in (True, False):
        claim = np.zeros(4, dtype=np.uint64)
        olib().orc_field_dot(0, p64(A), p64(B), ctypes.c_size_t(n), p64(claim))
        if not honest:
            claim = rand_table(rng, 1)[0]
        otr = ol.Transcript(b"sq")
        want_polys = np.zeros((rounds, 2, 4), dtype=np.uint64)
        want_r = np.zeros((rounds, 4), dtype=np.uint64)
        want_fin = np.zeros((2, 4), dtype=np.uint64)
        a, b = A.copy(), B.copy()
        full = ctypes.c_size_t(hip.SIZE_MAX)
        assert olib().orc_sumcheck_quad(p64(np.ascontiguousarray(claim)), ctypes.c_size_t(rounds), p64(a), full, full, p64(b), full, full, otr.h, p64(want_polys),
                                        p64(want_r), p64(want_fin)) == 0
        tr = hip.Transcript(None, b"sq")
        a, b = A.copy(), B.copy()
        polys = np.zeros((rounds, 2, 4), dtype=np.uint64)
        r = np.zeros((rounds, 4), dtype=np.uint64)
        fin = np.zeros((2, 4), dtype=np.uint64)
        rc = hip.lib().sp_sumcheck_quad_host(None, hip.p64(np.ascontiguousarray(claim)), ctypes.c_size_t(rounds), hip.p64(a), hip.p64(b), tr.h, hip.p64(polys), hip.p64(r),
                                             hip.p64(fin))
        assert rc == 0, hip.lib().sp_last_error()
        assert (r == want_r).all() and (polys == want_polys).all() and (fin == want_fin).all()
        assert (tr.squeeze(b"after") == otr.squeeze(b"after", fid=0)).all()


def test_null_arguments_are_refused():
    L = hip.lib()
    assert L.sp_sumcheck_cubic3_host(None, None, None, ctypes.c_size_t(3), None, None, None, None, None, None, None) != 0
    assert L.sp_sumcheck_quad_host(None, None, ctypes.c_size_t(0), None, None, None, None, None, None) != 0
