"""GPU parity for the NeutronNova rows of SURVEY 8(a) that are built at kernel level: witness folding (a22), shared-weights MSM /
fold_commitments (a16, a20) and the outer-pow cubic evaluation (a6) — C ABI vs oracle/neutronnova.hpp, bit-exact."""
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


def test_weights_from_r(ctx):
    rng = np.random.default_rng(SEED)
    for ell, n in ((0, 1), (1, 2), (3, 7), (5, 32), (8, 256)):
        r = ol.random_field_array(rng, ell) if ell else np.zeros((0, 4), dtype=np.uint64)
        want = np.zeros((n, 4), dtype=np.uint64)
        olib().orc_weights_from_r(p64(r) if ell else None, ctypes.c_size_t(ell), ctypes.c_size_t(n), p64(want))
        assert (hip.weights_from_r(r, n) == want).all()


@pytest.mark.parametrize("n_inst,dim,bits", [(2, 100, False), (7, 5000, True), (32, 4096, True), (5, 3000, False)])
def test_fold_tables_matches_fold_multiple(ctx, n_inst, dim, bits):
    rng = np.random.default_rng(SEED + n_inst)
    if bits:  # SHA-like witnesses: zeros, ones and a few larger values exercise all three branches of the fast path (mod.rs:615-631)
        Ws = np.zeros((n_inst, dim, 4), dtype=np.uint64)
        pick = rng.integers(0, 10, size=(n_inst, dim))
        Ws[pick >= 5] = to_mont(1)
        big = pick == 9
        Ws[big] = ol.random_field_array(rng, int(big.sum()))
    else:
        Ws = ol.random_field_array(rng, n_inst * dim).reshape(n_inst, dim, 4)
    r_bs = ol.random_field_array(rng, max(1, int(np.ceil(np.log2(n_inst)))))
    w = hip.weights_from_r(r_bs, n_inst)
    want = np.zeros((dim, 4), dtype=np.uint64)
    olib().orc_fold_witnesses(p64(w), p64(np.ascontiguousarray(Ws.reshape(-1, 4))), ctypes.c_size_t(n_inst), ctypes.c_size_t(dim), p64(want))
    tabs = [hip.Table.from_host(ctx, Ws[i]) for i in range(n_inst)]
    out = hip.Table.zeros(ctx, dim)
    hip.fold_tables(ctx, tabs, w, dim, out)
    assert (out.read() == want).all()


@pytest.mark.parametrize("n,rows", [(2, 3), (7, 5), (32, 16), (64, 4), (16, 80), (5, 130)])  # rows >= 64: the throughput kernels
def test_msm_shared_weights(ctx, n, rows):
    # msm.rs:228-356; also what fold_commitments computes per Hyrax row (hyrax_pc.rs:775-790)
    rng = np.random.default_rng(SEED + 100 + n)
    gens = np.zeros((rows * n, 8), dtype=np.uint64)
    olib().orc_from_label(b"fold", ctypes.c_size_t(rows * n), p64(gens))
    bases = gens.reshape(rows, n, 8)
    w = ol.random_field_array(rng, n)
    w[0] = to_mont(1)  # the boolean-weight peel (msm.rs:254-256)
    if n > 3:
        w[3] = 0
    want = np.zeros((rows, 8), dtype=np.uint64)
    olib().orc_msm_shared_weights(p64(w), ctypes.c_size_t(n), p64(np.ascontiguousarray(bases.reshape(-1))), ctypes.c_size_t(rows), p64(want))
    assert (hip.msm_shared_weights(ctx, w, bases) == want).all()


@pytest.mark.parametrize("ell,left_bits", [(4, 1), (8, 3), (12, 5), (12, 6), (3, 3), (5, 6)])
def test_eval_cubic_outer_pow(ctx, ell, left_bits):
    # src/sumcheck.rs:366-498; the last two cases take the 4-table fallback (len < left, :378-385)
    rng = np.random.default_rng(SEED + 200 + ell)
    n = 1 << ell
    length = n // 2
    left = 1 << left_bits
    A, B, C = (ol.random_field_array(rng, n) for _ in range(3))
    if length < left:
        pl = ol.random_field_array(rng, n)  # the pow table as a 4th bound table
        pr = np.zeros((0, 4), dtype=np.uint64)
        want = np.zeros((3, 4), dtype=np.uint64)
        olib().orc_eval_cubic_outer_pow(p64(pl), ctypes.c_size_t(n), None, ctypes.c_size_t(0), p64(A), p64(B), p64(C), ctypes.c_size_t(n), p64(want))
        got = hip.eval_cubic_outer_pow(ctx, hip.Table.from_host(ctx, pl), None, *(hip.Table.from_host(ctx, x) for x in (A, B, C)))
    else:
        right = length // left
        pl = ol.random_field_array(rng, left)
        pr = ol.random_field_array(rng, 2 * right)
        want = np.zeros((3, 4), dtype=np.uint64)
        olib().orc_eval_cubic_outer_pow(p64(pl), ctypes.c_size_t(left), p64(pr), ctypes.c_size_t(2 * right), p64(A), p64(B), p64(C), ctypes.c_size_t(n), p64(want))
        got = hip.eval_cubic_outer_pow(ctx, hip.Table.from_host(ctx, pl), hip.Table.from_host(ctx, pr), *(hip.Table.from_host(ctx, x) for x in (A, B, C)))
    assert (got == want).all()
