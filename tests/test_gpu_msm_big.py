"""GPU parity of the general multi-block Pippenger (spartan2_amd/csrc/kernels_pippenger.hpp, sp_msm / sp_msm_small_u64 from 4096 points up and
sp_msm_points at any size): DlogGroupExt::vartime_multiscalar_mul -> msm (src/provider/msm.rs:59-222) and vartime_multiscalar_mul_small (:367-409)
against the oracle's restatement at n = 2^12, 2^16 and 2^20, every window width the library uses, the structure the reference special-cases (zeros, ones,
order - 1, repeated and opposite bases), and the size-independent identity MSM(s, t_i * H) = (<s, t>) * H at full size."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import lib as olib, p64
from spartan2_amd import hip

pytestmark = pytest.mark.gpu
P = ol.MODULI[0]


@pytest.fixture(scope="module")
def env():
    ctx = hip.Context(0)
    L = olib()
    okey = ctypes.c_void_p(L.orc_hyrax_setup(b"ck", ctypes.c_size_t(4)))
    ck_aff = np.zeros((4, 8), dtype=np.uint64)
    h_aff = np.zeros(8, dtype=np.uint64)
    L.orc_hyrax_key_export(okey, p64(ck_aff), p64(h_aff))
    key = hip.CommitmentKey(ctx, ck_aff, h_aff)
    rng = np.random.default_rng(0xB16)
    n = 1 << 20
    t = ol.random_field_array(rng, n)  # the points' discrete logs to the base H
    pts = np.zeros((n, 8), dtype=np.uint64)
    for lo in range(0, n, 1 << 16):  # inputs only: t_i * H by the fixed-base table walk (itself pinned by tests/test_gpu_group.py)
        pts[lo:lo + (1 << 16)] = key.fixed_base_mul_h(t[lo:lo + (1 << 16)])
    yield ctx, key, t, pts, h_aff
    L.orc_hyrax_free(okey)
    ctx.close()


def oracle_msm(scalars, bases, threads=0):
    out = np.zeros(8, dtype=np.uint64)
    assert olib().orc_msm(p64(np.ascontiguousarray(scalars)), p64(np.ascontiguousarray(bases)), ctypes.c_size_t(scalars.shape[0]), ctypes.c_size_t(threads), p64(out)) == 0
    return out


def structured_scalars(rng, n):
    s = ol.random_field_array(rng, n)
    s[1] = ol.to_mont(1)  # scalar == 1 (peeled by the reference, msm.rs:93-95)
    s[2] = 0
    s[3] = ol.to_mont(P - 1)  # folds to 1 with the point negated
    s[4] = ol.to_mont((P + 1) // 2)  # the fold's boundary
    s[5] = ol.to_mont((1 << 255) - 19)
    s[n // 2:n // 2 + 64] = 0
    s[n // 3:n // 3 + 50] = ol.mont_array([int(x) for x in rng.integers(0, 1 << 40, size=50)])
    return s


@pytest.mark.parametrize("log_n,window", [(12, 0), (12, 8), (12, 10), (12, 12), (13, 13), (13, 14), (16, 0), (16, 10), (16, 14)])
def test_msm_points_matches_oracle(env, log_n, window):
    ctx, key, t, pts, h = env
    n = 1 << log_n
    rng = np.random.default_rng(100 * log_n + window)
    s = structured_scalars(rng, n)
    bases = pts[5:5 + n].copy()
    bases[8] = bases[7]  # P + P inside a bucket when the digits meet, P - P when they are opposite
    s[8] = s[7]
    bases[10] = bases[9]
    s[10] = ol.to_mont((P - ol.from_mont(s[9])) % P)
    want = oracle_msm(s, bases)
    got = hip.msm_points(ctx, hip.Table.from_host(ctx, s), 0, n, hip.Points(ctx, bases), 0, window)
    assert (got == want).all()
    if window == 0:  # the host-buffer entry point takes the same path from 4096 points up
        assert (hip.msm(ctx, s, bases) == want).all()


@pytest.mark.parametrize("n", [4096, 5000, 70001])
def test_msm_ragged_sizes_and_ranges(env, n):
    ctx, key, t, pts, h = env
    rng = np.random.default_rng(n)
    s = ol.random_field_array(rng, n)
    want = oracle_msm(s, pts[:n])
    tab, dev = hip.Table.from_host(ctx, s), hip.Points(ctx, pts[:n])
    assert (hip.msm_points(ctx, tab, 0, n, dev) == want).all()
    # point-range sharding: the ranges' partial sums add up to the whole
    cut = n // 3 + 7
    a, b = hip.msm_points(ctx, tab, 0, cut, dev, 0), hip.msm_points(ctx, tab, cut, n - cut, dev, cut)
    tot = np.zeros(8, dtype=np.uint64)
    olib().orc_point_add(p64(a), p64(b), p64(tot))
    assert (tot == want).all()


@pytest.mark.parametrize("bits", [1, 7, 20, 33, 64])
def test_msm_small_u64_big(env, bits):
    ctx, key, t, pts, h = env
    n = 1 << 14
    rng = np.random.default_rng(bits)
    u = rng.integers(0, 1 << min(bits, 63), size=n, dtype=np.uint64)
    if bits == 64:
        u = u * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
        u[0] = np.uint64(0xFFFFFFFFFFFFFFFF)
    want = np.zeros(8, dtype=np.uint64)
    assert olib().orc_msm_small(p64(u), p64(np.ascontiguousarray(pts[:n])), ctypes.c_size_t(n), p64(want)) == 0
    assert (hip.msm_small(ctx, u, pts[:n]) == want).all()


@pytest.mark.parametrize("kind", ["full", "u64"])
def test_msm_2_pow_20(env, kind):
    """n = 2^20 against the oracle (a few seconds of OpenMP), and the identity MSM(s, t_i H) = <s, t> H that needs no oracle MSM at all."""
    ctx, key, t, pts, h = env
    n = 1 << 20
    rng = np.random.default_rng(20)
    if kind == "full":
        s = structured_scalars(rng, n)
        got = hip.msm_points(ctx, hip.Table.from_host(ctx, s), 0, n, hip.Points(ctx, pts))
        assert (got == oracle_msm(s, pts)).all()
    else:
        u = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
        got = hip.msm_small(ctx, u, pts)
        want_small = np.zeros(8, dtype=np.uint64)
        assert olib().orc_msm_small(p64(u), p64(np.ascontiguousarray(pts)), ctypes.c_size_t(n), p64(want_small)) == 0
        assert (got == want_small).all()
        return
    dot = np.zeros(4, dtype=np.uint64)
    olib().orc_field_dot(0, p64(np.ascontiguousarray(s)), p64(np.ascontiguousarray(t)), ctypes.c_size_t(n), p64(dot))
    assert (got == key.fixed_base_mul_h(dot.reshape(1, 4))[0]).all()
