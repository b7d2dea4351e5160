"""GPU parity: the NeutronNova batched ZK sum-check drivers (prove_quad_batched_zk, prove_cubic_with_additive_term_batched_zk,
src/sumcheck.rs:702-917) through the C ABI against the oracle's restatement; the verifier circuit's process_round is the same caller-side
hook on both sides. Also the sum-check identity itself: the final claims reproduce the last round polynomial at the last challenge."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
P = ol.MODULI[0]


@pytest.fixture(scope="module")
def ctx():
    from spartan2_amd import hip

    c = hip.Context(0)
    yield c
    c.close()


# tables <= 2^13 bind + evaluate in one launch with one product per lane, larger ones as two launches: 15 rounds run both forms
@pytest.mark.parametrize("num_rounds", [1, 4, 9, 13, 15])
def test_quad_batched(ctx, num_rounds):
    from spartan2_amd import hip

    rng = np.random.default_rng(100 + num_rounds)
    n = 1 << num_rounds
    T = [ol.random_field_array(rng, n) for _ in range(4)]  # A0 A1 B0 B1
    if num_rounds >= 4:  # sparse tails, as z / poly_ABC have them
        T[0][n // 2 + 3 :] = 0
        T[2][n - n // 4 :] = 0
    dot = lambda a, b: ol.to_mont(sum(x * y for x, y in zip(ol.ints_of(a), ol.ints_of(b))) % P)
    claims = np.stack([dot(T[0], T[2]), dot(T[1], T[3])])
    want_r, want_fin = ol.prove_quad_batched(claims, num_rounds, *T, 7, ol.batched_transcript_hook(ol.Transcript(b"q")))
    tabs = [hip.Table.from_host(ctx, t) for t in T]
    got_r, got_fin = hip.sumcheck_quad_batched(ctx, claims, num_rounds, *tabs, 7, ol.batched_transcript_hook(ol.Transcript(b"q")))
    assert (want_r == got_r).all() and (want_fin == got_fin).all()
    # the bound tables are the multilinear extensions at r: A0(r) by direct evaluation
    r = ol.ints_of(got_r)
    vals = ol.ints_of(T[0])
    for rj in r:
        h = len(vals) // 2
        vals = [(vals[i] + rj * (vals[h + i] - vals[i])) % P for i in range(h)]
    assert vals[0] == ol.from_mont(got_fin[0])


@pytest.mark.parametrize("num_rounds", [2, 5, 10, 15])
def test_cubic_outer_pow_batched(ctx, num_rounds):
    from spartan2_amd import hip

    rng = np.random.default_rng(200 + num_rounds)
    n = 1 << num_rounds
    ell, left, right = ol.tensor_decomp(n)
    tau = ol.random_field_array(rng, 1)[0]
    E = ol.pow_split_evals(tau, ell, left, right)
    pl, pr = E[:left].copy(), E[left:].copy()
    step = [ol.random_field_array(rng, n) for _ in range(3)]
    core = [ol.random_field_array(rng, n) for _ in range(3)]
    # core is a satisfying branch (claim 0): C = A o B
    a, b = ol.ints_of(core[0]), ol.ints_of(core[1])
    core[2] = ol.mont_array([x * y % P for x, y in zip(a, b)])
    e = ol.ints_of(E)
    sa, sb, sc = (ol.ints_of(t) for t in step)
    t_out = ol.to_mont(sum(e[k % left] * e[left + k // left] % P * ((sa[k] * sb[k] - sc[k]) % P) for k in range(n)) % P)
    want_r, want_fin, want_base = ol.prove_cubic_outer_pow_batched(num_rounds, pl, pr, step, core, t_out, 3, ol.batched_transcript_hook(ol.Transcript(b"c")))
    tl, tr_ = hip.Table.from_host(ctx, pl), hip.Table.from_host(ctx, pr)
    ts = [hip.Table.from_host(ctx, t) for t in step]
    tc = [hip.Table.from_host(ctx, t) for t in core]
    got_r = hip.sumcheck_cubic_outer_pow_batched(ctx, num_rounds, tl, tr_, ts, tc, t_out, 3, ol.batched_transcript_hook(ol.Transcript(b"c")))
    assert (want_r == got_r).all()
    got_fin = np.stack([t.read(0, 1)[0] for t in ts + tc])
    assert (want_fin == got_fin).all()
    assert (tl.read(0, 1)[0] == want_base).all()
