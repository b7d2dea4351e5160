"""GPU parity: the NeutronNova NIFS rounds through the C ABI (sp_nifs_*) against the oracle's restatement of
NeutronNovaNIFS::prove (oracle/nifs.hpp, both branches), bit-exact on every round polynomial, the folded layers, T_out and eq(r_b, rho).
The per-round `process_round` (verifier-circuit commit, outside the data path) is the same caller-supplied transcript hook on both sides."""
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


def _layers(rng, n_padded, total, n_large, full_random=False):
    if full_random:
        A = np.stack([ol.random_field_array(rng, total) for _ in range(n_padded)])
        B = np.stack([ol.random_field_array(rng, total) for _ in range(n_padded)])
        C = np.stack([ol.random_field_array(rng, total) for _ in range(n_padded)])
        return A, B, C
    a = rng.integers(-5, 6, size=(n_padded, total)).astype(object)
    b = rng.integers(0, 2, size=(n_padded, total)).astype(object)
    a[0, 0], b[0, 0] = (1 << 62) - 1, -((1 << 62) - 1)  # extremes of the small range
    a[1, 0], b[1, 0] = -((1 << 62) - 1), (1 << 62) - 1
    for _ in range(n_large):
        i, k = int(rng.integers(n_padded)), int(rng.integers(total))
        a[i, k] = int.from_bytes(rng.bytes(32), "little") % P
        b[i, k] = int.from_bytes(rng.bytes(32), "little") % P
    c = (a * b) % P
    arr = lambda m: np.stack([ol.mont_array([int(v) % P for v in row]) for row in m])
    return arr(a), arr(b), arr(c)


def _run_gpu(ctx, left, right, E, rhos, A, B, C, small, hook):
    from spartan2_amd import hip

    n_padded, total = A.shape[0], left * right
    nifs = hip.Nifs(ctx, n_padded, left, right)
    for which, M in enumerate((A, B, C)):
        for b in range(n_padded):
            v = nifs.layer(which, b)
            v.write(0, M[b])
            v.free()
    nifs.begin(E, rhos, small_values=small)
    polys, r_bs = [], []
    for t in range(rhos.shape[0]):
        co = nifs.round(t)
        polys.append(co)
        r = hook(t, co)
        r_bs.append(r)
        nifs.challenge(r)
    oa, ob, oc = (hip.Table.zeros(ctx, total) for _ in range(3))
    T, eq = nifs.finish(oa, ob, oc)
    out = dict(polys=np.stack(polys), r_bs=np.stack(r_bs), A=oa.read(0, total), B=ob.read(0, total), C=oc.read(0, total), T_out=T, eq_rho_at_rb=eq)
    nifs.free()
    return out


@pytest.mark.parametrize("n_padded,num_cons,n_large,small,full_random", [
    (2, 16, 0, False, False), (2, 16, 0, True, False), (4, 64, 3, True, False), (8, 64, 4, False, False), (8, 1 << 9, 5, True, False),
    (4, 1 << 16, 7, True, False),   # left = 256: the factored (block-per-x_out) kernels
    (4, 1 << 16, 7, False, False),
    (8, 1 << 17, 0, False, True),   # left = 512, uniformly random layers (every position large): field kernels
    (4, 1 << 16, 0, True, True),    # small-value request on data that is all large: everything goes through the corrections
])
def test_nifs_rounds_bit_exact(ctx, n_padded, num_cons, n_large, small, full_random):
    rng = np.random.default_rng(77 + n_padded + num_cons + n_large)
    ell, left, right = ol.tensor_decomp(num_cons)
    total = left * right
    A, B, C = _layers(rng, n_padded, total, n_large, full_random)
    tau = ol.random_field_array(rng, 1)[0]
    E = ol.pow_split_evals(tau, ell, left, right)
    ell_b = n_padded.bit_length() - 1
    rhos = ol.random_field_array(rng, ell_b)
    want = ol.nifs_prove_core(left, right, E, rhos, A, B, C, small, ol.transcript_round_hook(ol.Transcript(b"nifs")))
    got = _run_gpu(ctx, left, right, E, rhos, A, B, C, small, ol.transcript_round_hook(ol.Transcript(b"nifs")))
    for key in ("polys", "r_bs", "T_out", "eq_rho_at_rb", "A", "B", "C"):
        assert (want[key] == got[key]).all(), key


def test_pow_split_evals_and_to_small(ctx):
    from spartan2_amd import hip

    rng = np.random.default_rng(5)
    tau = ol.random_field_array(rng, 1)[0]
    for n in (16, 1 << 9, 1 << 15):
        ell, left, right = ol.tensor_decomp(n)
        assert (hip.pow_split_evals(tau, ell, left, right) == ol.pow_split_evals(tau, ell, left, right)).all()
    SM = (1 << 62) - 1
    vals = [0, 1, 5, P - 3, SM, P - SM, SM + 1, P - SM - 1, P - 1, 1 << 64, 1 << 200] + [int.from_bytes(rng.bytes(32), "little") % P for _ in range(300)]
    vals += [int(x) for x in rng.integers(-1000, 1000, size=700)]
    arr = ol.mont_array([v % P for v in vals])
    t = hip.Table.from_host(ctx, arr)
    got, got_large = hip.to_small_vec_or_zero(ctx, t, len(vals))
    want, want_large = ol.to_small_vec_or_zero(arr)
    assert (got == want).all() and (got_large == want_large).all()


def test_nifs_layers_from_multiply_vec_and_fold(ctx):
    """Two satisfying instances of the one-block SHA-256 shape: sp_multiply_vec writes Az/Bz/Cz straight into the NIFS layer views
    (src/neutronnova_zk.rs:576-596); the rounds then run on them and the folded claim holds (Python integers)."""
    from spartan2_amd import frontend, hip
    from spartan2_amd.host import pad_shape

    insts = [frontend.sha256_circuit(m) for m in (b"abc", b"abd")]  # one shape, two satisfying assignments
    oshape = ol.OracleShape(insts[0])
    mats, dims = pad_shape(insts[0])
    shape = hip.Shape(ctx, mats, dims)
    N, M = oshape.num_cons, oshape.num_vars
    ell, left, right = ol.tensor_decomp(N)
    total = left * right
    assert total == N
    nifs = hip.Nifs(ctx, 2, left, right)
    want_layers = []
    for b, inst in enumerate(insts):
        W = np.zeros((M, 4), dtype=np.uint64)
        W[oshape.num_shared : oshape.num_shared + len(inst.witness)] = ol.mont_array([int(x) for x in inst.witness])
        z = np.concatenate([W, ol.mont_array([1] + [int(x) for x in inst.publics])])
        views = [nifs.layer(w, b) for w in range(3)]
        shape.multiply_vec(hip.Table.from_host(ctx, z), *views)
        want = [np.zeros((N, 4), dtype=np.uint64) for _ in range(3)]
        assert ol.lib().orc_shape_multiply_vec(oshape.h, ol.p64(z), *(ol.p64(w) for w in want)) == 0
        for v, w in zip(views, want):
            assert (v.read(0, N) == w).all()
            v.free()
        want_layers.append(want)
    rng = np.random.default_rng(9)
    E = ol.pow_split_evals(ol.random_field_array(rng, 1)[0], ell, left, right)
    rhos = ol.random_field_array(rng, 1)
    A, B, C = (np.stack([want_layers[b][q] for b in range(2)]) for q in range(3))
    for small in (False, True):
        for which, Mx in enumerate((A, B, C)):  # the rounds consume the layers: rewrite them
            for b in range(2):
                v = nifs.layer(which, b)
                v.write(0, Mx[b])
                v.free()
        nifs.begin(E, rhos, small_values=small)
        tr = ol.Transcript(b"nifs")
        hook = ol.transcript_round_hook(tr)
        co = nifs.round(0)
        nifs.challenge(hook(0, co))
        oa, ob, oc = (hip.Table.zeros(ctx, total) for _ in range(3))
        T, _ = nifs.finish(oa, ob, oc)
        want = ol.nifs_prove_core(left, right, E, rhos, A, B, C, small, ol.transcript_round_hook(ol.Transcript(b"nifs")))
        assert (oa.read(0, total) == want["A"]).all() and (oc.read(0, total) == want["C"]).all() and (T == want["T_out"]).all()
        fa, fb, fc, e = ol.ints_of(want["A"]), ol.ints_of(want["B"]), ol.ints_of(want["C"]), ol.ints_of(E)
        lhs = sum(e[k % left] * e[left + k // left] % P * ((fa[k] * fb[k] - fc[k]) % P) for k in range(total)) % P
        assert lhs == ol.from_mont(T)
