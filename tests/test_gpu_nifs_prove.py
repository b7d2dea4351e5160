"""GPU parity: NeutronNovaNIFS::prove as a whole (host driver spartan2_amd/host/neutronnova_nifs.cpp over the C ABI) against the oracle's
restatement (oracle/nifs.hpp nifs_prove): transcript preamble (U, T, tau, rho), layers by sp_multiply_vec, the rounds, fold_multiple,
fold_blinds, the X fold and fold_commitments[_partial] — every output bit-exact. `process_round` is the same caller-side hook on both sides."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
P = ol.MODULI[0]


@pytest.fixture(scope="module")
def env():
    from spartan2_amd import hip

    ctx = hip.Context(0)
    L = ol.lib()
    okey = ctypes.c_void_p(L.orc_hyrax_setup(b"ck", ctypes.c_size_t(2048)))
    ck_aff = np.zeros((2048, 8), dtype=np.uint64)
    h_aff = np.zeros(8, dtype=np.uint64)
    L.orc_hyrax_key_export(okey, ol.p64(ck_aff), ol.p64(h_aff))
    ck = hip.CommitmentKey(ctx, ck_aff, h_aff)
    yield ctx, okey, ck
    L.orc_hyrax_free(okey)
    ctx.close()


def _padded_witness(oshape, inst):
    W = np.zeros((oshape.num_vars, 4), dtype=np.uint64)
    w = ol.mont_array([int(x) for x in inst.witness])
    s, p, r = oshape.num_shared_unpadded, oshape.num_precommitted_unpadded, oshape.num_rest_unpadded
    W[:s] = w[:s]
    W[oshape.num_shared : oshape.num_shared + p] = w[s : s + p]
    W[oshape.num_shared + oshape.num_precommitted : oshape.num_shared + oshape.num_precommitted + r] = w[s + p :]
    return W


def _setup(ctx, okey, insts, rng):
    from spartan2_amd import hip
    from spartan2_amd.host import pad_shape

    oshape = ol.OracleShape(insts[0])
    mats, dims = pad_shape(insts[0])
    shape = hip.Shape(ctx, mats, dims)
    rows = oshape.num_vars // 2048
    assert rows * 2048 == oshape.num_vars
    Ws = np.stack([_padded_witness(oshape, i) for i in insts])
    X = np.stack([ol.mont_array([int(x) for x in i.publics]) for i in insts])
    r_W = np.stack([ol.random_field_array(rng, rows) for _ in insts])
    comms = np.zeros((len(insts), rows, 8), dtype=np.uint64)
    for k in range(len(insts)):
        assert ol.lib().orc_hyrax_commit(okey, ol.p64(Ws[k]), ctypes.c_size_t(oshape.num_vars), ol.p64(r_W[k]), 1, ol.p64(comms[k])) == 0
    return oshape, shape, dims, Ws, X, r_W, comms


def _compare(ctx, okey, ck, oshape, shape, dims, Ws, X, r_W, comms, small):
    from spartan2_amd import hip, host

    want = ol.nifs_prove(oshape, okey, comms, X, Ws, r_W, small, ol.Transcript(b"neutronnova_prove"), ol.transcript_round_hook(ol.Transcript(b"vc")))
    tabs = [hip.Table.from_host(ctx, w) for w in Ws]
    otr = ol.Transcript(b"vc")
    got = host.nifs_prove(ctx, shape, dims, ck, comms, X, tabs, r_W, small, hip.Transcript(ctx, b"neutronnova_prove"), ol.transcript_round_hook(otr))
    N, nv = oshape.num_cons, oshape.num_vars
    for key in ("polys", "r_bs", "E_eq", "tail", "folded_rW", "folded_X", "folded_comm"):
        assert (want[key] == got[key]).all(), key
    for key, n in (("A", N), ("B", N), ("C", N), ("folded_W", nv)):
        assert (want[key] == got[key].read(0, n)).all(), key
    return want


def test_nifs_prove_padding_and_partial_commitment_fold(env):
    """3 instances of one synthetic shape with shared + precommitted + rest segments: padded to 4 with clones of instance 0 (:549-552);
    rest rows of the folded commitment come from the folded blind (fold_commitments_partial, hyrax_pc.rs:820-874)."""
    from spartan2_amd import frontend

    ctx, okey, ck = env
    rng = np.random.default_rng(31)
    insts = [frontend.synthetic_circuit(60, 5, num_public=3, shared_permille=150, precommitted_permille=450, witness_seed=s) for s in (11, 12, 13)]
    setup = _setup(ctx, okey, insts, rng)
    oshape = setup[0]
    assert oshape.num_rest > 0 and oshape.num_shared > 0
    for small in (True, False):
        want = _compare(ctx, okey, ck, *setup, small)
    # the rest segment of the folded witness is re-zeroed (:1227-1231)
    assert not want["folded_W"][oshape.num_shared + oshape.num_precommitted :].any()


def test_nifs_prove_sha256_steps_and_folded_commitment_opens(env):
    """Two one-block SHA-256 step instances (precommitted-only, as benches/sha256_neutronnova.rs builds them): full fold_commitments;
    the folded commitment is the Hyrax commitment of the folded witness under the folded blind, and the folded instance satisfies the
    folded claim T_out = sum_k E[k] (A B - C)[k]."""
    from spartan2_amd import frontend

    ctx, okey, ck = env
    rng = np.random.default_rng(32)
    insts = [frontend.sha256_circuit(m) for m in (b"abc", b"abd")]
    setup = _setup(ctx, okey, insts, rng)
    oshape = setup[0]
    want = _compare(ctx, okey, ck, *setup, True)
    recommit = np.zeros_like(want["folded_comm"])
    assert ol.lib().orc_hyrax_commit(okey, ol.p64(want["folded_W"]), ctypes.c_size_t(oshape.num_vars), ol.p64(want["folded_rW"]), 0, ol.p64(recommit)) == 0
    assert (recommit == want["folded_comm"]).all()
    _, left, right = ol.tensor_decomp(oshape.num_cons)
    fa, fb, fc, e = ol.ints_of(want["A"]), ol.ints_of(want["B"]), ol.ints_of(want["C"]), ol.ints_of(want["E_eq"])
    lhs = sum(e[k % left] * e[left + k // left] % P * ((fa[k] * fb[k] - fc[k]) % P) for k in range(oshape.num_cons)) % P
    assert lhs == ol.from_mont(want["tail"][0])
