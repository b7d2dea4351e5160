"""GPU: SURVEY 8(f) rank 4 on the device-backed drivers — the vk digest and the serialised proof of the product equal the oracle's, verify accepts the
deserialised bytes and rejects tampered ones, on circuits with every witness-segment mix and with verifier challenges (the BASELINE configurations run the
same assertions at their own size in tests/test_gpu_configs.py and tests/test_gpu_neutronnova_zk.py); and the incremental-commit cache of SpartanZkSNARK
(PCS::commit_without_blind / commit_incremental, src/provider/pcs/hyrax_pc.rs:533-607, src/spartan_zk.rs:335-366) through the C ABI against the oracle."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import pywire
from spartan2_amd import frontend, hip, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


CIRCUITS = {
    "precommitted_only": lambda: frontend.synthetic_circuit(40, 0xD1, num_public=3),
    "all_three_segments": lambda: frontend.synthetic_circuit(300, 0xBEEF, num_public=2, shared_permille=200, precommitted_permille=500),
    "rest_only_cubic": lambda: frontend.cubic_circuit(),
    "sha256_one_block": lambda: frontend.sha256_circuit(b"abc"),
}


@pytest.mark.parametrize("name", list(CIRCUITS))
def test_digest_and_proof_bytes_equal_the_oracles(ctx, name):
    inst = CIRCUITS[name]()
    tape = ol.make_tape(91, 8192)
    osp = ol.OracleSpartan(inst)
    used = osp.prep_prove(tape)
    want, _, _ = osp.prove(tape[used:])
    gsp = host.SpartanSNARK(ctx, inst)
    assert gsp.vk_digest.tobytes() == osp.export_keys()[4].tobytes()
    assert gsp.prep_prove(tape) == used
    got, _, _ = gsp.prove(tape[used:])
    assert (got == want).all()
    data = gsp.proof_to_bytes(got)
    assert data == osp.proof_to_bytes(want)
    L = gsp.proof_layout()
    assert data == pywire.spartan_proof_bytes(got, L["rows_shared"], L["rows_precommitted"], L["rows_rest"], L["num_public"], L["num_challenges"], L["rounds_x"],
                                              L["rounds_y"], L["z_len"])
    assert gsp.verify_bytes(data) == 0
    assert osp.verify_words(osp.proof_from_bytes(data)) == 0
    # tampering: a flipped bit inside a scalar -> a failed check; malformed bytes / a proof of another shape -> check 1
    rng = np.random.default_rng(5)
    for pos in [len(data) - 1, len(data) - 40, len(data) // 2] + [int(x) for x in rng.integers(0, len(data), size=6)]:
        bad = bytearray(data)
        bad[pos] ^= 1
        assert gsp.verify_bytes(bytes(bad)) != 0, pos
    assert gsp.verify_bytes(data[:-1]) == 1 and gsp.verify_bytes(data + b"\0") == 1 and gsp.verify_bytes(b"") == 1
    other = host.SpartanSNARK(ctx, frontend.synthetic_circuit(41, 0xD1, num_public=4))
    assert other.verify_bytes(data) == 1
    other.close()
    gsp.close()


def test_challenge_circuit_proof_bytes(ctx):
    """num_challenges > 0 (src/bellpepper/r1cs.rs:429-461): the instance's challenges travel in the proof"""
    from challenge_circuit import ChallengeCircuit

    inst = ChallengeCircuit(120, seed=3)
    synth = inst.synthesize(ol.to_mont, ol.from_mont)
    tape = ol.make_tape(17, 8192)
    osp = ol.OracleSpartan(inst)
    used = osp.prep_prove(tape, is_small=False)  # the rest segment (challenge * x) is full-width
    want, _, _ = osp.prove(tape[used:], synthesize=synth)
    gsp = host.SpartanSNARK(ctx, inst)
    assert gsp.prep_prove(tape, is_small=False) == used
    got, _, _ = gsp.prove(tape[used:], synthesize=synth)
    assert (got == want).all() and gsp.proof_layout()["num_challenges"] > 0
    data = gsp.proof_to_bytes(got)
    assert data == osp.proof_to_bytes(want) and gsp.verify_bytes(data) == 0
    gsp.close()


@pytest.mark.parametrize("width,n", [(2048, 5 * 2048), (2048, 3 * 2048 + 100), (32, 96)])
def test_commit_without_blind_and_incremental(ctx, width, n):
    L = ol.lib()
    okey = ctypes.c_void_p(L.orc_hyrax_setup(b"ck", ctypes.c_size_t(width)))
    ck_aff = np.zeros((width, 8), dtype=np.uint64)
    h_aff = np.zeros(8, dtype=np.uint64)
    L.orc_hyrax_key_export(okey, ol.p64(ck_aff), ol.p64(h_aff))
    key = hip.CommitmentKey(ctx, ck_aff, h_aff)
    rng = np.random.default_rng(width + n)
    rows = -(-n // width)
    # a witness of bits with one all-zero row and one row of full-width values
    bits = rng.integers(0, 2, size=n)
    v = np.zeros((n, 4), dtype=np.uint64)
    v[bits == 1] = ol.to_mont(1)
    if rows > 2:
        v[width:2 * width] = 0
    v[-min(width, n) // 2:] = ol.random_field_array(rng, min(width, n) // 2)
    want_raw = np.zeros((rows, 8), dtype=np.uint64)
    assert L.orc_hyrax_commit_without_blind(okey, ol.p64(v), ctypes.c_size_t(n), 0, ol.p64(want_raw)) == 0
    t = hip.Table.from_host(ctx, v)
    raw = key.commit_without_blind(t, 0, n, is_small=False)
    assert (raw == want_raw).all()
    if rows > 2:
        assert not raw[1].any()  # the identity
    blinds = ol.random_field_array(rng, rows)
    zeros = hip.Table.zeros(ctx, n)
    # first prove of SpartanZkSNARK: raw + zero delta + blinds == the plain commitment
    first = key.commit_incremental(raw, zeros, 0, n, blinds)
    assert (first == key.commit(t, 0, n, blinds, is_small=False)).all()
    # later proves: a sparse delta against the cached rows
    delta = np.zeros((n, 4), dtype=np.uint64)
    P = ol.MODULI[0]
    for k in rng.integers(0, n, size=7):
        delta[k] = ol.to_mont(int(rng.integers(1, 1 << 62)) * (-1 if k % 2 else 1) % P)
    want = np.zeros((rows, 8), dtype=np.uint64)
    assert L.orc_hyrax_commit_incremental(okey, ol.p64(want_raw), ctypes.c_size_t(rows), ol.p64(delta), ctypes.c_size_t(n), ol.p64(blinds), ol.p64(want)) == 0
    td = hip.Table.from_host(ctx, delta)
    got = key.commit_incremental(raw, td, 0, n, blinds)
    assert (got == want).all()
    # == the commitment of v + delta
    vd = np.stack([ol.to_mont((ol.from_mont(v[i]) + ol.from_mont(delta[i])) % P) for i in np.nonzero(delta.any(axis=1))[0]])
    v2 = v.copy()
    v2[delta.any(axis=1)] = vd
    assert (got == key.commit(hip.Table.from_host(ctx, v2), 0, n, blinds, is_small=False)).all()
    # fewer cached rows than rows: the missing ones count as the identity
    short = key.commit_incremental(raw[:1], td, 0, n, blinds)
    want_s = np.zeros((rows, 8), dtype=np.uint64)
    assert L.orc_hyrax_commit_incremental(okey, ol.p64(want_raw[:1].copy()), ctypes.c_size_t(1), ol.p64(delta), ctypes.c_size_t(n), ol.p64(blinds), ol.p64(want_s)) == 0
    assert (short == want_s).all()
    L.orc_hyrax_free(okey)
