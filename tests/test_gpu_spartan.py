"""GPU parity for the whole hot path: SpartanSNARK::{setup, prep_prove, prove} driven through the C ABI vs the CPU oracle on the
same (R1CS, witness, randomness tape): identical proof words (scalars as canonical Montgomery limbs, points as canonical affine
coordinates), and the oracle's restated verifier (src/spartan.rs:469-578) accepts the GPU proof. Tampered GPU proofs are rejected."""
import numpy as np
import pytest

import oracle_lib as ol
from spartan2_amd import frontend, hip, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def run_both(ctx, inst, seed):
    tape = ol.make_tape(seed, 8192)
    osp = ol.OracleSpartan(inst)
    used_o = osp.prep_prove(tape)
    want, used_o2, _ = osp.prove(tape[used_o:])
    gsp = host.SpartanSNARK(ctx, inst)
    used_g = gsp.prep_prove(tape)
    assert used_g == used_o
    got, used_g2, phases = gsp.prove(tape[used_g:])
    assert used_g2 == used_o2
    return osp, gsp, want, got, tape, used_g


@pytest.mark.parametrize("which", ["synthetic_small", "synthetic_3k", "sha256_1block", "sha256_3blocks"])
def test_prove_matches_oracle_bit_exact(ctx, which):
    inst = {
        "synthetic_small": lambda: frontend.synthetic_circuit(5, 7, num_public=2),
        "synthetic_3k": lambda: frontend.synthetic_circuit(40, 0xDEADBEEF, num_public=5),
        "sha256_1block": lambda: frontend.sha256_circuit(b"abc"),
        "sha256_3blocks": lambda: frontend.sha256_circuit(bytes(range(150))),
    }[which]()
    osp, gsp, want, got, tape, used = run_both(ctx, inst, 11)
    # setup parity: same digest substitute, same keys
    ck, h, ck_s, h_s, dig = osp.export_keys()
    assert (gsp.vk_digest == dig).all()
    # prep parity: commitment rows and cached products
    for a, b in zip(gsp.prep_export(), osp.prep_export()):
        assert (a == b).all()
    assert len(got) == len(want)
    assert (got == want).all()
    assert osp.verify_words(got) == 0
    # prove again on the same prep state with fresh randomness (the benches' warm-up + timed pattern, sha256_spartan.rs:224-243)
    tape2 = ol.make_tape(12, 4096)
    want2 = osp.prove(tape2)[0]
    got2 = gsp.prove(tape2)[0]
    assert (got2 == want2).all() and not (got2 == got).all()
    assert osp.verify_words(got2) == 0


def test_reference_e2e_cubic_circuit_on_the_gpu(ctx):
    """The reference's own end-to-end test (src/spartan.rs:653-688, T256HyraxEngine): rest-only CubicCircuit, is_small = false,
    verify() must return the public output [15]."""
    inst = frontend.cubic_circuit()
    tape = ol.make_tape(3, 4096)
    osp = ol.OracleSpartan(inst)
    assert osp.prep_prove(tape, is_small=False) == 0
    want = osp.prove(tape)[0]
    gsp = host.SpartanSNARK(ctx, inst)
    assert gsp.prep_prove(tape, is_small=False) == 0
    got = gsp.prove(tape)[0]
    assert (got == want).all()
    assert osp.verify_words(got) == 0
    assert ol.from_mont(got[8:12]) == 15


@pytest.mark.parametrize("cut", [(300, 400), (0, 0), (1000, 0), (250, 0), (0, 600)])
def test_shared_precommitted_rest_segments_match_oracle(ctx, cut):
    # shared_witness / precommitted_witness / rest commitments (bellpepper/r1cs.rs:306-538) in every combination
    inst = frontend.synthetic_circuit(6, 21, num_public=2, shared_permille=cut[0], precommitted_permille=cut[1])
    osp, gsp, want, got, _, _ = run_both(ctx, inst, 4)
    for a, b in zip(gsp.prep_export(), osp.prep_export()):
        assert (a == b).all()
    assert (got == want).all()
    assert osp.verify_words(got) == 0


def test_tampered_gpu_proof_is_rejected(ctx):
    inst = frontend.synthetic_circuit(9, 3, num_public=2)
    osp, gsp, want, got, _, _ = run_both(ctx, inst, 5)
    rng = np.random.default_rng(1)
    for pos in [0, len(got) // 3, len(got) // 2, len(got) - 1] + list(rng.integers(0, len(got), size=4)):
        bad = got.copy()
        bad[int(pos)] ^= np.uint64(1 << 7)
        assert osp.verify_words(bad) != 0


def test_wrong_witness_length_is_an_error(ctx):
    inst = frontend.synthetic_circuit(5, 7, num_public=2)
    gsp = host.SpartanSNARK(ctx, inst)
    inst.witness = inst.witness[:-1]
    with pytest.raises(hip.SpartanHipError) as e:
        gsp.prep_prove(ol.make_tape(1, 64))
    assert "rc=-2" in str(e.value)  # SpartanError::InvalidWitnessLength


def test_prove_is_identical_on_every_driver_path(ctx, monkeypatch):
    """The latency machinery must not change a single proof word: the default driver (comm_LZ as an MSM over the row commitments started
    mid-sum-check by the helper thread, rounds launched ahead of their challenge through the device-memory mailbox, resident tail) against
    SPARTAN_LZ_DIRECT=1 (the reference's order: bind W with L, then the MSM over the key) and SPARTAN_MAIL_DEV=0 (launch after each
    challenge, host-memory mailbox). Several rows of W so that the row variables exist."""
    inst = frontend.sha256_circuit(bytes(range(150)))
    tape = ol.make_tape(23, 8192)

    def prove_with(c, **flags):
        sn = host.SpartanSNARK(c, inst)
        used = sn.prep_prove(tape)
        if flags:
            sn.set_flags(**flags)
        words, _, _ = sn.prove(tape[used:])
        again, _, _ = sn.prove(tape[used:])  # the prep state is reusable: second prove on it is the same proof
        assert (words == again).all()
        return words

    base = prove_with(ctx)
    osp = ol.OracleSpartan(inst)  # ... and every path is compared with the ORACLE's proof, not only with each other
    used_o = osp.prep_prove(tape)
    want, _, _ = osp.prove(tape[used_o:])
    assert (base == want).all()
    # the row tables of the prepared witness at each of their build times (prep_prove's SPARTAN_PREP_TABLES): never, inside prep_prove, queued there
    for mode in ("off", "sync", "prep"):
        monkeypatch.setenv("SPARTAN_PREP_TABLES", mode)
        assert (prove_with(ctx) == want).all()
        assert (prove_with(ctx, reference_order=True) == want).all()
    monkeypatch.delenv("SPARTAN_PREP_TABLES")
    # the reference-order driver: one thread, statement order of src/spartan.rs:226-466, PCS::prove as ONE call of sp_hyrax_prove (table walks over the
    # window tables of the key beside the commitment's hashing) — and the same call with the bucket MSMs it falls back to
    assert (prove_with(ctx, reference_order=True) == base).all()
    monkeypatch.setenv("SPARTAN_KEY_TABLES", "0")
    assert (prove_with(ctx, reference_order=True) == base).all()
    monkeypatch.delenv("SPARTAN_KEY_TABLES")
    monkeypatch.setenv("SPARTAN_LZ_DIRECT", "1")
    assert (prove_with(ctx) == base).all()
    monkeypatch.delenv("SPARTAN_LZ_DIRECT")
    monkeypatch.setenv("SPARTAN_MAIL_DEV", "0")  # read when a context is created
    c2 = hip.Context(0)
    try:
        assert (prove_with(c2) == base).all()
    finally:
        c2.close()
