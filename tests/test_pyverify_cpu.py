"""CPU: the oracle's SpartanSNARK proofs under an independent verifier — tests/pyverify.py, a Python-integer SpartanSNARK::verify written from
src/spartan.rs:469-578 and the files it calls (its header lists them), sharing no code with oracle/ or spartan2_amd/. It pins, against a third
implementation: the Keccak transcript and every encoding absorbed into it, the vk digest, both sum-checks, the matrix evaluations over the padded
shape, SparsePolynomial::evaluate, and the Hyrax / linear-IPA opening (hyrax_pc.rs:480-531, ipa.rs:173-221) with textbook Jacobian arithmetic."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
import pyverify as pv
import pywire
from spartan2_amd import frontend
from test_oracle_wire import CASES, _layout, _prove

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_python_keccak_and_transcript_known_answers():
    # src/provider/keccak.rs:148-157 (test_keccak_example) and the two standard digests
    assert pv.keccak256((0xFFFFFFFF).to_bytes(4, "little")).hex() == "29045a592007d0c246ef02c2223570da9522d0cf0f73282c79a1bc8f0bb2c238"
    assert pv.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert pv.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # multi-block input against the oracle's Keccak (pinned by the same vectors in tests/test_oracle_kats.py)
    data = bytes(range(256)) * 3
    out = np.zeros(32, dtype=np.uint8)
    ol.lib().orc_keccak256(ol.p8(np.frombuffer(data, dtype=np.uint8).copy()), len(data), ol.p8(out))
    assert pv.keccak256(data) == out.tobytes()


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_proofs_pass_the_python_verifier(name):
    inst = CASES[name]()
    sp, words = _prove(inst, 3)
    ck, h, ck_s, h_s, dig = sp.export_keys()
    lay = _layout(sp)
    publics = pv.verify(inst, ck, h, ck_s, h_s, words, lay)  # recomputes the vk digest itself (pywire.spartan_vk_digest)
    assert publics == [int(v) for v in inst.publics]
    assert pv.verify(inst, ck, h, ck_s, h_s, words, lay, vk_digest=dig.tobytes()) == publics


@pytest.mark.parametrize("name", list(CASES))
def test_python_verifier_reads_the_wire_bytes(name):
    """the proof as the reference's bincode bytes (oracle writer) read back by an independent Python reader and verified: pins the byte layout in the
    reading direction too, and the parsed fields equal the ones taken from the flat words"""
    inst = CASES[name]()
    sp, words = _prove(inst, 3)
    ck, h, ck_s, h_s, dig = sp.export_keys()
    data = sp.proof_to_bytes(words)
    assert pv.parse_proof_bytes(data) == pv.parse_proof(words, **_layout(sp))
    assert pv.verify_bytes(inst, ck, h, ck_s, h_s, data, vk_digest=dig.tobytes()) == [int(v) for v in inst.publics]
    for cut in (1, 33, len(data) // 2):
        with pytest.raises(pv.VerifyError):
            pv.verify_bytes(inst, ck, h, ck_s, h_s, data[:-cut], vk_digest=dig.tobytes())
    bad = bytearray(data)
    bad[-32:] = pywire.P_SCALAR.to_bytes(32, "little")  # z_beta := p: not canonical
    with pytest.raises(pv.VerifyError, match="non-canonical"):
        pv.verify_bytes(inst, ck, h, ck_s, h_s, bytes(bad), vk_digest=dig.tobytes())
    with pytest.raises(pv.VerifyError):
        pv.verify_bytes(inst, ck, h, ck_s, h_s, data + b"\x00", vk_digest=dig.tobytes())


def test_python_verifier_rejects_what_the_reference_rejects():
    """one flipped bit per proof section: each is caught by the check the reference would fail at (spartan.rs:511-514, :548-551; ipa.rs:200-217)"""
    inst = CASES["all_three_segments"]()
    sp, words = _prove(inst, 4)
    ck, h, ck_s, h_s, dig = sp.export_keys()
    lay = _layout(sp)
    rows = lay["rows_shared"] + lay["rows_pre"] + lay["rows_rest"]
    o_public = 8 * rows
    o_outer = o_public + 4 * (lay["num_public"] + lay["num_challenges"])
    o_claims = o_outer + 12 * lay["lx"]
    o_inner = o_claims + 12
    o_evalw = o_inner + 8 * lay["ly"]
    o_z = o_evalw + 8 + 16
    cases = [(o_public, "outer sum-check: final claim"),  # a public value: the transcript, hence tau, changes
             (o_outer + 4, "outer sum-check: final claim"), (o_claims, "outer sum-check: final claim"), (o_inner + 4, "inner sum-check: final claim"),
             (o_evalw, "inner sum-check: final claim"), (o_evalw + 4, "inner product argument: first equation"),  # blind_eval_W: comm_eval is absorbed, r moves
             (o_z + 4 * 7, "inner product argument: first equation"), (len(words) - 8, "inner product argument: first equation"),  # z_delta
             (len(words) - 4, "inner product argument: second equation")]  # z_beta
    for off, why in cases:
        bad = words.copy()
        bad[off] ^= np.uint64(2)
        with pytest.raises(pv.VerifyError, match=why):
            pv.verify(inst, ck, h, ck_s, h_s, bad, lay, vk_digest=dig.tobytes())
        assert sp.verify_words(bad) != 0  # the oracle's verifier agrees
    # a commitment row moved off the curve / another key digest
    bad = words.copy()
    bad[0] ^= np.uint64(1)
    with pytest.raises(pv.VerifyError):
        pv.verify(inst, ck, h, ck_s, h_s, bad, lay, vk_digest=dig.tobytes())
    with pytest.raises(pv.VerifyError):
        pv.verify(inst, ck, h, ck_s, h_s, words, lay, vk_digest=bytes(32))


def test_challenge_circuit_proof_passes_the_python_verifier():
    """a circuit with verifier challenges (bellpepper/r1cs.rs:429-461): validate() re-derives them from the transcript"""
    from challenge_circuit import ChallengeCircuit

    inst = ChallengeCircuit(40)
    syn = inst.synthesize(ol.to_mont, ol.from_mont)
    sp = ol.OracleSpartan(inst)
    tape = ol.make_tape(6, 8192)
    used = sp.prep_prove(tape, is_small=False)
    words, _, _ = sp.prove(tape[used:], synthesize=syn)
    assert sp.verify_words(words) == 0
    ck, h, ck_s, h_s, dig = sp.export_keys()
    lay = _layout(sp)
    assert lay["num_challenges"] > 0
    pv.verify(inst, ck, h, ck_s, h_s, words, lay)
    bad = words.copy()
    bad[8 * (lay["rows_shared"] + lay["rows_pre"] + lay["rows_rest"]) + 4 * lay["num_public"]] ^= np.uint64(1)
    with pytest.raises(pv.VerifyError, match="Challenges do not match"):
        pv.verify(inst, ck, h, ck_s, h_s, bad, lay)


def test_golden_proof_passes_the_python_verifier():
    """the frozen proof of tests/golden/spartan_small.json (regenerated here by the oracle, compared by hash) verifies under the Python verifier"""
    with open(os.path.join(GOLD, "spartan_small.json")) as f:
        gold = json.load(f)
    inst = frontend.synthetic_circuit(6, 0xDEADBEEF, num_public=3)
    tape = np.frombuffer(hashlib.shake_256(b"golden-tape").digest(64 * 4096), dtype=np.uint8).reshape(4096, 64).copy()
    sp = ol.OracleSpartan(inst)
    used = sp.prep_prove(tape)
    words, _, _ = sp.prove(tape[used:])
    assert hashlib.sha256(words.tobytes()).hexdigest() == gold["proof_sha256"]
    ck, h, ck_s, h_s, dig = sp.export_keys()
    assert dig.tobytes().hex() == gold["vk_digest"]
    pv.verify(inst, ck, h, ck_s, h_s, words, _layout(sp))


def test_config_2_proof_of_the_oracle_at_its_own_size():
    """BASELINE config 2 (sha256_spartan, 2048-byte message, 2^20 constraints): the oracle's proof under the Python verifier, vk digest recomputed in Python
    over the 62 MB of matrix bytes"""
    inst = frontend.sha256_circuit(bytes(2048))
    sp, words = _prove(inst, 3)
    ck, h, ck_s, h_s, dig = sp.export_keys()
    assert pywire.spartan_vk_digest(inst, ck, h, ck_s, h_s) == dig.tobytes()
    assert pv.verify(inst, ck, h, ck_s, h_s, words, _layout(sp), vk_digest=dig.tobytes()) == [int(v) for v in inst.publics]
