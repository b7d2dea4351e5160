"""End-to-end prove -> verify on the CPU oracle (the integration level of the reference's test pyramid:
src/spartan.rs:653-688 test_snark), plus tamper rejection. CPU only."""
import numpy as np
import pytest

import oracle_lib as ol
from spartan2_amd import frontend


@pytest.fixture(scope="module")
def small():
    inst = frontend.synthetic_circuit(12, 0xDEADBEEF, num_public=4)
    sp = ol.OracleSpartan(inst)
    tape = ol.make_tape(1, 4096)
    used = sp.prep_prove(tape)
    words, used2, _ = sp.prove(tape[used:])
    return sp, words


def test_oracle_prove_verify_accepts(small):
    sp, words = small
    assert sp.verify_words(words) == 0


def test_oracle_rejects_single_bit_tampering(small):
    sp, words = small
    rng = np.random.default_rng(3)
    n = len(words)
    # one flipped bit in each proof section must be rejected
    for pos in [0, n // 7, n // 3, n // 2, (2 * n) // 3, n - 40, n - 1] + list(rng.integers(0, n, size=6)):
        bad = words.copy()
        bad[int(pos)] ^= np.uint64(1)
        assert sp.verify_words(bad) != 0, int(pos)


def test_oracle_proof_is_deterministic_given_the_tape():
    inst = frontend.synthetic_circuit(5, 7, num_public=2)
    tape = ol.make_tape(9, 4096)
    outs = []
    for _ in range(2):
        sp = ol.OracleSpartan(inst)
        used = sp.prep_prove(tape)
        outs.append(sp.prove(tape[used:])[0])
    assert (outs[0] == outs[1]).all()
    sp2 = ol.OracleSpartan(inst)
    tape2 = ol.make_tape(10, 4096)
    used = sp2.prep_prove(tape2)
    assert not (sp2.prove(tape2[used:])[0] == outs[0]).all()


def test_oracle_sha256_one_block_prove_verify():
    inst = frontend.sha256_circuit(bytes(range(40)))  # one compression, ~26k constraints
    assert 26000 < inst.num_cons < 27000 and inst.num_public == 256
    sp = ol.OracleSpartan(inst)
    assert sp.shape.num_cons == 1 << 15 and sp.shape.num_vars == 1 << 15
    tape = ol.make_tape(2, 4096)
    used = sp.prep_prove(tape)
    words, _, _ = sp.prove(tape[used:])
    assert sp.verify_words(words) == 0


def test_reference_e2e_cubic_circuit_public_output_15():
    """KAT (6) of SURVEY 8c: the reference's test_snark (src/spartan.rs:653-688) on T256HyraxEngine proves the rest-only
    CubicCircuit x^3 + x + 5 = y, x = 2 and expects verify() to return [15]."""
    inst = frontend.cubic_circuit()
    assert (inst.num_cons, inst.num_shared, inst.num_precommitted, inst.num_rest, inst.num_public) == (4, 0, 0, 4, 1)
    assert list(inst.publics) == [15]
    sp = ol.OracleSpartan(inst)
    assert (sp.shape.num_cons, sp.shape.num_vars) == (4, 2048)
    tape = ol.make_tape(3, 4096)
    used = sp.prep_prove(tape, is_small=False)
    assert used == 0  # nothing to commit at prep time: no shared / precommitted variables
    words, _, _ = sp.prove(tape)
    assert sp.verify_words(words) == 0
    # public values travel in the proof; the verifier returns them (src/spartan.rs:577)
    rows = 1
    assert ol.from_mont(words[8 * rows : 8 * rows + 4]) == 15


@pytest.mark.parametrize("cut", [(300, 400), (0, 0), (1000, 0), (250, 0), (0, 600)])
def test_oracle_prove_verify_with_shared_precommitted_rest_segments(cut):
    inst = frontend.synthetic_circuit(6, 21, num_public=2, shared_permille=cut[0], precommitted_permille=cut[1])
    sp = ol.OracleSpartan(inst)
    tape = ol.make_tape(4, 8192)
    used = sp.prep_prove(tape)
    words, _, _ = sp.prove(tape[used:])
    assert sp.verify_words(words) == 0
    bad = words.copy()
    bad[3] ^= np.uint64(1)
    assert sp.verify_words(bad) != 0
