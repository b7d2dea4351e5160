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
