"""GPU parity: circuits with verifier challenges (VERDICT r1 missing #3) — the prover squeezes the challenges after the precommitted commitment,
re-synthesizes the rest of the witness through the circuit's callback, commits it, and carries the challenges in the instance
(bellpepper/r1cs.rs:429-461, src/r1cs/mod.rs:1516-1549). Proof == the oracle's; both verifiers accept; a wrong challenge is rejected."""
import numpy as np
import pytest

import oracle_lib as ol
from challenge_circuit import ChallengeCircuit
from spartan2_amd import hip, host

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [1, 300])
def test_prove_with_verifier_challenge_matches_oracle(K):
    ctx = hip.Context(0)
    inst = ChallengeCircuit(K)
    syn = inst.synthesize(ol.to_mont, ol.from_mont)
    tape = ol.make_tape(40 + K, 8192)
    osp = ol.OracleSpartan(inst)
    used = osp.prep_prove(tape, is_small=False)
    want, used2, _ = osp.prove(tape[used:], synthesize=syn)
    gsp = host.SpartanSNARK(ctx, inst)
    assert gsp.prep_prove(tape, is_small=False) == used
    got, gused2, _ = gsp.prove(tape[used:], synthesize=syn)
    assert gused2 == used2 and len(got) == len(want)
    assert (got == want).all()
    assert osp.verify_words(got) == 0 and gsp.verify(got) == 0
    # a second prove on the same prep state (fresh randomness -> the same challenge only if the commitments agree: they do not)
    tape2 = ol.make_tape(41 + K, 4096)
    got2 = gsp.prove(tape2, synthesize=syn)[0]
    assert (got2 == osp.prove(tape2, synthesize=syn)[0]).all()
    # the challenge is part of the instance: tampering with it is rejected by both verifiers at the instance check
    rows = (gsp.dims["num_precommitted"] + gsp.dims["num_rest"]) // 2048
    ch_off = 8 * rows + 4 * inst.num_public
    bad = got.copy()
    bad[ch_off] ^= np.uint64(1)
    assert osp.verify_words(bad) == 1 and gsp.verify(bad) == 1
    # without the callback the prover refuses
    with pytest.raises(hip.SpartanHipError):
        gsp.prove(tape2)
    gsp.close()
    ctx.close()
