"""CPU: the oracle's NeutronNovaZkSNARK (oracle/neutronnova_zk.hpp: verifier circuit, multi-round commitments, NovaNIFS, relaxed Spartan, folded opening)
is self-consistent: prove -> serialize -> deserialize -> verify accepts; a flipped bit anywhere in the proof is rejected. This is what pins the ZK
wrapper's restatement in the absence of a runnable reference (the GPU tests then compare the product's proof with this one word for word)."""
import numpy as np
import pytest

import oracle_lib as ol
from spartan2_amd import frontend


def _insts(n, n_groups=8):
    steps = [frontend.synthetic_circuit(n_groups, 0xA5, num_public=1, witness_seed=50 + i) for i in range(n)]
    core = frontend.synthetic_circuit(n_groups, 0xA5, num_public=1, witness_seed=999)
    return steps, core


@pytest.mark.parametrize("n", [2, 3])
def test_prove_verify_and_tamper(n):
    steps, core = _insts(n)
    nn = ol.OracleNeutronNova(steps, core)
    assert nn.info["vc_public"] == 6 and nn.info["vc_rounds"] == nn.info["nb"] + 1 + nn.info["nx"] + 1 + nn.info["ny"] + 1 + 2
    tape = ol.make_tape(17 + n, 16384)
    words, used, _ = nn.prove(tape)
    assert nn.verify_words(words) == 0
    # deterministic in the tape
    again, used2, _ = nn.prove(tape)
    assert used == used2 and (again == words).all()
    rng = np.random.default_rng(n)
    for pos in [0, 9, len(words) // 5, len(words) // 3, len(words) // 2, 2 * len(words) // 3, len(words) - 1] + list(rng.integers(0, len(words), size=8)):
        bad = words.copy()
        bad[int(pos)] ^= np.uint64(1 << 3)
        assert nn.verify_words(bad) != 0, int(pos)
