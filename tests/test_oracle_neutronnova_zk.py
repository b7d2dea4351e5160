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


def _assert_counts(info):
    """The verifier circuit's shape against the counts derived by hand from src/zk.rs (tests/golden/reference_kats.json): what narrows the common-mode
    hole of two restatements (oracle, product) written from one reading of the reference."""
    c = ol.verifier_circuit_counts(info["nb"], info["nx"], info["ny"], 32)
    assert info["vc_rounds"] == c["rounds"]
    assert info["vc_cons_unpadded"] == c["constraints"]
    assert info["vc_vars"] == c["vars_padded"]
    assert info["vc_public"] == c["public"]
    assert info["vc_cons"] == 1 << (c["constraints"] - 1).bit_length()


@pytest.mark.parametrize("n,groups", [(2, 8), (4, 3), (8, 3), (16, 8)])
def test_verifier_circuit_counts_match_the_hand_derivation(n, groups):
    steps, core = _insts(n, groups)
    _assert_counts(ol.OracleNeutronNova(steps, core).info)


def test_verifier_circuit_counts_at_config_3():
    # 32 one-compression SHA-256 step circuits + core (benches/sha256_neutronnova.rs): nb = 5, nx = 15, ny = 16
    circs = [frontend.sha256_step_circuit(bytes([i]) * 64) for i in range(32)]
    info = ol.OracleNeutronNova(circs, frontend.sha256_step_circuit(bytes(64))).info
    assert (info["nb"], info["nx"], info["ny"]) == (5, 15, 16)
    _assert_counts(info)
    assert info["vc_cons_unpadded"] == 2 + 4 * 4 + 4 + 3 + 8 * 14 + 10 + 7 + 6 * 15 + 12 + 64
