"""CPU: the NeutronNova verifier circuit of the oracle against an independent Python synthesis (tests/pyvcircuit.py, written from src/zk.rs:473-943,
src/bellpepper/r1cs.rs:606-693 and src/r1cs/mod.rs:1555-1672): the MATRICES, not only their counts — the whole NeutronNovaVerifierKey digest
(keys | S_step | S_core after SplitR1CSShape::equalize | vc_shape | vc_shape_regular | vc keys, src/neutronnova_zk.rs:1305-1333) recomputed in Python
equals the oracle's, so every coefficient, column index, row pointer and dimension of the verifier circuit's three matrices agrees byte for byte."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
import pyvcircuit as pvc
from spartan2_amd import frontend


@pytest.fixture(scope="module")
def gens():
    g = np.zeros((2049, 8), dtype=np.uint64)
    ol.lib().orc_from_label(b"ck", ctypes.c_size_t(2049), ol.p64(g))
    return g


@pytest.mark.parametrize("n,groups,core_groups", [(2, 8, 8), (3, 8, 2), (5, 3, 3), (16, 8, 8), (2, 30, 30)])
def test_vk_digest_recomputed_in_python(gens, n, groups, core_groups):
    steps = [frontend.synthetic_circuit(groups, 0xA5, num_public=1, witness_seed=50 + i) for i in range(n)]
    core = frontend.synthetic_circuit(core_groups, 0xA5, num_public=1, witness_seed=999)
    nn = ol.OracleNeutronNova(steps, core)
    dig, sh = pvc.nn_vk_digest(steps[0], core, n, gens)
    assert dig == nn.digest().tobytes()
    info = nn.info
    assert (sh["num_rounds"], sh["num_cons_unpadded"], sh["num_cons"], sum(sh["vars_padded"]), sh["num_public"]) == (
        info["vc_rounds"], info["vc_cons_unpadded"], info["vc_cons"], info["vc_vars"], info["vc_public"])
    c = ol.verifier_circuit_counts(info["nb"], info["nx"], info["ny"], 32)  # the hand-derived counts (tests/golden/reference_kats.json)
    assert (sh["num_rounds"], sh["num_cons_unpadded"], sum(sh["vars_padded"]), sh["num_public"]) == (c["rounds"], c["constraints"], c["vars_padded"], c["public"])


def test_python_circuit_at_config_3_dimensions():
    """nb = 5, nx = 15, ny = 16 (32 Sha256StepCircuit instances, benches/sha256_neutronnova.rs) without building the SHA circuits"""
    sh = pvc.multiround_shape(pvc.VerifierCircuit(5, 15, 16, 32))
    c = ol.verifier_circuit_counts(5, 15, 16, 32)
    assert (sh["num_rounds"], sh["num_cons_unpadded"], sum(sh["vars_padded"]), sh["num_public"]) == (c["rounds"], c["constraints"], c["vars_padded"], c["public"])
    assert sh["chals_per_round"] == [1] * 5 + [0] + [1] * (15 + 1 + 16) + [0, 0, 0]
    # every row pointer array ends at its number of entries; every column is inside the matrix
    for data, idx, ptr, cols in sh["mats"]:
        assert len(ptr) == sh["num_cons"] + 1 and ptr[-1] == len(idx) == len(data) and max(idx) < cols


def test_the_row_order_is_part_of_the_digest():
    """What the third-party assumption (LinearCombination::iter: inputs first, then aux, by index) moves: with the terms in the order the gadgets
    write them the matrices hold the same entries per row but other bytes."""
    sh = pvc.multiround_shape(pvc.VerifierCircuit(1, 3, 4, 32))

    class InsertionLC(pvc.LC):
        def __init__(self):
            super().__init__()
            self.order = []

        def add(self, var, coeff=1):
            if var not in self.order:
                self.order.append(var)
            return super().add(var, coeff)

        def terms(self):
            return [(v, (self.inputs if v[0] == "in" else self.aux)[v[1]]) for v in self.order]

    saved = pvc.LC
    try:
        pvc.LC = InsertionLC
        other = pvc.multiround_shape(pvc.VerifierCircuit(1, 3, 4, 32))
    finally:
        pvc.LC = saved
    assert pvc.multiround_shape_bytes(other) != pvc.multiround_shape_bytes(sh)
    for (d0, i0, p0, _), (d1, i1, p1, _) in zip(sh["mats"], other["mats"]):
        assert p0 == p1
        for r in range(len(p0) - 1):
            assert sorted(zip(i0[p0[r]:p0[r + 1]], d0[p0[r]:p0[r + 1]])) == sorted(zip(i1[p1[r]:p1[r + 1]], d1[p1[r]:p1[r + 1]]))


def test_golden_neutronnova_digest_recomputed_in_python(gens):
    """the frozen vk digest of tests/golden/neutronnova_small.json (which the GPU suite holds the product to, without the oracle) is the SHA-256 that the
    Python synthesis + Python equalize + Python framing produce"""
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "neutronnova_small.json")) as f:
        gold = json.load(f)
    steps0 = frontend.synthetic_circuit(8, 0xA5, num_public=1, witness_seed=50)
    core = frontend.synthetic_circuit(2, 0xA5, num_public=1, witness_seed=7)
    assert pvc.nn_vk_digest(steps0, core, 3, gens)[0].hex() == gold["vk_digest"]
