"""GPU vs the committed golden fixtures (tests/golden/*.json): these run without consulting the oracle library at all, so the
C-ABI results are pinned to frozen data as well as to the live oracle (tests/test_gpu_*.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

from spartan2_amd import frontend, hip, host

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _arr(hexstr, shape):
    return np.frombuffer(bytes.fromhex(hexstr), dtype=np.uint64).reshape(shape)


def _inputs(ell):
    import make_golden_impl  # only its input derivation (SHAKE256 -> from_uniform) is used here

    return make_golden_impl.sumcheck_inputs(ell)


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def test_sumcheck_against_golden(ctx):
    with open(os.path.join(GOLD, "sumcheck_small.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        ell = case["ell"]
        A, B, C, taus = _inputs(ell)
        tr = hip.Transcript(ctx, b"golden")
        polys, r, fin = hip.sumcheck_cubic3(ctx, np.zeros(4, dtype=np.uint64), taus, *(hip.Table.from_host(ctx, x) for x in (A, B, C)), tr)
        assert (polys == _arr(case["cubic_polys"], (ell, 3, 4))).all()
        assert (r == _arr(case["cubic_r"], (ell, 4))).all()
        assert (fin == _arr(case["cubic_final"], (3, 4))).all()
        assert (tr.squeeze(b"after") == _arr(case["transcript_after"], (4,))).all()
        trq = hip.Transcript(ctx, b"golden-quad")
        qp, qr, qf = hip.sumcheck_quad(ctx, _arr(case["quad_claim"], (4,)).copy(), ell, hip.Table.from_host(ctx, A), hip.Table.from_host(ctx, B), trq)
        assert (qp == _arr(case["quad_polys"], (ell, 2, 4))).all() and (qr == _arr(case["quad_r"], (ell, 4))).all()
        assert (qf == _arr(case["quad_final"], (2, 4))).all()


def test_spartan_proof_against_golden(ctx):
    with open(os.path.join(GOLD, "spartan_small.json")) as f:
        gold = json.load(f)
    inst = frontend.synthetic_circuit(6, 0xDEADBEEF, num_public=3)
    assert inst.num_cons == gold["num_cons"] and inst.num_aux == gold["num_aux"]
    tape = np.frombuffer(hashlib.shake_256(b"golden-tape").digest(64 * 4096), dtype=np.uint8).reshape(4096, 64).copy()
    sn = host.SpartanSNARK(ctx, inst)
    used = sn.prep_prove(tape)
    words, used2, _ = sn.prove(tape[used:])
    assert (used, used2, len(words)) == (gold["tape_blocks_prep"], gold["tape_blocks_prove"], gold["proof_words"])
    assert hashlib.sha256(words.tobytes()).hexdigest() == gold["proof_sha256"]
    assert words[:64].tobytes().hex() == gold["proof_head"] and words[-16:].tobytes().hex() == gold["proof_tail"]
