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
    # the wire formats against frozen data: the vk digest (SHA-256 over SpartanVerifierKey::write_bytes) and the proof as bincode bytes
    assert sn.vk_digest.tobytes().hex() == gold["vk_digest"]
    wire = sn.proof_to_bytes(words)
    assert len(wire) == gold["wire_len"] and hashlib.sha256(wire).hexdigest() == gold["wire_sha256"] and wire[:64].hex() == gold["wire_head"]
    assert sn.verify_bytes(wire) == 0
    # and the product's proof under the independent Python-integer verifier (tests/pyverify.py: written from src/spartan.rs:469-578, no oracle, no
    # C++): keys as the product derives them, the vk digest recomputed in Python from the circuit
    import pyverify

    g, g_s = host.from_label(b"ck", 2049), host.from_label(b"ck_s", 2)
    lay = sn.proof_layout()
    lay = dict(rows_shared=lay["rows_shared"], rows_pre=lay["rows_precommitted"], rows_rest=lay["rows_rest"], num_public=lay["num_public"],
               num_challenges=lay["num_challenges"], lx=lay["rounds_x"], ly=lay["rounds_y"], nz=lay["z_len"])
    assert pyverify.verify(inst, g[:2048], g[2048], g_s[0], g_s[1], words, lay) == [int(v) for v in inst.publics]
    assert pyverify.verify_bytes(inst, g[:2048], g[2048], g_s[0], g_s[1], wire) == [int(v) for v in inst.publics]  # from the product's bincode bytes
    bad = words.copy()
    bad[-1] ^= np.uint64(1)
    with pytest.raises(pyverify.VerifyError):
        pyverify.verify(inst, g[:2048], g[2048], g_s[0], g_s[1], bad, lay)


def test_nifs_rounds_against_golden(ctx):
    """sp_nifs_* with the round hook run on the PRODUCT's transcript (hip.Transcript): polynomials, challenges, T_out and the folded layers
    match the frozen vectors — both value paths (field layers, i64 mirrors)."""
    import make_golden_impl
    import oracle_lib as ol  # pure-Python Montgomery helpers only (from_mont)

    with open(os.path.join(GOLD, "nifs_small.json")) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        n_inst, num_cons = case["n_inst"], case["num_cons"]
        left, right, E, rhos, A, B, C = make_golden_impl.nifs_inputs(n_inst, num_cons)
        total, ell_b = left * right, rhos.shape[0]
        for small in (False, True):
            nifs = hip.Nifs(ctx, n_inst, left, right)
            for which, M in enumerate((A, B, C)):
                for b in range(n_inst):
                    v = nifs.layer(which, b)
                    v.write(0, M[b])
                    v.free()
            nifs.begin(E, rhos, small_values=small)
            tr = hip.Transcript(ctx, b"golden-nifs")
            polys, r_bs = [], []
            for t in range(ell_b):
                co = nifs.round(t)
                for i in range(4):  # a scalar enters the transcript as its canonical value, 32 bytes big-endian (src/provider/traits.rs:282-286)
                    tr.absorb(b"p", ol.from_mont(co[i]).to_bytes(32, "big"))
                r = tr.squeeze(b"c")
                nifs.challenge(r)
                polys.append(co)
                r_bs.append(r)
            oa, ob, oc = (hip.Table.zeros(ctx, total) for _ in range(3))
            T_out, eq = nifs.finish(oa, ob, oc)
            assert (np.stack(polys) == _arr(case["polys"], (ell_b, 4, 4))).all() and (np.stack(r_bs) == _arr(case["r_bs"], (ell_b, 4))).all()
            assert (T_out == _arr(case["T_out"], (4,))).all() and (eq == _arr(case["eq_rho_at_rb"], (4,))).all()
            for name, tab in (("A", oa), ("B", ob), ("C", oc)):
                assert hashlib.sha256(tab.read(0, total).tobytes()).hexdigest() == case[name + "_sha256"], name
            nifs.free()


def test_neutronnova_proof_against_golden(ctx):
    """NeutronNovaZkSNARK on the device-backed driver against frozen data (no oracle in the process): vk digest - which covers the equalized step / core
    shapes and the verifier circuit's matrices -, tape use, proof words and their bincode bytes; its own verifier accepts words and bytes."""
    with open(os.path.join(GOLD, "neutronnova_small.json")) as f:
        gold = json.load(f)
    steps = [frontend.synthetic_circuit(8, 0xA5, num_public=1, witness_seed=50 + i) for i in range(3)]
    core = frontend.synthetic_circuit(2, 0xA5, num_public=1, witness_seed=7)
    tape = np.frombuffer(hashlib.shake_256(b"golden-tape-nn").digest(64 * 32768), dtype=np.uint8).reshape(32768, 64).copy()
    nn = host.NeutronNovaZkSNARK(ctx, steps, core)
    assert nn.info == gold["info"] and nn.vk_digest.tobytes().hex() == gold["vk_digest"]
    used0 = nn.prep_prove(tape)
    for reference_order in (False, True):
        words, used1, _ = nn.prove(tape[used0:], reference_order=reference_order)
        assert [used0, used1] == gold["tape_blocks"] and len(words) == gold["proof_words"]
        assert hashlib.sha256(words.tobytes()).hexdigest() == gold["proof_sha256"]
        assert words[:64].tobytes().hex() == gold["proof_head"] and words[-16:].tobytes().hex() == gold["proof_tail"]
        if not reference_order:
            nn.close()
            nn = host.NeutronNovaZkSNARK(ctx, steps, core)  # (a prove rerandomizes the prep state in place: the second driver starts from a fresh one)
            assert nn.prep_prove(tape) == used0
    wire = nn.proof_to_bytes(words)
    assert len(wire) == gold["wire_len"] and hashlib.sha256(wire).hexdigest() == gold["wire_sha256"]
    assert nn.verify(words) == 0 and nn.verify_bytes(wire) == 0
    import pynnverify  # NeutronNovaZkSNARK::verify in Python integers (no oracle, no product code beyond the generators)

    pubs = pynnverify.verify_bytes(steps[0], core, len(steps), host.from_label(b"ck", 2049), wire)
    assert pubs == ([[int(v) for v in s.publics] for s in steps], [int(v) for v in core.publics])
    nn.close()


def test_neutronnova_rest_variables_against_golden(ctx):
    """The reference's own NeutronNova test in small (circuits that live in synthesize: REST variables, src/neutronnova_zk.rs:2357-2503) on the device-backed
    driver against frozen data, no oracle in the process: vk digest, tape use, proof words, bincode bytes; its own verifier and the Python one accept."""
    with open(os.path.join(GOLD, "neutronnova_rest.json")) as f:
        gold = json.load(f)
    steps = [frontend.sha256_rest_circuit(bytes([i]) * 32) for i in range(2)]
    tape = np.frombuffer(hashlib.shake_256(b"golden-tape-nn-rest").digest(64 * 32768), dtype=np.uint8).reshape(32768, 64).copy()
    nn = host.NeutronNovaZkSNARK(ctx, steps, steps[0])
    assert nn.info == gold["info"] and nn.vk_digest.tobytes().hex() == gold["vk_digest"]
    used0 = nn.prep_prove(tape)
    words, used1, _ = nn.prove(tape[used0:])
    assert [used0, used1] == gold["tape_blocks"] and len(words) == gold["proof_words"]
    assert hashlib.sha256(words.tobytes()).hexdigest() == gold["proof_sha256"]
    wire = nn.proof_to_bytes(words)
    assert len(wire) == gold["wire_len"] and hashlib.sha256(wire).hexdigest() == gold["wire_sha256"]
    assert nn.verify(words) == 0 and nn.verify_bytes(wire) == 0
    import pynnverify

    assert pynnverify.verify_bytes(steps[0], steps[0], 2, host.from_label(b"ck", 2049), wire) == ([[0], [0]], [0])
    nn.close()
