"""CPU: the oracle's wire formats and key digests (oracle/wire.hpp, oracle/wire_formats.hpp) against what pins them:
SHA-256 against FIPS 180-4 vectors and hashlib; the bincode framing and the vk digest stream against the independent Python writer
tests/pywire.py (written from the reference's struct definitions, src/digest.rs:22-77, src/spartan.rs:62-137); round trips; rejection of
malformed bytes. The byte layout of third-party point / field types is the one documented assumption (oracle/wire.hpp header)."""
import hashlib

import numpy as np
import pytest

import oracle_lib as ol
import pywire
from spartan2_amd import frontend


def test_sha256_fips_vectors_and_hashlib():
    # FIPS 180-4 / NIST example vectors
    assert ol.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    assert ol.sha256(b"").hex() == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    assert ol.sha256(b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq").hex() == "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"
    assert ol.sha256(b"a" * 1000000).hex() == "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0"
    rng = np.random.default_rng(5)
    for n in list(range(0, 130)) + [255, 256, 257, 4095, 65536 + 7]:
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert ol.sha256(data) == hashlib.sha256(data).digest(), n


def _layout(sp):
    sh = sp.shape
    rows = lambda n: -(-n // 2048)
    nz = min(2048, sh.num_vars)
    return dict(rows_shared=rows(sh.num_shared), rows_pre=rows(sh.num_precommitted), rows_rest=rows(sh.num_rest), num_public=sh.num_public,
                num_challenges=sh.num_challenges, lx=sh.num_cons.bit_length() - 1, ly=sh.num_vars.bit_length(), nz=nz)


def _prove(inst, seed, synthesize=None):
    sp = ol.OracleSpartan(inst)
    tape = ol.make_tape(seed, 8192)
    used = sp.prep_prove(tape)
    words, _, _ = sp.prove(tape[used:], synthesize=synthesize)
    assert sp.verify_words(words) == 0
    return sp, words


CASES = {
    "precommitted_only": lambda: frontend.synthetic_circuit(12, 0xDEADBEEF, num_public=4),
    "all_three_segments": lambda: frontend.synthetic_circuit(9, 0xBEEF, num_public=2, shared_permille=200, precommitted_permille=500),
    "rest_only": lambda: frontend.cubic_circuit(),  # the reference's own e2e circuit (src/spartan.rs:587-651)
}


def _case(name):
    return CASES[name]()


@pytest.mark.parametrize("name", list(CASES))
def test_spartan_proof_bytes_match_the_python_writer_and_round_trip(name):
    inst = _case(name)
    sp, words = _prove(inst, 3)
    data = sp.proof_to_bytes(words)
    assert data == pywire.spartan_proof_bytes(words, **_layout(sp))
    back = sp.proof_from_bytes(data)
    assert back is not None and (back == words).all()
    assert sp.verify_words(back) == 0
    # bincode rejects trailing bytes and truncated input; a scalar >= the modulus is not a field element
    assert sp.proof_from_bytes(data + b"\0") is None
    for cut in (1, 31, 32, 97, len(data) // 2):
        assert sp.proof_from_bytes(data[:-cut]) is None
    bad = bytearray(data)
    bad[-32:] = (pywire.P_SCALAR).to_bytes(32, "little")  # z_beta := p
    assert sp.proof_from_bytes(bytes(bad)) is None
    bad = bytearray(data)
    bad[-1] ^= 0x40  # still canonical (top byte of p is 0xff): decodes, but to another proof, which the verifier rejects
    other = sp.proof_from_bytes(bytes(bad))
    assert other is not None and sp.verify_words(other) != 0
    # a length prefix larger than the input must not allocate or read out of bounds
    bad = bytearray(data)
    off = 1 if _layout(sp)["rows_shared"] == 0 else 0
    bad[off + 1:off + 9] = (1 << 60).to_bytes(8, "little")
    assert sp.proof_from_bytes(bytes(bad)) is None


def test_spartan_proof_accepts_any_jacobian_representative():
    """The reference writes E::GE in whatever representative its arithmetic left; readers must map (X, Y, Z) to the same group element."""
    inst = frontend.synthetic_circuit(6, 21, num_public=1)
    sp, words = _prove(inst, 8)
    data = bytearray(sp.proof_to_bytes(words))
    lay = _layout(sp)
    assert lay["rows_shared"] == 0 and lay["rows_pre"] >= 1
    off = 1 + 1 + 8  # None tag, Some tag, Vec length: first point of comm_W_precommitted
    P = pywire.P_BASE
    x, y, z = (int.from_bytes(data[off + 32 * i:off + 32 * i + 32], "little") for i in range(3))
    assert z == 1
    lam = 0x1234567890ABCDEF1234567
    X, Y, Z = x * lam * lam % P, y * pow(lam, 3, P) % P, lam
    for i, v in enumerate((X, Y, Z)):
        data[off + 32 * i:off + 32 * i + 32] = v.to_bytes(32, "little")
    back = sp.proof_from_bytes(bytes(data))
    assert back is not None and (back == words).all()
    data[off:off + 32] = ((X + 1) % P).to_bytes(32, "little")  # off the curve
    assert sp.proof_from_bytes(bytes(data)) is None


@pytest.mark.parametrize("name", list(CASES))
def test_spartan_vk_digest_is_sha256_of_the_reference_stream(name):
    inst = _case(name)
    sp = ol.OracleSpartan(inst)
    ck, h, ck_s, h_s, dig = sp.export_keys()
    assert dig.tobytes() == pywire.spartan_vk_digest(inst, ck, h, ck_s, h_s)
    # the key as a serde value: vk_ee | ck_s | S, every part bincode; starts with num_cols and the Vec length of the generators
    vk = sp.vk_bytes()
    assert vk[:16] == (2048).to_bytes(8, "little") * 2
    w = pywire.Writer()
    w.hyrax_key(ck, h)
    w.hyrax_key(ck_s, h_s)
    assert vk.startswith(w.bytes())


def test_vk_digest_separates_keys_and_shapes():
    a = ol.OracleSpartan(frontend.synthetic_circuit(6, 1, num_public=1)).export_keys()[4].tobytes()
    b = ol.OracleSpartan(frontend.synthetic_circuit(6, 2, num_public=1)).export_keys()[4].tobytes()
    c = ol.OracleSpartan(frontend.synthetic_circuit(6, 1, num_public=1)).export_keys()[4].tobytes()
    assert a == c and a != b


@pytest.mark.parametrize("groups,core_groups", [(8, 8), (30, 2)])
def test_neutronnova_proof_bytes_match_the_python_writer_and_round_trip(groups, core_groups):
    """(30, 2): two rows of precommitted step variables against one in the core — after equalize the core's rows split 1 | 1 where a step's split 2 | 0"""
    steps = [frontend.synthetic_circuit(groups, 0xA5, num_public=1, witness_seed=50 + i) for i in range(3)]
    core = frontend.synthetic_circuit(core_groups, 0xA5, num_public=1, witness_seed=999)
    nn = ol.OracleNeutronNova(steps, core)
    words, _, _ = nn.prove(ol.make_tape(23, 16384))
    assert nn.verify_words(words) == 0
    data = nn.proof_to_bytes(words)
    back = nn.proof_from_bytes(data)
    assert back is not None and (back == words).all()
    assert nn.proof_from_bytes(data[:-1]) is None and nn.proof_from_bytes(data + b"\1") is None
    info, W = nn.info, 32
    rows = lambda n: -(-n // 2048)
    (ds, _, _), (dc, _, _) = pywire.equalize(pywire.pad_shape(steps[0]), pywire.pad_shape(core))
    vc = ol.verifier_circuit_rounds(info["nb"], info["nx"], info["ny"], W)
    want = pywire.nn_proof_bytes(
        words, rows(ds["num_shared"]), rows(ds["num_precommitted"]), rows(ds["num_rest"]), len(steps), ds["num_public"], dc["num_public"],
        min(2048, ds["num_shared"] + ds["num_precommitted"] + ds["num_rest"]), rows_pre_core=rows(dc["num_precommitted"]), rows_rest_core=rows(dc["num_rest"]),
        vc_rows_per_round=[v // W for v in vc["vars_padded"]], vc_public=info["vc_public"], vc_chals_per_round=vc["challenges"], vc_cons_rows=info["vc_cons"] // W,
        vc_io=sum(vc["challenges"]) + info["vc_public"], lx=info["vc_cons"].bit_length() - 1, ly=(1 << (info["vc_vars"] - 1).bit_length()).bit_length(), width=W)
    assert data == want
