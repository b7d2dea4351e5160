"""Builds the frozen vectors under tests/golden/ from the CPU oracle (which tests/test_oracle_kats.py pins against the
reference's own KATs). The reference itself is Rust and cannot run in this image, so these are oracle outputs frozen as data:
inputs are derived from fixed seeds, outputs are hex strings of canonical Montgomery limbs."""
import ctypes
import hashlib

import numpy as np

import oracle_lib as ol
from oracle_lib import lib as olib, p64


def _hex(arr):
    return np.ascontiguousarray(arr, dtype=np.uint64).tobytes().hex()


def _field_from_seed(seed: bytes, n: int):
    """n field elements: element i = from_uniform(SHAKE256(seed || i)[:64]) — reproducible without numpy's RNG."""
    out = np.zeros((n, 4), dtype=np.uint64)
    for i in range(n):
        raw = np.frombuffer(hashlib.shake_256(seed + i.to_bytes(4, "little")).digest(64), dtype=np.uint8).copy()
        olib().orc_field_from_uniform(0, ol.p8(raw), p64(out[i]))
    return out


def sumcheck_inputs(ell):
    n = 1 << ell
    A = _field_from_seed(b"golden-A", n)
    B = _field_from_seed(b"golden-B", n)
    C = np.zeros_like(A)
    for i in range(n):
        olib().orc_field_binop(0, 2, p64(A[i]), p64(B[i]), p64(C[i]))
    taus = _field_from_seed(b"golden-tau", ell)
    return A, B, C, taus


def sumcheck_small():
    out = {"note": "oracle outputs (oracle/sumcheck.hpp) on SHAKE256-derived inputs; see tests/make_golden_impl.py", "cases": []}
    for ell in (3, 6, 9):
        A, B, C, taus = sumcheck_inputs(ell)
        claim = np.zeros(4, dtype=np.uint64)
        tr = ol.Transcript(b"golden")
        polys = np.zeros((ell, 3, 4), dtype=np.uint64)
        r = np.zeros((ell, 4), dtype=np.uint64)
        fin = np.zeros((3, 4), dtype=np.uint64)
        a, b, c = A.copy(), B.copy(), C.copy()
        assert olib().orc_sumcheck_cubic3(p64(claim), p64(taus), ctypes.c_size_t(ell), p64(a), p64(b), p64(c), tr.h, p64(polys), p64(r), p64(fin)) == 0
        after = tr.squeeze(b"after")
        # quadratic: claim = <A, B>
        qclaim = np.zeros(4, dtype=np.uint64)
        olib().orc_field_dot(0, p64(A), p64(B), ctypes.c_size_t(1 << ell), p64(qclaim))
        trq = ol.Transcript(b"golden-quad")
        qpolys = np.zeros((ell, 2, 4), dtype=np.uint64)
        qr = np.zeros((ell, 4), dtype=np.uint64)
        qfin = np.zeros((2, 4), dtype=np.uint64)
        a, b = A.copy(), B.copy()
        full = ctypes.c_size_t(2**64 - 1)
        assert olib().orc_sumcheck_quad(p64(qclaim), ctypes.c_size_t(ell), p64(a), full, full, p64(b), full, full, trq.h, p64(qpolys), p64(qr), p64(qfin)) == 0
        out["cases"].append({"ell": ell, "cubic_polys": _hex(polys), "cubic_r": _hex(r), "cubic_final": _hex(fin), "transcript_after": _hex(after),
                             "quad_claim": _hex(qclaim), "quad_polys": _hex(qpolys), "quad_r": _hex(qr), "quad_final": _hex(qfin)})
    return out


def spartan_small():
    """One full proof of a seeded synthetic circuit: digest of the proof words + the first words, enough to pin the whole path."""
    from spartan2_amd import frontend

    inst = frontend.synthetic_circuit(6, 0xDEADBEEF, num_public=3)
    tape = np.frombuffer(hashlib.shake_256(b"golden-tape").digest(64 * 4096), dtype=np.uint8).reshape(4096, 64).copy()
    sp = ol.OracleSpartan(inst)
    used = sp.prep_prove(tape)
    words, used2, _ = sp.prove(tape[used:])
    assert sp.verify_words(words) == 0
    wire = sp.proof_to_bytes(words)
    return {"note": "oracle proof (oracle/spartan.hpp) of frontend.synthetic_circuit(6, 0xDEADBEEF, 3) with tape = SHAKE256('golden-tape'); vk_digest = SHA-256 "
                    "over SpartanVerifierKey::write_bytes, wire_* = the proof as bincode bytes of SpartanSNARK (oracle/wire.hpp states the framing)",
            "num_cons": inst.num_cons, "num_aux": inst.num_aux, "tape_blocks_prep": used, "tape_blocks_prove": used2, "proof_words": len(words),
            "proof_sha256": hashlib.sha256(words.tobytes()).hexdigest(), "proof_head": _hex(words[:64]), "proof_tail": _hex(words[-16:]),
            "vk_digest": sp.export_keys()[4].tobytes().hex(), "wire_len": len(wire), "wire_sha256": hashlib.sha256(wire).hexdigest(), "wire_head": wire[:64].hex()}


def neutronnova_small():
    """One NeutronNovaZkSNARK proof: three step circuits of 8 groups and a core circuit of 2 groups (so SplitR1CSShape::equalize is in the path), seeded
    tape. Pins the vk digest (keys | equalized shapes | verifier-circuit shapes), the proof words and their bincode bytes."""
    from spartan2_amd import frontend

    steps = [frontend.synthetic_circuit(8, 0xA5, num_public=1, witness_seed=50 + i) for i in range(3)]
    core = frontend.synthetic_circuit(2, 0xA5, num_public=1, witness_seed=7)
    tape = np.frombuffer(hashlib.shake_256(b"golden-tape-nn").digest(64 * 32768), dtype=np.uint8).reshape(32768, 64).copy()
    nn = ol.OracleNeutronNova(steps, core)
    words, used, _ = nn.prove(tape)
    assert nn.verify_words(words) == 0
    wire = nn.proof_to_bytes(words)
    return {"note": "oracle proof (oracle/neutronnova_zk.hpp) of 3 x synthetic_circuit(8, 0xA5, 1, witness_seed 50 + i) + core synthetic_circuit(2, 0xA5, 1, witness_seed 7), "
                    "tape = SHAKE256('golden-tape-nn'); vk_digest = SHA-256 over NeutronNovaVerifierKey::write_bytes, wire_* = the proof as bincode bytes",
            "info": nn.info, "tape_blocks": [int(used[0]), int(used[1])], "proof_words": len(words), "proof_sha256": hashlib.sha256(words.tobytes()).hexdigest(),
            "proof_head": _hex(words[:64]), "proof_tail": _hex(words[-16:]), "vk_digest": nn.digest().tobytes().hex(), "wire_len": len(wire),
            "wire_sha256": hashlib.sha256(wire).hexdigest()}


def neutronnova_rest():
    """The reference's own NeutronNova test in small (test_neutron_sha256, src/neutronnova_zk.rs:2357-2503: circuits that live in synthesize): two SHA-256
    circuits over 32-byte preimages [i; 32] with REST variables only, core = the first one. Pins what the rest-segment path adds: the rest rows committed
    inside prove, the full-width witness fold."""
    from spartan2_amd import frontend

    steps = [frontend.sha256_rest_circuit(bytes([i]) * 32) for i in range(2)]
    tape = np.frombuffer(hashlib.shake_256(b"golden-tape-nn-rest").digest(64 * 32768), dtype=np.uint8).reshape(32768, 64).copy()
    nn = ol.OracleNeutronNova(steps, steps[0])
    words, used, _ = nn.prove(tape)
    assert nn.verify_words(words) == 0
    wire = nn.proof_to_bytes(words)
    return {"note": "oracle proof of 2 x sha256_rest_circuit([i; 32]) + core = the first, tape = SHAKE256('golden-tape-nn-rest')",
            "info": nn.info, "tape_blocks": [int(used[0]), int(used[1])], "proof_words": len(words), "proof_sha256": hashlib.sha256(words.tobytes()).hexdigest(),
            "vk_digest": nn.digest().tobytes().hex(), "wire_len": len(wire), "wire_sha256": hashlib.sha256(wire).hexdigest()}


# ---- NeutronNova NIFS rounds (oracle/nifs.hpp) -----------------------------------------------------------------------------------
def nifs_inputs(n_inst, num_cons):
    """Layers with SHA-like small entries (A in {-2..2}, B bits, C = A o B) and a few full-size ones; E from a SHAKE-derived tau."""
    ell, left, right = ol.tensor_decomp(num_cons)
    total = left * right
    P = ol.MODULI[0]
    raw = np.frombuffer(hashlib.shake_256(b"golden-nifs-%d-%d" % (n_inst, num_cons)).digest(2 * n_inst * total), dtype=np.uint8).reshape(2, n_inst, total)
    a = (raw[0] % 5).astype(np.int64) - 2
    b = (raw[1] & 1).astype(np.int64)
    lut = {v: ol.to_mont(v % P) for v in range(-2, 3)}
    A = np.stack([np.stack([lut[int(v)] for v in row]) for row in a])
    B = np.stack([np.stack([lut[int(v)] for v in row]) for row in b])
    C = np.stack([np.stack([lut[int(v)] for v in row]) for row in a * b])
    big = _field_from_seed(b"golden-nifs-big", 4)  # two full-size pairs: positions that take the large-value corrections
    for j, (i, k) in enumerate(((0, 1), (n_inst - 1, total - 2))):
        A[i, k], B[i, k] = big[2 * j], big[2 * j + 1]
        olib().orc_field_binop(0, 2, p64(A[i, k]), p64(B[i, k]), p64(C[i, k]))
    tau = _field_from_seed(b"golden-nifs-tau", 1)[0]
    E = ol.pow_split_evals(tau, ell, left, right)
    rhos = _field_from_seed(b"golden-nifs-rho", n_inst.bit_length() - 1)
    return left, right, E, rhos, A, B, C


def nifs_small():
    out = {"note": "oracle outputs (oracle/nifs.hpp nifs_prove_core, cached-i64 branch) on SHAKE256-derived inputs; round hook = transcript "
                   "b'golden-nifs': absorb the 4 coefficients under b'p', squeeze b'c' (tests/oracle_lib.py transcript_round_hook)", "cases": []}
    for n_inst, num_cons in ((4, 64), (8, 512)):
        left, right, E, rhos, A, B, C = nifs_inputs(n_inst, num_cons)
        o = ol.nifs_prove_core(left, right, E, rhos, A, B, C, True, ol.transcript_round_hook(ol.Transcript(b"golden-nifs")))
        out["cases"].append({"n_inst": n_inst, "num_cons": num_cons, "polys": _hex(o["polys"]), "r_bs": _hex(o["r_bs"]), "T_out": _hex(o["T_out"]),
                             "eq_rho_at_rb": _hex(o["eq_rho_at_rb"]), "A_sha256": hashlib.sha256(o["A"].tobytes()).hexdigest(),
                             "B_sha256": hashlib.sha256(o["B"].tobytes()).hexdigest(), "C_sha256": hashlib.sha256(o["C"].tobytes()).hexdigest(),
                             "A_head": _hex(o["A"][:2]), "C_head": _hex(o["C"][:2])})
    return out
