"""GPU parity for SURVEY 8(f) rank 1: NeutronNovaZkSNARK::{setup, prep_prove, prove, verify} on the device-backed driver (spartan2_amd/host/neutronnova_zk.cpp:
verifier circuit + process_round commitments, NeutronNovaNIFS, batched outer / inner sum-checks, NovaNIFS with a random relaxed instance,
RelaxedR1CSSpartanProof, folded Hyrax opening) against the oracle's restatement (oracle/neutronnova_zk.hpp): identical vk digest, identical proof words
on the same circuits and randomness tape, and the oracle's NeutronNovaZkSNARK::verify accepts the device's proof. BASELINE config 3 at its own size:
32 Sha256StepCircuit instances + the core circuit (benches/sha256_neutronnova.rs)."""
import numpy as np
import pytest

import oracle_lib as ol
from spartan2_amd import frontend, hip, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def _both(ctx, steps, core, seed):
    onn = ol.OracleNeutronNova(steps, core)
    tape = ol.make_tape(seed, 32768)
    want, used, secs = onn.prove(tape)
    gnn = host.NeutronNovaZkSNARK(ctx, steps, core)
    assert gnn.info == onn.info and (gnn.vk_digest == onn.digest()).all()
    # ... and equal to the digest recomputed in Python from an independent synthesis of the verifier circuit (tests/pyvcircuit.py, from src/zk.rs:473-943):
    # the product's verifier-circuit MATRICES, the equalized step / core shapes and the key framing, byte for byte
    import pyvcircuit

    assert gnn.vk_digest.tobytes() == pyvcircuit.nn_vk_digest(steps[0], core, len(steps), host.from_label(b"ck", 2049))[0]
    c = ol.verifier_circuit_counts(gnn.info["nb"], gnn.info["nx"], gnn.info["ny"], 32)  # hand-derived from src/zk.rs (tests/golden/reference_kats.json)
    assert (gnn.info["vc_rounds"], gnn.info["vc_cons_unpadded"], gnn.info["vc_vars"], gnn.info["vc_public"]) == (c["rounds"], c["constraints"], c["vars_padded"], c["public"])
    assert gnn.prep_prove(tape) == used[0]
    got, used_g, phases = gnn.prove(tape[used[0]:])
    assert used_g == used[1] and len(got) == len(want)
    return onn, gnn, want, got, tape, used, phases


@pytest.mark.parametrize("n,groups", [(2, 8), (3, 30), (8, 30)])
def test_prove_matches_oracle_synthetic_steps(ctx, n, groups):
    steps = [frontend.synthetic_circuit(groups, 0xA5, num_public=1, witness_seed=50 + i) for i in range(n)]
    core = frontend.synthetic_circuit(groups, 0xA5, num_public=1, witness_seed=999)
    onn, gnn, want, got, tape, used, _ = _both(ctx, steps, core, 60 + n)
    assert (got == want).all()
    assert onn.verify_words(got) == 0 and gnn.verify(got) == 0
    # the prep state is rerandomized in place by every prove: a second prove is a different, equally valid proof
    got2, _, _ = gnn.prove(tape[used[0] + used[1]:])
    assert not (got2 == got).all() and onn.verify_words(got2) == 0 and gnn.verify(got2) == 0
    bad = got.copy()
    bad[len(bad) // 2] ^= np.uint64(1 << 9)
    assert onn.verify_words(bad) != 0 and gnn.verify(bad) != 0
    gnn.close()


@pytest.mark.parametrize("n,groups,core_groups", [(3, 8, 2), (2, 2, 9), (2, 30, 2), (3, 2, 30)])
def test_step_and_core_shapes_of_different_size(ctx, n, groups, core_groups):
    """SplitR1CSShape::equalize in setup (src/neutronnova_zk.rs:1413, src/r1cs/mod.rs:913-971): a core circuit with fewer (or more) constraints than the
    step circuit — in the last two cases by more than a commitment row, so that step and core split the equalized variables into precommitted | rest
    segments of different sizes (4096 | 0 against 2048 | 2048). Same vk digest, same proof words as the oracle, both verifiers accept, the wire bytes round-trip; both drivers (side jobs and the
    reference-order one) produce it. (tests/test_equalize_cpu.py pins the equalized matrices against a Python restatement.)"""
    steps = [frontend.synthetic_circuit(groups, 0xA5, num_public=1, witness_seed=50 + i) for i in range(n)]
    core = frontend.synthetic_circuit(core_groups, 0xA5 + (core_groups > groups), num_public=1, witness_seed=7)
    onn, gnn, want, got, tape, used, _ = _both(ctx, steps, core, 75 + n)
    assert (got == want).all()
    assert onn.verify_words(got) == 0 and gnn.verify(got) == 0
    assert gnn.proof_to_bytes(got) == onn.proof_to_bytes(want) and gnn.verify_bytes(gnn.proof_to_bytes(got)) == 0
    # ... and a third verifier, Python integers written from the reference's verify alone (tests/pynnverify.py), accepts the product's bytes
    import pynnverify

    pubs = pynnverify.verify_bytes(steps[0], core, n, host.from_label(b"ck", 2049), gnn.proof_to_bytes(got))
    assert pubs == ([[int(v) for v in s.publics] for s in steps], [int(v) for v in core.publics])
    gnn.close()
    with pytest.raises(Exception, match="at least two step circuits"):  # zero NIFS rounds: the reference's setup panics (src/zk.rs:637-641)
        host.NeutronNovaZkSNARK(ctx, steps[:1], core)
    ref = host.NeutronNovaZkSNARK(ctx, steps, core)
    assert ref.prep_prove(tape) == used[0]
    again, _, _ = ref.prove(tape[used[0]:], reference_order=True)
    assert (again == want).all()
    ref.close()
    # one shared commitment serves every circuit: a different padded shared segment cannot be equalized
    sh = lambda g: frontend.synthetic_circuit(g, 0x77, num_public=1, shared_permille=900, precommitted_permille=1000, witness_seed=5)
    with pytest.raises(Exception, match="different padded shared segments"):
        host.NeutronNovaZkSNARK(ctx, [sh(30), sh(30)], sh(8))


@pytest.mark.parametrize("n,groups", [(2, 8), (5, 30)])
def test_verify_rejects_what_the_oracle_rejects(ctx, n, groups):
    """NeutronNovaZkSNARK::verify on the device-backed driver (src/neutronnova_zk.rs:2096-2343): accepts the oracle's proof and its own; a low bit
    flipped at positions spread over every section of the proof (instances, opening argument, per-round commitments, public values, challenges, NIFS
    commitment, random instance, both relaxed sum-checks, direct openings) is rejected with the SAME check index the oracle's verifier reports; a
    truncated proof and a non-canonical scalar are rejected as malformed."""
    steps = [frontend.synthetic_circuit(groups, 0x3C, num_public=1, witness_seed=150 + i) for i in range(n)]
    core = frontend.synthetic_circuit(groups, 0x3C, num_public=1, witness_seed=1999)
    onn, gnn, want, got, _, _, _ = _both(ctx, steps, core, 160 + n)
    assert (got == want).all() and gnn.verify(want) == 0
    rng = np.random.default_rng(7 + n)
    # scalars only: a flipped coordinate bit leaves the curve (the oracle's loader does not check that; the device driver answers 1)
    positions = sorted(set(int(x) for x in rng.integers(0, len(got), size=60)) | {0, 5, len(got) - 1, len(got) - 5, len(got) // 2})
    agree, seen = 0, {}
    for pos in positions:
        bad = got.copy()
        bad[pos] ^= np.uint64(1)
        want_rc, got_rc = onn.verify_words(bad), gnn.verify(bad)
        assert got_rc != 0, pos
        assert want_rc != 0, pos
        if got_rc != 1:  # 1 = the stricter encoding checks (off-curve point, non-canonical limb) fire before the oracle's first check would
            assert got_rc == want_rc, (pos, got_rc, want_rc)
            agree += 1
        seen[got_rc] = seen.get(got_rc, 0) + 1
    print("verify: failed-check histogram over", len(positions), "tampered proofs:", dict(sorted(seen.items())))
    assert agree >= 10 and {2, 4, 6} <= set(seen)
    assert gnn.verify(got[:-4]) == 1
    bad = got.copy()
    bad[-4:] = np.uint64(0xFFFFFFFFFFFFFFFF)  # blind_E >= the modulus
    assert gnn.verify(bad) == 1
    # the same on serialised proofs: every tampered proof that still decodes fails the check its flat form fails; malformed bytes fail check 1
    data = gnn.proof_to_bytes(got)
    assert data == onn.proof_to_bytes(got) and gnn.verify_bytes(data) == 0
    for pos in positions[:12]:
        bad = got.copy()
        bad[pos] ^= np.uint64(1)
        rc = gnn.verify(bad)
        if rc != 1:
            assert gnn.verify_bytes(gnn.proof_to_bytes(bad)) == rc, pos
    assert gnn.verify_bytes(data[:-1]) == 1 and gnn.verify_bytes(data + b"\0") == 1 and gnn.verify_bytes(b"") == 1
    b = bytearray(data)
    b[-32:] = bytes([0xFF]) * 32
    assert gnn.verify_bytes(bytes(b)) == 1
    gnn.close()


@pytest.mark.parametrize("n,groups", [(3, 30), (8, 30)])
def test_reference_order_driver_is_the_same_proof(ctx, n, groups):
    """nnz_prove_reference_order: one thread, ABI calls in the statement order of src/neutronnova_zk.rs:1609-2093, PCS::prove as ONE sp_hyrax_prove —
    what an unchanged neutronnova_zk.rs over the shim gets. Word for word the oracle's proof, like the default driver's."""
    steps = [frontend.synthetic_circuit(groups, 0xA5, num_public=1, witness_seed=50 + i) for i in range(n)]
    core = frontend.synthetic_circuit(groups, 0xA5, num_public=1, witness_seed=999)
    onn = ol.OracleNeutronNova(steps, core)
    tape = ol.make_tape(60 + n, 32768)
    want, used, _ = onn.prove(tape)
    gnn = host.NeutronNovaZkSNARK(ctx, steps, core)
    assert gnn.prep_prove(tape) == used[0]
    got, used_g, phases = gnn.prove(tape[used[0]:], reference_order=True)
    assert used_g == used[1] and (got == want).all()
    assert onn.verify_words(got) == 0 and gnn.verify(got) == 0
    gnn.close()


def test_shared_and_precommitted_segments(ctx):
    mk = lambda ws: frontend.synthetic_circuit(30, 0x77, num_public=2, shared_permille=300, precommitted_permille=1000, witness_seed=ws)
    # every circuit shares step 0's shared witness (src/neutronnova_zk.rs:1485-1488): build the others with the same shared segment
    steps = [mk(5), mk(5), mk(5)]
    core = mk(5)
    onn, gnn, want, got, _, _, _ = _both(ctx, steps, core, 71)
    assert (got == want).all() and onn.verify_words(got) == 0 and gnn.verify(got) == 0
    data = gnn.proof_to_bytes(got)  # Some(comm_W_shared) on the wire
    assert data[0] == 1 and data == onn.proof_to_bytes(want) and gnn.verify_bytes(data) == 0
    gnn.close()


@pytest.mark.parametrize("kind", ["synthetic_rest_only", "reference_test_circuit_7x32B"])
def test_rest_variables(ctx, kind):
    """Circuits that live in SpartanCircuit::synthesize — REST variables, committed inside prove (bellpepper/r1cs.rs:463-500), no matvec cache in the reference
    (can_cache_matvec, src/neutronnova_zk.rs:1524) — as the reference's own test has them (test_neutron_sha256, :2480-2503: a SHA-256 circuit entirely in
    synthesize, 2 / 7 / 32 / 64 instances over 32- and 64-byte preimages [i; len], core = the first step circuit): same proof words as the oracle, every
    verifier accepts (the Python one from the bincode bytes), both drivers agree, and the shape the reference's fold gets wrong is refused."""
    if kind == "synthetic_rest_only":
        mk = lambda ws: frontend.synthetic_circuit(8, 0xA5, num_public=1, precommitted_permille=0, witness_seed=ws)
        steps, core = [mk(11 + i) for i in range(3)], mk(99)
    else:
        steps = [frontend.sha256_rest_circuit(bytes([i]) * 32) for i in range(7)]
        core = steps[0]
    assert steps[0].num_rest > 0 and steps[0].num_precommitted == 0
    onn, gnn, want, got, tape, used, _ = _both(ctx, steps, core, 91)
    assert (got == want).all()
    assert onn.verify_words(got) == 0 and gnn.verify(got) == 0
    data = gnn.proof_to_bytes(got)
    assert data == onn.proof_to_bytes(want) and gnn.verify_bytes(data) == 0
    import pynnverify

    pubs = pynnverify.verify_bytes(steps[0], core, len(steps), host.from_label(b"ck", 2049), data)
    assert pubs == ([[int(v) for v in s.publics] for s in steps], [int(v) for v in core.publics])
    bad = got.copy()
    bad[3] ^= np.uint64(2)  # a coordinate of the first step's first rest row (no shared / precommitted rows in front of it)
    assert onn.verify_words(bad) != 0 and gnn.verify(bad) != 0
    gnn.close()
    ref = host.NeutronNovaZkSNARK(ctx, steps, core)
    assert ref.prep_prove(tape) == used[0]
    again, _, _ = ref.prove(tape[used[0]:], reference_order=True)
    assert (again == want).all()
    ref.close()
    if kind == "synthetic_rest_only":
        mk3 = lambda ws: frontend.synthetic_circuit(30, 0x77, num_public=2, shared_permille=200, precommitted_permille=500, witness_seed=ws)
        with pytest.raises(Exception, match="drops the rest segment"):
            host.NeutronNovaZkSNARK(ctx, [mk3(5), mk3(5)], mk3(5))


def test_reference_test_sizes_64_instances(ctx):
    """the largest case of test_neutron_sha256 (src/neutronnova_zk.rs:2489-2501): 64 instances of the two-block circuit (64-byte preimages), on the device
    alone: prove, verify, bytes round trip, a tampered proof refused"""
    steps = [frontend.sha256_rest_circuit(bytes([i]) * 64) for i in range(64)]
    gnn = host.NeutronNovaZkSNARK(ctx, steps, steps[0])
    tape = ol.make_tape(6464, 65536)
    used0 = gnn.prep_prove(tape)
    words, _, phases = gnn.prove(tape[used0:])
    assert gnn.info["nb"] == 6 and gnn.verify(words) == 0
    data = gnn.proof_to_bytes(words)
    assert gnn.verify_bytes(data) == 0 and (gnn.proof_from_bytes(data) == words).all()
    bad = words.copy()
    bad[len(bad) // 3] ^= np.uint64(1 << 5)
    assert gnn.verify(bad) != 0
    print("64 x sha256(64 B) in synthesize, prove phases (ms):", {k: round(v, 3) for k, v in phases.items()})
    gnn.close()


def test_c3_sha256_neutronnova_32_steps(ctx):
    """BASELINE config 3: 32 step circuits (block [i; 64], one compression each) + the core circuit, T256HyraxEngine shapes."""
    steps = [frontend.sha256_step_circuit(bytes([i]) * 64) for i in range(32)]
    core = frontend.sha256_step_circuit(bytes(64))
    onn, gnn, want, got, _, _, phases = _both(ctx, steps, core, 3232)
    assert gnn.info["nb"] == 5 and gnn.info["nx"] == 15 and gnn.info["ny"] == 16
    assert (got == want).all()
    assert onn.verify_words(got) == 0 and gnn.verify(got) == 0
    # wire formats at config 3: the vk digest is the reference's SHA-256 stream on both sides (compared in _both), the proof's
    # bincode bytes equal the oracle's, and verify accepts the deserialised bytes
    data = gnn.proof_to_bytes(got)
    assert data == onn.proof_to_bytes(want)
    assert gnn.verify_bytes(data) == 0 and (gnn.proof_from_bytes(data) == got).all()
    assert onn.verify_words(onn.proof_from_bytes(data)) == 0
    # the third verifier at the benchmark's own size: Python integers from the reference's verify alone, vk digest recomputed from the 33 SHA shapes
    import pynnverify

    pubs = pynnverify.verify_bytes(steps[0], core, 32, host.from_label(b"ck", 2049), data)
    assert pubs == ([[0]] * 32, [0])
    print("C3 prove phases (ms):", {k: round(v, 3) for k, v in phases.items()}, "proof bytes:", len(data))
    gnn.close()
