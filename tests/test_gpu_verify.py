"""GPU: the device-backed SpartanSNARK::verify (host driver `verify`, src/spartan.rs:469-578) accepts the proofs the GPU prover makes and
rejects tampered ones with the same failed-check index as the oracle's restated verifier."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from spartan2_amd import hip

    c = hip.Context(0)
    yield c
    c.close()


def _prove(ctx, inst, seed):
    from spartan2_amd import host

    sn = host.SpartanSNARK(ctx, inst)
    tape = ol.make_tape(seed, 4096)
    sn.prep_prove(tape)
    words, _, _ = sn.prove(ol.make_tape(seed + 1, 4096))
    return sn, words


@pytest.mark.parametrize("which", ["sha256_1block", "synthetic_segments", "cubic"])
def test_verify_accepts_and_rejects_like_the_oracle(ctx, which):
    from spartan2_amd import frontend

    inst = {"sha256_1block": lambda: frontend.sha256_circuit(b"abc"),
            "synthetic_segments": lambda: frontend.synthetic_circuit(60, 9, num_public=3, shared_permille=200, precommitted_permille=500),
            "cubic": frontend.cubic_circuit}[which]()
    sn, words = _prove(ctx, inst, 1234)
    osp = ol.OracleSpartan(inst)
    assert sn.verify(words) == 0 and osp.verify_words(words) == 0
    d = sn.dims
    rows = (((d["num_shared"] + 2047) // 2048) if d["num_shared_unpadded"] else 0) + (((d["num_precommitted"] + 2047) // 2048) if d["num_precommitted_unpadded"] else 0) \
        + (d["num_rest"] + 2047) // 2048
    lx = (d["num_cons"] - 1).bit_length()
    off_pub = 8 * rows
    off_outer = off_pub + 4 * d["num_public"]
    off_claims = off_outer + 12 * lx
    off_inner = off_claims + 12
    n = len(words)
    # one flipped bit per proof section: publics, an outer polynomial, a claim, an inner polynomial, eval_W, z_vec, z_beta
    for pos in [p for p in (off_pub if d["num_public"] else None, off_outer + 5, off_claims + 1, off_inner + 2, n - 8 - 4 * min(2048, d["num_shared"] + d["num_precommitted"] + d["num_rest"]) - 16 - 8,
                            n - 8 - 3, n - 1) if p is not None]:
        bad = words.copy()
        bad[pos] ^= np.uint64(1)
        want = osp.verify_words(bad)
        got = sn.verify(bad)
        assert want != 0 and got == want, (pos, want, got)
    sn.close()


def test_verify_2kib_bench_instance(ctx):
    """The bench instance (2^20 constraints): prove then verify on the device."""
    from spartan2_amd import frontend

    sn, words = _prove(ctx, frontend.sha256_circuit(bytes(2048)), 77)
    assert sn.verify(words) == 0
    bad = words.copy()
    bad[len(bad) // 2] ^= np.uint64(4)
    assert sn.verify(bad) != 0
    sn.close()
