"""GPU: error behaviour of the C ABI — every entry point returns -(SpartanError class) with a message instead of computing on bad input
(src/errors.rs:13-110: InvalidInputLength = -1, InvalidWitnessLength = -2, DivisionByZero = -3, InternalError = -5), mirroring the
reference's own checks (msm.rs:194-198, r1cs/mod.rs:578-600, multilinear.rs:96-99, neutronnova_zk.rs:718-720)."""
import ctypes

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


def _rc(excinfo):
    return int(str(excinfo.value).split("rc=")[1].split(":")[0])


def test_length_checks(ctx):
    from spartan2_amd import hip, host

    rng = np.random.default_rng(1)
    t8, t16 = hip.Table.from_host(ctx, ol.random_field_array(rng, 8)), hip.Table.from_host(ctx, ol.random_field_array(rng, 16))
    tr = hip.Transcript(ctx, b"e")
    with pytest.raises(hip.SpartanHipError) as e:  # tables of different length (multilinear.rs / sumcheck.rs asserts)
        hip.sumcheck_cubic3(ctx, np.zeros(4, dtype=np.uint64), ol.random_field_array(rng, 3), t8, t8, t16, tr)
    assert _rc(e) == -1
    with pytest.raises(hip.SpartanHipError) as e:  # rounds do not match the table length
        hip.sumcheck_quad(ctx, np.zeros(4, dtype=np.uint64), 5, t8, t8, tr)
    assert _rc(e) == -1
    with pytest.raises(hip.SpartanHipError) as e:  # write past the allocation
        t8.write(4, ol.random_field_array(rng, 8))
    assert _rc(e) == -1
    with pytest.raises(hip.SpartanHipError) as e:  # fold_multiple: all W vectors must have the same length (r1cs/mod.rs:595-600)
        hip.fold_tables(ctx, [t8, t16], ol.random_field_array(rng, 2), 16, hip.Table.zeros(ctx, 16))
    assert _rc(e) == -1
    with pytest.raises(hip.SpartanHipError) as e:  # PowPolynomial::split_evals: left * right must be 2^ell (power.rs:67)
        hip.pow_split_evals(ol.to_mont(3), 4, 4, 8)
    assert _rc(e) == -1
    g = host.from_label(b"ck", 9)
    key = hip.CommitmentKey(ctx, g[:8], g[8])
    with pytest.raises(hip.SpartanHipError) as e:  # MSM: more scalars than bases (msm.rs:194-198)
        key.msm(ol.random_field_array(rng, 9))
    assert _rc(e) == -1
    with pytest.raises(hip.SpartanHipError) as e:  # commit range outside the table
        key.commit(t8, 4, 8, ol.random_field_array(rng, 1))
    assert _rc(e) == -1
    assert (hip.msm(ctx, np.zeros((0, 4), dtype=np.uint64), np.zeros((0, 8), dtype=np.uint64)) == 0).all()  # empty MSM = identity (msm.rs:190-192)


def test_nifs_protocol_and_division_by_zero(ctx):
    from spartan2_amd import hip

    rng = np.random.default_rng(2)
    ell, left, right = ol.tensor_decomp(16)
    nifs = hip.Nifs(ctx, 2, left, right)
    E = ol.pow_split_evals(ol.to_mont(5), ell, left, right)
    with pytest.raises(hip.SpartanHipError) as e:  # one rho per folding round
        nifs.begin(E, ol.random_field_array(rng, 3))
    assert _rc(e) == -1
    nifs.begin(E, ol.random_field_array(rng, 1))
    with pytest.raises(hip.SpartanHipError) as e:  # rounds in order
        nifs.round(1)
    assert _rc(e) == -1
    with pytest.raises(hip.SpartanHipError) as e:  # challenge without a pending polynomial
        nifs.challenge(ol.to_mont(1))
    assert _rc(e) == -1
    out = [hip.Table.zeros(ctx, 16) for _ in range(3)]
    with pytest.raises(hip.SpartanHipError) as e:  # finish before the rounds
        nifs.finish(*out)
    assert _rc(e) == -1
    nifs.begin(E, np.zeros((1, 4), dtype=np.uint64))  # rho = 0: finish_round! divides by rho (neutronnova_zk.rs:709-710)
    with pytest.raises(hip.SpartanHipError) as e:
        nifs.round(0)
    assert _rc(e) == -3
    with pytest.raises(hip.SpartanHipError) as e:  # n_padded must be a power of two
        hip.Nifs(ctx, 3, left, right)
    assert _rc(e) == -1


def test_batched_hook_failure_aborts_the_sumcheck(ctx):
    from spartan2_amd import hip

    rng = np.random.default_rng(3)
    T = [hip.Table.from_host(ctx, ol.random_field_array(rng, 8)) for _ in range(4)]

    def bad_hook(rnd, cs, cc):
        raise RuntimeError("process_round failed")

    with pytest.raises(hip.SpartanHipError) as e:  # the caller's process_round error surfaces as InternalError
        hip.sumcheck_quad_batched(ctx, np.zeros((2, 4), dtype=np.uint64), 3, *T, 0, bad_hook)
    assert _rc(e) == -5
