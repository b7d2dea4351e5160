"""GPU parity for the entry points added in round 2 to close the ABI against the reference surface (VERDICT r1 "small entry points"):
vartime_scalar_mul (wNAF-5, msm.rs:779-867), the two-term fold_commitments (hyrax_pc.rs:757-776), rerandomize_commitment (hyrax_pc.rs:321-344),
multiply_vec_batched (r1cs/mod.rs:1130-1166), evaluation_points_zero_check_round0 (sumcheck.rs:1163-1271)."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol
from spartan2_amd import frontend, hip, host

pytestmark = pytest.mark.gpu
P = ol.MODULI[0]


@pytest.fixture(scope="module")
def ctx():
    c = hip.Context(0)
    yield c
    c.close()


def _points(rng, n):
    g = host.from_label(b"gaps", 4)
    ks = ol.random_field_array(rng, n)
    out = np.zeros((n, 8), dtype=np.uint64)
    for i in range(n):
        ol.lib().orc_point_mul(ol.p64(g[i % 4]), ol.p64(ks[i]), ol.p64(out[i]))
    return out


def _mul(pt, k):
    o = np.zeros(8, dtype=np.uint64)
    ol.lib().orc_point_mul(ol.p64(np.ascontiguousarray(pt)), ol.p64(np.ascontiguousarray(k)), ol.p64(o))
    return o


def _add(a, b):
    o = np.zeros(8, dtype=np.uint64)
    ol.lib().orc_point_add(ol.p64(np.ascontiguousarray(a)), ol.p64(np.ascontiguousarray(b)), ol.p64(o))
    return o


@pytest.mark.parametrize("n", [1, 5, 100])  # host side of the library below 48 points, one device lane per point above
def test_vartime_scalar_mul_and_two_term_fold(ctx, n):
    rng = np.random.default_rng(900 + n)
    pts, qs = _points(rng, n), _points(rng, n)
    for w in (ol.random_field_array(rng, 1)[0], ol.to_mont(1), ol.to_mont(0), ol.to_mont(P - 1), ol.to_mont(31), ol.to_mont(1 << 200)):
        got = hip.vartime_scalar_mul(ctx, pts, w)
        want = np.stack([_mul(pt, w) for pt in pts])
        assert (got == want).all()
        fold = hip.fold_commitments2(ctx, qs, pts, w)
        assert (fold == np.stack([_add(q, m) for q, m in zip(qs, want)])).all()


@pytest.mark.parametrize("n", [1, 16, 17, 40])
def test_two_term_fold_with_ladders_built_ahead(ctx, n):
    """sp_fold_commitments2_begin / _finish (the doubling ladders of q's rows built on the polling host threads before the weight is known, the weight's
    non-adjacent form walked over them) against the oracle's p + w q and the one-call form: random weights, 0, 1, p - 1 (the longest NAF), 2^200, a weight
    with long runs of ones (carries through the NAF), identity rows on either side, a dropped job."""
    rng = np.random.default_rng(1900 + n)
    pts, qs = _points(rng, n), _points(rng, n)
    if n > 2:
        pts[1] = 0  # q row = the identity
        qs[2] = 0   # p row = the identity
    ones = ol.to_mont((1 << 255) - (1 << 13) + 5)
    for w in (ol.random_field_array(rng, 1)[0], ol.to_mont(1), ol.to_mont(0), ol.to_mont(P - 1), ol.to_mont(1 << 200), ones, ol.to_mont(3)):
        got = hip.fold_commitments2_split(ctx, qs, pts, w)
        want = np.stack([_add(q, _mul(pt, w)) for q, pt in zip(qs, pts)])
        assert (got == want).all()
        assert (hip.fold_commitments2(ctx, qs, pts, w) == want).all()
    assert hip.fold_commitments2_split(ctx, qs, pts, ones, drop=True) is None
    assert (hip.fold_commitments2_split(ctx, qs, pts, ones) == np.stack([_add(q, _mul(pt, ones)) for q, pt in zip(qs, pts)])).all()


def test_rerandomize_commitment(ctx):
    rng = np.random.default_rng(31)
    g = host.from_label(b"ck", 65)
    key = hip.CommitmentKey(ctx, g[:64], g[64])
    for rows in (3, 40):
        comm = _points(rng, rows)
        r_old, r_new = ol.random_field_array(rng, rows), ol.random_field_array(rng, rows)
        got = key.rerandomize(comm, r_old, r_new)
        for i in range(rows):
            d = ol.to_mont((ol.from_mont(r_new[i]) - ol.from_mont(r_old[i])) % P)
            assert (got[i] == _add(comm[i], _mul(g[64], d))).all()
        # rerandomizing back returns the original commitment
        assert (key.rerandomize(got, r_new, r_old) == comm).all()


def test_multiply_vec_batched(ctx):
    inst = frontend.synthetic_circuit(12, 9, num_public=3)
    mats, dims = host.pad_shape(inst)
    oshape = ol.OracleShape(inst)
    shape = hip.Shape(ctx, mats, dims)
    N, ncols = dims["num_cons"], oshape.num_vars + oshape.num_extra
    rng = np.random.default_rng(3)
    zs = [ol.random_field_array(rng, ncols) for _ in range(5)]
    outs = [[hip.Table.zeros(ctx, N) for _ in zs] for _ in range(3)]
    shape.multiply_vec_batched([hip.Table.from_host(ctx, z) for z in zs], *outs)
    for k, z in enumerate(zs):
        want = [np.zeros((N, 4), dtype=np.uint64) for _ in range(3)]
        assert ol.lib().orc_shape_multiply_vec(oshape.h, ol.p64(z), *(ol.p64(w) for w in want)) == 0
        for m in range(3):
            assert (outs[m][k].read(0, N) == want[m]).all()


@pytest.mark.parametrize("ell", [1, 2, 3, 9, 14, 19])
def test_zero_check_round0(ctx, ell):
    rng = np.random.default_rng(70 + ell)
    n = 1 << ell
    A, B = ol.random_field_array(rng, n), ol.random_field_array(rng, n)
    for zero_tau in (False, True):
        taus = ol.random_field_array(rng, ell)
        if zero_tau:
            taus[0] = 0  # the tau = 0 fallback (:1244-1268)
        want = np.zeros((3, 4), dtype=np.uint64)
        assert ol.lib().orc_zero_check_round0(ol.p64(taus), ctypes.c_size_t(ell), ol.p64(A), ol.p64(B), ol.p64(want)) == 0
        got = hip.eval_cubic_zero_check_round0(ctx, taus, hip.Table.from_host(ctx, A), hip.Table.from_host(ctx, B))
        assert (got == want).all()
    # on a satisfying triple the zero-check shortcut equals the general round-1 evaluation (the property the shortcut rests on)
    if ell >= 2:
        a, b = ol.ints_of(A), ol.ints_of(B)
        C = ol.mont_array([x * y % P for x, y in zip(a, b)])
        taus = ol.random_field_array(rng, ell)
        tr = hip.Transcript(ctx, b"zc")
        polys, _, _ = hip.sumcheck_cubic3(ctx, np.zeros(4, dtype=np.uint64), taus, hip.Table.from_host(ctx, A), hip.Table.from_host(ctx, B), hip.Table.from_host(ctx, C), tr)
        zc = hip.eval_cubic_zero_check_round0(ctx, taus, hip.Table.from_host(ctx, A), hip.Table.from_host(ctx, B))
        # compressed poly of round 1: (c0, c2, c3); eval_0 = c0, and the cubic through (0, eval_0), (1, -eval_0), (2, eval_2), (3, eval_3) has these c2, c3
        e0, e2, e3 = (ol.from_mont(x) for x in zc)
        e1 = (-e0) % P
        inv2, inv6 = pow(2, -1, P), pow(6, -1, P)
        c3 = (e3 - 3 * e2 + 3 * e1 - e0) * inv6 % P
        c2 = ((e2 - 2 * e1 + e0) * inv2 - 3 * c3) % P
        assert [ol.from_mont(x) for x in polys[0]] == [e0, c2, c3]


@pytest.mark.parametrize("n_groups", [3, 40, 700])  # N = 2^9 (general first evaluation), 2^12, 2^16 (products path, factored eq tables)
def test_round0_products_fused_into_the_matrix_vector_product(ctx, n_groups):
    """sp_multiply_vec_incremental_round0 + sp_sumcheck_cubic3_round0 == sp_multiply_vec_incremental + sp_sumcheck_cubic3, word for word."""
    inst = frontend.synthetic_circuit(n_groups, 77, num_public=3, shared_permille=200, precommitted_permille=500)
    mats, dims = host.pad_shape(inst)
    shape = hip.Shape(ctx, mats, dims)
    N = dims["num_cons"]
    M = dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"]
    rng = np.random.default_rng(n_groups)
    z = ol.random_field_array(rng, M + 1 + dims["num_public"])
    zc = z.copy()
    zc[dims["num_shared"] + dims["num_precommitted"] :] = 0
    cached = [hip.Table.zeros(ctx, N) for _ in range(3)]
    shape.multiply_vec(hip.Table.from_host(ctx, zc), *cached)
    zt = hip.Table.from_host(ctx, z)
    plain = [hip.Table.zeros(ctx, N) for _ in range(3)]
    shape.multiply_vec_incremental(zt, *cached, *plain)
    fused = [hip.Table.zeros(ctx, N) for _ in range(3)]
    p0, p1 = hip.Table.zeros(ctx, N // 2), hip.Table.zeros(ctx, N // 2)
    shape.multiply_vec_incremental_round0(zt, *cached, *fused, p0, p1)
    for a, b in zip(plain, fused):
        assert (a.read(0, N) == b.read(0, N)).all()
    a, b, c = (ol.ints_of(t.read(0, N)) for t in plain)
    h = N // 2
    assert ol.ints_of(p0.read(0, h)) == [(a[i] * b[i] - c[i]) % P for i in range(h)]
    assert ol.ints_of(p1.read(0, h)) == [(a[i + h] - a[i]) * (b[i + h] - b[i]) % P for i in range(h)]
    ell = N.bit_length() - 1
    taus = ol.random_field_array(rng, ell)
    claim = ol.random_field_array(rng, 1)[0]  # z is not a satisfying assignment: any claim will do for prover-side equality
    want = hip.sumcheck_cubic3(ctx, claim, taus, *plain, hip.Transcript(ctx, b"r0"))
    got = hip.sumcheck_cubic3_round0(ctx, claim, taus, *fused, p0, p1, hip.Transcript(ctx, b"r0"))
    for w, g in zip(want, got):
        assert (w == g).all()


def test_table_view_and_device_ptr(ctx):
    """sp_table_view: a non-owning window that the fold / copy entry points accept like a table; sp_table_device_ptr: address + capacity."""
    rng = np.random.default_rng(21)
    v = ol.random_field_array(rng, 64)
    t = hip.Table.from_host(ctx, v)
    L = hip.lib()
    views = []
    for g in range(4):
        h = ctypes.c_void_p()
        hip.check(L.sp_table_view(t.h, ctypes.c_size_t(16 * g), ctypes.c_size_t(16), ctypes.byref(h)))
        views.append(hip.Table(ctx, h))
    assert all((views[g].read(0, 16) == v[16 * g : 16 * g + 16]).all() for g in range(4))
    # sum of the four windows with unit weights = fold_multiple over views (the combine step of the sharded NIFS driver)
    out = hip.Table.zeros(ctx, 16)
    hip.fold_tables(ctx, views, np.stack([ol.to_mont(1)] * 4), 16, out)
    want = ol.mont_array([sum(ol.from_mont(v[16 * g + j]) for g in range(4)) % P for j in range(16)])
    assert (out.read(0, 16) == want).all()
    # writes through a view land in the parent
    views[1].write(0, v[:4])
    assert (t.read(16, 4) == v[:4]).all()
    ptr, cap = ctypes.c_void_p(), ctypes.c_size_t()
    hip.check(L.sp_table_device_ptr(t.h, ctypes.byref(ptr), ctypes.byref(cap)))
    assert ptr.value and cap.value >= 64 * 32
    bad = ctypes.c_void_p()
    assert L.sp_table_view(t.h, ctypes.c_size_t(60), ctypes.c_size_t(16), ctypes.byref(bad)) == -1
    for x in views:
        x.free()
    assert (t.read(0, 4) == v[:4]).all()  # freeing views leaves the storage alone
    t.free()


def test_long_absorbs_hashed_on_the_library_thread_equal_the_oracle_transcript(ctx):
    """With sp_transcript_set_async, sp_transcript_absorb hands inputs of >= 4 KiB to the library's hashing thread and returns (the caller's next calls run beside the Keccak blocks);
    every later use of the transcript joins first. The squeezed challenges must be the oracle transcript's (src/provider/keccak.rs:70-99) whatever mix of
    short / long absorbs, dom_seps, clones and squeezes follows."""
    rng = np.random.default_rng(77)
    tr, otr = hip.Transcript(ctx, b"async"), ol.Transcript(b"async")
    tr.set_async(True)
    for step, n in enumerate([10, 5000, 70000, 3, 4096, 4095, 200000]):
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        tr.absorb(b"blob", data)
        otr.absorb(b"blob", data)
        if step % 3 == 1:
            tr.dom_sep(b"sep")
            otr.dom_sep(b"sep")
        if step % 2 == 0:
            assert (tr.squeeze(b"c") == otr.squeeze(b"c")).all()
    big = rng.integers(0, 256, size=100000, dtype=np.uint8).tobytes()
    tr.absorb(b"last", big)  # still hashing when the clone is taken: the clone must wait for it
    otr.absorb(b"last", big)
    want = otr.squeeze(b"z")
    tr2 = tr.clone()
    assert (tr2.squeeze(b"z") == want).all() and (tr.squeeze(b"z") == want).all()


@pytest.mark.parametrize("ell", [12, 13, 15, 20])  # 2^(ell - 10) high entries: fewer than, exactly and more than one block's eight
def test_eq_table_begun_two_coordinates_early(ctx, ell):
    """sp_eq_table_begin / _finish (the half tables of the first ell - K coordinates built ahead, K = 2, 3, 4, the last K applied in the one launch
    behind the last challenge) give the table of sp_eq_table; a `_finish` without a `_begin`, or with a different prefix, is the plain call; fewer
    than two or more than four missing coordinates are refused."""
    from spartan2_amd import hip

    rng = np.random.default_rng(4200 + ell)
    r = ol.random_field_array(rng, ell)
    r[ell - 1] = 0  # a zero and a one among the late coordinates
    r[ell - 3] = ol.to_mont(1)
    want = hip.Table.eq(ctx, r).read()
    out = hip.Table.zeros(ctx, 1 << ell)
    for k in (2, 3, 4):
        out.write(0, np.zeros((1 << ell, 4), dtype=np.uint64))
        hip.Table.eq_begin(ctx, r[: ell - k], ell)
        out.eq_finish(r)
        assert (out.read() == want).all(), k
    for k in (1, 5):
        with pytest.raises(hip.SpartanHipError):
            hip.Table.eq_begin(ctx, r[: ell - k], ell)
    out2 = hip.Table.zeros(ctx, 1 << ell)
    out2.eq_finish(r)  # nothing begun
    assert (out2.read() == want).all()
    r2 = r.copy()
    r2[0] = ol.random_field_array(rng, 1)[0]
    hip.Table.eq_begin(ctx, r[: ell - 2], ell)
    out2.eq_finish(r2)  # another point than the one begun
    assert (out2.read() == hip.Table.eq(ctx, r2).read()).all()


@pytest.mark.parametrize("env,targets", [
    ("SPARTAN_FOLD_STAGE2=1", "tests/test_gpu_configs.py tests/test_gpu_sumcheck.py -k 'c1_c2 or 2_pow_21 or streaming'"),
    ("SPARTAN_PIP_MINW=2", "tests/test_gpu_msm_big.py -k 'msm_points_matches_oracle or ragged'"),
    ("SPARTAN_COMB_MINW=2", "tests/test_gpu_configs.py::test_c4_full_scalar_commit_2048_rows_matches_oracle"),
    ("SPARTAN_FBTABLES_OLD=1", "tests/test_gpu_group.py -k 'fbtables_every_entry or fixed_base_tables_multi_mul_matches'"),
    ("SPARTAN_HAND_N_CUBIC=256 SPARTAN_HAND_N_QUAD=512", "tests/test_gpu_sumcheck.py tests/test_gpu_configs.py -k 'cubic or quad or c1_c2 or randomised or two_round'"),
    ("SPARTAN_HAND_N_CUBIC=64 SPARTAN_HAND_N_QUAD=128 SPARTAN_WALKERS=0", "tests/test_gpu_sumcheck.py -k 'cubic or quad'"),
    ("SPARTAN_VC_SPLIT=0", "tests/test_gpu_neutronnova_zk.py -k 'oracle'"),
    ("SPARTAN_WALKERS=0", "tests/test_gpu_neutronnova_zk.py tests/test_gpu_group.py tests/test_gpu_abi_gaps.py -k '(oracle or commit_split or ladders) and not switched'"),
])
def test_switched_code_paths_in_a_process_of_their_own(env, targets):
    """Code paths behind switches that are read once per process - the second stage folded into the streaming producers (measured, off by default), the
    two-waves-per-SIMD forms of the comb / Pippenger bucket kernels (spill-free, measured behind the three-wave forms), round 5's table build, the resident
    tail's hand-over at its first one-block step with the host's rounds on the walkers (and an intermediate size without walkers), the round commitments of
    the ZK verifier circuit through the device walk, a process without walkers - run the parity
    tests that exercise them in a child process with the switch set: they stay bit-exact against the oracle although no default run takes them."""
    import shlex
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("SPARTAN_TEST_CHILD"):
        pytest.skip("already inside a child process of this test (a -k expression matched its own parameter id)")
    child_env = dict(os.environ, SPARTAN_TEST_CHILD="1", **dict(kv.split("=") for kv in env.split()))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + shlex.split(targets), cwd=root, env=child_env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "no tests ran" not in tail, tail
