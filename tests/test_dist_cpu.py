"""N > 1 path on CPU: two gloo ranks exercise the sharding + barrier + max-over-ranks logic bench.py uses on the GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from spartan2_amd import dist as spd, frontend

    g = spd.Group(backend="gloo")
    lo, hi = spd.shard_range(5, g.rank, g.world)  # 5 step instances over 2 ranks: 3 + 2
    # each rank builds only its own instances (per-instance sharding, no exchange of R1CS data)
    sizes = [frontend.synthetic_circuit(2 + i, 100 + i, num_public=1).num_cons for i in range(lo, hi)]
    g.barrier()
    elapsed = 0.5 + 0.25 * g.rank  # pretend rank 1 is the slow one
    emax = g.max_over_ranks(elapsed)
    total_units = g.sum_over_ranks(float(sum(sizes)))
    value = total_units / emax
    q.put((g.rank, lo, hi, emax, total_units, value))
    g.close()


def test_two_rank_gloo_sharding_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 5)]
    assert res[0][3] == res[1][3] == 0.75  # max over ranks
    assert res[0][4] == res[1][4] and res[0][5] == res[1][5] == res[0][4] / 0.75


def _worker_group_ops(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes

    import numpy as np

    import oracle_lib as ol
    from oracle_lib import lib as olib, p64
    from spartan2_amd import dist as spd

    g = spd.Group(backend="gloo")
    rng = np.random.default_rng(5)  # same data on every rank (bases replicated, witness rows addressed by range)
    rows, cols = 5, 2048
    v = np.zeros((rows * cols, 4), dtype=np.uint64)
    v[rng.integers(0, 2, size=rows * cols) == 1] = ol.to_mont(1)
    blinds = ol.random_field_array(rng, rows)
    key = ctypes.c_void_p(olib().orc_hyrax_setup(b"ck", ctypes.c_size_t(cols)))

    def commit_rows(lo, hi):  # the CPU oracle stands in for CommitmentKey.commit (tests may use the oracle; the product path uses the GPU)
        out = np.zeros((hi - lo, 8), dtype=np.uint64)
        if hi > lo:
            olib().orc_hyrax_commit(key, p64(np.ascontiguousarray(v[lo * cols : hi * cols])), ctypes.c_size_t((hi - lo) * cols),
                                    p64(np.ascontiguousarray(blinds[lo:hi])), 1, p64(out))
        return out

    full = spd.commit_rows_sharded(g, rows, commit_rows)
    want = commit_rows(0, rows)
    ok_commit = bool((full == want).all())
    # point-range sharded MSM
    n = 300
    gens = np.zeros((n, 8), dtype=np.uint64)
    olib().orc_from_label(b"ck", ctypes.c_size_t(n), p64(gens))
    sc = ol.random_field_array(rng, n)

    def msm(lo, hi):
        out = np.zeros(8, dtype=np.uint64)
        olib().orc_msm(p64(np.ascontiguousarray(sc[lo:hi])), p64(np.ascontiguousarray(gens[lo:hi])), ctypes.c_size_t(hi - lo), ctypes.c_size_t(1), p64(out))
        return out

    def point_sum(pts):
        acc = np.zeros(8, dtype=np.uint64)
        for p_ in pts:
            nxt = np.zeros(8, dtype=np.uint64)
            olib().orc_point_add(p64(acc), p64(np.ascontiguousarray(p_)), p64(nxt))
            acc = nxt
        return acc

    got = spd.msm_point_range_sharded(g, n, msm, point_sum)
    ok_msm = bool((got == msm(0, n)).all())
    q.put((g.rank, ok_commit, ok_msm))
    g.close()


def test_two_rank_row_sharded_commit_and_point_range_msm():
    """Hyrax rows shard by row (all-gather of rows, no reduction); a single MSM shards by point range
    (all-gather of partial points + local adds). Exercised with two gloo ranks; per-rank compute is the CPU oracle here."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_group_ops, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]


def test_shard_range_is_a_partition():
    from spartan2_amd.dist import shard_range, whole_job_throughput

    for n in (0, 1, 7, 8, 256):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                cover += list(range(lo, hi))
                assert 0 <= hi - lo <= (n + world - 1) // world
            assert cover == list(range(n))
    assert whole_job_throughput(100.0, 10, 2.0, 8) == 4000.0


def _worker_nifs(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np

    import oracle_lib as ol
    from pynifs import P, PyNifs
    from spartan2_amd import dist as spd

    g = spd.Group(backend="gloo")
    n_inst, num_cons = 8, 16
    n_local = n_inst // world
    ell, left, right = ol.tensor_decomp(num_cons)
    total = left * right
    rng = np.random.default_rng(42)  # every rank derives the same batch and keeps only its shard
    a = rng.integers(-3, 4, size=(n_inst, total)).astype(object)
    b = rng.integers(0, 2, size=(n_inst, total)).astype(object)
    c = (a * b) % P
    arr = lambda m: np.stack([ol.mont_array([int(v) % P for v in row]) for row in m])
    A, B, C = arr(a), arr(b), arr(c)
    E = ol.pow_split_evals(ol.to_mont(123456789), ell, left, right)
    rhos = ol.mont_array([1000003 + 17 * i for i in range(3)])
    nifs = PyNifs(n_local, left, right)
    for i in range(n_local):
        for which, M in enumerate((A, B, C)):
            PyNifs.write_layer(nifs.layer(which, i), M[rank * n_local + i])
    add = lambda x, y: ol.mont_array([(u + v) % P for u, v in zip(ol.ints_of(x), ol.ints_of(y))])
    hook = ol.transcript_round_hook(ol.Transcript(b"vc"))  # deterministic: every rank runs it
    root, r_bs, polys = spd.nifs_rounds_sharded(g, nifs, lambda n: PyNifs(n, left, right), E, rhos, n_local, False, hook, add, PyNifs.read_layer, PyNifs.write_layer)
    out = None
    if rank == 0:
        fa, fb, T_out, eq = root.finish_ab()
        want = ol.nifs_prove_core(left, right, E, rhos, A, B, C, False, ol.transcript_round_hook(ol.Transcript(b"vc")))
        out = (bool((np.stack(polys) == want["polys"]).all()), bool((np.stack(r_bs) == want["r_bs"]).all()), bool((fa == want["A"]).all()),
               bool((fb == want["B"]).all()), bool((T_out == want["T_out"]).all()), bool((eq == want["eq_rho_at_rb"]).all()), len(polys))
    q.put((rank, out))
    g.close()


def test_two_rank_gloo_nifs_rounds_sharded_match_the_unsharded_oracle():
    """8 instances over 2 ranks: two shard-local rounds with a 2-element exchange each, hand-off of one layer pair per rank, last round on
    rank 0 — round polynomials, challenges, folded layers and T_out equal the oracle's unsharded NeutronNovaNIFS::prove."""
    import mp_util

    res = mp_util.run_ranks(_worker_nifs, 2, timeout=300)
    assert res[1] is None
    assert res[0] == (True, True, True, True, True, True, 3)


def _worker_sumcheck(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes

    import numpy as np

    import oracle_lib as ol
    import pysumcheck as ps
    from oracle_lib import lib as olib, p64
    from spartan2_amd import dist as spd

    g = spd.Group(backend="gloo")
    ell = 5
    n = 1 << ell
    rng = np.random.default_rng(99)
    A, B = ol.random_field_array(rng, n), ol.random_field_array(rng, n)
    C = ol.mont_array([x * y % ps.P for x, y in zip(ol.ints_of(A), ol.ints_of(B))])
    taus = ol.random_field_array(rng, ell)
    claim = np.zeros(4, dtype=np.uint64)
    tr = ol.Transcript(b"sc")
    cubic = lambda cl, p, ts, a, b, c, sc, red: ps.cubic_sharded(tr, cl, p, ts, a, b, c, sc, red)
    polys, r, fin = spd.sumcheck_cubic3_sharded(g, cubic, claim, taus, *(spd.slice_of(T, rank, world) for T in (A, B, C)), lambda arr: arr)
    qclaim = np.zeros(4, dtype=np.uint64)
    olib().orc_field_dot(0, p64(A), p64(B), ctypes.c_size_t(n), p64(qclaim))
    trq = ol.Transcript(b"sq")
    quad = lambda cl, rounds, a, b, red: ps.quad_sharded(trq, cl, rounds, a, b, red)
    qpolys, qr, qfin = spd.sumcheck_quad_sharded(g, quad, qclaim, ell, *(spd.slice_of(T, rank, world) for T in (A, B)), lambda arr: arr)
    # the oracle's unsharded sum-checks on the full tables
    otr = ol.Transcript(b"sc")
    wp, wr, wf = np.zeros((ell, 3, 4), dtype=np.uint64), np.zeros((ell, 4), dtype=np.uint64), np.zeros((3, 4), dtype=np.uint64)
    a, b, c = A.copy(), B.copy(), C.copy()
    assert olib().orc_sumcheck_cubic3(p64(claim), p64(taus), ctypes.c_size_t(ell), p64(a), p64(b), p64(c), otr.h, p64(wp), p64(wr), p64(wf)) == 0
    oq = ol.Transcript(b"sq")
    qp, qrr, qf = np.zeros((ell, 2, 4), dtype=np.uint64), np.zeros((ell, 4), dtype=np.uint64), np.zeros((2, 4), dtype=np.uint64)
    a, b = A.copy(), B.copy()
    full = ctypes.c_size_t((1 << 64) - 1)
    assert olib().orc_sumcheck_quad(p64(qclaim), ctypes.c_size_t(ell), p64(a), full, full, p64(b), full, full, oq.h, p64(qp), p64(qrr), p64(qf)) == 0
    q.put((rank, tuple(bool(x) for x in ((polys == wp).all(), (r == wr).all(), (fin == wf).all(), (tr.squeeze(b"z") == otr.squeeze(b"z")).all(),
                                         (qpolys == qp).all(), (qr == qrr).all(), (qfin == qf).all()))))
    g.close()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sumcheck_by_table_slice_matches_the_unsharded_oracle(world):
    """Tables sharded on their last log2(world) variables; one 2-element exchange per round; every rank finishes the last rounds on the gathered
    values and ends with the same polynomials, challenges, final evaluations and transcript state as the oracle's unsharded sum-checks."""
    import mp_util

    res = mp_util.run_ranks(_worker_sumcheck, world, timeout=300)
    for r in range(world):
        assert res[r] == (True,) * 7, (r, res[r])


# ---- the C++ exchange layer's callback backend (spartan2_amd/host/comm.hpp) over gloo ---------------------------------------------------------
def _comm_worker(rank, world, port, q):
    import os
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np

    from spartan2_amd import dist as spd, host

    g = spd.Group(backend="gloo")
    comm = host.Comm(rank, world, "torch")
    small = np.arange(12, dtype=np.uint64) + 1000 * rank  # a round's three field elements
    got = comm.allgather(small)
    ok = all((got[r] == np.arange(12, dtype=np.uint64) + 1000 * r).all() for r in range(world))
    big = np.full(2048 * 4 + 16, rank + 7, dtype=np.uint64)  # the opening's record: partial L.W + two partial points
    got = comm.allgather(big)
    ok = ok and all((got[r] == r + 7).all() for r in range(world))
    st = comm.stats()
    q.put((rank, (bool(ok), st["exchanges"], st["bytes_gathered"])))
    comm.close()
    g.close()


@pytest.mark.parametrize("world", [2, 4])
def test_cpp_exchange_layer_all_gather_over_gloo(world):
    import mp_util

    res = mp_util.run_ranks(_comm_worker, world)
    for r in range(world):
        assert res[r] == (True, 2, (12 * 8 + (2048 * 4 + 16) * 8) * world)
