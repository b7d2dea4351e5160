"""GPU: sum-check by evaluation-table slice (spartan2_amd.dist.sumcheck_{cubic3,quad}_sharded over sp_sumcheck_*_sharded) with two ranks — two
processes sharing the test box's GPU, gloo for the 2-3-element exchange per round — against the oracle's UNSHARDED sum-checks on the full
tables: identical round polynomials, challenges, final evaluations and transcript state."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, ell):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import oracle_lib as ol
    from oracle_lib import lib as olib, p64
    from spartan2_amd import dist as spd, hip

    P = ol.MODULI[0]
    g = spd.Group(backend="gloo")
    ctx = hip.Context(0)
    n = 1 << ell
    rng = np.random.default_rng(777)
    A, B = ol.random_field_array(rng, n), ol.random_field_array(rng, n)
    C = ol.mont_array([x * y % P for x, y in zip(ol.ints_of(A), ol.ints_of(B))])  # satisfying triple: claim 0
    taus = ol.random_field_array(rng, ell)
    claim = np.zeros(4, dtype=np.uint64)
    tr = hip.Transcript(ctx, b"sc")
    tabs = [hip.Table.from_host(ctx, spd.slice_of(T, rank, world)) for T in (A, B, C)]
    cubic = lambda cl, p, ts, a, b, c, sc, red: hip.sumcheck_cubic3_sharded(ctx, cl, p, ts, a, b, c, tr, sc, red)
    polys, r, fin = spd.sumcheck_cubic3_sharded(g, cubic, claim, taus, *tabs, lambda arr: hip.Table.from_host(ctx, arr))
    after = tr.squeeze(b"after")
    # quadratic on fresh tables: claim = <A, B>
    qclaim = np.zeros(4, dtype=np.uint64)
    olib().orc_field_dot(0, p64(A), p64(B), ctypes.c_size_t(n), p64(qclaim))
    trq = hip.Transcript(ctx, b"sq")
    qt = [hip.Table.from_host(ctx, spd.slice_of(T, rank, world)) for T in (A, B)]
    quad = lambda cl, rounds, a, b, red: hip.sumcheck_quad_sharded(ctx, cl, rounds, a, b, trq, red)
    qpolys, qr, qfin = spd.sumcheck_quad_sharded(g, quad, qclaim, ell, *qt, lambda arr: hip.Table.from_host(ctx, arr))
    out = None
    if rank == 0:
        otr = ol.Transcript(b"sc")
        wp, wr, wf = np.zeros((ell, 3, 4), dtype=np.uint64), np.zeros((ell, 4), dtype=np.uint64), np.zeros((3, 4), dtype=np.uint64)
        a, b, c = A.copy(), B.copy(), C.copy()
        assert olib().orc_sumcheck_cubic3(p64(claim), p64(taus), ctypes.c_size_t(ell), p64(a), p64(b), p64(c), otr.h, p64(wp), p64(wr), p64(wf)) == 0
        oq = ol.Transcript(b"sq")
        qp, qrr, qf = np.zeros((ell, 2, 4), dtype=np.uint64), np.zeros((ell, 4), dtype=np.uint64), np.zeros((2, 4), dtype=np.uint64)
        a, b = A.copy(), B.copy()
        full = ctypes.c_size_t((1 << 64) - 1)
        assert olib().orc_sumcheck_quad(p64(qclaim), ctypes.c_size_t(ell), p64(a), full, full, p64(b), full, full, oq.h, p64(qp), p64(qrr), p64(qf)) == 0
        out = tuple(bool(x) for x in ((polys == wp).all(), (r == wr).all(), (fin == wf).all(), (after == otr.squeeze(b"after")).all(), (qpolys == qp).all(),
                                      (qr == qrr).all(), (qfin == qf).all()))
    q.put((rank, out))
    ctx.close()
    g.close()


@pytest.mark.parametrize("ell", [4, 12, 16])
def test_two_ranks_on_one_gpu(ell):
    import mp_util

    res = mp_util.run_ranks(_worker, 2, (ell,))
    assert res[1] is None and res[0] == (True,) * 7
