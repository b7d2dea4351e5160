"""GPU: row-sharded Hyrax commit and point-range-sharded MSM (spartan2_amd.dist) with two ranks — two processes on the test box's GPU, gloo
for the all-gather — against the oracle's unsharded commit / MSM: every row commitment and the MSM result bit-exact."""
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


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import oracle_lib as ol
    from oracle_lib import lib as olib, p64
    from spartan2_amd import dist as spd, hip

    g = spd.Group(backend="gloo")
    ctx = hip.Context(0)
    okey = ctypes.c_void_p(olib().orc_hyrax_setup(b"ck", ctypes.c_size_t(2048)))
    ck_aff, h_aff = np.zeros((2048, 8), dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    olib().orc_hyrax_key_export(okey, p64(ck_aff), p64(h_aff))
    key = hip.CommitmentKey(ctx, ck_aff, h_aff)
    rng = np.random.default_rng(21)  # same data on every rank; each commits only its rows / sums only its points
    rows, cols = 7, 2048
    v = ol.random_field_array(rng, rows * cols)  # full-width scalars: the digit path
    v[cols : 2 * cols] = 0
    v[3 * cols : 4 * cols][rng.integers(0, 2, size=cols) == 0] = 0
    blinds = ol.random_field_array(rng, rows)
    table = hip.Table.from_host(ctx, v)
    got_rows = spd.commit_rows_sharded(g, rows, lambda lo, hi: key.commit(table, lo * cols, (hi - lo) * cols, blinds[lo:hi]) if hi > lo else np.zeros((0, 8), dtype=np.uint64))
    scal = ol.random_field_array(rng, 2048)
    got_msm = spd.msm_point_range_sharded(g, 2048, lambda lo, hi: hip.msm(ctx, scal[lo:hi], ck_aff[lo:hi]), hip.point_sum)
    # a large MSM over caller-supplied bases by point range (SURVEY 8(e) "implement point-range for the single-large-MSM case"): each rank runs the
    # multi-block Pippenger (sp_msm_points) on its range of the device-resident operands, the affine partial sums are gathered and added
    nbig = 1 << 15
    big_pts = np.concatenate([key.fixed_base_mul_h(ol.random_field_array(rng, 1 << 14)) for _ in range(nbig >> 14)])
    big_s = ol.random_field_array(rng, nbig)
    big_tab, big_dev = hip.Table.from_host(ctx, big_s), hip.Points(ctx, big_pts)
    got_big = spd.msm_point_range_sharded(g, nbig, lambda lo, hi: hip.msm_points(ctx, big_tab, lo, hi - lo, big_dev, lo), hip.point_sum)
    out = None
    if rank == 0:
        want_big = np.zeros(8, dtype=np.uint64)
        assert olib().orc_msm(p64(big_s), p64(np.ascontiguousarray(big_pts)), ctypes.c_size_t(nbig), ctypes.c_size_t(0), p64(want_big)) == 0
        assert (got_big == want_big).all(), "point-range sharded Pippenger"
        want_rows = np.zeros((rows, 8), dtype=np.uint64)
        assert olib().orc_hyrax_commit(okey, p64(v), ctypes.c_size_t(rows * cols), p64(blinds), 0, p64(want_rows)) == 0
        want_msm = np.zeros(8, dtype=np.uint64)
        assert olib().orc_msm(p64(scal), p64(ck_aff), ctypes.c_size_t(2048), ctypes.c_size_t(1), p64(want_msm)) == 0
        out = (bool((got_rows == want_rows).all()), bool((got_msm == want_msm).all()))
    q.put((rank, out))
    ctx.close()
    g.close()


def test_two_ranks_on_one_gpu():
    import mp_util

    res = mp_util.run_ranks(_worker, 2)
    assert res[1] is None and res[0] == (True, True)
