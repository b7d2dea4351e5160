"""GPU: the sharded NIFS rounds (spartan2_amd.dist.nifs_rounds_sharded over sp_nifs_begin_shard / round_sums / round_finish / fold_pending /
resume) with two ranks — two processes sharing the one GPU of the test box, gloo for the exchange — against the oracle's UNSHARDED
NeutronNovaNIFS::prove: identical round polynomials, challenges, folded layers, T_out. Also the per-shard C fold."""
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


def _worker(rank, world, port, q, n_inst, num_cons, small):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import oracle_lib as ol
    from spartan2_amd import dist as spd, hip

    P = ol.MODULI[0]
    g = spd.Group(backend="gloo")
    ctx = hip.Context(0)
    n_local = n_inst // world
    ell, left, right = ol.tensor_decomp(num_cons)
    total = left * right
    rng = np.random.default_rng(4242)
    lut = np.stack([ol.to_mont(v) for v in (0, 1, 2, P - 1)])
    ai, bi = rng.integers(0, 4, size=(n_inst, total)), rng.integers(0, 2, size=(n_inst, total))
    vals = np.array([0, 1, 2, -1])
    A, B = lut[ai], lut[bi]
    C = lut[np.select([vals[ai] * bi == 0, vals[ai] * bi == 1, vals[ai] * bi == 2], [0, 1, 2], 3)]
    E = ol.pow_split_evals(ol.to_mont(987654321), ell, left, right)
    ell_b = n_inst.bit_length() - 1
    rhos = ol.mont_array([2000003 + 31 * i for i in range(ell_b)])
    nifs = hip.Nifs(ctx, n_local, left, right)
    for i in range(n_local):
        for which, M in enumerate((A, B, C)):
            v = nifs.layer(which, i)
            v.write(0, M[rank * n_local + i])
            v.free()
    add = lambda x, y: ol.mont_array([(u + w) % P for u, w in zip(ol.ints_of(x), ol.ints_of(y))])
    hook = ol.transcript_round_hook(ol.Transcript(b"vc"))
    root, r_bs, polys = spd.nifs_rounds_sharded(g, nifs, lambda n: hip.Nifs(ctx, n, left, right), E, rhos, n_local, small, hook, add,
                                                lambda v: v.read(0, total), lambda v, a: v.write(0, a))
    # per-shard C fold with this rank's slice of weights_from_r (every rank knows all challenges: the hook is deterministic; ranks != 0 re-derive
    # the last ones from rank 0's transcript in a real deployment — here rank 0 shares them)
    all_r = spd._all_gather_rows(g, np.stack(r_bs) if rank == 0 else np.zeros((ell_b, 4), dtype=np.uint64), [ell_b] * world).reshape(world, ell_b, 4)[0]
    w = hip.weights_from_r(all_r, n_inst)
    part = hip.Table.zeros(ctx, total)
    views = [nifs.layer(2, i) for i in range(n_local)]
    hip.fold_tables(ctx, views, w[rank * n_local : (rank + 1) * n_local], total, part)
    parts = spd._all_gather_rows(g, part.read(0, total), [total] * world).reshape(world, total, 4)
    out = None
    if rank == 0:
        oa, ob = hip.Table.zeros(ctx, total), hip.Table.zeros(ctx, total)
        T_out, eq = root.finish(oa, ob, None)
        tabs = [hip.Table.from_host(ctx, p) for p in parts]
        oc = hip.Table.zeros(ctx, total)
        hip.fold_tables(ctx, tabs, np.stack([ol.to_mont(1)] * world), total, oc)
        want = ol.nifs_prove_core(left, right, E, rhos, A, B, C, small, ol.transcript_round_hook(ol.Transcript(b"vc")))
        out = tuple(bool(x) for x in ((np.stack(polys) == want["polys"]).all(), (np.stack(r_bs) == want["r_bs"]).all(), (oa.read(0, total) == want["A"]).all(),
                                      (ob.read(0, total) == want["B"]).all(), (oc.read(0, total) == want["C"]).all(), (T_out == want["T_out"]).all(),
                                      (eq == want["eq_rho_at_rb"]).all()))
    q.put((rank, out))
    ctx.close()
    g.close()


@pytest.mark.parametrize("n_inst,num_cons,small", [(8, 64, False), (8, 1 << 16, True), (4, 1 << 9, False)])
def test_two_ranks_on_one_gpu(n_inst, num_cons, small):
    import mp_util

    res = mp_util.run_ranks(_worker, 2, (n_inst, num_cons, small))
    assert res[1] is None and res[0] == (True,) * 7
