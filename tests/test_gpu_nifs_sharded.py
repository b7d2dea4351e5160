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


def _worker_cpp(rank, world, port, q, n_inst, groups, small):
    """The C++ driver (spartan2_amd/host/neutronnova_nifs.cpp nifs_prove_sharded) over the callback exchange backend: the whole of
    NeutronNovaNIFS::prove - preamble, layers by SpMV on the rank's own instances, rounds, hand-off, C / witness / commitment folds."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ctypes

    import oracle_lib as ol
    import test_gpu_nifs_prove as tnp
    from spartan2_amd import dist as spd, frontend, hip, host

    g = spd.Group(backend="gloo")
    ctx = hip.Context(0)
    L = ol.lib()
    okey = ctypes.c_void_p(L.orc_hyrax_setup(b"ck", ctypes.c_size_t(2048)))
    ck_aff, h_aff = np.zeros((2048, 8), dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    L.orc_hyrax_key_export(okey, ol.p64(ck_aff), ol.p64(h_aff))
    ck = hip.CommitmentKey(ctx, ck_aff, h_aff)
    insts = [frontend.synthetic_circuit(groups, 5, num_public=2, shared_permille=150, precommitted_permille=450, witness_seed=100 + s) for s in range(n_inst)]
    oshape, shape, dims, Ws, X, r_W, comms = tnp._setup(ctx, okey, insts, np.random.default_rng(77))
    want = ol.nifs_prove(oshape, okey, comms, X, Ws, r_W, small, ol.Transcript(b"neutronnova_prove"), ol.transcript_round_hook(ol.Transcript(b"vc")))
    n_local = n_inst // world
    lo = rank * n_local
    comm = host.Comm(rank, world, "torch")
    tabs = [hip.Table.from_host(ctx, w) for w in Ws[lo : lo + n_local]]
    got = host.nifs_prove_sharded(ctx, comm, shape, dims, ck, comms[lo : lo + n_local], X[lo : lo + n_local], tabs, r_W[lo : lo + n_local], small,
                                  hip.Transcript(ctx, b"neutronnova_prove"), ol.transcript_round_hook(ol.Transcript(b"vc")))
    N, nv = oshape.num_cons, oshape.num_vars
    ok = [bool((want[k] == got[k]).all()) for k in ("polys", "r_bs", "E_eq", "tail", "folded_rW", "folded_X", "folded_comm")]
    ok += [bool((want[k] == got[k].read(0, m)).all()) for k, m in (("A", N), ("B", N), ("C", N), ("folded_W", nv))]
    q.put((rank, (tuple(ok), comm.stats()["exchanges"])))
    comm.close()
    ctx.close()
    g.close()


@pytest.mark.parametrize("n_inst,groups,small", [(4, 60, True), (8, 30, False)])
def test_cpp_driver_two_ranks_on_one_gpu(n_inst, groups, small):
    import mp_util

    res = mp_util.run_ranks(_worker_cpp, 2, (n_inst, groups, small))
    for r in (0, 1):
        ok, exchanges = res[r]
        assert ok == (True,) * 11, (r, ok)
    # instance data + c_vals + one exchange per local round + 2 layers + C partial + witness partial
    assert res[0][1] == res[1][1] == 2 + (n_inst // 2).bit_length() - 1 + 4


def test_cpp_driver_one_rank_rccl():
    """World of one over a real one-rank RCCL communicator: the production exchange backend, every all-gather a device / pinned-host round trip."""
    import ctypes

    import oracle_lib as ol
    import test_gpu_nifs_prove as tnp
    from spartan2_amd import frontend, hip, host

    ctx = hip.Context(0)
    L = ol.lib()
    okey = ctypes.c_void_p(L.orc_hyrax_setup(b"ck", ctypes.c_size_t(2048)))
    ck_aff, h_aff = np.zeros((2048, 8), dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    L.orc_hyrax_key_export(okey, ol.p64(ck_aff), ol.p64(h_aff))
    ck = hip.CommitmentKey(ctx, ck_aff, h_aff)
    insts = [frontend.synthetic_circuit(40, 5, num_public=2, precommitted_permille=1000, witness_seed=300 + s) for s in range(4)]
    oshape, shape, dims, Ws, X, r_W, comms = tnp._setup(ctx, okey, insts, np.random.default_rng(78))
    want = ol.nifs_prove(oshape, okey, comms, X, Ws, r_W, True, ol.Transcript(b"neutronnova_prove"), ol.transcript_round_hook(ol.Transcript(b"vc")))
    comm = host.Comm(0, 1, "rccl", device=0)
    tabs = [hip.Table.from_host(ctx, w) for w in Ws]
    got = host.nifs_prove_sharded(ctx, comm, shape, dims, ck, comms, X, tabs, r_W, True, hip.Transcript(ctx, b"neutronnova_prove"),
                                  ol.transcript_round_hook(ol.Transcript(b"vc")))
    for key in ("polys", "r_bs", "E_eq", "tail", "folded_rW", "folded_X", "folded_comm"):
        assert (want[key] == got[key]).all(), key
    for key, m in (("A", oshape.num_cons), ("B", oshape.num_cons), ("C", oshape.num_cons), ("folded_W", oshape.num_vars)):
        assert (want[key] == got[key].read(0, m)).all(), key
    # layers prepared once (prep_prove's cached_step_matvec / cached_step_i64) serve every prove: the rounds only read them
    prepared = host.nifs_prepare(ctx, shape, dims, X, tabs, True)
    for _ in range(2):
        again = host.nifs_prove_sharded(ctx, comm, shape, dims, ck, comms, X, tabs, r_W, True, hip.Transcript(ctx, b"neutronnova_prove"),
                                        ol.transcript_round_hook(ol.Transcript(b"vc")), prepared=prepared)
        for key in ("polys", "r_bs", "tail", "folded_rW", "folded_comm"):
            assert (want[key] == again[key]).all(), key
        for key, m in (("A", oshape.num_cons), ("C", oshape.num_cons), ("folded_W", oshape.num_vars)):
            assert (want[key] == again[key].read(0, m)).all(), key
    host.nifs_free(prepared)
    comm.close()
    L.orc_hyrax_free(okey)
    ctx.close()
