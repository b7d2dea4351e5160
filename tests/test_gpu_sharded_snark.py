"""GPU: ONE SpartanSNARK proof sharded over several ranks (spartan2_amd/host/sharded_snark.cpp: row-sharded commitment and Az/Bz/Cz, slice-sharded
sum-checks with one exchange per round, column-sharded poly_ABC, point-range MSMs) — bit-identical to the CPU oracle's UNSHARDED proof. The test
box has one GPU, so 2 and 4 ranks share it and exchange through gloo (the callback backend of the C++ exchange layer); the RCCL backend itself
is exercised with a one-rank communicator (ncclCommInitRank / ncclAllGather from C++) — RCCL refuses two ranks on one device."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, which, gather_log2):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # where the sum-checks hand over from slices (one exchange per round) to gathered tables (no exchange): read once per process by the driver
    os.environ["SPARTAN_SHARD_GATHER_LOG2"] = str(gather_log2)
    import oracle_lib as ol
    from spartan2_amd import dist as spd, frontend, hip, host

    g = spd.Group(backend="gloo")
    ctx = hip.Context(0)
    comm = host.Comm(rank, world, "torch")
    inst = {"synthetic": lambda: frontend.synthetic_circuit(150, 0xDEADBEEF, num_public=5),
            "segments": lambda: frontend.synthetic_circuit(220, 21, num_public=3, shared_permille=300, precommitted_permille=400),
            "sha256": lambda: frontend.sha256_circuit(bytes(range(150)))}[which]()
    tape = ol.make_tape(77, 8192)
    sn = host.ShardedSpartanSNARK(ctx, comm, inst)
    used = sn.prep_prove(tape)
    got, used2, phases = sn.prove(tape[used:])
    again, _, _ = sn.prove(tape[used:])  # the prep state is reusable
    out = None
    everyone = comm.allgather(got)
    same_on_all_ranks = bool((everyone == got).all())
    if rank == 0:
        osp = ol.OracleSpartan(inst)
        assert osp.prep_prove(tape) == used
        want, ou2, _ = osp.prove(tape[used:])
        out = (bool(len(want) == len(got) and (want == got).all()), bool((again == got).all()), ou2 == used2, osp.verify_words(got) == 0, same_on_all_ranks,
               int(phases["exchanges"]))
    q.put((rank, out))
    sn.close()
    comm.close()
    ctx.close()
    g.close()


# gather_log2: 0 = hand over at one element per rank (an exchange in every slice round: the round-1 protocol); 8 = a few slice rounds, then one bulk
# hand-over of 2^8-element tables; 16 (the default) = these small instances are gathered at once and every rank runs the sum-checks alone
@pytest.mark.parametrize("world,which,gather_log2", [(2, "synthetic", 0), (4, "synthetic", 0), (2, "segments", 8), (4, "sha256", 8), (4, "synthetic", 8),
                                                     (2, "sha256", 16), (4, "segments", 5)])
def test_sharded_prove_is_the_unsharded_proof(world, which, gather_log2):
    import mp_util

    res = mp_util.run_ranks(_worker, world, (which, gather_log2))
    assert res[0][:5] == (True,) * 5, res[0]
    if gather_log2 == 0:
        assert res[0][5] > 20  # one exchange per local sum-check round + commitment + finals + opening
    elif gather_log2 == 16:
        assert res[0][5] <= 16  # commitment rows, 3 + 2 table hand-overs, the opening record (per prove): no per-round exchange at all


def test_rccl_backend_one_rank_and_world_of_one_prove():
    """The production exchange backend (ncclCommInitRank + ncclAllGather, called from C++) on a one-rank communicator, under the sharded driver
    with world = 1: the proof equals the oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from spartan2_amd import frontend, hip, host

    ctx = hip.Context(0)
    comm = host.Comm(0, 1, "rccl", device=0)
    x = np.arange(1000, dtype=np.uint64)
    assert (comm.allgather(x)[0] == x).all()
    big = np.arange(1 << 16, dtype=np.uint64)  # staging buffers grow
    assert (comm.allgather(big)[0] == big).all()
    inst = frontend.synthetic_circuit(60, 5, num_public=3)
    tape = ol.make_tape(5, 8192)
    sn = host.ShardedSpartanSNARK(ctx, comm, inst)
    used = sn.prep_prove(tape)
    got, _, phases = sn.prove(tape[used:])
    osp = ol.OracleSpartan(inst)
    osp.prep_prove(tape)
    want = osp.prove(tape[used:])[0]
    assert (want == got).all()
    assert comm.stats()["exchanges"] >= 4
    sn.close()
    comm.close()
    ctx.close()
