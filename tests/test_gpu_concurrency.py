"""GPU: several independent proofs in flight on ONE device (one sp_ctx, one host thread and one helper thread each): the challenge mailbox, the
self-validating result slots and the resident-tail lease under contention. Every proof must equal the single-context proof and no kernel may
run into its mailbox watchdog (r2: a release-only fence in the resident tail made 3 of 160 concurrent proofs time out)."""
import threading

import numpy as np
import pytest

import oracle_lib as ol
from spartan2_amd import frontend, hip, host

pytestmark = pytest.mark.gpu


def test_eight_proofs_in_flight_are_all_the_same_proof():
    inst = frontend.sha256_circuit(bytes(2048))
    tape, step = ol.make_tape(1, 4096), ol.make_tape(2, 4096)
    from spartan2_amd.dist import cpu_budget

    # one polling owner thread per context (+ a mostly sleeping helper): stay within half of the CPU quota of the box (16 on the bench boxes: 8)
    P, per = max(2, min(8, cpu_budget() // 2)), 20
    ctxs = [hip.Context(0) for _ in range(P)]
    snarks = [host.SpartanSNARK(c, inst) for c in ctxs]
    for sn in snarks:
        sn.prep_prove(tape)
    ref = snarks[0].prove(step)[0]
    for sn in snarks[1:]:  # first proves allocate the contexts' workspaces: one at a time, before the proofs in flight
        assert (sn.prove(step)[0] == ref).all()
    errors, bad = [], []

    def worker(i):
        for k in range(per):
            try:
                if not (snarks[i].prove(step)[0] == ref).all():
                    bad.append((i, k))
            except Exception as e:  # noqa: BLE001
                errors.append((i, k, str(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for sn in snarks:
        sn.close()
    for c in ctxs:
        c.close()
    assert not errors, errors[:3]
    assert not bad, bad[:3]
