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
    """In a process of its own, as bench.py runs the same leg: the library's callers (the C++ drivers, a Rust host) do not carry PyTorch or the
    oracle's OpenMP pool, whose threads compete with the polling owner threads for the CPU quota of the box."""
    import json
    import os
    import subprocess
    import sys

    from spartan2_amd.dist import cpu_budget

    # one polling owner thread per context (+ two mostly sleeping helpers): stay within half of the CPU quota of the box (16 on the bench boxes: 8)
    P, per = max(2, min(8, cpu_budget() // 2)), 20
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "concurrency_stress.py"), "--contexts", str(P), "--proofs", str(per), "--json"], capture_output=True,
                       text=True, timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert line, (r.stdout + r.stderr)[-2000:]
    out = json.loads(line[-1])
    assert out["proofs"] == P * per and not out["errors"] and out["mismatches"] == 0, out
