#!/usr/bin/env python3
"""P independent proofs in flight on one GPU (one sp_ctx + host thread + helper thread each), every proof compared with the first: stress for the
mailbox / slot protocols and the resident-tail lease under contention. Prints one line per failure and a summary."""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from spartan2_amd import frontend, hip, host

ap = argparse.ArgumentParser()
ap.add_argument("--contexts", type=int, default=8)
ap.add_argument("--proofs", type=int, default=40)
ap.add_argument("--message-bytes", type=int, default=2048)
ap.add_argument("--tape-seed", type=int, default=1)
ap.add_argument("--step-seed", type=int, default=2)
ap.add_argument("--device", type=int, default=0)
ap.add_argument("--json", action="store_true", help="print one JSON object (bench.py's extra leg runs this tool in a process of its own)")
ap.add_argument("--torch-first", action="store_true", help="import torch and touch the device first, as bench.py does")
ap.add_argument("--main-ctx", action="store_true", help="keep one more context with a proved instance alive beside the P workers, as bench.py does")
ap.add_argument("--stats-pass", action="store_true", help="with --main-ctx: run an instrumented pass on the main context first, as bench.py does")
args = ap.parse_args()
if args.torch_first:
    import torch

    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
inst = frontend.sha256_circuit(bytes(args.message_bytes))
tape = np.random.default_rng(args.tape_seed).integers(0, 256, size=(4096, 64), dtype=np.uint8)
step = np.random.default_rng(args.step_seed).integers(0, 256, size=(4096, 64), dtype=np.uint8)
if args.main_ctx:
    mctx = hip.Context(args.device)
    msn = host.SpartanSNARK(mctx, inst)
    msn.prep_prove(tape)
    for _ in range(5):
        msn.prove(step)
    if args.stats_pass:
        mctx.reset_stats(True)
        mctx.stats_filter("")
        for _ in range(3):
            msn.prove(step)
        mctx.reset_stats(False)
ctxs = [hip.Context(args.device) for _ in range(args.contexts)]
snarks = [host.SpartanSNARK(c, inst) for c in ctxs]
for sn in snarks:
    sn.prep_prove(tape)
ref = snarks[0].prove(step)[0]
for sn in snarks[1:]:  # every context's first prove allocates its workspaces: outside the concurrent phase
    assert (sn.prove(step)[0] == ref).all()
errors, mismatches = [], 0
lock = threading.Lock()


def worker(i):
    global mismatches
    for k in range(args.proofs):
        try:
            w = snarks[i].prove(step)[0]
            if not (w == ref).all():
                with lock:
                    mismatches += 1
        except Exception as e:  # noqa: BLE001
            with lock:
                errors.append((i, k, str(e)))


ts = [threading.Thread(target=worker, args=(i,)) for i in range(args.contexts)]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
dt = time.perf_counter() - t0
# challenge-mailbox diagnostics summed over the contexts: answers taken from the host-memory mirror (the device line had not answered), watchdog trips
mail = [c.mail_stats() for c in ctxs]
mail_sum = {"mirror_answers": sum(m[0] for m in mail), "watchdog_trips": sum(m[1] for m in mail), "ring_in_device_memory": bool(mail[0][4]),
            "last_mirror_answer_wanted_vs_device_line": [(m[2], m[3]) for m in mail if m[0]][:4]}
for e in errors[:12]:
    print("ERROR ctx %d proof %d: %s" % e)
n = args.contexts * args.proofs
if args.json:
    import hashlib
    import json

    print(json.dumps({"proofs_in_flight": args.contexts, "proofs": n, "seconds": dt, "ms_per_proof_amortised": dt / n * 1e3, "errors": [e[2] for e in errors[:3]], "error_count": len(errors), "mail": mail_sum,
                      "mismatches": mismatches, "proof_sha256": hashlib.sha256(np.ascontiguousarray(ref).tobytes()).hexdigest(), "num_cons": inst.num_cons}))
    sys.exit(0)
print(f"{n} proofs, {len(errors)} errors, {mismatches} mismatches, {dt / n * 1e3:.3f} ms per proof amortised; mailbox {mail_sum}")
sys.exit(1 if errors or mismatches else 0)
