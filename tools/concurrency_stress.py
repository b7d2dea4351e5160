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
args = ap.parse_args()
inst = frontend.sha256_circuit(bytes(args.message_bytes))
tape = np.random.default_rng(1).integers(0, 256, size=(4096, 64), dtype=np.uint8)
step = np.random.default_rng(2).integers(0, 256, size=(4096, 64), dtype=np.uint8)
ctxs = [hip.Context(0) for _ in range(args.contexts)]
snarks = [host.SpartanSNARK(c, inst) for c in ctxs]
for sn in snarks:
    sn.prep_prove(tape)
ref = snarks[0].prove(step)[0]
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
for e in errors[:12]:
    print("ERROR ctx %d proof %d: %s" % e)
n = args.contexts * args.proofs
print(f"{n} proofs, {len(errors)} errors, {mismatches} mismatches, {dt / n * 1e3:.3f} ms per proof amortised")
sys.exit(1 if errors or mismatches else 0)
