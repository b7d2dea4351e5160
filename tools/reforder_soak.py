#!/usr/bin/env python3
"""Soak of the announced opening (reference-order driver) against the headline driver on ONE context: N proves, the driver switched every few proves
(the two paths share the context's helper thread, the auxiliary stream's lanes and the mapped landing buffers), every proof compared with the first.
Usage: tools/reforder_soak.py [N=20000]. Prints the slowest proves of either kind and the number of mismatches (must be 0)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from spartan2_amd import frontend, hip, host

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
inst = frontend.sha256_circuit(bytes(2048))
ctx = hip.Context(0)
sn = host.SpartanSNARK(ctx, inst)
tape = np.random.default_rng(1).integers(0, 256, size=(4096, 64), dtype=np.uint8)
sn.prep_prove(tape)
step = np.random.default_rng(2).integers(0, 256, size=(4096, 64), dtype=np.uint8)
base = sn.prove(step)[0]
bad = 0
times = {True: [], False: []}
ref = False
for i in range(N):
    if i % 3 == 0:
        ref = not ref
        sn.set_flags(reference_order=ref)
    t0 = time.perf_counter()
    w = sn.prove(step)[0]
    times[ref].append(time.perf_counter() - t0)
    if not (w == base).all():
        bad += 1
        print(f"prove {i} (reference_order={ref}): MISMATCH", file=sys.stderr)
for k, name in ((False, "headline"), (True, "reference order")):
    t = np.array(times[k]) * 1e3
    print(f"{name}: {len(t)} proves, median {np.median(t):.3f} ms, p99 {np.percentile(t, 99):.3f}, max {t.max():.3f}, over 2 ms: {int((t > 2).sum())}")
print(f"mismatches: {bad}")
sys.exit(1 if bad else 0)
