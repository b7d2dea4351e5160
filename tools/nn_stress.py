#!/usr/bin/env python3
"""Repeated NeutronNovaZkSNARK::prove / verify on one prep state (the helper thread's jobs on the second context, the tape-position checks, the mapped
fixed-base slots under many repetitions): every proof must verify on the device-backed verifier, a sample of them on the oracle's."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import oracle_lib as ol
from spartan2_amd import frontend, hip, host

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = hip.Context(0)
steps = [frontend.sha256_step_circuit(bytes([i]) * 64) for i in range(32)]
core = frontend.sha256_step_circuit(bytes(64))
nn = host.NeutronNovaZkSNARK(ctx, steps, core)
onn = ol.OracleNeutronNova(steps, core)
tape = np.random.default_rng(5).integers(0, 256, size=(32768, 64), dtype=np.uint8)
used = nn.prep_prove(tape)
bad = 0
t0 = time.perf_counter()
for k in range(reps):
    w, _, _ = nn.prove(tape[used + (k % 7):])
    rc = nn.verify(w)
    if rc != 0:
        bad += 1
        print("proof", k, "rejected by the device-backed verifier:", rc)
    if k % 50 == 0 and onn.verify_words(w) != 0:
        bad += 1
        print("proof", k, "rejected by the oracle's verifier")
dt = time.perf_counter() - t0
print(f"{reps} prove + verify pairs, {bad} failures, {dt / reps * 1e3:.2f} ms per pair")
sys.exit(1 if bad else 0)
