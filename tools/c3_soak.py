#!/usr/bin/env python3
"""NeutronNovaZkSNARK::prove at BASELINE config 3, N consecutive proves: the distribution of the step times and the phases of every step above twice the
median (a polling host thread that keeps a library helper thread off its core shows here as one 15-20 ms prove in a few dozen).
usage: c3_soak.py [N=300]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from spartan2_amd import frontend, hip, host

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = hip.Context(0)
circs = [frontend.sha256_step_circuit(bytes([i]) * 64) for i in range(32)]
core = frontend.sha256_step_circuit(bytes(64))
nn = host.NeutronNovaZkSNARK(ctx, circs, core)
tape = np.random.default_rng(0xC3).integers(0, 256, size=(32768, 64), dtype=np.uint8)
used = nn.prep_prove(tape)
for _ in range(3):
    nn.prove(tape[used:])
ts, phs = [], []
for _ in range(n):
    t0 = time.perf_counter()
    _, _, ph = nn.prove(tape[used:])
    ts.append((time.perf_counter() - t0) * 1e3)
    phs.append(ph)
a = np.array(ts)
med = float(np.median(a))
print(f"{n} proves: mean {a.mean():.3f} median {med:.3f} min {a.min():.3f} p90 {np.percentile(a, 90):.3f} p99 {np.percentile(a, 99):.3f} max {a.max():.3f} ms; over 2x median: {(a > 2 * med).sum()}")
for i in np.nonzero(a > 2 * med)[0][:10]:
    print(f"  step {i}: {a[i]:.2f} ms", {k: round(v, 2) for k, v in phs[i].items()})
nn.close()
ctx.close()
