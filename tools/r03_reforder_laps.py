#!/usr/bin/env python3
"""Reference-order driver at BASELINE config 2 with the host-side laps of sp_hyrax_prove printed (SPARTAN_HOST_LAPS=1 in the environment)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from spartan2_amd import frontend, hip, host

inst = frontend.sha256_circuit(bytes(2048))
ctx = hip.Context(0)
sn = host.SpartanSNARK(ctx, inst)
tape = np.random.default_rng(1).integers(0, 256, size=(4096, 64), dtype=np.uint8)
used = sn.prep_prove(tape)
step = np.random.default_rng(2).integers(0, 256, size=(4096, 64), dtype=np.uint8)
base = sn.prove(step)[0]
sn.set_flags(reference_order=True)
for i in range(6):
    t0 = time.perf_counter()
    w, _, ph = sn.prove(step)
    dt = time.perf_counter() - t0
    print(f"prove {i}: {dt * 1e3:.3f} ms identical={bool((w == base).all())} phases={ {k: round(v, 3) for k, v in ph.items()} }", file=sys.stderr)
