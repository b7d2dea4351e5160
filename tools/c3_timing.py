#!/usr/bin/env python3
"""NeutronNovaZkSNARK::prove at BASELINE config 3 without PyTorch in the process (bench.py --workload c3 is the same loop with it):
usage: c3_timing.py [--torch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if "--torch" in sys.argv:
    import torch

    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
from spartan2_amd import frontend, hip, host

ctx = hip.Context(0)
circs = [frontend.sha256_step_circuit(bytes([i]) * 64) for i in range(32)]
core = frontend.sha256_step_circuit(bytes(64))
nn = host.NeutronNovaZkSNARK(ctx, circs, core)
tape = np.random.default_rng(0xC3).integers(0, 256, size=(32768, 64), dtype=np.uint8)
used = nn.prep_prove(tape)
for _ in range(3):
    nn.prove(tape[used:])
t0 = time.perf_counter()
n = 10
acc = {}
for _ in range(n):
    _, _, ph = nn.prove(tape[used:])
    for k, v in ph.items():
        acc[k] = acc.get(k, 0) + v / n
print(f"{'with' if '--torch' in sys.argv else 'without'} torch: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per prove", {k: round(v, 2) for k, v in acc.items()})
nn.close()
ctx.close()
