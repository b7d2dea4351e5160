#!/usr/bin/env python3
"""Bisects what bench.py's preamble does to its 8-proofs-in-flight leg (debug aid)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as B
from spartan2_amd import dist as spd

mode = sys.argv[1] if len(sys.argv) > 1 else ""
torch.cuda.set_device(0)
if "pin" in mode:
    print("pinned cpus", B._pin_to_gpu_numa(0))
group = spd.Group(backend="nccl") if "dist" in mode else None
from spartan2_amd import frontend, hip, host
inst = frontend.sha256_circuit(bytes(2048))
tape = np.random.default_rng(1).integers(0, 256, size=(4096, 64), dtype=np.uint8)
step = np.random.default_rng(2).integers(0, 256, size=(4096, 64), dtype=np.uint8)
if "main" in mode:
    ctx = hip.Context(0)
    sn0 = host.SpartanSNARK(ctx, inst)
    sn0.prep_prove(tape)
    sn0.set_flags(prefix_cache=False)
    for _ in range(22):
        sn0.prove(step)
    if "barrier" in mode and group:
        group.barrier()
P = 8
ctxs = [hip.Context(0) for _ in range(P)]
snarks = [host.SpartanSNARK(c, inst) for c in ctxs]
for sn in snarks:
    sn.prep_prove(tape)
    if "nowarm" not in mode:
        sn.prove(step)
ref = snarks[0].prove(step)[0]
per = 20
def worker(i):
    if "stagger" in mode:
        time.sleep(i * 0.00017)
    for _ in range(per):
        w = snarks[i].prove(step)[0]
        if "cmp" in mode:
            assert (w == ref).all()
threads = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in threads: t.start()
for t in threads: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(mode or "plain", f"{dt / (P * per) * 1e3:.3f} ms per proof amortised", flush=True)
os._exit(0)
