#!/usr/bin/env python3
"""poly_ABC alone on the config-2 shape (sha256_spartan 2 KiB), a few launches on a random rx table: the command behind the counter passes of
profiles/r02_polyabc_pmc.txt (rocprofv3 --pmc ... -- python tools/polyabc_prof.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from spartan2_amd import frontend, hip, host

ctx = hip.Context(0)
inst = frontend.sha256_circuit(bytes(int(os.environ.get("MSG", "2048"))))
mats, dims = host.pad_shape(inst)
shape = hip.Shape(ctx, mats, dims)
N = dims["num_cons"]
M = dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"]
rng = np.random.default_rng(5)
v = rng.integers(0, 1 << 63, size=(N, 4), dtype=np.uint64)
v[:, 3] &= np.uint64((1 << 62) - 1)
rx = hip.Table.from_host(ctx, v)
out = hip.Table.zeros(ctx, 2 * M)
r = v[7].copy()
for _ in range(3):
    shape.poly_abc(rx, r, 2 * M, out)
ctx.synchronize()
ctx.reset_stats(True)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    shape.poly_abc(rx, r, 2 * M, out)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"poly_ABC: {dt*1e6:.1f} us wall per call; event-timed {ctx.kernel_stats('poly_abc')[0] / reps * 1e3:.1f} us")
ctx.close()
