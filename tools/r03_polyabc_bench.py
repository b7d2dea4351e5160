#!/usr/bin/env python3
"""poly_ABC at config 2 (or argv[1] message bytes): HIP-event time of the default one-pass kernel class, 20 launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from spartan2_amd import frontend, hip, host

inst = frontend.sha256_circuit(bytes(int(sys.argv[1]) if len(sys.argv) > 1 else 2048))
ctx = hip.Context(0)
mats, dims = host.pad_shape(inst)
shape = hip.Shape(ctx, mats, dims)
N = dims["num_cons"]
M = dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"]
ell = N.bit_length() - 1
rng = np.random.default_rng(1)
r_x = rng.integers(0, 1 << 62, size=(ell, 4), dtype=np.uint64)
r = rng.integers(0, 1 << 62, size=4, dtype=np.uint64)
rx = hip.Table.eq(ctx, r_x)
out = hip.Table.zeros(ctx, 2 * M)
shape.poly_abc(rx, r, 2 * M, out)
ctx.reset_stats(True)
ctx.stats_filter("")
for _ in range(20):
    shape.poly_abc(rx, r, 2 * M, out)
ms, n, b = ctx.kernel_stats("poly_abc")
print(f"poly_abc: {ms / max(n, 1) * 1e3:.1f} us ({n} launches)")
ctx.reset_stats(False)
