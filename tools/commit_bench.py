#!/usr/bin/env python3
"""Hyrax commit throughput (PCS::commit, hyrax_pc.rs:207-303) at BASELINE config 4's shape: 2^22 full-width scalars = 2048 row MSMs of 2048
points over one base vector, on one MI355X; also the bit-witness commit of config 2 (2^20 bits). Prints points/s and per-kernel times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from spartan2_amd import hip, host

ctx = hip.Context(0)
g = host.from_label(b"ck", 2049)
key = hip.CommitmentKey(ctx, g[:2048], g[2048])
rng = np.random.default_rng(7)
for log_n, kind in ((18, "full"), (22, "full"), (20, "bits")):
    n = 1 << log_n
    if kind == "full":
        v = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
        v[:, 3] &= np.uint64((1 << 63) - 1)  # < 2^255 < p: a canonical element (its Montgomery reading is just another uniform element)
    else:
        one = np.array([1, 0xFFFFFFFF00000000, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFE], dtype=np.uint64)  # R mod p = Montgomery form of 1
        v = np.zeros((n, 4), dtype=np.uint64)
        v[rng.integers(0, 2, size=n) == 1] = one
    rows = n // 2048
    blinds = rng.integers(0, 1 << 62, size=(rows, 4), dtype=np.uint64)
    t = hip.Table.from_host(ctx, v)
    key.commit(t, 0, n, blinds)  # warm-up
    ctx.reset_stats(True)
    t0 = time.perf_counter()
    key.commit(t, 0, n, blinds)
    dt = time.perf_counter() - t0
    ks = {k: ctx.kernel_stats(k)[0] for k in ("msm_rows_comb", "msm_rows_sort", "msm_rows_bucket_sum", "msm_rows_window_reduce", "msm_rows_horner", "msm_binary_rows", "fixed_base")}
    ctx.reset_stats(False)
    print(f"commit 2^{log_n} {kind}: {rows} rows, {dt*1e3:.1f} ms, {n/dt/1e6:.1f} M (scalar, base) pairs/s; kernel ms: " + ", ".join(f"{k}={v:.2f}" for k, v in ks.items() if v))
    t.free()
ctx.close()
