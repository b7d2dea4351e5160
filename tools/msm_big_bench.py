#!/usr/bin/env python3
"""One general Pippenger MSM (sp_msm_points, kernels_pippenger.hpp) on device-resident operands: wall time and per-stage HIP-event time per window width.
    python tools/msm_big_bench.py [log2 n = 20] [windows = 0,10,12,13,14]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from spartan2_amd import hip, host  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
windows = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,10,12,13,14").split(",")]
n = 1 << log_n
ctx = hip.Context(0)
gens = host.from_label(b"ck", 5)
key = hip.CommitmentKey(ctx, gens[:4], gens[4])
rng = np.random.default_rng(1)


def rand_fe(k):
    v = rng.integers(0, 1 << 62, size=(k, 4), dtype=np.uint64) * np.uint64(4) + rng.integers(0, 4, size=(k, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 63) - 1)
    return v


pts = np.zeros((n, 8), dtype=np.uint64)
for lo in range(0, n, 1 << 16):
    pts[lo:lo + (1 << 16)] = key.fixed_base_mul_h(rand_fe(min(1 << 16, n - lo)))
dev = hip.Points(ctx, pts)
tab = hip.Table.from_host(ctx, rand_fe(n))
for w in windows:
    hip.msm_points(ctx, tab, 0, n, dev, 0, w)  # warm-up (workspaces)
    ctx.reset_stats(True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        hip.msm_points(ctx, tab, 0, n, dev, 0, w)
    wall = (time.perf_counter() - t0) / reps
    st = {k: ctx.kernel_stats(k) for k in ("msm_big_sort", "msm_big_buckets", "msm_big_window")}
    ctx.reset_stats(False)
    stages = {k: v[0] / max(v[1], 1) for k, v in st.items()}
    wn = w if w else hip.lib().sp_msm_pippenger_window(ctypes.c_size_t(n)) if hasattr(hip.lib(), "sp_msm_pippenger_window") else 0
    nwin = -(-257 // wn) if wn else 0
    print(f"n=2^{log_n} window={w or ('auto:%d' % wn)}: wall {wall * 1e3:.3f} ms = {n / wall / 1e6:.1f} M pairs/s; stages (ms): sort {stages['msm_big_sort']:.3f}, "
          f"buckets {stages['msm_big_buckets']:.3f}, window sums {stages['msm_big_window']:.3f}" + (f"; {n * nwin / (stages['msm_big_buckets'] * 1e-3) / 1e9:.2f} G mixed additions/s in the bucket kernel" if nwin and stages['msm_big_buckets'] > 0 else ""), flush=True)
ctx.close()
