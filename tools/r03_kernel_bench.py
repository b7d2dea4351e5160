#!/usr/bin/env python3
"""Round-3 micro-benchmarks through the C ABI (HIP events of the library's own instrumentation):
  rowmat   bind_with_delayed 512 x 2048 and 2048 x 2048: the one-launch streaming kernel against the two-stage form
  walk     one table-walk MSM over 2049 window tables (the key + h): k_multi_mul_wide (decides, read once
           per process: run twice), one walk and two walks side by side (the two lanes)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

from spartan2_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=("rowmat", "walk"))
args = ap.parse_args()
ctx = hip.Context(0)
rng = np.random.default_rng(3)


def rand_fe(n):
    v = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 61) - 1)
    return v


if args.what == "rowmat":
    for rows, cols in ((512, 2048), (2048, 2048), (128, 2048)):
        t = hip.Table.from_host(ctx, rand_fe(rows * cols))
        L = rand_fe(rows)
        for tall in ("1",):
            hip.rowmat_vec(ctx, t, rows, cols, L)
            ctx.reset_stats(True)
            ctx.stats_filter("rowmat_vec")
            for _ in range(20):
                hip.rowmat_vec(ctx, t, rows, cols, L)
            ms, n, b = ctx.kernel_stats("rowmat_vec")
            ctx.reset_stats(False)
            print(f"rowmat_vec {rows} x {cols} tall={tall}: {ms / n * 1e3:.1f} us per call, {b / n / (ms / n * 1e-3) / 1e9:.0f} GB/s algorithmic ({b / n / 1e6:.1f} MB)")
        t.free()
else:
    import ctypes

    import oracle_lib as ol

    n = 2049
    pts = np.zeros((n, 8), dtype=np.uint64)
    ol.lib().orc_from_label(b"walk_bench", ctypes.c_size_t(n), ol.p64(pts))
    tabs = hip.FixedBaseTables(ctx, pts)
    sc = rand_fe(n)
    tabs.multi_mul(sc)
    ctx.reset_stats(True)
    ctx.stats_filter("multi_mul")
    t0 = time.perf_counter()
    for _ in range(20):
        tabs.multi_mul(sc)
    wall = (time.perf_counter() - t0) / 20
    ms, k, _ = ctx.kernel_stats("multi_mul")
    ctx.reset_stats(False)
    print(f"walk over {n} tables : kernel {ms / k * 1e3:.1f} us, call {wall * 1e6:.1f} us")
