#!/usr/bin/env python3
"""Per-kernel timing of the NeutronNova NIFS rounds (sp_nifs_*) at BASELINE config 3 size (32 step instances x 2^15 constraints) and at a
2^20-constraint size, with achieved algorithmic GB/s per kernel class. Layers are synthetic small values (bits / small signed), i.e. what SHA
step circuits produce. Usage: python tools/nifs_bench.py [n_instances log2_cons reps]..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from spartan2_amd import hip, host


def small_layers(rng, n, total):
    """Montgomery limbs of small values without per-element Python: values in {0,1,2,3} * R mod p via a 4-entry lookup."""
    P = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF
    R = 1 << 256
    lut = np.array([[((v * R % P) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in range(4)], dtype=np.uint64)
    return lut[rng.integers(0, 4, size=(n, total))]


def run(ctx, n_inst, log_cons, reps, small):
    rng = np.random.default_rng(1)
    ell, left, right = host.tensor_decomp(1 << log_cons)
    total = left * right
    nifs = hip.Nifs(ctx, n_inst, left, right)
    layers = [small_layers(rng, n_inst, total) for _ in range(3)]
    E = hip.pow_split_evals(np.array([3, 5, 7, 11], dtype=np.uint64), ell, left, right)
    ell_b = n_inst.bit_length() - 1
    rhos = np.array([[17 + i, 1, 2, 3] for i in range(ell_b)], dtype=np.uint64)
    oa, ob, oc = (hip.Table.zeros(ctx, total) for _ in range(3))
    walls = []
    for rep in range(reps + 1):
        for which in range(3):
            for b in range(n_inst):
                v = nifs.layer(which, b)
                v.write(0, layers[which][b])
                v.free()
        if rep == 1:
            ctx.reset_stats(True)
        ctx.synchronize()
        t0 = time.perf_counter()
        nifs.begin(E, rhos, small_values=small)
        t1 = time.perf_counter()
        for t in range(ell_b):
            nifs.round(t)
            nifs.challenge(np.array([1000 + t, 2, 3, 4], dtype=np.uint64))
        nifs.finish(oa, ob, oc)
        ctx.synchronize()
        t2 = time.perf_counter()
        if rep:
            walls.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    out = {"instances": n_inst, "log2_cons": log_cons, "small_values": small, "begin_ms": float(np.mean([w[0] for w in walls])),
           "rounds_and_finish_ms": float(np.mean([w[1] for w in walls])), "kernels": {}}
    for k in ("nifs_to_small", "nifs_cvals", "nifs_round0", "nifs_round0_small", "nifs_fold_prove", "nifs_fold", "fold_tables"):
        ms, launches, nbytes = ctx.kernel_stats(k)
        if launches:
            out["kernels"][k] = {"launches_per_run": launches / reps, "ms_per_run": ms / reps, "alg_GBps": nbytes / ms / 1e6, "frac_of_8TBps": nbytes / ms / 1e6 / 8000}
    ctx.reset_stats(False)
    nifs.free()
    return out


if __name__ == "__main__":
    ctx = hip.Context(0)
    cfgs = [(32, 15, 5), (64, 20, 2)] if len(sys.argv) < 4 else [tuple(int(x) for x in sys.argv[i : i + 3]) for i in range(1, len(sys.argv) - 2, 3)]
    for n_inst, lc, reps in cfgs:
        for small in (False, True):
            print(json.dumps(run(ctx, n_inst, lc, reps, small)))
    ctx.close()
