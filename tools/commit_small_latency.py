#!/usr/bin/env python3
"""Latency of one narrow-key commitment (sp_hyrax_commit_small, width-32 key + blind: the per-round commitment of the ZK verifier circuit,
hyrax_pc.rs:221-260): copy in, one launch of per-scalar table walks (five cooperative addition levels), copy out, 33 additions + one inversion on the
host. Measured 96 us (58 us of it the kernel). A one-launch form — scalars read from a mapped page, the 33 points added by the last block, result polled
from a mapped slot — was measured at 150 us: six more dependent addition levels on the device (about 10 us each) cost far more than the 33 host additions
(13 us) and the two small copies they replace, so the tree stays split where it is."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes

import numpy as np

import oracle_lib as ol
from oracle_lib import lib as olib, p64
from spartan2_amd import hip

width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = hip.Context(0)
gs = np.zeros((width + 1, 8), dtype=np.uint64)
olib().orc_from_label(b"narrow_key_bench", ctypes.c_size_t(width + 1), p64(gs))
k = hip.CommitmentKey(ctx, gs[:width], gs[width])
rng = np.random.default_rng(1)
sc = ol.random_field_array(rng, width)
blind = ol.random_field_array(rng, 1)[0]
for _ in range(20):
    ref = k.commit_small(sc, blind)
t0 = time.perf_counter()
for _ in range(reps):
    k.commit_small(sc, blind)
dt = (time.perf_counter() - t0) / reps
ctx.reset_stats(True)
for _ in range(50):
    k.commit_small(sc, blind)
st = ctx.kernel_stats("fixed_base")
ctx.reset_stats(False)
print("width %d: %.1f us per commitment; fixed_base kernel (ms, launches, bytes) over 50 calls: %s" % (width, dt * 1e6, st))
