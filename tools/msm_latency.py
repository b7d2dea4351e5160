"""Latency of one 1024-point MSM with full-width scalars through the C ABI (upload, digits, sort, bucket sums, window reduce, host Horner):
the shape of comm_LZ / delta on the sha256_spartan path. Usage: python tools/msm_latency.py [n]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle_lib as ol
from oracle_lib import lib as olib, p64
from spartan2_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = hip.Context(0)
g = np.zeros((n, 8), dtype=np.uint64)
olib().orc_from_label(b"ck", ctypes.c_size_t(n), p64(g))
rng = np.random.default_rng(1)
s = ol.random_field_array(rng, n)
want = np.zeros(8, dtype=np.uint64)
olib().orc_msm(p64(s), p64(g), ctypes.c_size_t(n), ctypes.c_size_t(1), p64(want))
for _ in range(5):
    got = hip.msm(ctx, s, g)
assert (got == want).all()
t0 = time.perf_counter()
K = 50
for _ in range(K):
    hip.msm(ctx, s, g)
print(f"n={n}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per MSM (bit-exact vs oracle)")
