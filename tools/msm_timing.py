"""Times the device-resident-key MSM (n = 2048 full scalars), the fixed-base rows and the Hyrax commit of a bit witness."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spartan2_amd import hip, host
ctx = hip.Context(0)
g = host.from_label(b"ck", 2049)
key = hip.CommitmentKey(ctx, g[:2048], g[2048])
rng = np.random.default_rng(3)
raw = rng.integers(0, 2**63, size=(2048, 4), dtype=np.uint64)
raw[:, 3] &= np.uint64((1 << 62) - 1)
for _ in range(3):
    key.msm(raw)
ctx.reset_stats(True)
t = time.perf_counter()
K = 20
for _ in range(K):
    key.msm(raw)
dt = (time.perf_counter() - t) / K
print(f"msm_ck n=2048: {dt*1e3:.3f} ms per call")
for k in ("msm_sort", "msm_bucket_sum", "msm_window_reduce"):
    ms, n, _ = ctx.kernel_stats(k)
    print(f"  {k}: {ms/max(n,1)*1e3:.1f} us avg over {n}")
ctx.reset_stats(True)
t = time.perf_counter()
for _ in range(K):
    key.fixed_base_mul_h(raw[:84])
print(f"fixed_base_mul_h n=84: {(time.perf_counter()-t)/K*1e3:.3f} ms per call; kernel {ctx.kernel_stats('fixed_base')[0]/K*1e3:.1f} us")
