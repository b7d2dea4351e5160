"""Per-round cost of the batched ZK sum-check drivers (sp_sumcheck_cubic_outer_pow_batched / sp_sumcheck_quad_batched) with a trivial hook."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from spartan2_amd import hip, host
ctx=hip.Context(0)
rng=np.random.default_rng(1)
def rnd(n):
    v=rng.integers(0,1<<63,size=(n,4),dtype=np.uint64); v[:,3]&=np.uint64((1<<62)-1); return v
ell=15; n=1<<ell
_,left,right=host.tensor_decomp(n)
E=hip.pow_split_evals(rnd(1)[0],ell,left,right)
const=rnd(1)[0]
for trial in range(3):
    pl,pr=hip.Table.from_host(ctx,E[:left]),hip.Table.from_host(ctx,E[left:])
    ts=[hip.Table.from_host(ctx,rnd(n)) for _ in range(3)]; tc=[hip.Table.from_host(ctx,rnd(n)) for _ in range(3)]
    ctx.synchronize(); t0=time.perf_counter()
    hip.sumcheck_cubic_outer_pow_batched(ctx,ell,pl,pr,ts,tc,const,0,lambda r,a,b: const)
    ctx.synchronize(); t1=time.perf_counter()
    A=[hip.Table.from_host(ctx,rnd(2*n)) for _ in range(4)]
    ctx.synchronize(); t2=time.perf_counter()
    hip.sumcheck_quad_batched(ctx,np.stack([const,const]),ell+1,*A,0,lambda r,a,b: const)
    ctx.synchronize(); t3=time.perf_counter()
    print(f"cubic batched {ell} rounds: {(t1-t0)*1e3:.3f} ms ({(t1-t0)/ell*1e6:.0f} us/round); quad batched {ell+1} rounds: {(t3-t2)*1e3:.3f} ms ({(t3-t2)/(ell+1)*1e6:.0f} us/round)")
