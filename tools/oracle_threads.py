"""CPU oracle prove() of the bench instance at several OpenMP thread counts (the cpu_baseline leg of bench.py uses all cores)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from spartan2_amd import frontend

inst = frontend.sha256_circuit(bytes(2048))
osp = ol.OracleSpartan(inst)
osp.prep_prove(ol.make_tape(1, 4096))
ref = None
for th in (64, 32, 16, 8, 4, 1):
    n = ol.lib().orc_set_threads(th)
    osp.prove(ol.make_tape(2, 4096))
    w, _, secs = osp.prove(ol.make_tape(2, 4096))
    ref = w if ref is None else ref
    print(f"threads {n:4d}: {secs * 1e3:8.1f} ms  same proof: {bool((w == ref).all())}")
