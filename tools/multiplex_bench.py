#!/usr/bin/env python3
"""P proofs in flight on T polling threads (ss_prove_multiplexed: every proof on a stack of its own, the library's wait hook switches between them):
amortised time per proof, every proof compared with a plain single prove. Prints one line per (P, T) or one JSON object (--json)."""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from spartan2_amd import frontend, hip, host

ap = argparse.ArgumentParser()
ap.add_argument("--sweep", default="8x1,8x2,16x2,16x4,24x4,32x4", help="comma-separated contexts x threads")
ap.add_argument("--proofs", type=int, default=40, help="proofs per context")
ap.add_argument("--message-bytes", type=int, default=2048)
ap.add_argument("--device", type=int, default=0)
ap.add_argument("--json", action="store_true")
args = ap.parse_args()
inst = frontend.sha256_circuit(bytes(args.message_bytes))
tape = np.random.default_rng(1).integers(0, 256, size=(4096, 64), dtype=np.uint8)
step = np.random.default_rng(2).integers(0, 256, size=(4096, 64), dtype=np.uint8)
pairs = [tuple(int(x) for x in p.split("x")) for p in args.sweep.split(",")]
nmax = max(p for p, _ in pairs)
ctxs = [hip.Context(args.device) for _ in range(nmax)]
snarks = [host.SpartanSNARK(c, inst) for c in ctxs]
for sn in snarks:
    sn.prep_prove(tape)
ref = snarks[0].prove(step)[0]
for sn in snarks[1:]:  # every context's first prove allocates its workspaces: outside the measured phase
    assert (sn.prove(step)[0] == ref).all()
L = host.lib()
pub = np.ascontiguousarray(inst.publics, dtype=np.uint64)
results = []
for P, T in pairs:
    pks = (ctypes.c_void_p * P)(*[s.pk for s in snarks[:P]])
    pss = (ctypes.c_void_p * P)(*[s.ps for s in snarks[:P]])
    out = np.zeros((P, len(ref)), dtype=np.uint64)
    stats = (ctypes.c_double * 4)()
    rc = L.ss_prove_multiplexed(pks, pss, ctypes.c_size_t(P), hip.p64(pub) if len(pub) else None, ctypes.c_size_t(len(pub)), hip.p8(step), ctypes.c_size_t(step.shape[0]),
                                ctypes.c_size_t(args.proofs), ctypes.c_size_t(T), hip.p64(out), ctypes.c_size_t(len(ref)), stats)
    err = L.ss_last_error().decode() if rc else ""
    mism = int((out != ref).any(axis=1).sum()) if rc == 0 else -1
    n = P * args.proofs
    r = {"proofs_in_flight": P, "polling_threads": T, "proofs": n, "seconds": stats[0], "ms_per_proof_amortised": stats[0] / n * 1e3, "context_switches": int(stats[1]), "host_busy_ms_per_proof": (stats[0] * P - stats[3]) / n * 1e3 if T >= P else None,
         "mismatches": mism, "rc": rc, "error": err}
    results.append(r)
    if not args.json:
        print(f"{P} in flight on {T} threads: {r['ms_per_proof_amortised']:.3f} ms per proof amortised ({n} proofs, {mism} mismatches, rc {rc} {err}), "
              f"{int(stats[1]) / max(n, 1):.0f} switches per proof, in polls {stats[3] / n * 1e3:.3f} ms per proof", flush=True)
if args.json:
    L.ss_last_error.restype = ctypes.c_char_p
    print(json.dumps({"results": results, "proof_sha256": hashlib.sha256(np.ascontiguousarray(ref).tobytes()).hexdigest(), "num_cons": inst.num_cons}))
sys.exit(1 if any(r["rc"] or r["mismatches"] for r in results) else 0)
