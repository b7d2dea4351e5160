#!/usr/bin/env python3
"""The command behind profiles/<round>_kernel_stats.md (tools/profile_job.sh) (run under rocprofv3 --kernel-trace, and under the two --pmc passes).
  prove   config-2 (sha256 2048 B) SpartanSNARK::prove only, the headline driver: 2 warm-up + N proves. Nothing else in the process, so
          every launch in the trace belongs to a C2 prove and a (kernel, grid) pair names ONE call site.
  solo    the same kernels at the same sizes, one ABI call at a time with a device sync in between: nothing else on the GPU while a kernel
          runs. The difference between the two columns of the report is what the overlap inside a prove costs the kernel."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from spartan2_amd import frontend, hip, host

ap = argparse.ArgumentParser()
ap.add_argument("mode", choices=("prove", "solo"))
ap.add_argument("--proves", type=int, default=10)
ap.add_argument("--reference-order", action="store_true")
args = ap.parse_args()
ctx = hip.Context(0)
inst = frontend.sha256_circuit(bytes(2048))
rng = np.random.default_rng(11)


def rand_fe(n):
    v = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 61) - 1)
    return v


if args.mode == "prove":
    snark = host.SpartanSNARK(ctx, inst)
    tape = np.random.default_rng(1).integers(0, 256, size=(4096, 64), dtype=np.uint8)
    snark.prep_prove(tape)
    step = np.random.default_rng(2).integers(0, 256, size=(4096, 64), dtype=np.uint8)
    snark.set_flags(prefix_cache=False, reference_order=args.reference_order)
    for _ in range(2 + args.proves):
        snark.prove(step)
    print("proves:", args.proves, "reference_order:", args.reference_order)
else:
    mats, dims = host.pad_shape(inst)
    shape = hip.Shape(ctx, mats, dims)
    N, M = dims["num_cons"], dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"]
    print("num_cons", N, "num_vars", M)
    reps = 10
    # PCS::prove's L^T W: 512 x 2048
    rows = M // 2048
    t = hip.Table.from_host(ctx, rand_fe(M))
    L = rand_fe(rows)
    for _ in range(reps):
        hip.rowmat_vec(ctx, t, rows, 2048, L)
        ctx.synchronize()
    # bind_and_prepare_poly_ABC (full size, as the prover's)
    rx = hip.Table.eq(ctx, rand_fe(N.bit_length() - 1))
    r = rand_fe(1)[0]
    out = hip.Table.zeros(ctx, 2 * M)
    for _ in range(reps):
        shape.poly_abc(rx, r, 2 * M, out)
        ctx.synchronize()
    # outer sum-check on three 2^20 tables, inner on two 2^21 tables (dense, then with the prover's effective ranges)
    ell = N.bit_length() - 1
    for _ in range(5):
        A, B, C = (hip.Table.eq(ctx, rand_fe(ell)) for _ in range(3))
        ctx.synchronize()
        hip.sumcheck_cubic3(ctx, np.zeros(4, dtype=np.uint64), rand_fe(ell), A, B, C, hip.Transcript(ctx, b"solo"))
        ctx.synchronize()
        for x in (A, B, C):
            x.free()
    ly = (2 * M).bit_length() - 1
    num_extra = 1 + len(inst.publics) if hasattr(inst, "publics") else 2
    for eff in (False, True):
        for _ in range(5):
            A, B = (hip.Table.eq(ctx, rand_fe(ly)) for _ in range(2))
            if eff:
                A.set_len(2 * M, M, num_extra)
                B.set_len(2 * M, M, num_extra)
            ctx.synchronize()
            hip.sumcheck_quad(ctx, np.zeros(4, dtype=np.uint64), ly, A, B, hip.Transcript(ctx, b"solo"))
            ctx.synchronize()
            A.free()
            B.free()
    print("solo done")
ctx.close()
