"""The process's polling host threads (spartan2_amd/csrc/walk_pool.hpp) without a GPU: every part of a parallel region runs exactly once with the walkers
polling, asleep or absent, under several owner threads at once, and a table walk adds up its entries (tests/native/pool_check.hip, host code only);
the same region through the C ABI (sp_host_parallel_for) with a Python callback."""
import ctypes
import os
import shutil
import subprocess

import pytest

from spartan2_amd import hip

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pool_check(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    exe = str(tmp_path_factory.mktemp("pool") / "pool_check")
    subprocess.run(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-pthread", "-w", "-o", exe, os.path.join(HERE, "native", "pool_check.hip")], check=True, capture_output=True, timeout=900)
    return exe


@pytest.mark.parametrize("walkers,hot", [("3", "1"), ("3", "0"), ("0", "0"), ("31", "1")])
def test_parallel_regions_and_walks(pool_check, walkers, hot):
    out = subprocess.run([pool_check, hot], capture_output=True, text=True, timeout=900, env=dict(os.environ, SPARTAN_WALKERS=walkers))
    assert out.returncode == 0 and ": 0 mismatches" in out.stdout, out.stdout + out.stderr


def test_parallel_for_through_the_abi():
    L = hip.lib()
    FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint)
    hits = []

    def body(arg, part, nparts):
        hits.append((part, nparts))

    cb = FN(body)
    L.sp_walkers_keep_hot(ctypes.c_uint64(2000))
    for nparts in (1, 2, 9, 32):
        hits.clear()
        assert L.sp_host_parallel_for(ctypes.c_uint(nparts), cb, None) == 0
        assert sorted(hits) == [(p, nparts) for p in range(nparts)]
    assert L.sp_host_parallel_for(ctypes.c_uint(4), FN(), None) != 0  # a null function is refused
    assert L.sp_walkers() >= 0


def test_two_term_fold_over_ladders_without_a_gpu():
    """sp_fold_commitments2_begin / _finish is host work when the process has walkers (the doubling ladders and the non-adjacent-form walk run on the polling
    threads): p + w q against the oracle's point arithmetic, no context, no device."""
    import numpy as np

    import oracle_lib as ol

    L = hip.lib()
    if L.sp_walkers() == 0:
        pytest.skip("no walkers in this process: the call falls back to the device form")
    rng = np.random.default_rng(77)

    def points(n, label):
        g = np.zeros((n, 8), dtype=np.uint64)
        ol.lib().orc_from_label(label, ctypes.c_size_t(n), ol.p64(g))
        return g

    def mul(pt, k):
        o = np.zeros(8, dtype=np.uint64)
        ol.lib().orc_point_mul(ol.p64(np.ascontiguousarray(pt)), ol.p64(np.ascontiguousarray(k)), ol.p64(o))
        return o

    def add(a, b):
        o = np.zeros(8, dtype=np.uint64)
        ol.lib().orc_point_add(ol.p64(np.ascontiguousarray(a)), ol.p64(np.ascontiguousarray(b)), ol.p64(o))
        return o

    for n in (1, 5, 17):
        q, p = points(n, b"fold2_q"), points(n, b"fold2_p")
        if n > 2:
            q[1] = 0
            p[2] = 0
        for w in (ol.random_field_array(rng, 1)[0], ol.to_mont(0), ol.to_mont(1), ol.to_mont((1 << 255) - (1 << 13) + 5)):
            job = ctypes.c_void_p()
            assert L.sp_fold_commitments2_begin(None, hip.p64(q), ctypes.c_size_t(n), ctypes.byref(job)) == 0
            out = np.zeros_like(p)
            assert L.sp_fold_commitments2_finish(None, job, hip.p64(p), hip.p64(np.ascontiguousarray(w)), hip.p64(out)) == 0
            assert (out == np.stack([add(pp, mul(qq, w)) for pp, qq in zip(p, q)])).all()
    job = ctypes.c_void_p()
    assert L.sp_fold_commitments2_begin(None, hip.p64(points(3, b"fold2_q")), ctypes.c_size_t(3), ctypes.byref(job)) == 0
    L.sp_fold_commitments2_drop(job)
