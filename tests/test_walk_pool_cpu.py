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
