#!/usr/bin/env python3
"""Per-call cost of the exchange layer's all-gather on a one-rank RCCL communicator (staging copies + ncclAllGather + synchronise): a lower bound of
what one sum-check round of a sharded prove pays per exchange at N > 1."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # in this image the communicator initialises only beside PyTorch's bundled RCCL / HIP runtime (the system librccl fails in rocmwrap)
from spartan2_amd import hip, host
ctx = hip.Context(0)
comm = host.Comm(0, 1, "rccl", device=0)
L = host.lib()
send = np.arange(12, dtype=np.uint64)
recv = np.zeros(12, dtype=np.uint64)
for n in (96, 4096, 65536):
    s = np.zeros(n // 8, dtype=np.uint64); r = np.zeros(n // 8, dtype=np.uint64)
    for _ in range(20):
        L.ssc_comm_allgather(comm.h, s.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), r.ctypes.data_as(ctypes.c_void_p))
    t0 = time.perf_counter()
    for _ in range(500):
        L.ssc_comm_allgather(comm.h, s.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), r.ctypes.data_as(ctypes.c_void_p))
    print(n, "bytes:", (time.perf_counter() - t0) / 500 * 1e6, "us per all-gather (one-rank RCCL communicator)", comm.stats())
comm.close(); ctx.close()
