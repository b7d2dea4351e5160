"""Spawn helpers for the multi-process tests (gloo ranks): collect one result per rank, but fail as soon as a rank dies instead of waiting
for a queue timeout."""
import queue
import socket
import time

import torch.multiprocessing as mp


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(target, world, extra_args=(), timeout=600):
    """target(rank, world, port, q, *extra_args) must q.put((rank, result)). Returns {rank: result}."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(extra_args)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    deadline = time.time() + timeout
    try:
        while len(res) < world:
            try:
                r, out = q.get(timeout=1.0)
                res[r] = out
            except queue.Empty:
                dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
                if dead:
                    raise AssertionError(f"rank(s) died before reporting: {dead}")
                if time.time() > deadline:
                    raise AssertionError(f"timed out after {timeout} s with results from ranks {sorted(res)}")
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0, p.exitcode
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    return res
