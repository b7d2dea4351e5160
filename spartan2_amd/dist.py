"""Multi-GPU plumbing for the Python harness: one process per GPU under torch.distributed.run. torch.distributed carries the barrier / max-over-ranks of
the timed region ("nccl" = RCCL on the GPUs, "gloo" in the CPU tests) and hands the C++ exchange layer its RCCL unique id once.

The DATA-PATH exchanges of the sharded prover (per-round sums of the slice-sharded sum-checks, commitment rows, partial MSM points) do not go
through this module: they are ncclAllGather calls made from C++ (spartan2_amd/host/comm.hpp, driven by sharded_snark.cpp). The functions below that
exchange through torch.distributed + numpy (`*_sharded`) are the round-1 prototypes, kept as the gloo-testable reference of the same partitioning
(tests/test_dist_cpu.py, tests/test_gpu_{nifs,sumcheck,commit}_sharded.py)."""
import os

import torch


# PyTorch-on-ROCm keeps its upstream spellings: the RCCL backend is registered as "nccl" and HIP devices are device type "cuda". Nothing CUDA is involved.
RCCL_BACKEND = "nccl"
HIP_DEVICE_TYPE = "cuda"


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous, balanced shard [lo, hi) of n_units independent units (proofs / step instances) for this rank."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class Group:
    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist

            backend = backend or (RCCL_BACKEND if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == RCCL_BACKEND:
                kw["device_id"] = torch.device("cuda", self.local_rank)
            dist.init_process_group(backend=backend, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        dev = HIP_DEVICE_TYPE if (torch.cuda.is_available() and self.dist.get_backend() == RCCL_BACKEND) else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        dev = HIP_DEVICE_TYPE if (torch.cuda.is_available() and self.dist.get_backend() == RCCL_BACKEND) else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def whole_job_throughput(units_per_rank_step: float, steps: int, elapsed_max: float, world: int) -> float:
    """value of bench.py: units all ranks processed / max-over-ranks time."""
    return world * units_per_rank_step * steps / elapsed_max


# ---- sharded group work ---------------------------------------------------------------------------------------
import numpy as np


def _all_gather_rows(group: Group, local: np.ndarray, counts):
    """all-gather of variable-length (rows, 8) uint64 blocks; `counts[r]` rows come from rank r."""
    if group.dist is None:
        return local
    width = local.shape[1]
    mx = max(counts) if counts else 0
    dev = HIP_DEVICE_TYPE if (torch.cuda.is_available() and group.dist.get_backend() == RCCL_BACKEND) else "cpu"
    buf = torch.zeros((mx, width), dtype=torch.int64, device=dev)
    if local.shape[0]:
        buf[: local.shape[0]] = torch.from_numpy(local.view(np.int64).copy()).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(group.world)]
    group.dist.all_gather(outs, buf)
    parts = [o[:c].cpu().numpy().view(np.uint64) for o, c in zip(outs, counts)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, width), dtype=np.uint64)


def commit_rows_sharded(group: Group, n_rows: int, commit_rows_fn):
    """Hyrax commitment of n_rows rows sharded BY ROW (independent units, bases replicated): rank r commits rows
    [lo_r, hi_r) with commit_rows_fn(lo, hi) -> (hi-lo, 8) affine rows; one all-gather of 64-byte rows assembles the commitment
    on every rank. No reduction is needed (PCS::commit's rows are independent, hyrax_pc.rs:230-300)."""
    lo, hi = shard_range(n_rows, group.rank, group.world)
    local = np.ascontiguousarray(commit_rows_fn(lo, hi), dtype=np.uint64).reshape(hi - lo, 8)
    counts = [shard_range(n_rows, r, group.world)[1] - shard_range(n_rows, r, group.world)[0] for r in range(group.world)]
    return _all_gather_rows(group, local, counts)


def msm_point_range_sharded(group: Group, n_points: int, msm_fn, point_sum_fn):
    """One large MSM sharded BY POINT RANGE: rank r computes the partial sum over points [lo_r, hi_r) with msm_fn(lo, hi) -> (8,)
    affine; partials are all-gathered (RCCL has no EC-add op) and added locally with point_sum_fn((world, 8)) -> (8,)."""
    lo, hi = shard_range(n_points, group.rank, group.world)
    part = np.ascontiguousarray(msm_fn(lo, hi), dtype=np.uint64).reshape(1, 8)
    allp = _all_gather_rows(group, part, [1] * group.world)
    return point_sum_fn(allp)


def _field_sum(parts: np.ndarray, add_fn):
    """sum of (world, k, 4) field elements over axis 0 with the caller's field addition (RCCL has no modular-add reduction: ranks all-gather
    their few elements and add locally, in rank order — exact arithmetic, so every rank gets identical totals)."""
    acc = parts[0]
    for r in range(1, parts.shape[0]):
        acc = add_fn(acc, parts[r])
    return acc


def nifs_rounds_sharded(group: Group, nifs, make_nifs, E_eq, rhos, n_local: int, small_values: bool, hook, add_fn, read_layer, write_layer):
    """NeutronNovaNIFS::prove rounds (src/neutronnova_zk.rs:779-1206) with the 2^ell_b instances sharded over the ranks, n_local (a power of
    two >= 2) consecutive instances per rank — SURVEY.md 8(e), one process per GPU.

    Per round of the first log2(n_local) rounds every rank evaluates its own instance pairs (`nifs.round_sums`, the data-parallel part) and
    the ranks exchange 2 field elements (all-gather + local adds); every rank then runs the O(1) finish and the deterministic `hook`
    (the caller's process_round) redundantly, so no broadcast is needed. After those rounds each rank is left with ONE folded layer per
    matrix; they are gathered on rank 0 (the only bulk exchange: 2 layers per rank, once), which runs the remaining log2(world) rounds on
    a fresh object from `make_nifs(world)`. Returns on rank 0 (nifs_root, r_bs, polys); on other ranks (None, r_bs_local, polys_local)
    where the lists stop at the hand-off. `nifs` must have its A/B/C layers filled. add_fn(a, b): field addition on (k, 4) limb arrays;
    read_layer(table) -> (total, 4) array and write_layer(table, array) move a layer through the host (gloo) or could stay on device (RCCL)."""
    world, rank = group.world, group.rank
    ell_b = rhos.shape[0]
    assert n_local >= 2 and n_local & (n_local - 1) == 0 and n_local * world == 1 << ell_b
    nifs.begin_shard(E_eq, rhos, rank * n_local, small_values)
    cv = _all_gather_rows(group, nifs.cvals(), [n_local] * world)
    nifs.set_cvals(cv)
    local_rounds = n_local.bit_length() - 1
    polys, r_bs = [], []
    for t in range(local_rounds):
        part = nifs.round_sums(t).reshape(1, 8)
        allp = _all_gather_rows(group, part, [1] * world).reshape(world, 2, 4)
        sums = _field_sum(allp, add_fn)
        co = nifs.round_finish(t, sums)
        polys.append(co)
        r = np.ascontiguousarray(hook(t, co), dtype=np.uint64)
        r_bs.append(r)
        nifs.challenge(r)
    if world == 1:
        return nifs, r_bs, polys
    # hand-off: apply the pending fold, ship the single remaining A / B layer of every rank to rank 0
    nifs.fold_pending()
    layers = []
    for which in (0, 1):
        v = nifs.current_layer(which, 0)
        layers.append(read_layer(v))
        v.free()
    total = layers[0].shape[0]
    mine = np.concatenate(layers).reshape(2 * total, 4)
    gathered = _all_gather_rows(group, mine, [2 * total] * world).reshape(world, 2, total, 4)
    T_cur, acc_eq = nifs.state()
    if rank != 0:
        return None, r_bs, polys
    root = make_nifs(world)
    for b in range(world):
        for which in (0, 1):
            v = root.layer(which, b)
            write_layer(v, gathered[b, which])
            v.free()
    root.resume(E_eq, rhos, local_rounds, np.stack(r_bs), T_cur, acc_eq, cv)
    for t in range(local_rounds, ell_b):
        co = root.round(t)
        polys.append(co)
        r = np.ascontiguousarray(hook(t, co), dtype=np.uint64)
        r_bs.append(r)
        root.challenge(r)
    return root, r_bs, polys


# ---- sum-check by evaluation-table slice ----------------------------------------------------------------------------
_P = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF  # scalar field of the bench engine (src/provider/pt256.rs:55)
_R = 1 << 256


def _to_int(limbs):  # Montgomery limbs -> integer value
    v = sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(4)))
    return v * pow(_R, -1, _P) % _P


def _to_limbs(v):
    m = v % _P * _R % _P
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def slice_of(table: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Rank `rank`'s slice of a table sharded on its last log2(world) variables: Z_g[j] = Z[(j << k) | g]."""
    return np.ascontiguousarray(table[rank::world])


def sumcheck_cubic3_sharded(group: Group, cubic_fn, claim, taus, A, B, C, make_table):
    """SumcheckProof::prove_cubic_with_three_inputs (src/sumcheck.rs:502-571) with the three tables sharded by slice over the ranks.

    cubic_fn(claim, p, taus, A, B, C, scale, reduce) -> (polys, r, final (3,4), claim_out, p_out) is sp_sumcheck_cubic3_sharded bound to the
    rank's context and transcript (every rank runs the same deterministic transcript). Per round the ranks exchange 2-3 field elements
    (all-gather + local adds in rank order). After the ell - k local rounds each rank holds one value per table; they are gathered into
    2^k-element tables on EVERY rank (`make_table(array)`), which all finish the last k rounds redundantly — no further exchange.
    Returns (polys, r, final) of the whole ell-round sum-check."""
    world, rank = group.world, group.rank
    k = world.bit_length() - 1
    assert 1 << k == world
    taus = np.ascontiguousarray(taus, dtype=np.uint64).reshape(-1, 4)
    ell = taus.shape[0]
    one = _to_limbs(1)
    if world == 1:
        polys, r, fin, _, _ = cubic_fn(claim, one, taus, A, B, C, None, None)
        return polys, r, fin
    # scale = eq(taus[ell-k..ell), bits of rank), MSB of the k-bit rank index = first of those variables
    sc = 1
    for i in range(k):
        t = _to_int(taus[ell - k + i])
        sc = sc * (t if (rank >> (k - 1 - i)) & 1 else (1 - t)) % _P

    def reduce(sums):
        allp = _all_gather_rows(group, np.ascontiguousarray(sums, dtype=np.uint64).reshape(1, -1), [1] * world).reshape(world, -1, 4)
        return np.stack([_to_limbs(sum(_to_int(allp[g, i]) for g in range(world))) for i in range(allp.shape[1])])

    polys1, r1, fin_loc, claim1, p1 = cubic_fn(claim, one, taus[: ell - k], A, B, C, _to_limbs(sc), reduce)
    gathered = _all_gather_rows(group, fin_loc.reshape(1, 12), [1] * world).reshape(world, 3, 4)
    TA, TB, TC = (make_table(np.ascontiguousarray(gathered[:, q])) for q in range(3))
    polys2, r2, fin, _, _ = cubic_fn(claim1, p1, taus[ell - k :], TA, TB, TC, None, None)
    return np.concatenate([polys1, polys2]), np.concatenate([r1, r2]), fin


def sumcheck_quad_sharded(group: Group, quad_fn, claim, rounds, A, B, make_table):
    """SumcheckProof::prove_quad (src/sumcheck.rs:190-247) on tables sharded by slice; quad_fn(claim, rounds, A, B, reduce) ->
    (polys, r, final (2,4), claim_out) = sp_sumcheck_quad_sharded."""
    world = group.world
    k = world.bit_length() - 1
    assert 1 << k == world
    if world == 1:
        polys, r, fin, _ = quad_fn(claim, rounds, A, B, None)
        return polys, r, fin

    def reduce(sums):
        allp = _all_gather_rows(group, np.ascontiguousarray(sums, dtype=np.uint64).reshape(1, -1), [1] * world).reshape(world, -1, 4)
        return np.stack([_to_limbs(sum(_to_int(allp[g, i]) for g in range(world))) for i in range(allp.shape[1])])

    polys1, r1, fin_loc, claim1 = quad_fn(claim, rounds - k, A, B, reduce)
    gathered = _all_gather_rows(group, fin_loc.reshape(1, 8), [1] * world).reshape(world, 2, 4)
    TA, TB = (make_table(np.ascontiguousarray(gathered[:, q])) for q in range(2))
    polys2, r2, fin, _ = quad_fn(claim1, k, TA, TB, None)
    return np.concatenate([polys1, polys2]), np.concatenate([r1, r2]), fin


def cpu_budget() -> int:
    """CPUs this process may actually burn: the affinity mask capped by the cgroup's CFS quota (cpu.max). Every context in flight keeps one owner
    thread polling; past the quota the whole cgroup is throttled and a throttled owner cannot answer its resident kernel in time."""
    import math

    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()
            if quota != "max":
                n = min(n, max(1, math.floor(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // p_))
    except (OSError, ValueError):
        pass
    return n
