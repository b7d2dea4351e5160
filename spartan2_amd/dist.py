"""Multi-GPU plumbing: one process per GPU under torch.distributed.run; the path shards by independent units (whole proofs
at config 2, step instances at the NeutronNova configs) with no data-path collective. torch.distributed is used only for the
barrier and the max-over-ranks of the timed region ("nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""
import os

import torch


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

            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl":
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
        dev = "cuda" if (torch.cuda.is_available() and self.dist.get_backend() == "nccl") else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return float(x)
        dev = "cuda" if (torch.cuda.is_available() and self.dist.get_backend() == "nccl") else "cpu"
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
