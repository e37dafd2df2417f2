"""Multi-GPU scaffolding of bench.py: one process per GPU (torch.distributed, backend nccl == RCCL on ROCm; gloo on CPU
for the tests).  By default reads are the sharded unit — every rank maps its own batch against the replicated filter +
index — so the data path has no collective; only the timing protocol (barrier, max over ranks) and the result roll-up
use one.  The bin-sharded mode with its two all-to-alls lives in sharded.py."""
import os


class Group:
    def __init__(self, backend=None, force=False):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = "cpu"
        if self.world > 1 or force:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = "cuda"
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def shard_seed(config_id, rank):
    """Seed of the reads simulated by `rank` (0x5EED0000 + config; ranks get disjoint streams)."""
    return 0x5EED0000 + config_id + 1000 * rank


def whole_job_rate(reads_per_rank, steps, world, elapsed_max):
    """BASELINE metric: reads of ALL ranks divided by the slowest rank's time."""
    return reads_per_rank * world * steps / elapsed_max
