"""Batched-image mode across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI, or "gloo" in CPU tests), independent images sharded by rank.

Images are independent units: there is no data-path collective.  The only
collectives are (i) the barrier bracketing a timed region, (ii) a MAX all-reduce of the elapsed
time, (iii) a SUM all-reduce of {images done, pixels done, checksum} as the end-of-batch
consistency check and, optionally (bench.py --bcast-filter), (iv) a broadcast of rank 0's prepared
filter W in place of every rank recomputing it (setup, outside the timed region).  Partitioning follows the reference's calculate_distribution
(fft/fft_mpi.cpp:89-100) applied to images instead of rows.
"""
import time


def calculate_distribution(total, parts):
    """counts[g] = total // parts + (g < total % parts); displs = prefix sums (fft/fft_mpi.cpp:89-100)."""
    base, rem = divmod(int(total), int(parts))
    counts = [base + (1 if g < rem else 0) for g in range(parts)]
    displs = [0] * parts
    for g in range(1, parts):
        displs[g] = displs[g - 1] + counts[g - 1]
    return counts, displs


class Comm:
    """Minimal wrapper so the same code runs single-process, under gloo (CPU tests) and under RCCL."""

    def __init__(self, backend=None, device=None):
        import os
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29511")
                dist.init_process_group(backend=backend or "nccl", rank=self.rank, world_size=self.world)
            self.dist = dist

    def _tensor(self, vals, dtype):
        import torch
        dev = self.device if (self.dist is not None and self.dist.get_backend() == "nccl") else "cpu"
        return torch.tensor(vals, dtype=dtype, device=dev)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def allreduce_max(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = self._tensor([float(x)], torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allreduce_min(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = self._tensor([float(x)], torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def broadcast(self, tensor, src=0):
        """dist.broadcast of a tensor in place (RCCL over xGMI under "nccl"); the data-carrying collective of the batched mode
        (rank 0's prepared filter W to every rank: the MPI_Bcast / Scatterv of fft/fft_mpi.cpp:334-378).  Single process: no-op."""
        if self.dist is not None:
            self.dist.broadcast(tensor, src=src)
        return tensor

    def gather_objects(self, obj):
        """list of every rank's picklable object on rank 0 (None elsewhere); plumbing for tests and result collection"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def allreduce_sum(self, vals):
        if self.dist is None:
            return [float(v) for v in vals]
        import torch
        t = self._tensor([float(v) for v in vals], torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def timed_steps(comm, step_fn, sync_fn, steps, warmup):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device sync on
    both sides; returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step_fn()
    sync_fn()
    comm.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    comm.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    return comm.allreduce_max(dt)
