"""Batched-image mode across the GPUs of one node: one process per GPU (torch.distributed,
backend "nccl" = RCCL over xGMI, or "gloo" in CPU tests), independent images sharded by rank.

Images are independent units: there is no data-path collective.  The only
collectives are (i) the barrier bracketing a timed region, (ii) a MAX all-reduce of the elapsed
time, (iii) a SUM all-reduce of {images done, pixels done, checksum} as the end-of-batch
consistency check and, optionally (bench.py --bcast-filter), (iv) a broadcast of rank 0's prepared
filter W in place of every rank recomputing it (setup, outside the timed region).  Partitioning follows the reference's calculate_distribution
(fft/fft_mpi.cpp:89-100) applied to images instead of rows.
"""
import os
import sys
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
    """Minimal wrapper so the same code runs single-process, under gloo (CPU tests) and under RCCL.

    Under "nccl" the process group is bound to this rank's device at creation (`device_id`: the communicator is built
    eagerly on that device, so a barrier can never pick another one) and every collective has a deadline
    (FDR_DIST_TIMEOUT_S, default 300 s).  A collective that raises ends the process with a one-line message and a
    non-zero exit code -- the launcher then stops the other ranks -- instead of leaving them waiting."""

    def __init__(self, backend=None, device=None, timeout_s=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = device
        self.backend = backend or "nccl"
        # FDR_DIST_SINGLE=1: build the process group even for ONE rank -- the GPU test box has one GPU, and this is how the
        # "nccl" set-up below (device-bound communicator, deadline, barrier on the bound device) meets real RCCL there
        if self.world > 1 or os.environ.get("FDR_DIST_SINGLE") == "1":
            import datetime
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL on this pool: dmabuf IPC only
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29511")
                if timeout_s is None:
                    timeout_s = float(os.environ.get("FDR_DIST_TIMEOUT_S", "300"))
                kw = {"timeout": datetime.timedelta(seconds=timeout_s)}
                if self.backend == "nccl" and device is not None:
                    kw["device_id"] = device
                self._guard("init_process_group", lambda: dist.init_process_group(
                    backend=self.backend, rank=self.rank, world_size=self.world, **kw))
            else:
                self.backend = dist.get_backend()
            self.dist = dist

    def _guard(self, what, fn):
        try:
            return fn()
        except BaseException as e:  # noqa: BLE001 -- any failure of a collective ends this rank, loudly
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                raise
            sys.stderr.write("fdr.batch: rank %d/%d: %s failed (backend %s): %s: %s\n"
                             % (self.rank, self.world, what, self.backend, type(e).__name__, e))
            sys.stderr.flush()
            os._exit(13)  # no destructors: tearing a broken process group down can itself wait for the peers

    def _tensor(self, vals, dtype):
        import torch
        dev = self.device if (self.dist is not None and self.backend == "nccl") else "cpu"
        return torch.tensor(vals, dtype=dtype, device=dev)

    def barrier(self):
        if self.dist is None:
            return
        if self.backend == "nccl" and self.device is not None and getattr(self.device, "index", None) is not None:
            self._guard("barrier", lambda: self.dist.barrier(device_ids=[self.device.index]))
        else:
            self._guard("barrier", self.dist.barrier)

    def _allreduce(self, vals, op, what):
        import torch
        t = self._tensor([float(v) for v in vals], torch.float64)
        self._guard(what, lambda: self.dist.all_reduce(t, op=op))
        return [float(v) for v in t.tolist()]

    def allreduce_max(self, x):
        if self.dist is None:
            return float(x)
        return self._allreduce([x], self.dist.ReduceOp.MAX, "all_reduce(MAX)")[0]

    def allreduce_min(self, x):
        if self.dist is None:
            return float(x)
        return self._allreduce([x], self.dist.ReduceOp.MIN, "all_reduce(MIN)")[0]

    def broadcast(self, tensor, src=0):
        """dist.broadcast of a tensor in place (RCCL over xGMI under "nccl"); the data-carrying collective of the batched mode
        (rank 0's prepared filter W to every rank: the MPI_Bcast / Scatterv of fft/fft_mpi.cpp:334-378).  Single process: no-op."""
        if self.dist is not None:
            self._guard("broadcast", lambda: self.dist.broadcast(tensor, src=src))
        return tensor

    def gather_objects(self, obj):
        """list of every rank's picklable object on rank 0 (None elsewhere); plumbing for tests and result collection"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self._guard("gather_object", lambda: self.dist.gather_object(obj, out, dst=0))
        return out

    def allreduce_sum(self, vals):
        if self.dist is None:
            return [float(v) for v in vals]
        return self._allreduce(vals, self.dist.ReduceOp.SUM, "all_reduce(SUM)")

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self._guard("destroy_process_group", self.dist.destroy_process_group)


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
