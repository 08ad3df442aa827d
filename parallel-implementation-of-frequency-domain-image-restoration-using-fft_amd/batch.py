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

    The batched mode has no data-path collective -- only barriers, scalar all-reduces and the optional set-up broadcast of
    the filter -- so the rendezvous itself must never be what loses a multi-GPU run:
      * the DEFAULT process group is always gloo (TCP on localhost: nothing GPU-specific can break it);
      * under backend "nccl" an RCCL group over the same ranks is created beside it and proven with one all-reduce on this
        rank's device; a MIN all-reduce over gloo then decides for ALL ranks together: RCCL if every rank's test succeeded
        (the normal case: the collectives of the timed region then run over RCCL / xGMI), else gloo for everyone, with the
        reason kept in `fallback_reason` (bench.py prints it in `config.collectives`);
      * every collective has a deadline (FDR_DIST_TIMEOUT_S, default 300 s; the RCCL proof: FDR_RCCL_PROBE_TIMEOUT_S, default
        120 s; TORCH_NCCL_BLOCKING_WAIT makes a timed-out RCCL call raise instead of aborting the process), and a collective
        that raises ends the process with a one-line message and exit code 13 -- the launcher then stops the other ranks --
        instead of leaving them waiting."""

    def __init__(self, backend=None, device=None, timeout_s=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.group = None          # None = the default (gloo) group; else the RCCL group
        self.device = device
        self.backend = backend or "nccl"
        self.fallback_reason = None
        # FDR_DIST_SINGLE=1: build the process group even for ONE rank -- the GPU test box has one GPU, and this is how the
        # "nccl" set-up below (RCCL group beside gloo, proof all-reduce, consensus) meets real RCCL there
        if self.world > 1 or os.environ.get("FDR_DIST_SINGLE") == "1":
            import datetime
            import torch.distributed as dist
            self.dist = dist
            want = self.backend
            if timeout_s is None:
                timeout_s = float(os.environ.get("FDR_DIST_TIMEOUT_S", "300"))
            if not dist.is_initialized():
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL on this pool: dmabuf IPC only
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29511")
                os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
                if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
                    # one node: gloo on the loopback interface -- its default is to resolve the HOST NAME, which containers
                    # do not always do
                    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
                self.backend = "gloo"
                td = datetime.timedelta(seconds=timeout_s)
                if want == "nccl":
                    try:
                        dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world, timeout=td)
                    except BaseException as e:  # noqa: BLE001 -- no gloo here: RCCL alone, as before round 4
                        if isinstance(e, (KeyboardInterrupt, SystemExit)):
                            raise
                        sys.stderr.write("fdr.batch: rank %d/%d: gloo group unavailable (%s: %s); RCCL alone\n"
                                         % (self.rank, self.world, type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
                        self.backend = "nccl"
                        kw = {"device_id": device} if device is not None else {}
                        self._guard("init_process_group", lambda: dist.init_process_group(
                            backend="nccl", rank=self.rank, world_size=self.world, timeout=td, **kw))
                        return
                    self._try_rccl(datetime.timedelta(seconds=float(os.environ.get("FDR_RCCL_PROBE_TIMEOUT_S", "120"))))
                else:
                    self._guard("init_process_group", lambda: dist.init_process_group(
                        backend="gloo", rank=self.rank, world_size=self.world, timeout=td))
            else:  # a group the caller made: use it as it is
                self.backend = dist.get_backend()

    def _try_rccl(self, timeout):
        """RCCL group beside the gloo one; used only if EVERY rank could build it and all-reduce on it."""
        import torch
        dist = self.dist
        ok, why, pg = 1.0, "", None
        try:
            if self.device is None:
                raise RuntimeError("no device given for the RCCL group")
            pg = dist.new_group(backend="nccl", timeout=timeout)
            t = torch.ones(1, dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
            torch.cuda.synchronize(self.device)
            if int(t.item()) != self.world:
                raise RuntimeError("proof all-reduce returned %r for %d ranks" % (t.item(), self.world))
        except BaseException as e:  # noqa: BLE001
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                raise
            ok, why = 0.0, "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
        flag = torch.tensor([ok], dtype=torch.float64)
        self._guard("all_reduce(MIN) of the RCCL proof", lambda: dist.all_reduce(flag, op=dist.ReduceOp.MIN))
        if flag.item() >= 1.0:
            self.group, self.backend = pg, "nccl"
        else:
            self.fallback_reason = why or "another rank could not build or use the RCCL group"
            sys.stderr.write("fdr.batch: rank %d/%d: RCCL group unavailable (%s); the barriers and scalar all-reduces of this run go "
                             "over gloo (there is no data-path collective)\n" % (self.rank, self.world, self.fallback_reason))
            sys.stderr.flush()

    def collectives(self):
        """What carries the collectives: for bench.py's `config.collectives`."""
        if self.dist is None:
            return "none (single process)"
        if self.backend == "nccl":
            return "rccl"
        return "gloo" if not self.fallback_reason else "gloo (rccl unavailable: %s)" % self.fallback_reason

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
            self._guard("barrier", lambda: self.dist.barrier(group=self.group, device_ids=[self.device.index]))
        else:
            self._guard("barrier", lambda: self.dist.barrier(group=self.group))

    def _allreduce(self, vals, op, what):
        import torch
        t = self._tensor([float(v) for v in vals], torch.float64)
        self._guard(what, lambda: self.dist.all_reduce(t, op=op, group=self.group))
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
        (rank 0's prepared filter W to every rank: the MPI_Bcast / Scatterv of fft/fft_mpi.cpp:334-378).  Single process: no-op.
        Over gloo (CPU tests, rehearsals, the fallback) a device tensor is staged through the host."""
        if self.dist is None:
            return tensor
        if self.backend != "nccl" and getattr(tensor, "is_cuda", False):
            h = tensor.detach().cpu()
            self._guard("broadcast", lambda: self.dist.broadcast(h, src=src, group=self.group))
            tensor.copy_(h)
            return tensor
        self._guard("broadcast", lambda: self.dist.broadcast(tensor, src=src, group=self.group))
        return tensor

    def gather_objects(self, obj):
        """list of every rank's picklable object on rank 0 (None elsewhere); plumbing for tests and result collection"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self._guard("gather_object", lambda: self.dist.gather_object(obj, out, dst=0))  # (objects: always over gloo)
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
