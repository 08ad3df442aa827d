"""Worker for the GPU test test_rccl_single_rank_smoke: the collectives bench.py issues under --backend nccl (= RCCL) --
barrier, MAX / MIN / SUM all-reduce of float64 device tensors, broadcast of an int32 device block -- on a one-rank
process group, so that the tensors, dtypes and devices the batched mode hands to RCCL are exercised on the one GPU a test
box has (two ranks on one GPU are refused by RCCL; the multi-rank logic itself runs under gloo in tests/test_host.py)."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", FDR_DIST_SINGLE="1", FDR_DIST_TIMEOUT_S="120")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    batch = importlib.import_module(PKG + ".batch")
    # Comm's OWN set-up, as bench.py --gpus N runs it on every rank: a gloo default group, an RCCL group beside it proven
    # with one all-reduce on the device, the gloo consensus, then every collective of the run on the RCCL group -- through
    # the guard that turns a failure into a message and a non-zero exit
    comm = batch.Comm(backend="nccl", device=dev)
    assert comm.dist is not None and dist.is_initialized() and dist.get_backend() == "gloo"
    assert comm.backend == "nccl" and comm.group is not None and comm.collectives() == "rccl", (comm.backend, comm.fallback_reason)
    comm.barrier()
    mx, mn = comm.allreduce_max(1.25), comm.allreduce_min(-3.5)
    sm = comm.allreduce_sum([1, 2.5, 4])
    blk = torch.arange(1 << 16, dtype=torch.int32, device=dev)
    comm.broadcast(blk, src=0)
    torch.cuda.synchronize()
    calls = []
    dt = batch.timed_steps(comm, lambda: calls.append(1), torch.cuda.synchronize, steps=3, warmup=1)
    ok = (mx == 1.25 and mn == -3.5 and sm == [1.0, 2.5, 4.0] and int(blk.to(torch.int64).sum().item()) == (1 << 16) * ((1 << 16) - 1) // 2
          and len(calls) == 4 and dt >= 0.0)
    print(json.dumps({"ok": bool(ok), "backend": comm.backend, "collectives": comm.collectives()}))
    comm.close()


if __name__ == "__main__":
    main()
