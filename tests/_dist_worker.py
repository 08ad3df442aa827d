"""Worker for test_host.py::test_two_rank_gloo: exercises the N > 1 host logic of the batched mode
(rank sharding, barrier-bracketed timing, MAX / SUM all-reduces) on CPU with the gloo backend."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd"


def main():
    batch = importlib.import_module(PKG + ".batch")
    comm = batch.Comm(backend="gloo")
    total_images = int(sys.argv[1])
    counts, displs = batch.calculate_distribution(total_images, comm.world)
    mine = list(range(displs[comm.rank], displs[comm.rank] + counts[comm.rank]))
    done = []

    def step():
        time.sleep(0.002 * (comm.rank + 1))  # rank 1 is slower: MAX must pick it up
        done.extend(mine)

    elapsed = batch.timed_steps(comm, step, lambda: None, steps=3, warmup=1)
    tot = comm.allreduce_sum([len(mine), sum(mine)])
    # the data-carrying collective of bench.py --bcast-filter: rank 0's block to every rank, then a hash all-reduce (MIN == MAX
    # == rank 0's) proves every rank holds rank 0's bits
    import torch
    blk = torch.arange(4096, dtype=torch.int32) * (7 if comm.rank == 0 else 0) + (comm.rank * 1000)
    comm.broadcast(blk, src=0)
    h = int(blk.to(torch.int64).sum().item()) & ((1 << 52) - 1)
    hmin, hmax = int(comm.allreduce_min(h)), int(comm.allreduce_max(h))
    if comm.rank == 0:
        print(json.dumps({"world": comm.world, "elapsed": elapsed, "images": tot[0], "index_sum": tot[1],
                          "steps_seen": len(done) // max(len(mine), 1),
                          "bcast_ok": hmin == hmax == h == int((torch.arange(4096, dtype=torch.int64) * 7).sum().item())}))
    comm.close()


if __name__ == "__main__":
    main()
