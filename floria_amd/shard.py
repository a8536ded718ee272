"""Multi-GPU sharding of the per-block path: one process per GPU, work queue broadcast once.

Blocks of different contigs share nothing and blocks of one contig share only the read-only pileup
(graph_processing.rs:345-362; floria.rs:229), so the path shards with NO data-path collective: rank 0 builds
the queue descriptor (contig -> rank, longest-processing-time first on an estimated cost), broadcasts it
(`torch.distributed.broadcast`: RCCL over xGMI on the GPU box, gloo in the CPU tests) and every rank phases its
own contigs.  The only other collectives are the barrier and the MAX-reduce of the timed region in bench.py.
"""
import numpy as np


def lpt_assign(costs, world):
    """Longest-processing-time-first assignment: returns int32 [n] rank of every item (deterministic)."""
    costs = np.asarray(costs, np.float64)
    order = np.lexsort((np.arange(len(costs)), -costs))
    load = np.zeros(world, np.float64)
    owner = np.zeros(len(costs), np.int32)
    for i in order:
        r = int(np.argmin(load))          # first minimum: deterministic ties
        owner[i] = r
        load[r] += costs[i]
    return owner


def broadcast_queue(owner, dist=None, device="cpu"):
    """Broadcast rank 0's queue descriptor to every rank (no-op without an initialised process group)."""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(owner, np.int32)
    import torch
    n = torch.tensor([len(owner) if dist.get_rank() == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=0)
    t = torch.zeros(int(n.item()), dtype=torch.int32, device=device)
    if dist.get_rank() == 0:
        t.copy_(torch.from_numpy(np.asarray(owner, np.int32)))
    dist.broadcast(t, src=0)
    return t.cpu().numpy()


def my_items(owner, rank):
    return np.nonzero(np.asarray(owner) == rank)[0]
