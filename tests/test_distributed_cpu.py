"""`-m "not gpu"`: the N>1 path (queue broadcast + sharding + max-over-ranks reduce) with world_size 2 on gloo.
The compute backend in this CPU test is the oracle (no GPU here); bench.py uses the same shard module with the
HIP library on every rank."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from floria_amd import shard, synth
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_contigs = 6
    owner = shard.lpt_assign([100 + i for i in range(n_contigs)], world) if rank == 0 else None
    owner = shard.broadcast_queue(owner, dist)
    mine = shard.my_items(owner, rank)
    blocks = 0
    parts = {}
    for i in mine:
        c = synth.make_config_contig(1, int(i), scale=0.3)
        s, e = oracle.block_ranges(c.snp_pos, 10000)
        r = oracle.phase_blocks(c.pileup, s, e, oracle.make_params(0.03125))
        blocks += len(s)
        parts[int(i)] = r.part
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([blocks], dtype=torch.int64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), owner=owner, mine=mine, tmax=t.numpy(), total=tot.numpy(),
             **{f"part{k}": v for k, v in parts.items()})
    dist.destroy_process_group()


def test_two_rank_sharding_covers_every_contig_once(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert np.array_equal(z[0]["owner"], z[1]["owner"])                  # every rank saw rank 0's queue
    assert sorted(list(z[0]["mine"]) + list(z[1]["mine"])) == list(range(6))
    assert float(z[0]["tmax"][0]) == 2.0 and int(z[0]["total"][0]) == int(z[1]["total"][0]) > 0
    # the sharded results equal a single-process run
    sys.path.insert(0, ROOT)
    from floria_amd import synth
    from oracle import oracle
    for r in range(world):
        for i in z[r]["mine"]:
            c = synth.make_config_contig(1, int(i), scale=0.3)
            s, e = oracle.block_ranges(c.snp_pos, 10000)
            ref = oracle.phase_blocks(c.pileup, s, e, oracle.make_params(0.03125))
            assert np.array_equal(ref.part, z[r][f"part{int(i)}"])
