#!/usr/bin/env python
"""(GPU) the host-pileup entry points on random BATCHES of random pileups: pipelined from pinned CSR pileups, pipelined from the compact wire form, and the plain
upload + resident call — random chunk counts — against the oracle contig by contig.   usage: scripts/upload_fuzz.py [first seed = 0] [count = 200]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib
from oracle import oracle
from tests.helpers import random_pileup

oracle.build()
ctx = lib.FloriaHip(0)
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = runs = 0
for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(660000 + seed)
    nct = int(rng.integers(1, 9))
    mixed = rng.random() < 0.2                                     # a 4-allele or q = 0 contig in the batch: the pipelined call phases again with the matching kernels
    piles, bc, bs, be = [], [], [], []
    for i in range(nct):
        al = 4 if (mixed and i == nct - 1) else 2
        p = random_pileup(rng, int(rng.integers(2, 250)), int(rng.integers(2, 90)), int(rng.integers(1, 5)), max_len=int(rng.integers(1, 60)), alleles=al,
                          q0_frac=0.1 if (mixed and i == 0) else 0.0, err=float(rng.choice([0.0, 0.05, 0.2])), drop=float(rng.choice([0.0, 0.1, 0.5])))
        S = int(p.last.max())
        nb = int(rng.integers(1, 5))
        s = np.sort(rng.integers(1, S + 1, size=nb)); e = np.minimum(S, s + rng.integers(0, 40, size=nb))
        piles.append(p); bc += [i] * nb; bs += list(s); be += list(e)
    bc, bs, be = (np.asarray(x, np.uint32) for x in (bc, bs, be))
    eps = float(rng.choice([0.03125, 0.04]))
    P, B = int(rng.integers(1, 6)), int(rng.integers(1, 11))
    par = lib.make_params(eps, P, B)
    # round 6: a third of the batches in the reference's arithmetic (every route pipelines there too), half of those with HOST-GIVEN set orders for some contigs - random
    # permutations of every read's cells: whatever the order, the device must add in it exactly as the oracle does
    arith = int(rng.random() < 0.34)
    if arith and rng.random() < 0.5:
        for i, p in enumerate(piles):
            if rng.random() < 0.6:
                p.set_order = np.concatenate([rng.permutation(int(p.read_off[r + 1] - p.read_off[r])) for r in range(p.n_reads)]).astype(np.uint32)
    oracle.set_arith_mode(arith); ctx.set_option("arith", arith)
    want = []
    for i in range(nct):
        m = bc == i
        want.append(oracle.phase_blocks(piles[i], bs[m], be[m], oracle.make_params(eps, P, B), threads=4))
    ctx.set_option("upload_chunks", int(rng.integers(0, 6)))
    pin_arena, pinned = lib.pin_pileups(piles)
    arena, parr, _ = lib.pack_pileups(piles)
    hs = ctx.upload_batch(piles)
    got = {"pinned": ctx.phase_pileups_batch(pinned, bc, bs, be, par), "packed": ctx.phase_pileups_batch(parr, bc, bs, be, par),
           "pageable": ctx.phase_pileups_batch(piles, bc, bs, be, par), "resident": ctx.phase_blocks_batch(hs, bc, bs, be, par)}
    for h in hs:
        h.free()
    for how, r in got.items():
        runs += 1
        off = {i: 0 for i in range(nct)}
        ok = True
        for b in range(len(bc)):
            i = int(bc[b]); k = off[i]; off[i] += 1
            ro = want[i]
            ok = ok and ro.best_ploidy[k] == r.best_ploidy[b] and np.array_equal(ro.block(k)[0], r.block(b)[0]) and np.array_equal(ro.block(k)[1], r.block(b)[1]) \
                and np.array_equal(ro.mec[k].view(np.uint64), r.mec[b].view(np.uint64)) and ro.ploidies_tried[k] == r.ploidies_tried[b]
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed} route {how} contigs {nct} mixed {mixed} eps {eps} P {P} B {B} arith {arith} set orders {[p.set_order is not None for p in piles]}")
ctx.set_option("upload_chunks", 0); ctx.set_option("arith", 0); oracle.set_arith_mode(0)
print(f"seeds {s0}..{s0 + cnt - 1}: {runs} batch calls over four routes, {bad} mismatches")
