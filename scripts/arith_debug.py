#!/usr/bin/env python
"""(GPU) reference-arithmetic mode of the library against oracle arithmetic mode 1 on the random pileups of tests/test_gpu_arith.py: what differs, block by block."""
import sys
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib
from oracle import oracle
from tests.helpers import random_pileup

oracle.build()
ctx = lib.FloriaHip(0)
ctx.set_option("arith", 1)
oracle.set_arith_mode(1)
NON_DYADIC = (0.04, 0.05, 0.0437)
seeds = [int(x) for x in sys.argv[1:]] or range(18)
for seed in seeds:
    rng = np.random.default_rng(7000 + seed)
    alleles = 4 if seed % 3 == 2 else 2
    pile = random_pileup(rng, int(rng.integers(5, 150)), int(rng.integers(4, 60)), int(rng.integers(1, 5)), max_len=int(rng.integers(2, 40)),
                         alleles=alleles, q0_frac=0.1 if seed % 4 == 1 else 0.0, err=float(rng.choice([0.0, 0.05, 0.2])))
    S = int(pile.last.max())
    nb = int(rng.integers(1, 6))
    s = np.sort(rng.integers(1, S + 1, size=nb))
    e = np.minimum(S, s + rng.integers(0, 25, size=nb))
    eps = NON_DYADIC[seed % 3]
    P, B, sens, stop = int(rng.integers(1, 7)), int(rng.integers(1, 13)), int(rng.integers(1, 4)), int(rng.integers(0, 2))
    ro = oracle.phase_blocks(pile, s, e, oracle.make_params(eps, P, B, sens, stop), threads=1)
    rg = ctx.phase_blocks(pile, s, e, lib.make_params(eps, P, B, sens, stop))
    for b in range(ro.n_blocks):
        same = ro.best_ploidy[b] == rg.best_ploidy[b] and np.array_equal(ro.block(b)[1], rg.block(b)[1]) and np.array_equal(ro.mec[b].view(np.uint64), rg.mec[b].view(np.uint64))
        if not same:
            print(f"seed {seed} eps {eps} P {P} B {B} sens {sens} stop {stop} block {b} [{s[b]},{e[b]}] reads {len(ro.block(b)[0])}: best {ro.best_ploidy[b]} / {rg.best_ploidy[b]} tried {ro.ploidies_tried[b]} / {rg.ploidies_tried[b]}")
            print("   mec oracle", [x.hex() for x in ro.mec[b]])
            print("   mec hip   ", [x.hex() for x in rg.mec[b]])
            print("   part oracle", ro.block(b)[1][:60])
            print("   part hip   ", rg.block(b)[1][:60])
    print(f"seed {seed}: done, margin {ro.min_prune_margin!r} / {rg.min_prune_margin!r}")
