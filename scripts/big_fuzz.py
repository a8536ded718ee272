#!/usr/bin/env python
"""(GPU) LARGE random pileups (hundreds to thousands of reads per block, reads of up to 400 cells: several LDS tiles per read, long hash windows) through the default
S1 path and the general-insert path against the oracle.   usage: scripts/big_fuzz.py [first seed = 0] [count = 100]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib
from oracle import oracle
from tests.helpers import random_pileup

oracle.build()
ctx = lib.FloriaHip(0)
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(555000 + seed)
    alleles = 4 if rng.random() < 0.15 else 2
    n_snps = int(rng.integers(100, 900))
    pile = random_pileup(rng, int(rng.integers(400, 3000)), n_snps, int(rng.integers(1, 6)), max_len=int(rng.choice([40, 120, 400])),
                         alleles=alleles, q0_frac=0.05 if rng.random() < 0.2 else 0.0, err=float(rng.choice([0.01, 0.05, 0.15])), drop=float(rng.choice([0.0, 0.2])))
    S = int(pile.last.max())
    s = np.asarray([1, max(1, S // 4), max(1, S // 2)], np.uint32)
    e = np.asarray([S, min(S, S // 4 + 300), min(S, S // 2 + 150)], np.uint32)
    eps = float(rng.choice([0.03125, 0.04]))
    ro = oracle.phase_blocks(pile, s, e, oracle.make_params(eps, 5, 10), threads=16)
    for nb in (0, 1):
        ctx.set_option("no_bulk", nb)
        rg = ctx.phase_blocks(pile, s, e, lib.make_params(eps, 5, 10))
        same = (np.array_equal(ro.best_ploidy, rg.best_ploidy) and np.array_equal(ro.part, rg.part) and np.array_equal(ro.mec.view(np.uint64), rg.mec.view(np.uint64))
                and np.array_equal(ro.ploidies_tried, rg.ploidies_tried) and (ro.min_prune_margin == rg.min_prune_margin or abs(ro.min_prune_margin - rg.min_prune_margin) <= 1e-11))
        if not same:
            bad += 1
            print(f"MISMATCH seed {seed} no_bulk {nb} eps {eps} alleles {alleles} reads {pile.n_reads} snps {n_snps}: best {ro.best_ploidy} / {rg.best_ploidy}")
ctx.set_option("no_bulk", 0)
print(f"seeds {s0}..{s0 + cnt - 1} (large pileups, default and general insert path): {bad} mismatches")
