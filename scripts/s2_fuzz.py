#!/usr/bin/env python
"""(GPU) S1 -> haplogroups of overlapping blocks -> S2 (floria_hip_reassign) on random pileups, both arithmetics, ascending and random visiting orders,
against the oracle (mode 0 / mode 1).   usage: scripts/s2_fuzz.py [first seed = 0] [count = 300]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib
from oracle import oracle
from tests.helpers import random_pileup

oracle.build()
ctx = lib.FloriaHip(0)
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 300
EPS = (0.04, 0.05, 0.0437, 0.03125)
bad = runs = 0
for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(77000 + seed)
    alleles = 4 if rng.random() < 0.25 else 2
    pile = random_pileup(rng, int(rng.integers(8, 200)), int(rng.integers(6, 100)), int(rng.integers(1, 5)), max_len=int(rng.integers(2, 60)),
                         alleles=alleles, q0_frac=0.1 if rng.random() < 0.3 else 0.0, err=float(rng.choice([0.0, 0.05, 0.2])))
    S = int(pile.last.max())
    nb = int(rng.integers(2, 7))
    s = np.sort(rng.integers(1, S + 1, size=nb))
    e = np.minimum(S, s + rng.integers(3, 50, size=nb))
    eps = EPS[int(rng.integers(0, len(EPS)))]
    for mode in (0, 1):
        oracle.set_arith_mode(mode); ctx.set_option("arith", mode)
        r = ctx.phase_blocks(pile, s, e, lib.make_params(eps, 4, 8))
        groups, ranges = [], []
        for b in range(r.n_blocks):
            for part in r.partitions(b):
                if len(part):
                    groups.append(part); ranges.append((int(s[b]), int(e[b])))
        if not groups:
            continue
        members = np.unique(np.concatenate(groups))
        for order in (None, rng.permutation(members).astype(np.uint32)):
            go = oracle.reassign(pile, groups, ranges, eps, read_order=order)
            gg = ctx.reassign(pile, groups, ranges, eps, read_order=order)
            runs += 1
            if not (go.n_groups == gg.n_groups and np.array_equal(go.range, gg.range) and np.array_equal(go.grp_off, gg.grp_off) and np.array_equal(go.grp_read, gg.grp_read)):
                bad += 1
                print(f"MISMATCH seed {seed} mode {mode} eps {eps} order {'given' if order is not None else 'ascending'}: groups {go.n_groups} / {gg.n_groups}")
oracle.set_arith_mode(0)
print(f"seeds {s0}..{s0 + cnt - 1}: {runs} S2 runs, {bad} mismatches")
