#!/usr/bin/env python
"""(GPU) the canonical S1 path under random launch knobs (kernel choice, insert path, stage plans, workgroup sizes, grid sizes) on random pileups against the
oracle: every knob combination must give the oracle's bits.   usage: scripts/knob_fuzz.py [first seed = 0] [count = 300]"""
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
KNOBS = {"beam_path": (0, 0, 1, 2, 3), "no_bulk": (0, 0, 1), "no_specialized": (0, 1), "no_p1_shortcut": (0, 1), "opt_global": (0, 0, 1), "opt_threads": (0, 128, 512, 1024),
         "speculate": (-1, 0, 0, 1, 2, 3), "groups": (0, 2), "slots": (0, 0, 48),
         "tail_overlap": (0, 1, 1), "tail_waves": (1, 2, 8), "opt_block_order": (0, 0, 1)}
DEFAULT = {"beam_path": 0, "no_bulk": 0, "no_specialized": 0, "no_p1_shortcut": 0, "opt_global": 0, "opt_threads": 0, "speculate": -1, "groups": 0, "slots": 0, "tail_overlap": 0, "tail_waves": 2, "opt_block_order": 0}
bad = 0
for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(990000 + seed)
    big = rng.random() < 0.15
    alleles = 4 if rng.random() < 0.2 else 2
    pile = random_pileup(rng, int(rng.integers(3, 700 if big else 200)), int(rng.integers(2, 300 if big else 100)), int(rng.integers(1, 6)), max_len=int(rng.integers(1, 120 if big else 50)),
                         alleles=alleles, q0_frac=0.15 if rng.random() < 0.25 else 0.0, err=float(rng.choice([0.0, 0.02, 0.1, 0.3])), drop=float(rng.choice([0.0, 0.1, 0.4])))
    S = int(pile.last.max())
    nb = int(rng.integers(1, 8))
    s = np.sort(rng.integers(1, S + 1, size=nb))
    e = np.minimum(S, s + rng.integers(0, 120 if big else 50, size=nb))
    eps = float(rng.choice([0.03125, 0.04, 0.0625, 0.1]))
    if rng.random() < 0.5:
        P, B = 5, 10                                    # the ploidy-specialised instances
    else:
        P, B = int(rng.integers(1, 9)), int(rng.integers(1, 14))
    sens, stop = int(rng.integers(1, 4)), int(rng.integers(0, 2))
    knobs = {k: (v[int(rng.integers(0, len(v)))]) for k, v in KNOBS.items()}
    for k, v in knobs.items():
        ctx.set_option(k, v)
    ro = oracle.phase_blocks(pile, s, e, oracle.make_params(eps, P, B, sens, stop), threads=4)
    rg = ctx.phase_blocks(pile, s, e, lib.make_params(eps, P, B, sens, stop))
    same = (np.array_equal(ro.best_ploidy, rg.best_ploidy) and np.array_equal(ro.part, rg.part) and np.array_equal(ro.mec.view(np.uint64), rg.mec.view(np.uint64))
            and np.array_equal(ro.ploidies_tried, rg.ploidies_tried) and (ro.min_prune_margin == rg.min_prune_margin or abs(ro.min_prune_margin - rg.min_prune_margin) <= 1e-11))
    if not same:
        bad += 1
        print(f"MISMATCH seed {seed} margins {ro.min_prune_margin!r} / {rg.min_prune_margin!r} eps {eps} P {P} B {B} sens {sens} stop {stop} alleles {alleles} knobs {knobs}: best {ro.best_ploidy} / {rg.best_ploidy}")
for k, v in DEFAULT.items():
    ctx.set_option(k, v)
print(f"seeds {s0}..{s0 + cnt - 1} under random launch knobs: {bad} mismatches")
