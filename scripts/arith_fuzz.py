#!/usr/bin/env python
"""(GPU) random pileups through S1 in both arithmetics against the oracle (mode 0 / mode 1), many seeds: a wider net than the test suite's.
(Round 5: the replay / HBM-table / claim-table options of the reference arithmetic are drawn at random per seed, so the fallback paths behind the home-bucket rule run too.)
usage: scripts/arith_fuzz.py [first seed = 0] [count = 300]"""
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
EPS = (0.04, 0.05, 0.0437, 0.011, 0.0999, 0.03125)
bad = 0
margin_bits = 0
worst = 0.0
for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(424242 + seed)
    alleles = 4 if rng.random() < 0.25 else 2
    pile = random_pileup(rng, int(rng.integers(3, 220)), int(rng.integers(2, 120)), int(rng.integers(1, 6)), max_len=int(rng.integers(1, 90)),
                         alleles=alleles, q0_frac=0.15 if rng.random() < 0.3 else 0.0, err=float(rng.choice([0.0, 0.02, 0.1, 0.3])), drop=float(rng.choice([0.0, 0.1, 0.5])))
    S = int(pile.last.max())
    nb = int(rng.integers(1, 7))
    s = np.sort(rng.integers(1, S + 1, size=nb))
    e = np.minimum(S, s + rng.integers(0, 60, size=nb))
    eps = EPS[int(rng.integers(0, len(EPS)))]
    P, B, sens, stop = int(rng.integers(1, 8)), int(rng.integers(1, 13)), int(rng.integers(1, 4)), int(rng.integers(0, 2))
    if seed % 8 == 7:             # round 6: the wide beams too (-p <= 8, -n <= 40: BASELINE config 5's shape)
        wr = np.random.default_rng(77 + seed)
        P, B = int(wr.integers(5, 9)), int(wr.integers(13, 41))
    # the reference arithmetic's own launch options, at random: every map / every read's set replayed insertion by insertion, tables in HBM, a small claim table
    krng = np.random.default_rng(99 + seed)
    kn = {"arith_replay": int(krng.random() < 0.3), "arith_hbm": int(krng.random() < 0.15), "fx_tags": int(krng.choice([0, 0, 128, 256]))}
    for k, v in kn.items():
        ctx.set_option(k, v)
    for mode in (0, 1):
        oracle.set_arith_mode(mode); ctx.set_option("arith", mode)
        ro = oracle.phase_blocks(pile, s, e, oracle.make_params(eps, P, B, sens, stop), threads=4)
        rg = ctx.phase_blocks(pile, s, e, lib.make_params(eps, P, B, sens, stop))
        same = (np.array_equal(ro.best_ploidy, rg.best_ploidy) and np.array_equal(ro.part, rg.part) and np.array_equal(ro.mec.view(np.uint64), rg.mec.view(np.uint64))
                and np.array_equal(ro.ploidies_tried, rg.ploidies_tried))
        if not same:
            bad += 1
            print(f"RESULT MISMATCH seed {seed} mode {mode} knobs {kn} eps {eps} P {P} B {B} sens {sens} stop {stop} alleles {alleles}: best {ro.best_ploidy} / {rg.best_ploidy}")
        if ro.min_prune_margin != rg.min_prune_margin:          # a diagnostic built from exp / log: device libm against host libm
            margin_bits += 1
            rel = abs(ro.min_prune_margin - rg.min_prune_margin)
            worst = max(worst, rel)
            if rel > 1e-11:
                bad += 1
                print(f"MARGIN MISMATCH seed {seed} mode {mode}: {ro.min_prune_margin!r} / {rg.min_prune_margin!r}")
oracle.set_arith_mode(0)
print(f"seeds {s0}..{s0 + cnt - 1}, both arithmetics: {bad} mismatches of results (partitions, ploidies, MEC bits) or of the pruning margin beyond 1e-11; the margin differs from the oracle's in its last bits in {margin_bits} runs (largest difference {worst:.3g})")
