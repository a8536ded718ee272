#!/usr/bin/env python
"""(GPU) the rows after S1 (hap-graph nodes and edges, S2, haploset COV / ERR, HAPQ) on random pileups against the oracle.
usage: scripts/f_rows_fuzz.py [first seed = 0] [count = 300]"""
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
EPS = 0.03125
bad = {"graph": 0, "s2": 0, "stats": 0, "hapq": 0}
runs = 0


def same_f64(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(31337 + seed)
    alleles = 4 if rng.random() < 0.3 else 2
    const_q = rng.random() < 0.3                                  # equal qualities: ties everywhere
    pile = random_pileup(rng, int(rng.integers(10, 300)), int(rng.integers(8, 120)), int(rng.integers(1, 5)), max_len=int(rng.integers(2, 50)), alleles=alleles,
                         q0_frac=0.1 if rng.random() < 0.3 else 0.0, err=float(rng.choice([0.0, 0.05, 0.3])), qlo=20 if const_q else 5, qhi=20 if const_q else 40)
    S = int(pile.last.max())
    bl = int(rng.integers(5, 60))
    snp_pos = np.cumsum(rng.integers(1, 400, size=S)).astype(np.uint64)            # genome positions of the SNPs
    s, e = lib.get_range_with_lengths(snp_pos, int(bl * 200))
    if len(s) == 0:
        continue
    rc = ctx.upload(pile)
    par = lib.make_params(EPS, 4, 8)
    r = ctx.phase_blocks_batch([rc], np.zeros(len(s), np.uint32), s, e, par)
    ro = oracle.phase_blocks(pile, s, e, oracle.make_params(EPS, 4, 8), threads=4)
    runs += 1
    g = ctx.hap_graph(r)
    cov, ew = oracle.hap_graph(pile, s, e, ro)
    if not (same_f64(cov, g.node_cov) and np.array_equal(ew, g.edge_w)):
        bad["graph"] += 1; print(f"GRAPH MISMATCH seed {seed}")
    groups, ranges = [], []
    for b in range(r.n_blocks):
        for part in r.partitions(b):
            if len(part):
                groups.append(part); ranges.append((int(s[b]), int(e[b])))
    if groups:
        go = oracle.reassign(pile, groups, ranges, EPS)
        gg = ctx.reassign(rc, groups, ranges, EPS)
        if not (go.n_groups == gg.n_groups and np.array_equal(go.range, gg.range) and np.array_equal(go.grp_off, gg.grp_off) and np.array_equal(go.grp_read, gg.grp_read)):
            bad["s2"] += 1; print(f"S2 MISMATCH seed {seed}")
        parts = [gg.group(k) for k in range(gg.n_groups)] + [np.zeros(0, np.uint32)]
        rngs = [tuple(int(x) for x in gg.range[k]) for k in range(gg.n_groups)] + [(1, 2)]
        st = ctx.haploset_stats([rc], [0] * len(parts), parts, rngs)
        for k in range(len(parts)):
            ref = oracle.haploset_stats(pile, parts[k], rngs[k][0], rngs[k][1])
            if not same_f64(ref, st[k]):
                bad["stats"] += 1; print(f"STATS MISMATCH seed {seed} group {k}: {ref} / {st[k]}"); break
        hq, rel, avg = ctx.hapq(rc, parts, rngs, snp_pos, int(bl * 200))
        ohq, orel, oavg = oracle.hapq(pile, parts, rngs, snp_pos, int(bl * 200))
        if not (np.array_equal(hq, ohq) and same_f64(rel, orel) and same_f64([avg], [oavg])):
            bad["hapq"] += 1; print(f"HAPQ MISMATCH seed {seed}")
    rc.free()
print(f"seeds {s0}..{s0 + cnt - 1}: {runs} pileups with blocks; mismatches {bad}")
