#!/usr/bin/env python
"""Which read does a haplogroup split drop?  (CPU, oracle only; DESIGN.md §6, VERDICT r4 #8.)

separate_broken_haplogroups (part_block_manip.rs:27-98) splits a haplogroup at a coverage gap and drops the first read behind the gap; "first" among the
reads that share that first_position is the iteration order of an FxHashSet.  Counts, over stitched haplogroups of BASELINE config 1 / 3 / 4 slices, the
splits, the reads they drop, and the S2 results that change when ties are taken in descending id order or in an emulated set order instead of ascending id.
usage: scripts/a14_sensitivity.py [contigs per config = 6]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from floria_amd import synth
from oracle import oracle, stitch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
EPS = 0.03125
rows = []
for cfg, scale, thin in ((1, 1.0, 1.0), (3, 0.1, 1.0), (4, 1.0, 1.0), (3, 0.1, 0.35), (4, 1.0, 0.02)):
    C = synth.CONFIGS[cfg]
    groups_n = splits = dropped = changed = [0, 0]
    groups_n = splits = dropped = 0
    changed = {1: 0, 2: 0}
    ties = 0
    for idx in range(n):
        c = synth.make_config_contig(cfg, idx, scale)
        pile = c.pileup
        if thin < 1.0:                     # a thinned copy: coverage gaps inside haplogroups, i.e. splits, become common
            rng = np.random.default_rng(100 + idx)
            keep = np.nonzero(rng.random(pile.n_reads) < thin)[0]
            from floria_amd.pileup import Pileup
            pile = Pileup.from_reads([pile.read(int(r)) for r in keep])
        s, e = oracle.block_ranges(c.snp_pos, C["block_length"])
        r = oracle.phase_blocks(pile, s, e, oracle.make_params(EPS, C["max_ploidy"], C["beam"]), threads=8)
        cov, ew = oracle.hap_graph(pile, s, e, r)
        cols = stitch.build_hap_graph(r, s, e, cov, ew)
        _, flows = stitch.lp_optimum(cols)
        edges = stitch.lp_edges(cols)
        fl = [(edges[i][0], edges[i][1], float(round(x))) for i, x in enumerate(flows)]
        paths = stitch.disjoint_paths(cols, fl)
        groups, ranges = [p[2] for p in paths], [(p[0], p[1]) for p in paths]
        res = {}
        for mode in (0, 1, 2):
            oracle.set_a14_tie_mode(mode)
            oracle.a14_dropped(True)
            g = oracle.reassign(pile, groups, ranges, EPS)
            d = oracle.a14_dropped(True)
            res[mode] = (g, d)
        oracle.set_a14_tie_mode(0)
        g0, d0 = res[0]
        groups_n += len(groups); splits += g0.n_groups - len(groups) if d0 else 0; dropped += d0
        sig = lambda g: [(tuple(g.range[k]), tuple(int(x) for x in g.group(k))) for k in range(g.n_groups)]
        for mode in (1, 2):
            changed[mode] += sig(res[mode][0]) != sig(g0)
    rows.append((cfg, scale, thin, n, groups_n, dropped, changed[1], changed[2]))
    print(f"config {cfg} (scale {scale}, reads kept {thin:.2f}): {n} contigs, {groups_n} stitched haplogroups, {dropped} reads dropped by splits; "
          f"contigs whose final haplosets change: descending ties {changed[1]}, emulated set order {changed[2]}")
