#!/usr/bin/env python
"""How much do floria's results depend on FxHash iteration order?  (CPU, oracle only)

S1: blocks whose result differs between the canonical order (ascending counter_id at opt_iterate, local_clustering.rs:304) and
    (a) descending order, (b) the emulated FxHashSet order (oracle.set_order_mode(2): fxhash 0.2.1 + hashbrown as published).
S2: reads that land in a different haplogroup when process_reads_for_final_parts (part_block_manip.rs:203) visits the reads in
    the emulated read_to_parts_map order instead of ascending counter_id.
usage: scripts/order_sensitivity.py [contigs per config = 6]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from floria_amd import synth
from oracle import oracle, stitch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
EPS = 0.03125
tot = {1: [0, 0], 2: [0, 0]}
s2_moved = s2_reads = 0
for cfg, scale in ((2, 0.05), (3, 0.1), (4, 1.0)):
    C = synth.CONFIGS[cfg]
    for idx in range(n):
        c = synth.make_config_contig(cfg, idx, scale)
        s, e = oracle.block_ranges(c.snp_pos, C["block_length"])
        par = oracle.make_params(EPS, C["max_ploidy"], C["beam"])
        oracle.set_order_mode(0)
        r0 = oracle.phase_blocks(c.pileup, s, e, par, threads=8)
        for mode in (1, 2):
            oracle.set_order_mode(mode)
            r = oracle.phase_blocks(c.pileup, s, e, par, threads=8 if mode == 1 else 1)
            diff = sum(1 for b in range(r.n_blocks) if r.best_ploidy[b] != r0.best_ploidy[b] or not np.array_equal(r.block(b)[1], r0.block(b)[1])
                       or not np.array_equal(r.mec[b].view(np.uint64), r0.mec[b].view(np.uint64)))
            tot[mode][0] += diff; tot[mode][1] += r.n_blocks
            if mode == 2:
                set_order = oracle.last_set_order(r)
                oracle.set_order_mode(0)
                cov, ew = oracle.hap_graph(c.pileup, s, e, r)
                cols = stitch.build_hap_graph(r, s, e, cov, ew)
                _, flows = stitch.lp_optimum(cols)
                edges = stitch.lp_edges(cols)
                fl = [(edges[i][0], edges[i][1], float(round(x))) for i, x in enumerate(flows)]
                paths, node_paths = stitch.disjoint_paths(cols, fl, return_nodes=True)
                nonempty = [b for b in range(r.n_blocks) if r.best_ploidy[b]]
                np_blocks = [[(nonempty[cc], rr) for cc, rr in p] for p in node_paths]
                order = oracle.s2_visit_order_emulated(r, set_order, np_blocks)
                groups, ranges = [p[2] for p in paths], [(p[0], p[1]) for p in paths]
                ga = oracle.reassign(c.pileup, groups, ranges, EPS)
                gb = oracle.reassign(c.pileup, groups, ranges, EPS, read_order=order)
                def where(g):
                    w = {}
                    for k in range(g.n_groups):
                        for x in g.group(k): w[int(x)] = tuple(g.range[k])
                    return w
                wa, wb = where(ga), where(gb)
                s2_reads += len(wa); s2_moved += sum(1 for x in wa if wb.get(x) != wa[x])
        oracle.set_order_mode(0)
print(f"S1, descending vs ascending order: {tot[1][0]} of {tot[1][1]} blocks differ")
print(f"S1, emulated FxHashSet order vs ascending: {tot[2][0]} of {tot[2][1]} blocks differ")
print(f"S2, emulated read_to_parts_map order vs ascending: {s2_moved} of {s2_reads} reads end in a different haplogroup")
