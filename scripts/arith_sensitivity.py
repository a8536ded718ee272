#!/usr/bin/env python
"""How much do floria's results depend on the ORDER OF ITS f64 ADDITIONS?  (CPU, oracle only)

The HIP path (and oracle arithmetic mode 0) carries every weighted sum as an exact (Q24 integer, #epsilon) pair and forms
Q * 2^-24 + m * eps once.  The reference keeps running f64 sums (`diff += epsilon` between `diff += w`, utils_frags.rs:32-75;
error_vec per partition, global_clustering.rs:196-202; `errors +=` per position, local_clustering.rs:218-260), in the iteration
order of its hash containers.  oracle.set_arith_mode(1) restates exactly that (orders from the FxHashSet emulator).  For a dyadic
epsilon both are exact and must agree; elsewhere this script counts the blocks (S1) and reads (S2) whose RESULT differs.

Where the results differ the two are compared as SOLUTIONS: the block's MEC at the chosen ploidy (lower is better), and the share of
reads whose cluster's majority strain is their own strain in the generator's truth (purity).

usage: scripts/arith_sensitivity.py [contigs per config = 6] [epsilon ... = 0.03125 0.04 0.05 0.0437]"""
import sys
sys.path.insert(0, ".")
import itertools
import numpy as np
from floria_amd import synth
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
eps_list = [float(x) for x in sys.argv[2:]] or [0.03125, 0.04, 0.05, 0.0437]


def purity(part, truth):
    """reads whose cluster's majority strain is their own"""
    ok = 0
    for k in np.unique(part):
        t = truth[part == k]
        ok += np.bincount(t).max()
    return ok


def same_up_to_labels(pa, pb, p):
    if len(pa) == 0:
        return True
    for perm in itertools.permutations(range(p)):
        if np.array_equal(np.asarray(perm, np.uint8)[pa], pb):
            return True
    return False


for eps in eps_list:
    for order_mode in (0, 2):
        blocks = ploidy_diff = part_diff = part_diff_mod_labels = mec_bits = 0
        mec_lower = mec_higher = pur_reads = pur0 = pur1 = 0
        mec_sum0 = mec_sum1 = 0.0
        s2_reads = s2_moved = 0
        for cfg, scale in ((2, 0.05), (3, 0.1), (4, 1.0)):
            C = synth.CONFIGS[cfg]
            for idx in range(n):
                c = synth.make_config_contig(cfg, idx, scale, keep_truth=True)
                s, e = oracle.block_ranges(c.snp_pos, C["block_length"])
                par = oracle.make_params(eps, C["max_ploidy"], C["beam"])
                oracle.set_order_mode(order_mode)
                res = []
                for am in (0, 1):
                    oracle.set_arith_mode(am)
                    res.append(oracle.phase_blocks(c.pileup, s, e, par, threads=1 if order_mode == 2 else 8))
                oracle.set_arith_mode(0)
                r0, r1 = res
                for b in range(r0.n_blocks):
                    blocks += 1
                    if r0.best_ploidy[b] != r1.best_ploidy[b]:
                        ploidy_diff += 1; part_diff += 1; part_diff_mod_labels += 1
                        continue
                    pa, pb = r0.block(b)[1], r1.block(b)[1]
                    if not np.array_equal(pa, pb):
                        part_diff += 1
                        if not same_up_to_labels(pa, pb, max(1, int(r0.best_ploidy[b]))):
                            part_diff_mod_labels += 1
                    if not np.array_equal(r0.mec[b].view(np.uint64), r1.mec[b].view(np.uint64)):
                        mec_bits += 1
                    if r0.best_ploidy[b] and r1.best_ploidy[b]:
                        m0, m1 = float(r0.mec[b][r0.best_ploidy[b] - 1]), float(r1.mec[b][r1.best_ploidy[b] - 1])
                        mec_sum0 += m0; mec_sum1 += m1
                        if m0 < m1 - 1e-9: mec_lower += 1
                        elif m0 > m1 + 1e-9: mec_higher += 1
                        ids = r0.block(b)[0]
                        tr = np.asarray(c.strain)[ids]
                        pur_reads += len(ids); pur0 += purity(r0.block(b)[1], tr); pur1 += purity(r1.block(b)[1], tr)
                if order_mode == 0:          # S2 on the canonical S1 result, one group per (block, partition) with the block's range
                    groups, ranges = [], []
                    for b in range(r0.n_blocks):
                        ids, part = r0.block(b)
                        for k in range(int(r0.best_ploidy[b])):
                            groups.append(ids[part == k]); ranges.append((int(s[b]), int(e[b])))
                    out = []
                    for am in (0, 1):
                        oracle.set_arith_mode(am)
                        g = oracle.reassign(c.pileup, groups, ranges, eps)
                        w = {}
                        for k in range(g.n_groups):
                            for x in g.group(k):
                                w[int(x)] = (tuple(g.range[k]), k)
                        out.append(w)
                    oracle.set_arith_mode(0)
                    s2_reads += len(out[0]); s2_moved += sum(1 for x in out[0] if out[1].get(x) != out[0][x])
        oracle.set_order_mode(0)
        print(f"eps={eps:<8g} order={'ascending' if order_mode == 0 else 'emulated FxHashSet'}: S1 {blocks} blocks: ploidy differs {ploidy_diff}, "
              f"partition differs {part_diff} (beyond a relabelling: {part_diff_mod_labels}), mec vector differs in some bit {mec_bits}"
              + (f"; S2 {s2_moved} of {s2_reads} reads land elsewhere" if order_mode == 0 else ""), flush=True)
        print(f"    as solutions: MEC at the chosen ploidy canonical lower in {mec_lower} / higher in {mec_higher} blocks, summed {mec_sum0:.3f} vs {mec_sum1:.3f}; "
              f"purity vs the generator's strains {pur0 / max(1, pur_reads):.5f} vs {pur1 / max(1, pur_reads):.5f} ({pur_reads} read placements)", flush=True)
