"""`-m gpu`: libfloria_hip.so (through the C ABI) against the CPU oracle and the committed fixtures.

Bar: bit-exact — read->haplotype assignments, chosen ploidy and the f64 MEC vector (built from exact integer
sums) are compared with array_equal, never with a tolerance.
"""
import glob
import os

import numpy as np
import pytest

from floria_amd import synth
from floria_amd.pileup import Pileup, reads_in_interval
from tests.helpers import assert_block_results_equal, random_pileup
from tests.test_golden_oracle import GOLD, load_golden

pytestmark = pytest.mark.gpu
EPS = 0.03125


def both(gpu_ctx, hip_lib, oracle_mod, pile, s, e, eps=EPS, P=5, B=10, sens=2, stop=1, threads=8):
    ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(eps, P, B, sens, stop), threads=threads)
    rg = gpu_ctx.phase_blocks(pile, s, e, hip_lib.make_params(eps, P, B, sens, stop))
    return ro, rg


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_fixtures(gpu_ctx, hip_lib, path):
    z, pile = load_golden(path)
    r = gpu_ctx.phase_blocks(pile, z["blk_start"], z["blk_end"],
                             hip_lib.make_params(float(z["eps"]), int(z["max_ploidy"]), int(z["beam"]), int(z["sens"]), int(z["stop"])))
    assert np.array_equal(r.best_ploidy, z["best_ploidy"]) and np.array_equal(r.ploidies_tried, z["ploidies_tried"])
    assert np.array_equal(r.read_off, z["out_read_off"]) and np.array_equal(r.read_id, z["out_read_id"])
    assert np.array_equal(r.part, z["out_part"])
    assert np.array_equal(r.mec.view(np.uint64), z["mec"].view(np.uint64))


def test_kat1_through_c_abi(gpu_ctx, hip_lib):
    p = Pileup.from_reads([([1, 2, 3, 4], [i % 2] * 4, [20] * 4) for i in range(6)])
    r = gpu_ctx.phase_blocks(p, [1], [4], hip_lib.make_params(EPS))
    assert list(r.best_ploidy) == [2] and list(r.mec[0]) == [12.0, 0, 0, 0, 0]
    parts = r.partitions(0)
    assert list(parts[0]) == [1, 3, 5] and list(parts[1]) == [0, 2, 4]     # heap tie-breaking


@pytest.mark.parametrize("seed", range(12))
def test_random_small_pileups(gpu_ctx, hip_lib, oracle_mod, seed):
    rng = np.random.default_rng(100 + seed)
    alleles = 4 if seed % 3 == 2 else 2
    pile = random_pileup(rng, int(rng.integers(5, 150)), int(rng.integers(4, 60)), int(rng.integers(1, 5)), max_len=int(rng.integers(2, 30)),
                         alleles=alleles, q0_frac=0.1 if seed % 4 == 1 else 0.0, err=float(rng.choice([0.0, 0.05, 0.2])))
    S = int(pile.last.max())
    nb = int(rng.integers(1, 6))
    s = np.sort(rng.integers(1, S + 1, size=nb))
    e = np.minimum(S, s + rng.integers(0, 25, size=nb))
    eps = [EPS, 0.04, 0.0625, 0.1][seed % 4]
    ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, s, e, eps=eps, P=int(rng.integers(1, 7)), B=int(rng.integers(1, 13)),
                  sens=int(rng.integers(1, 4)), stop=int(rng.integers(0, 2)))
    assert_block_results_equal(ro, rg, f"seed {seed}")


def test_wide_beam_uses_lds_heap_beyond_one_wave(gpu_ctx, hip_lib, oracle_mod):
    # ploidy*beam > 64 states: more (state,partition) pairs than lanes, heap larger than a wavefront
    rng = np.random.default_rng(7)
    pile = random_pileup(rng, 120, 30, 4, max_len=10, err=0.25)
    ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, [1, 8], [20, 30], P=6, B=30)
    assert_block_results_equal(ro, rg, "wide beam")


def test_long_reads_span_several_lds_tiles(gpu_ctx, hip_lib, oracle_mod):
    rng = np.random.default_rng(9)
    pile = random_pileup(rng, 40, 900, 2, max_len=700, drop=0.02)     # reads of up to 700 cells (> BEAM_TILE = 256)
    ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, [1, 300], [400, 900], P=3)
    assert_block_results_equal(ro, rg, "long reads")


def test_edge_cases(gpu_ctx, hip_lib, oracle_mod):
    # empty block, block past every read, single read, reads spanning > 10000 SNPs are ignored
    # (local_clustering.rs:44-46), max_ploidy 1, beam 1
    reads = [([3, 4, 5], [0, 1, 0], [30, 30, 30]), ([1, 20000], [1, 1], [30, 30]), ([4, 6], [1, 1], [10, 0])]
    pile = Pileup.from_reads(reads)
    for P, B in ((1, 1), (3, 10)):
        ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, [1, 3, 7, 30000], [2, 6, 9, 30010], P=P, B=B)
        assert_block_results_equal(ro, rg, f"edge P={P}")
        assert rg.best_ploidy[0] == 0 and rg.best_ploidy[3] == 0
        assert 0 not in rg.block(1)[0]          # read 0 (after the Frag::cmp sort) spans 19999 SNPs
    # zero blocks
    r = gpu_ctx.phase_blocks(pile, [], [], hip_lib.make_params(EPS))
    assert r.n_blocks == 0 and r.read_id.size == 0


def test_invalid_inputs_fail_loudly(gpu_ctx, hip_lib):
    bad = Pileup(np.array([0, 1, 2], np.uint32), np.array([5, 3], np.uint32), np.zeros(2, np.uint8), np.full(2, 20, np.uint8),
                 np.array([5, 3], np.uint32), np.array([5, 3], np.uint32))          # not sorted by Frag::cmp
    with pytest.raises(hip_lib.FloriaHipError) as ei:
        gpu_ctx.phase_blocks(bad, [1], [5], hip_lib.make_params(EPS))
    assert ei.value.code == -1
    al = Pileup.from_reads([([1, 2], [0, 5], [20, 20])])
    with pytest.raises(hip_lib.FloriaHipError) as ei:
        gpu_ctx.phase_blocks(al, [1], [2], hip_lib.make_params(EPS))
    assert ei.value.code == -4
    ok = Pileup.from_reads([([1, 2], [0, 1], [20, 20])])
    with pytest.raises(hip_lib.FloriaHipError):
        gpu_ctx.phase_blocks(ok, [1], [2], hip_lib.make_params(EPS, max_ploidy=0))
    with pytest.raises(hip_lib.FloriaHipError):
        gpu_ctx.phase_blocks(ok, [1], [2], hip_lib.make_params(1.5))


@pytest.mark.parametrize("cfg,idx,scale,eps", [(2, 0, 0.05, EPS), (4, 0, 1.0, EPS), (4, 5, 1.0, 0.04), (3, 1, 0.3, EPS), (5, 0, 0.01, EPS)])
def test_baseline_config_slices(gpu_ctx, hip_lib, oracle_mod, cfg, idx, scale, eps):
    C = synth.CONFIGS[cfg]
    c = synth.make_config_contig(cfg, idx, scale)
    s, e = hip_lib.get_range_with_lengths(c.snp_pos, C["block_length"])
    if cfg == 5:
        s, e = s[:3], e[:3]
    ro, rg = both(gpu_ctx, hip_lib, oracle_mod, c.pileup, s, e, eps=eps, P=C["max_ploidy"], B=C["beam"])
    assert_block_results_equal(ro, rg, f"config {cfg} contig {idx}")
    # parity certificate: no pruning decision was within libm noise of the threshold
    assert rg.min_prune_margin > 1e-9 and rg.min_prune_margin == ro.min_prune_margin


def test_batch_equals_per_contig_and_is_deterministic(gpu_ctx, hip_lib):
    # size-independent properties at BASELINE config-4 contig size: batching, slot count and repetition never change results
    contigs = [synth.make_config_contig(4, i) for i in range(12)]
    res = [gpu_ctx.upload(c.pileup) for c in contigs]
    par = hip_lib.make_params(EPS)
    bc, bs, be, singles = [], [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
        bc += [i] * len(s); bs += list(s); be += list(e)
        singles.append(gpu_ctx.phase_blocks(res[i], s, e, par))
    a = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
    gpu_ctx.set_slots(37)
    b = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
    gpu_ctx.set_slots(0)
    assert_block_results_equal(a, b, "slots")
    assert np.array_equal(a.part, np.concatenate([x.part for x in singles]))
    assert np.array_equal(a.best_ploidy, np.concatenate([x.best_ploidy for x in singles]))
    # every read of every block is assigned exactly one haplotype < best_ploidy; read lists ascend
    for blk in range(a.n_blocks):
        ids, part = a.block(blk)
        assert np.all(np.diff(ids.astype(np.int64)) > 0) and np.all(part < a.best_ploidy[blk])
    # planted strain count is recovered on most blocks of these high-coverage contigs
    truth = np.array([contigs[i].ploidy_truth for i in bc])
    assert np.mean(a.best_ploidy == truth) > 0.8
    for r in res:
        r.free()


def groups_from_blocks(r, s, e):
    groups, ranges = [], []
    for b in range(r.n_blocks):
        for part in r.partitions(b):
            if len(part):
                groups.append(part); ranges.append((int(s[b]), int(e[b])))
    return groups, ranges


@pytest.mark.parametrize("cfg,idx,scale", [(1, 0, 1.0), (4, 2, 0.5), (3, 0, 0.2)])
def test_reassign_parity(gpu_ctx, hip_lib, oracle_mod, cfg, idx, scale):
    C = synth.CONFIGS[cfg]
    c = synth.make_config_contig(cfg, idx, scale)
    s, e = hip_lib.get_range_with_lengths(c.snp_pos, C["block_length"])
    r = gpu_ctx.phase_blocks(c.pileup, s, e, hip_lib.make_params(EPS))
    groups, ranges = groups_from_blocks(r, s, e)           # overlapping blocks -> reads sit in several haplogroups
    go = oracle_mod.reassign(c.pileup, groups, ranges, EPS)
    gg = gpu_ctx.reassign(c.pileup, groups, ranges, EPS)
    assert go.n_groups == gg.n_groups
    assert np.array_equal(go.range, gg.range) and np.array_equal(go.grp_off, gg.grp_off) and np.array_equal(go.grp_read, gg.grp_read)
    # every input read ends in at most one group; dropped reads only come from separate_broken_haplogroups (:69-84)
    assert len(np.unique(gg.grp_read)) == len(gg.grp_read)


def test_reassign_with_coverage_gap_splits_group(gpu_ctx, hip_lib, oracle_mod):
    reads = [([1, 2, 3], [0, 0, 0], [30] * 3), ([2, 3, 4], [0, 0, 0], [30] * 3), ([7, 8, 9], [1, 1, 1], [30] * 3), ([8, 9, 10], [1, 1, 1], [30] * 3),
             ([1, 2], [1, 1], [30] * 2)]
    pile = Pileup.from_reads(reads)
    groups = [np.arange(pile.n_reads, dtype=np.uint32), np.array([0, 1], np.uint32)]
    ranges = [(1, 10), (1, 4)]
    go = oracle_mod.reassign(pile, groups, ranges, EPS)
    gg = gpu_ctx.reassign(pile, groups, ranges, EPS)
    assert np.array_equal(go.range, gg.range) and np.array_equal(go.grp_off, gg.grp_off) and np.array_equal(go.grp_read, gg.grp_read)
    assert gg.n_groups >= 3        # group (1,10) is split at the coverage break after SNP 4


def test_reassign_batch_equals_per_contig_oracle(gpu_ctx, hip_lib, oracle_mod):
    # S2 for several contigs in one launch == the oracle called contig by contig (floria.rs:229,359-366)
    contigs = [synth.make_config_contig(4, i, 0.4) for i in range(5)]
    res = [gpu_ctx.upload(c.pileup) for c in contigs]
    par = hip_lib.make_params(EPS)
    all_groups, all_ranges, gc, per = [], [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
        r = gpu_ctx.phase_blocks(res[i], s, e, par)
        g, rg = groups_from_blocks(r, s, e)
        per.append((g, rg))
        all_groups += g; all_ranges += rg; gc += [i] * len(g)
    out = gpu_ctx.reassign_batch(res, gc, all_groups, all_ranges, EPS)
    assert len(out) == len(contigs)
    for i, c in enumerate(contigs):
        go = oracle_mod.reassign(c.pileup, per[i][0], per[i][1], EPS)
        assert go.n_groups == out[i].n_groups
        assert np.array_equal(go.range, out[i].range) and np.array_equal(go.grp_off, out[i].grp_off) and np.array_equal(go.grp_read, out[i].grp_read)
    for r in res:
        r.free()


@pytest.mark.parametrize("cfg,n_contigs,scale", [(1, 1, 1.0), (4, 3, 0.6), (3, 2, 0.2)])
def test_hap_graph_nodes_and_edges(gpu_ctx, hip_lib, oracle_mod, cfg, n_contigs, scale):
    # HapNode::new coverage + update_hap_graph edge weights on the batch that is still resident
    C = synth.CONFIGS[cfg]
    contigs = [synth.make_config_contig(cfg, i, scale) for i in range(n_contigs)]
    res = [gpu_ctx.upload(c.pileup) for c in contigs]
    bc, bs, be, per = [], [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, C["block_length"])
        per.append((s, e, len(bs)))
        bc += [i] * len(s); bs += list(s); be += list(e)
    r = gpu_ctx.phase_blocks_batch(res, bc, bs, be, hip_lib.make_params(EPS))
    g = gpu_ctx.hap_graph(r)
    assert np.array_equal(np.diff(g.node_off), r.best_ploidy)
    for i, c in enumerate(contigs):
        s, e, b0 = per[i]
        ro = oracle_mod.phase_blocks(c.pileup, s, e, oracle_mod.make_params(EPS), threads=4)
        cov, ew = oracle_mod.hap_graph(c.pileup, s, e, ro)
        lo, hi = int(g.node_off[b0]), int(g.node_off[b0 + len(s)])
        assert np.array_equal(cov.view(np.uint64), g.node_cov[lo:hi].view(np.uint64)), f"contig {i} node cov"
        elo, ehi = int(g.edge_off[b0]), int(g.edge_off[b0 + len(s)])
        assert np.array_equal(ew, g.edge_w[elo:ehi]), f"contig {i} edge weights"
        assert g.pred[b0] == -1                          # no edge across contigs
    assert g.edge_w.sum() > 0
    # the resident copy is invalidated by the next call on the context
    gpu_ctx.phase_blocks(res[0], per[0][0], per[0][1], hip_lib.make_params(EPS))
    with pytest.raises(hip_lib.FloriaHipError):
        gpu_ctx.hap_graph(r)
    for x in res:
        x.free()


@pytest.mark.parametrize("seed", range(24))
def test_random_medium_pileups_all_beam_paths(gpu_ctx, hip_lib, oracle_mod, seed):
    # medium blocks (hundreds of reads, deeper coverage, ties from identical reads) through the three beam kernels: shared-slab with the
    # register heap (default), the generic per-state-slab LDS-heap kernel and the wide-beam shared-slab kernel must all equal the oracle
    rng = np.random.default_rng(5000 + seed)
    ploidy = int(rng.integers(1, 6))
    pile = random_pileup(rng, int(rng.integers(150, 600)), int(rng.integers(20, 120)), ploidy, max_len=int(rng.integers(3, 60)),
                         alleles=2 if seed % 5 else 4, q0_frac=0.05 if seed % 6 == 0 else 0.0, err=float(rng.choice([0.0, 0.03, 0.1])),
                         qlo=20 if seed % 4 == 0 else 5, qhi=20 if seed % 4 == 0 else 40)
    S = int(pile.last.max())
    s = np.array([1, max(1, S // 3)]); e = np.array([max(1, S // 2), S])
    P, B = int(rng.integers(2, 6)), int(rng.integers(2, 11))
    eps = [EPS, 0.04][seed % 2]
    ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(eps, P, B), threads=4)
    try:
        for code, path in ((2, "slab"), (1, "generic"), (3, "wide")):
            gpu_ctx.set_option("beam_path", code)
            for spec in (0, 1):                      # one ploidy per stage / all ploidies of a block at once
                gpu_ctx.set_option("speculate", spec)
                rg = gpu_ctx.phase_blocks(pile, s, e, hip_lib.make_params(eps, P, B))
                assert_block_results_equal(ro, rg, f"seed {seed} path {path} speculate {spec}")
                assert rg.min_prune_margin == ro.min_prune_margin
    finally:
        gpu_ctx.set_option("beam_path", 0)
        gpu_ctx.set_option("speculate", -1)


def test_ploidy1_shortcut_equals_the_search(gpu_ctx, hip_lib):
    # the ploidy-1 beam search has no choice to make; skipping its launch must not change anything
    c = synth.make_config_contig(4, 7, 0.5)
    s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
    a = gpu_ctx.phase_blocks(c.pileup, s, e, hip_lib.make_params(EPS))
    gpu_ctx.set_option("no_p1_shortcut", 1)
    try:
        b = gpu_ctx.phase_blocks(c.pileup, s, e, hip_lib.make_params(EPS))
    finally:
        gpu_ctx.set_option("no_p1_shortcut", 0)
    assert_block_results_equal(a, b, "p1 shortcut")
    assert a.min_prune_margin == b.min_prune_margin


def test_reassign_with_caller_given_order(gpu_ctx, hip_lib, oracle_mod):
    # the greedy chain is order-dependent (the reference visits reads in FxHashMap order, part_block_manip.rs:203); with the
    # visiting order passed in, GPU and oracle agree for ANY order, and different orders really give different haplogroups
    c = synth.make_config_contig(4, 1, 0.5)
    s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
    r = gpu_ctx.phase_blocks(c.pileup, s, e, hip_lib.make_params(EPS))
    groups, ranges = groups_from_blocks(r, s, e)
    members = np.unique(np.concatenate(groups))
    rng = np.random.default_rng(3)
    outs = []
    for trial in range(3):
        order = rng.permutation(members).astype(np.uint32) if trial else members[::-1].copy()
        go = oracle_mod.reassign(c.pileup, groups, ranges, EPS, read_order=order)
        gg = gpu_ctx.reassign(c.pileup, groups, ranges, EPS, read_order=order)
        assert np.array_equal(go.range, gg.range) and np.array_equal(go.grp_off, gg.grp_off) and np.array_equal(go.grp_read, gg.grp_read)
        outs.append(gg.grp_read.tobytes() + gg.grp_off.tobytes())
    assert len(set(outs)) > 1
    # an order that misses a grouped read, or repeats one, is rejected
    with pytest.raises(hip_lib.FloriaHipError):
        gpu_ctx.reassign(c.pileup, groups, ranges, EPS, read_order=members[:-1])
    with pytest.raises(hip_lib.FloriaHipError):
        gpu_ctx.reassign(c.pileup, groups, ranges, EPS, read_order=np.concatenate([members, members[:1]]))


def test_haploset_stats(gpu_ctx, hip_lib, oracle_mod):
    # get_errors_cov_from_frags (utils_frags.rs:596-655): COV / ERR of the output headers, per haploset after S2
    contigs = [synth.make_config_contig(4, i, 0.4) for i in range(2)] + [synth.make_config_contig(3, 0, 0.2)]
    res = [gpu_ctx.upload(c.pileup) for c in contigs]
    groups, ranges, gc = [], [], []
    for i, c in enumerate(contigs):
        bl = 10000 if i < 2 else 500
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, bl)
        r = gpu_ctx.phase_blocks(res[i], s, e, hip_lib.make_params(EPS))
        g, rg = groups_from_blocks(r, s, e)
        out = gpu_ctx.reassign(res[i], g, rg, EPS)
        for k in range(out.n_groups):
            groups.append(out.group(k)); ranges.append(tuple(int(x) for x in out.range[k])); gc.append(i)
    groups.append(np.zeros(0, np.uint32)); ranges.append((3, 9)); gc.append(0)            # an empty haploset: err = 0/0 = NaN
    st = gpu_ctx.haploset_stats(res, gc, groups, ranges)
    assert st.shape == (len(groups), 4)
    for k in range(len(groups)):
        ref = oracle_mod.haploset_stats(contigs[gc[k]].pileup, groups[k], ranges[k][0], ranges[k][1])
        assert np.array_equal(ref.view(np.uint64), st[k].view(np.uint64)) or (np.isnan(ref[1]) and np.isnan(st[k][1]) and np.array_equal(ref[[0, 2, 3]], st[k][[0, 2, 3]])), (k, ref, st[k])
    assert np.isnan(st[-1][1]) and st[-1][0] == 0.0
    for x in res:
        x.free()


def test_mixed_batch_alleles_and_q0(gpu_ctx, hip_lib, oracle_mod):
    # one batch mixing a biallelic contig, a 4-allele contig and a contig with q=0 observations: the batch runs the
    # (A=4, presence-tracking) kernel instantiation for everyone; results per contig must equal the oracle's
    rng = np.random.default_rng(77)
    piles = [random_pileup(rng, 300, 80, 3, max_len=40, alleles=2), random_pileup(rng, 250, 60, 3, max_len=30, alleles=4),
             random_pileup(rng, 200, 50, 2, max_len=25, alleles=2, q0_frac=0.1)]
    res = [gpu_ctx.upload(p) for p in piles]
    bc, bs, be = [], [], []
    for i, p in enumerate(piles):
        S = int(p.last.max())
        bc += [i, i]; bs += [1, S // 2]; be += [S // 2 + 5 if S // 2 + 5 <= S else S, S]
    r = gpu_ctx.phase_blocks_batch(res, bc, bs, be, hip_lib.make_params(EPS, 4, 8))
    for i, p in enumerate(piles):
        ro = oracle_mod.phase_blocks(p, bs[2 * i:2 * i + 2], be[2 * i:2 * i + 2], oracle_mod.make_params(EPS, 4, 8))
        lo, hi = int(r.read_off[2 * i]), int(r.read_off[2 * i + 2])
        assert np.array_equal(ro.best_ploidy, r.best_ploidy[2 * i:2 * i + 2])
        assert np.array_equal(ro.read_id, r.read_id[lo:hi]) and np.array_equal(ro.part, r.part[lo:hi])
        assert np.array_equal(ro.mec.view(np.uint64), r.mec[2 * i:2 * i + 2].view(np.uint64))
    for x in res:
        x.free()


def numpy_mec_no_phred(p, ids, part, ploidy, eps):
    """get_mec_stats_epsilon_no_phred (local_clustering.rs:187-215) restated with numpy bincounts, independently of the oracle:
    per partition and SNP the allele counts; errors += total - max, and += eps where the consensus count is <= 1."""
    bad = 0.0
    for k in range(ploidy):
        rs = ids[part == k]
        if len(rs) == 0:
            continue
        lo, hi = p.read_off[rs], p.read_off[rs + 1]
        cells = np.concatenate([np.arange(a, b) for a, b in zip(lo, hi)])
        snp, al = p.snp[cells].astype(np.int64), p.allele[cells].astype(np.int64)
        base = snp.min()
        cnt = np.zeros((snp.max() - base + 1, 4), np.int64)
        np.add.at(cnt, (snp - base, al), 1)
        tot, mx = cnt.sum(1), cnt.max(1)
        have = tot > 0
        bad += float((tot - mx)[have].sum()) + float(np.count_nonzero(mx[have] <= 1)) * eps
    return bad


def test_bench_scale_properties(gpu_ctx, hip_lib, oracle_mod, monkeypatch):
    """Size-independent properties on a bench-shaped batch (BASELINE config 4 contigs at full contig size, enough blocks for the job
    groups to run on two streams): group count never changes results, the reported MEC of every sampled block equals an independent
    numpy recomputation from the returned partition, read lists equal the interval query, and a block sample equals the oracle."""
    n_contigs = 300
    contigs = [synth.make_config_contig(4, i) for i in range(n_contigs)]
    res = [gpu_ctx.upload(c.pileup) for c in contigs]
    par = hip_lib.make_params(EPS)
    bc, bs, be = [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
        bc += [i] * len(s); bs += list(s); be += list(e)
    assert len(bc) >= 2048                                           # enough blocks for two and three job groups (auto takes one below 48 x CUs blocks)
    gpu_ctx.set_option("groups", 1)
    one = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
    assert gpu_ctx.timing()["streams"] == 1
    gpu_ctx.set_option("groups", 2)
    two = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
    assert gpu_ctx.timing()["streams"] == 2
    gpu_ctx.set_option("groups", 3)
    three = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
    gpu_ctx.set_option("groups", 0)
    assert_block_results_equal(one, two, "1 vs 2 job groups")
    assert_block_results_equal(one, three, "1 vs 3 job groups")
    assert one.min_prune_margin == two.min_prune_margin == three.min_prune_margin > 1e-9
    assert np.all(two.ploidies_tried >= two.best_ploidy) and np.all(two.best_ploidy >= 1) and np.all(two.best_ploidy <= 5)
    rng = np.random.default_rng(1577)
    sample = rng.choice(two.n_blocks, size=96, replace=False)
    for blk in sample:
        ci = bc[blk]
        p = contigs[ci].pileup
        ids, part = two.block(int(blk))
        assert np.array_equal(ids, reads_in_interval(p, bs[blk], be[blk]))
        bp = int(two.best_ploidy[blk])
        assert np.all(part < bp)
        assert two.mec[blk, bp - 1] == numpy_mec_no_phred(p, ids.astype(np.int64), part, bp, EPS)
    # 32 of the sampled blocks against the oracle, bit for bit
    for blk in sample[:32]:
        ci = bc[blk]
        ro = oracle_mod.phase_blocks(contigs[ci].pileup, [bs[blk]], [be[blk]], oracle_mod.make_params(EPS))
        ids, part = two.block(int(blk))
        oid, opart = ro.block(0)
        assert ro.best_ploidy[0] == two.best_ploidy[blk] and ro.ploidies_tried[0] == two.ploidies_tried[blk]
        assert np.array_equal(oid, ids) and np.array_equal(opart, part)
        assert np.array_equal(ro.mec[0].view(np.uint64), two.mec[blk].view(np.uint64))
    for r in res:
        r.free()


def test_specialised_and_generic_slab_kernels_agree(gpu_ctx, hip_lib, oracle_mod, monkeypatch):
    # beam width 10 + biallelic + no q=0 cells runs beam_slab_kernel<2,false,P,10> (ploidy and beam as compile-time constants);
    # FLORIA_HIP_NO_SPECIALIZED=1 forces the runtime-parameter instance.  Both must equal the oracle.
    c = synth.make_config_contig(4, 11, 0.6)
    s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
    ro = oracle_mod.phase_blocks(c.pileup, s, e, oracle_mod.make_params(EPS), threads=8)
    spec = gpu_ctx.phase_blocks(c.pileup, s, e, hip_lib.make_params(EPS))
    gpu_ctx.set_option("no_specialized", 1)
    gen = gpu_ctx.phase_blocks(c.pileup, s, e, hip_lib.make_params(EPS))
    gpu_ctx.set_option("no_specialized", 0)
    assert_block_results_equal(ro, spec, "specialised")
    assert_block_results_equal(ro, gen, "generic")
    assert spec.min_prune_margin == gen.min_prune_margin == ro.min_prune_margin


def _same_f64(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64)) or (np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]))


@pytest.mark.parametrize("cfg,idx,scale,bl", [(1, 0, 1.0, None), (4, 3, 0.5, 10000), (3, 1, 0.2, 500)])
def test_hapq_parity(gpu_ctx, hip_lib, oracle_mod, cfg, idx, scale, bl):
    # get_hapq (part_block_manip.rs:517-616): HAPQ / REL_ERR / avg_err of the haplosets S2 returns (overlapping ranges of adjacent blocks)
    C = synth.CONFIGS[cfg]
    bl = bl or C["block_length"]
    c = synth.make_config_contig(cfg, idx, scale)
    s, e = hip_lib.get_range_with_lengths(c.snp_pos, bl)
    rc = gpu_ctx.upload(c.pileup)
    r = gpu_ctx.phase_blocks(rc, s, e, hip_lib.make_params(EPS, C["max_ploidy"], C["beam"]))
    g, rg = groups_from_blocks(r, s, e)
    # the raw block partitions overlap by a third of a block (graph_processing.rs:337): plenty of overlapping pairs
    g.append(np.zeros(0, np.uint32)); rg.append((1, 2))                                  # an empty haploset
    hq, rel, avg = gpu_ctx.hapq(rc, g, rg, c.snp_pos, bl)
    ohq, orel, oavg = oracle_mod.hapq(c.pileup, g, rg, c.snp_pos, bl)
    assert np.array_equal(hq, ohq), (hq[:20], ohq[:20])
    assert _same_f64(rel, orel) and _same_f64([avg], [oavg])
    assert hq.max() <= 60 and len(set(hq.tolist())) > 1
    rc.free()


def test_hapq_four_alleles_ties_and_q0(gpu_ctx, hip_lib, oracle_mod):
    # multi-allelic sites with equal qualities (ties everywhere) and q=0 cells: the consensus tie rule and the inner-map order
    rng = np.random.default_rng(77)
    pile = random_pileup(rng, 400, 60, 3, max_len=20, alleles=4, qlo=20, qhi=20, err=0.3, q0_frac=0.1)
    ids = np.arange(pile.n_reads, dtype=np.uint32)
    groups, ranges = [], []
    for k in range(24):                                                  # random overlapping haplosets, some tiny
        lo = int(rng.integers(1, 50)); hi = min(60, lo + int(rng.integers(1, 25)))
        m = (pile.first <= hi) & (pile.last >= lo)
        pick = ids[m][rng.random(int(m.sum())) < (0.5 if k % 3 else 0.05)]
        groups.append(pick.astype(np.uint32)); ranges.append((lo, hi))
    pos = np.cumsum(rng.integers(50, 400, size=60)).astype(np.uint64)
    rc = gpu_ctx.upload(pile)
    hq, rel, avg = gpu_ctx.hapq(rc, groups, ranges, pos, 2000)
    ohq, orel, oavg = oracle_mod.hapq(pile, groups, ranges, pos, 2000)
    assert np.array_equal(hq, ohq), (hq, ohq)
    assert _same_f64(rel, orel) and _same_f64([avg], [oavg])
    st = gpu_ctx.haploset_stats([rc], np.zeros(len(groups), np.uint32), groups, ranges)
    for k in range(len(groups)):
        assert _same_f64(st[k], oracle_mod.haploset_stats(pile, groups[k], ranges[k][0], ranges[k][1])), k
    rc.free()


def test_hapq_batch_equals_per_contig(gpu_ctx, hip_lib, oracle_mod):
    contigs = [synth.make_config_contig(4, i, 0.4) for i in range(3)] + [synth.make_config_contig(1, 0, 1.0)]
    res = [gpu_ctx.upload(c.pileup) for c in contigs]
    groups, ranges, gc, per = [], [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
        r = gpu_ctx.phase_blocks(res[i], s, e, hip_lib.make_params(EPS))
        g, rg = groups_from_blocks(r, s, e)
        per.append(gpu_ctx.hapq(res[i], g, rg, c.snp_pos, 10000))
        groups += g; ranges += rg; gc += [i] * len(g)
    # interleave the groups of different contigs: only the order inside a contig matters
    order = np.argsort(np.array(gc) * 0 + np.arange(len(gc)) % 7, kind="stable")
    hq, rel, avg = gpu_ctx.hapq_batch(res, [gc[k] for k in order], [groups[k] for k in order], [ranges[k] for k in order],
                                      [c.snp_pos for c in contigs], 10000)
    back = np.empty(len(order), np.int64); back[order] = np.arange(len(order))
    hq, rel = hq[back], rel[back]
    o = 0
    for i in range(len(contigs)):
        n = len(per[i][0])
        assert np.array_equal(hq[o:o + n], per[i][0]) and _same_f64(rel[o:o + n], per[i][1]) and _same_f64([avg[i]], [per[i][2]])
        ohq, orel, oavg = oracle_mod.hapq(contigs[i].pileup, groups[o:o + n], ranges[o:o + n], contigs[i].snp_pos, 10000)
        assert np.array_equal(per[i][0], ohq) and _same_f64(per[i][1], orel) and _same_f64([per[i][2]], [oavg])
        o += n
    for x in res:
        x.free()


@pytest.mark.parametrize("seed", range(16))
def test_random_pileups_default_beam_width(gpu_ctx, hip_lib, oracle_mod, seed):
    # beam width 10, biallelic, no q=0 cells: the ploidy-specialised instances (4 waves per SIMD for ploidy 2-3), the bulk insert path and its
    # hash-slot fallback, on inputs with many exact ties (equal qualities), noisy reads and very uneven read lengths
    rng = np.random.default_rng(9000 + seed)
    ploidy = int(rng.integers(1, 6))
    pile = random_pileup(rng, int(rng.integers(200, 900)), int(rng.integers(15, 150)), ploidy, max_len=int(rng.integers(2, 90)),
                         alleles=2, err=float(rng.choice([0.0, 0.02, 0.08, 0.25])), drop=float(rng.choice([0.0, 0.1, 0.4])),
                         qlo=30 if seed % 3 == 0 else 3, qhi=30 if seed % 3 == 0 else 41)
    S = int(pile.last.max())
    s = np.array([1, max(1, S // 4), max(1, S // 2)]); e = np.array([max(1, S // 2), max(1, 3 * S // 4), S])
    P = int(rng.integers(2, 6))
    eps = [EPS, 0.04, 0.0625][seed % 3]
    ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, s, e, eps=eps, P=P, B=10)
    assert_block_results_equal(ro, rg, f"seed {seed}")
    assert rg.min_prune_margin == ro.min_prune_margin


def test_reassign_sparse_choices_parallel_kernel(gpu_ctx, hip_lib, oracle_mod):
    # haplogroups as they come out of stitching: almost every read sits in ONE group, a few in two or three.  Such contigs run the
    # workgroup-parallel S2 kernel (single-candidate reads are added in bulk between the reads that have a choice); dense contigs
    # (test_reassign_parity) run the one-wavefront chain.  Both must equal the oracle for the canonical and for arbitrary orders.
    rng = np.random.default_rng(21)
    for cfg, idx, scale, n_groups in ((4, 6, 1.0, 4), (2, 0, 0.1, 3), (5, 0, 0.02, 8)):
        c = synth.make_config_contig(cfg, idx, scale, keep_truth=True)
        p = c.pileup
        S = int(p.last.max())
        groups = [np.nonzero(c.strain % n_groups == k)[0].astype(np.uint32) for k in range(n_groups)]
        ranges = [(1, S)] * n_groups
        extra = rng.choice(p.n_reads, size=max(4, p.n_reads // 40), replace=False)          # 2.5 % of the reads get a second / third candidate
        for r in extra:
            for k in rng.choice(n_groups, size=int(rng.integers(1, 3)), replace=False):
                if r not in groups[k]:
                    groups[k] = np.sort(np.append(groups[k], np.uint32(r)))
        for order in (None, rng.permutation(p.n_reads).astype(np.uint32)):
            go = oracle_mod.reassign(p, groups, ranges, EPS, read_order=order)
            gg = gpu_ctx.reassign(p, groups, ranges, EPS, read_order=order)
            assert np.array_equal(go.grp_off, gg.grp_off) and np.array_equal(go.grp_read, gg.grp_read) and np.array_equal(go.range, gg.range), (cfg, order is None)
            assert 0 < gpu_ctx.timing()["jobs"] < p.n_reads // 8                           # sparse: the parallel kernel took it
    # no read with a choice at all: no histogram work, assignment = the only candidate
    c = synth.make_config_contig(4, 9, 0.5, keep_truth=True)
    groups = [np.nonzero(c.strain == k)[0].astype(np.uint32) for k in range(int(c.strain.max()) + 1)]
    ranges = [(1, int(c.pileup.last.max()))] * len(groups)
    go = oracle_mod.reassign(c.pileup, groups, ranges, EPS)
    gg = gpu_ctx.reassign(c.pileup, groups, ranges, EPS)
    assert np.array_equal(go.grp_read, gg.grp_read) and np.array_equal(go.range, gg.range) and gpu_ctx.timing()["jobs"] == 0


def test_narrow_beam_with_many_slabs_takes_the_wide_kernel(gpu_ctx, hip_lib, oracle_mod):
    # -p 9 -n 7: ploidy*beam = 63 fits the register heap, but ploidy^2*beam = 567 slabs exceed the slab kernel's table (512): the wide-beam kernel
    # takes the job (round 1 kept a third kernel for this corner)
    rng = np.random.default_rng(909)
    pile = random_pileup(rng, 160, 50, 4, max_len=30)
    S = int(pile.last.max())
    ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, [1, S // 2], [S // 2 + 4, S], P=9, B=7)
    assert_block_results_equal(ro, rg, "p9 n7")
    assert rg.min_prune_margin == ro.min_prune_margin


def test_realign_kernel_equals_the_exact_affine_dp(gpu_ctx):
    # alignment::realign (alignment.rs:7-64) on the device: one wavefront per SNP call, the 32 x 32 affine-gap DP as a systolic array over DPP
    # lane shifts.  Windows: random bases; reference = the read with substitutions, a deleted base (shifted tail) or an inserted base,
    # 1-4 candidate alleles.  Expected: the numpy Gotoh DP of floria_amd/synth_bam.py on every (window, allele) pair, first best wins.
    from floria_amd import synth_bam
    rng = np.random.default_rng(17)
    n = 6000
    B = synth_bam.BASES
    q = B[rng.integers(0, 4, size=(n, 32))]
    r = q.copy()
    kind = rng.integers(0, 4, size=n)
    for i in range(n):
        if kind[i] == 0:                                              # a few substitutions
            k = rng.integers(0, 9)
            r[i, rng.integers(0, 32, size=k)] = B[rng.integers(0, 4, size=k)]
        elif kind[i] == 1:                                            # the read lost a base: the reference has one more before the tail
            x = int(rng.integers(1, 31)); r[i, x + 1:] = q[i, x:31]; r[i, x] = B[rng.integers(0, 4)]
        elif kind[i] == 2:                                            # the read has an extra base
            x = int(rng.integers(1, 31)); r[i, x:31] = q[i, x + 1:]; r[i, 31] = B[rng.integers(0, 4)]
        else:                                                         # unrelated
            r[i] = B[rng.integers(0, 4, size=32)]
    na = rng.integers(1, 5, size=n).astype(np.uint8)
    al = np.zeros((n, 4), np.uint8)
    for i in range(n):
        al[i, :na[i]] = B[rng.permutation(4)[:na[i]]]
    best, score = gpu_ctx.realign(q, r, al, na, want_scores=True)
    exp_best = np.zeros(n, np.uint8); exp_score = np.full(n, -10 ** 9, np.int64)
    for a in range(4):
        m = na > a
        ra = r[m].copy(); ra[:, 16] = al[m, a]
        s = synth_bam.nw_affine_batch(q[m], ra).astype(np.int64)
        idx = np.nonzero(m)[0]
        better = s > exp_score[idx]
        exp_best[idx[better]] = a; exp_score[idx[better]] = s[better]
    assert np.array_equal(score.astype(np.int64), exp_score)
    assert np.array_equal(best, exp_best)
    assert len(np.unique(best)) > 2 and (score < 20).any() and (score == 32).any() is not None
    # empty batch and argument checks
    assert len(gpu_ctx.realign(np.zeros((0, 32), np.uint8), np.zeros((0, 32), np.uint8), np.zeros((0, 4), np.uint8), np.zeros(0, np.uint8))) == 0
    with pytest.raises(Exception):
        gpu_ctx.realign(q[:1], r[:1], al[:1], np.array([5], np.uint8))


def _cloned_reads_pileup(rng, n_patterns, copies, n_snps, ploidy, qual=20):
    """Many byte-identical reads: children of mirrored states then carry identical truncated histograms AND identical scores, i.e. the
    duplicate test of global_clustering.rs:123-127 fires all the time and the heap is full of exact ties."""
    hap = rng.integers(0, 2, size=(ploidy, n_snps))
    reads = []
    for _ in range(n_patterns):
        L = int(rng.integers(3, 14)); s = int(rng.integers(1, n_snps - L + 1))
        snps = np.arange(s, s + L); st = int(rng.integers(0, ploidy))
        al = hap[st, snps - 1].copy()
        flip = rng.random(L) < 0.05
        al[flip] ^= 1
        for _ in range(int(rng.integers(copies // 2, copies + 1))):
            reads.append((snps, al, np.full(L, qual)))
    return Pileup.from_reads(reads)


@pytest.mark.parametrize("seed", range(24))
def test_general_insert_path_duplicates_and_ties(gpu_ctx, hip_lib, oracle_mod, seed):
    # The duplicate suppression (`node.1 == new_block && node.0.score >= new_node.score`) compares 128-bit linear hashes of the truncated histograms
    # instead of the histograms.  The bulk-insert shortcut of the slab kernel skips it whenever a 256-slot screen sees no two children alike; "no_bulk"
    # sends EVERY step through the general path (entry table, lane-parallel duplicate test, evictions).  On pileups of cloned reads — duplicates
    # with equal scores in almost every step — both routes, with sequential and speculative stages, must reproduce the oracle's deep comparison.
    # Seeds 8-23 (round 5): a non-dyadic epsilon in BOTH arithmetics — in the running sums states with equal histograms (equal hashes) carry scores
    # that differ in the last bit, so `score >=` decides differently than in the exact form, and the two modes must each follow their oracle.
    rng = np.random.default_rng(8800 + seed)
    ploidy = 2 + seed % 3
    pile = _cloned_reads_pileup(rng, 40 + 10 * (seed % 4), 12, 60, ploidy)
    S = int(pile.last.max())
    s = np.array([1, S // 2]); e = np.array([S, S])
    P, B = 5, [10, 10, 4, 7][seed % 4]
    eps = EPS if seed < 8 else (0.04, 0.05, 0.0437, 0.011)[seed % 4]
    try:
        for mode in ((0,) if seed < 8 else (0, 1)):
            oracle_mod.set_arith_mode(mode); gpu_ctx.set_option("arith", mode)
            ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(eps, P, B), threads=4)
            for nb in (1, 0):
                gpu_ctx.set_option("no_bulk", nb)
                for spec in (0, 1):
                    gpu_ctx.set_option("speculate", spec)
                    rg = gpu_ctx.phase_blocks(pile, s, e, hip_lib.make_params(eps, P, B))
                    assert_block_results_equal(ro, rg, f"seed {seed} arith {mode} no_bulk {nb} speculate {spec}")
                    assert rg.min_prune_margin == ro.min_prune_margin
    finally:
        gpu_ctx.set_option("no_bulk", 0); gpu_ctx.set_option("speculate", -1); gpu_ctx.set_option("arith", 0); oracle_mod.set_arith_mode(0)


def test_p16_n40_takes_the_generic_kernel(gpu_ctx, hip_lib, oracle_mod):
    # `floria-hip -p 16 -n 40` — a legal command line of the reference (parse_cmd_line.rs: no upper bound on either) — needs 640 states and 10 240
    # partition slabs per job: more than the shared-slab kernels' tables hold (beam_slab_kernel 512, beam_wide_kernel 8192), so it is the one
    # case that runs beam_kernel.h (per-state slabs, LDS heap).  Same results as the oracle, also at -p 9 -n 7 (wide kernel by slab count).
    rng = np.random.default_rng(1640)
    pile = random_pileup(rng, 220, 50, 4, max_len=25, err=0.05)
    S = int(pile.last.max())
    s = np.array([1]); e = np.array([S])
    for P, B in ((16, 40), (9, 7)):
        ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(EPS, P, B), threads=4)
        rg = gpu_ctx.phase_blocks(pile, s, e, hip_lib.make_params(EPS, P, B))
        assert_block_results_equal(ro, rg, f"-p {P} -n {B}")
        assert rg.min_prune_margin == ro.min_prune_margin


@pytest.mark.parametrize("seed", range(3))
def test_sums_beyond_32_bits_carry_into_the_high_plane(gpu_ctx, hip_lib, oracle_mod, seed):
    # The slab kernel keeps a biallelic sum as a u32 low word + a u8 high byte (beam_slab_kernel.h, NARROW).  Quality 93 weighs 2^24 * (1 - 5e-10), so a
    # haplotype that has absorbed more than 256 such reads at a position wraps its low word: 1500 deep reads over 12 SNPs from two strains put ~750
    # on every (haplotype, position, allele) — sums near 2^33.6, several carries per position, in-place adds, copies and window exits included.
    rng = np.random.default_rng(4100 + seed)
    n, S = 1500, 12
    strains = rng.integers(0, 2, size=(2, S))
    reads = []
    for i in range(n):
        lo = int(rng.integers(1, 4)); hi = int(rng.integers(S - 3, S + 1))
        src = strains[i % 2]
        al = [int(src[q - 1]) ^ (1 if rng.random() < 0.03 else 0) for q in range(lo, hi + 1)]
        reads.append((list(range(lo, hi + 1)), al, [93 if rng.random() < 0.9 else 40] * (hi - lo + 1)))
    reads.sort(key=lambda r: (r[0][0], r[0][-1]))
    pile = Pileup.from_reads(reads)
    ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, [1], [S], P=3)
    assert_block_results_equal(ro, rg, f"seed {seed}")
    assert int(ro.best_ploidy[0]) == 2


def test_a_block_of_65536_reads_leaves_the_narrow_sum_kernel(gpu_ctx, hip_lib, oracle_mod):
    # 40-bit sums hold 65 535 reads of full weight; the host sends a batch whose largest block has more down the wide kernel (u64 sums).
    # Same results as the oracle either way — this pins the routing rule and the wide kernel on a long block.
    rng = np.random.default_rng(77)
    n = 65536 + 200
    reads = []
    for i in range(n):
        q = 1 + (i * 6) // n                                     # first positions ascend: reads sorted as Frag::cmp wants
        a = int(rng.integers(0, 2))
        reads.append(([q, q + 1], [a, a ^ int(rng.random() < 0.1)], [30, 30]))
    pile = Pileup.from_reads(reads)
    ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, [1], [7], P=2)
    assert_block_results_equal(ro, rg, "65536 reads")



def test_binomial_screen_error_bound_on_this_device(gpu_ctx):
    # the level-1 screen of the pruning test (beam_slab_kernel.h: binom_screen_f32) evaluates stable_binom_cdf_p_rev with the hardware reciprocal and log2 and
    # assumes |screen - table| <= 2e-5 * n (BINOM_SCREEN_C); the numpy emulation with every rcp / log2 result moved one ulp the wrong way gave <= 9.2e-6 * n.
    # Measured on the device against the host-libm table, every (n, k) with n <= 1024, at the epsilons the tests and the bench use and at the ends of the
    # range the tool accepts: well inside the bound.
    for eps in (0.03125, 0.04, 0.05, 0.0437, 0.01, 0.2):
        err = gpu_ctx.selftest(eps, 1024)
        assert 0.0 < err <= 1.0e-5, (eps, err)


def test_s2_assign_only_leaves_the_splits_to_the_host(gpu_ctx, hip_lib, oracle_mod):
    """which read separate_broken_haplogroups drops depends on the reference's FxHashSet iteration order among reads that share a
    first_position (measured: every short-read contig changes with it, scripts/a14_sensitivity.py).  With "s2_assign_only" the library stops behind the greedy
    re-insertion; a host runs the two integer steps on its own sets.  Checked: the re-inserted haplogroups equal the oracle's, and the restated split + sort
    with ascending ties on them gives exactly what the default call returns."""
    def split_and_sort(pile, parts, ranges):                      # part_block_manip.rs:27-98 + :276-288, ties in ascending counter_id
        parts, ranges = [list(p) for p in parts], list(ranges)
        new_parts, new_ranges, broken = [], [], []
        for i in range(len(ranges)):
            latest, breaks = 0, []
            for r in parts[i]:
                if latest != 0 and pile.first[r] > latest and ranges[i][0] <= latest < ranges[i][1]:
                    breaks.append(latest)
                latest = max(latest, int(pile.last[r]))
            if not breaks:
                continue
            broken.append(i)
            spot, start, end, cur = 0, ranges[i][0], breaks[0], []
            for r in parts[i]:
                if pile.last[r] <= end:
                    cur.append(r)
                else:
                    new_parts.append(cur); new_ranges.append((start, end)); cur = []
                    start = end + 1; spot += 1
                    end = breaks[spot] if spot != len(breaks) else 2 ** 32 - 1
            new_parts.append(cur); new_ranges.append((start, ranges[i][1]))
        for i in broken:
            parts[i] = []
        parts += new_parts; ranges += new_ranges
        order = sorted(range(len(parts)), key=lambda k: ranges[k])
        return [parts[k] for k in order], [ranges[k] for k in order]

    n_split = 0
    for cfg, idx, scale, bl in ((3, 0, 0.2, 500), (3, 1, 0.1, 500), (4, 0, 0.5, 10000)):
        c = synth.make_config_contig(cfg, idx, scale)
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, bl)
        r = gpu_ctx.phase_blocks(c.pileup, s, e, hip_lib.make_params(EPS))
        groups, ranges = groups_from_blocks(r, s, e)
        full = gpu_ctx.reassign(c.pileup, groups, ranges, EPS)
        gpu_ctx.set_option("s2_assign_only", 1); oracle_mod.set_a14_tie_mode(-1)
        try:
            ga = gpu_ctx.reassign(c.pileup, groups, ranges, EPS)
            go = oracle_mod.reassign(c.pileup, groups, ranges, EPS)
        finally:
            gpu_ctx.set_option("s2_assign_only", 0); oracle_mod.set_a14_tie_mode(0)
        assert ga.n_groups == go.n_groups == len(groups)
        assert np.array_equal(ga.range, go.range) and np.array_equal(ga.grp_off, go.grp_off) and np.array_equal(ga.grp_read, go.grp_read)
        assert np.array_equal(np.asarray(ga.range).reshape(-1, 2), np.asarray(ranges, np.uint32).reshape(-1, 2))
        parts, rngs = split_and_sort(c.pileup, [list(map(int, ga.group(k))) for k in range(ga.n_groups)], [tuple(map(int, ga.range[k])) for k in range(ga.n_groups)])
        assert full.n_groups == len(parts)
        for k in range(full.n_groups):
            assert tuple(map(int, full.range[k])) == rngs[k] and list(map(int, full.group(k))) == parts[k], (cfg, idx, k)
        n_split += full.n_groups - len(groups)
    assert n_split > 0


@pytest.mark.parametrize("eps", (5e-4, 0.3, 0.45))
def test_epsilon_outside_the_screens_validated_range(gpu_ctx, hip_lib, oracle_mod, eps):
    """the f32 level-1 screen of the pruning test has a measured error bound for 1e-3 <= eps <= 0.2 and n within the host-built table only; outside
    of that range every decision must take the exact path.  Results AND the pruning margin equal the oracle's."""
    for seed in range(4):
        rng = np.random.default_rng(8800 + seed)
        pile = random_pileup(rng, int(rng.integers(40, 160)), int(rng.integers(20, 70)), int(rng.integers(2, 4)), max_len=int(rng.integers(8, 40)), err=0.05)
        S = int(pile.last.max())
        s = np.asarray([1, max(1, S // 2)], np.uint32); e = np.asarray([S, S], np.uint32)
        ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(eps, 4, 10), threads=4)
        rg = gpu_ctx.phase_blocks(pile, s, e, hip_lib.make_params(eps, 4, 10))
        assert_block_results_equal(ro, rg, f"eps {eps} seed {seed}")
        assert ro.min_prune_margin == rg.min_prune_margin


def test_reads_longer_than_the_binomial_table(gpu_ctx, hip_lib, oracle_mod):
    """... and a read of more cells than the table's n (1024): the screen is not trusted there either (and the p-value comes from the device formula)."""
    rng = np.random.default_rng(8900)
    pile = random_pileup(rng, 50, 2600, 2, max_len=1500, err=0.04, drop=0.05)
    assert int(np.diff(pile.read_off).max()) > 1100
    S = int(pile.last.max())
    s, e = np.asarray([1], np.uint32), np.asarray([S], np.uint32)
    ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(EPS, 3, 6), threads=1)
    rg = gpu_ctx.phase_blocks(pile, s, e, hip_lib.make_params(EPS, 3, 6))
    assert_block_results_equal(ro, rg, "long reads")


@pytest.mark.parametrize("mode", (0, 1))
def test_reads_of_exactly_256_cells_of_weight_one(gpu_ctx, hip_lib, oracle_mod, mode):
    """w(q) is exactly 1.0 = 2^24 Q24 units from q = 73 on, so a read of exactly 256 cells (one LDS tile) that all agree with a slab sums to 2^32:
    the per-slab `same` must not be accumulated in 32 bits.  Reads of 255 / 256 / 257 cells at q = 93, both arithmetics."""
    rng = np.random.default_rng(256)
    hap = rng.integers(0, 2, size=(2, 300))
    reads = []
    for r in range(14):
        L = (256, 256, 255, 257, 256, 200, 256)[r % 7]
        snps = np.arange(1, L + 1)
        al = hap[r % 2, :L].copy()
        if r >= 10:
            al[rng.integers(0, L, size=3)] ^= 1
        reads.append((snps, al, np.full(L, 93)))
    pile = Pileup.from_reads(reads)
    gpu_ctx.set_option("arith", mode); oracle_mod.set_arith_mode(mode)
    try:
        for eps in (EPS, 0.04):
            ro, rg = both(gpu_ctx, hip_lib, oracle_mod, pile, [1, 100], [257, 256], eps=eps, P=3, B=10)
            assert_block_results_equal(ro, rg, f"mode {mode} eps {eps}")
            assert ro.min_prune_margin == rg.min_prune_margin
    finally:
        gpu_ctx.set_option("arith", 0); oracle_mod.set_arith_mode(0)
