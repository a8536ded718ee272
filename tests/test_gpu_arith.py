"""`-m gpu`: the opt-in reference-arithmetic mode of libfloria_hip.so (`set_option("arith", 1)`) against the oracle's
arithmetic mode 1 — the reference's running f64 sums in the iteration orders of its hash containers — bit for bit, at the
non-dyadic epsilons where that form and the canonical one part ways; and against the canonical mode at a dyadic epsilon, where they must not.
"""
import numpy as np
import pytest

from floria_amd import synth
from tests.helpers import assert_block_results_equal, random_pileup

pytestmark = pytest.mark.gpu
NON_DYADIC = (0.04, 0.05, 0.0437)


@pytest.fixture()
def arith(gpu_ctx, oracle_mod):
    gpu_ctx.set_option("arith", 1)
    oracle_mod.set_arith_mode(1)
    yield gpu_ctx
    oracle_mod.set_arith_mode(0)
    gpu_ctx.set_option("arith", 0)


def both(ctx, hip_lib, oracle_mod, pile, s, e, eps, P=5, B=10, sens=2, stop=1):
    ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(eps, P, B, sens, stop), threads=8)
    rg = ctx.phase_blocks(pile, s, e, hip_lib.make_params(eps, P, B, sens, stop))
    return ro, rg


@pytest.mark.parametrize("seed", range(18))
def test_random_small_pileups_in_reference_arithmetic(arith, hip_lib, oracle_mod, seed):
    rng = np.random.default_rng(7000 + seed)
    alleles = 4 if seed % 3 == 2 else 2
    pile = random_pileup(rng, int(rng.integers(5, 150)), int(rng.integers(4, 60)), int(rng.integers(1, 5)), max_len=int(rng.integers(2, 40)),
                         alleles=alleles, q0_frac=0.1 if seed % 4 == 1 else 0.0, err=float(rng.choice([0.0, 0.05, 0.2])))
    S = int(pile.last.max())
    nb = int(rng.integers(1, 6))
    s = np.sort(rng.integers(1, S + 1, size=nb))
    e = np.minimum(S, s + rng.integers(0, 25, size=nb))
    eps = NON_DYADIC[seed % 3]
    ro, rg = both(arith, hip_lib, oracle_mod, pile, s, e, eps, P=int(rng.integers(1, 7)), B=int(rng.integers(1, 13)),
                  sens=int(rng.integers(1, 4)), stop=int(rng.integers(0, 2)))
    assert_block_results_equal(ro, rg, f"seed {seed} eps {eps}")
    assert ro.min_prune_margin == rg.min_prune_margin


@pytest.mark.parametrize("eps", NON_DYADIC)
@pytest.mark.parametrize("cfg,scale", ((2, 0.05), (3, 0.1), (4, 1.0)))
def test_config_slices_in_reference_arithmetic(arith, hip_lib, oracle_mod, cfg, scale, eps):
    """The slices scripts/arith_sensitivity.py measures (where ~60 % of the blocks differ between the two arithmetics)."""
    C = synth.CONFIGS[cfg]
    for idx in range(2):
        c = synth.make_config_contig(cfg, idx, scale)
        s, e = oracle_mod.block_ranges(c.snp_pos, C["block_length"])
        ro, rg = both(arith, hip_lib, oracle_mod, c.pileup, s, e, eps, P=C["max_ploidy"], B=C["beam"])
        assert_block_results_equal(ro, rg, f"config {cfg} contig {idx} eps {eps}")
        assert ro.min_prune_margin == rg.min_prune_margin


def test_the_mode_is_not_the_canonical_form_at_a_non_dyadic_epsilon(gpu_ctx, hip_lib, oracle_mod):
    """... otherwise the tests above would prove nothing: at eps = 0.04 the two arithmetics give different results on some block of this slice,
    on the device exactly as in the oracle."""
    C = synth.CONFIGS[4]
    c = synth.make_config_contig(4, 0, 1.0)
    s, e = oracle_mod.block_ranges(c.snp_pos, C["block_length"])
    par = hip_lib.make_params(0.04, C["max_ploidy"], C["beam"])
    r0 = gpu_ctx.phase_blocks(c.pileup, s, e, par)
    gpu_ctx.set_option("arith", 1)
    try:
        r1 = gpu_ctx.phase_blocks(c.pileup, s, e, par)
    finally:
        gpu_ctx.set_option("arith", 0)
    assert not (np.array_equal(r0.part, r1.part) and np.array_equal(r0.mec.view(np.uint64), r1.mec.view(np.uint64)))


@pytest.mark.parametrize("eps", (0.03125, 0.0625))
def test_dyadic_epsilon_both_arithmetics_agree_on_the_device(gpu_ctx, hip_lib, eps):
    C = synth.CONFIGS[4]
    c = synth.make_config_contig(4, 1, 1.0)
    from floria_amd import lib
    s, e = lib.get_range_with_lengths(c.snp_pos, C["block_length"])
    par = hip_lib.make_params(eps, C["max_ploidy"], C["beam"])
    r0 = gpu_ctx.phase_blocks(c.pileup, s, e, par)
    gpu_ctx.set_option("arith", 1)
    try:
        r1 = gpu_ctx.phase_blocks(c.pileup, s, e, par)
    finally:
        gpu_ctx.set_option("arith", 0)
    assert_block_results_equal(r0, r1, f"eps {eps}")
    assert r0.min_prune_margin == r1.min_prune_margin


def test_batch_of_contigs_and_the_pipelined_entry_point(arith, hip_lib, oracle_mod):
    """Several contigs in one call (the cell orders are laid out per contig) and the host-pileup entry point (which must not pipeline in this mode)."""
    C = synth.CONFIGS[4]
    piles, bc, bs, be = [], [], [], []
    for idx in range(3):
        c = synth.make_config_contig(4, idx, 1.0)
        s, e = oracle_mod.block_ranges(c.snp_pos, C["block_length"])
        piles.append(c.pileup); bc += [idx] * len(s); bs += list(s); be += list(e)
    par = hip_lib.make_params(0.04, C["max_ploidy"], C["beam"])
    rg = arith.phase_pileups_batch(piles, np.asarray(bc, np.uint32), np.asarray(bs, np.uint32), np.asarray(be, np.uint32), par)
    off = 0
    for idx in range(3):
        n = bc.count(idx)
        ro = oracle_mod.phase_blocks(piles[idx], np.asarray(bs[off:off + n]), np.asarray(be[off:off + n]), oracle_mod.make_params(0.04, C["max_ploidy"], C["beam"]), threads=8)
        for b in range(n):
            ids_o, part_o = ro.block(b)
            ids_g, part_g = rg.block(off + b)
            assert np.array_equal(ids_o, ids_g) and np.array_equal(part_o, part_g), f"contig {idx} block {b}"
            assert ro.best_ploidy[b] == rg.best_ploidy[off + b]
            assert np.array_equal(ro.mec[b].view(np.uint64), rg.mec[off + b].view(np.uint64))
        off += n


@pytest.mark.parametrize("eps", NON_DYADIC)
def test_final_reassignment_in_reference_arithmetic(arith, hip_lib, oracle_mod, eps):
    """S2 (process_reads_for_final_parts): the candidates' keys (diff + 1., id, same) from running sums in set order, ascending and caller-given visiting orders."""
    from tests.test_gpu_parity import groups_from_blocks
    for cfg, idx, scale, bl in ((4, 0, 0.5, 10000), (3, 0, 0.2, 500)):
        c = synth.make_config_contig(cfg, idx, scale)
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, bl)
        r = arith.phase_blocks(c.pileup, s, e, hip_lib.make_params(eps))
        groups, ranges = groups_from_blocks(r, s, e)           # overlapping blocks -> reads sit in several haplogroups
        members = np.unique(np.concatenate(groups))
        for order in (None, members[::-1].copy(), np.random.default_rng(5).permutation(members).astype(np.uint32)):
            go = oracle_mod.reassign(c.pileup, groups, ranges, eps, read_order=order)
            gg = arith.reassign(c.pileup, groups, ranges, eps, read_order=order)
            assert go.n_groups == gg.n_groups
            assert np.array_equal(go.range, gg.range) and np.array_equal(go.grp_off, gg.grp_off) and np.array_equal(go.grp_read, gg.grp_read)


def test_config4_64_contigs_in_reference_arithmetic(arith, hip_lib, oracle_mod):
    """A batch the size of a small shard (64 contigs, ~460 blocks, every ploidy up to 5) at the BASELINE second pass's epsilon."""
    C = synth.CONFIGS[4]
    contigs = [synth.make_config_contig(4, i, 1.0) for i in range(64)]
    handles = arith.upload_batch([c.pileup for c in contigs])
    bc, bs, be, per = [], [], [], []
    for i, c in enumerate(contigs):
        s, e = oracle_mod.block_ranges(c.snp_pos, C["block_length"])
        per.append((s, e)); bc += [i] * len(s); bs += list(s); be += list(e)
    rg = arith.phase_blocks_batch(handles, np.asarray(bc, np.uint32), np.asarray(bs, np.uint32), np.asarray(be, np.uint32), hip_lib.make_params(0.04, C["max_ploidy"], C["beam"]))
    off = 0
    for i, c in enumerate(contigs):
        s, e = per[i]
        ro = oracle_mod.phase_blocks(c.pileup, s, e, oracle_mod.make_params(0.04, C["max_ploidy"], C["beam"]), threads=8)
        for b in range(len(s)):
            assert ro.best_ploidy[b] == rg.best_ploidy[off + b], f"contig {i} block {b}"
            assert np.array_equal(ro.block(b)[1], rg.block(off + b)[1]), f"contig {i} block {b}"
            assert np.array_equal(ro.mec[b].view(np.uint64), rg.mec[off + b].view(np.uint64)), f"contig {i} block {b}"
        off += len(s)
    for h in handles:
        h.free()


@pytest.mark.parametrize("knob,value", [("arith_hbm", 1), ("arith_replay", 1), ("fx_tags", 128), ("opt_global", 1), ("speculate", 0), ("speculate", 1), ("speculate", 2), ("opt_threads", 512), ("opt_threads", 1024), ("slots", 96), ("groups", 2)])
def test_launch_knobs_do_not_change_reference_arithmetic_results(arith, hip_lib, oracle_mod, knob, value):
    """Histogram in HBM instead of LDS, every stage plan, other workgroup sizes, a small persistent grid: the same bits (the oracle comparison of the
    default plan is test_config_slices_in_reference_arithmetic)."""
    piles, bc, bs, be = [], [], [], []
    for i, (cfg, scale) in enumerate(((4, 1.0), (3, 0.1), (4, 1.0), (2, 0.05))):
        c = synth.make_config_contig(cfg, i, scale)
        s, e = oracle_mod.block_ranges(c.snp_pos, synth.CONFIGS[cfg]["block_length"])
        piles.append(c.pileup); bc += [i] * len(s); bs += list(s); be += list(e)
    hs = arith.upload_batch(piles)
    par = hip_lib.make_params(0.0437, 5, 10)
    args = (hs, np.asarray(bc, np.uint32), np.asarray(bs, np.uint32), np.asarray(be, np.uint32), par)
    r0 = arith.phase_blocks_batch(*args)
    arith.set_option(knob, value)
    try:
        r1 = arith.phase_blocks_batch(*args)
    finally:
        arith.set_option(knob, {"speculate": -1}.get(knob, 0))
    assert_block_results_equal(r0, r1, f"{knob}={value}")
    for h in hs:
        h.free()


@pytest.mark.parametrize("seed", range(3))
def test_long_reads_in_reference_arithmetic(arith, hip_lib, oracle_mod, seed):
    """Reads of several hundred cells: more than one LDS tile of the generic beam kernel per read (the running sum continues across tiles), position maps that
    grow through many resizes, first-insertion keys with large cell ranks."""
    rng = np.random.default_rng(9100 + seed)
    pile = random_pileup(rng, 60 + 20 * seed, 900, 3, max_len=700, alleles=4 if seed == 2 else 2, err=0.05, drop=0.15)
    S = int(pile.last.max())
    s = np.asarray([1, S // 3, S // 2], np.uint32)
    e = np.asarray([S, min(S, S // 3 + 400), S], np.uint32)
    eps = NON_DYADIC[seed % 3]
    ro, rg = both(arith, hip_lib, oracle_mod, pile, s, e, eps, P=4, B=6)
    assert int(np.diff(pile.read_off).max()) > 256
    assert_block_results_equal(ro, rg, f"seed {seed} eps {eps}")
    assert ro.min_prune_margin == rg.min_prune_margin


@pytest.mark.parametrize("seed", range(6))
def test_sparse_partitions_whose_positions_span_more_than_their_map_has_buckets(arith, hip_lib, oracle_mod, seed):
    """The optimise kernel lists a position map's iteration order straight from the histogram when the partition's positions span fewer than the map has
    buckets (every key then sits in its home bucket, optimize_kernel.h) and replays the insertions otherwise.  Linked-read-like fragments - a handful of
    cells scattered over several hundred positions - make partitions of ~100 keys over ~800 positions (128 or 256 buckets): the replay, with real collisions;
    the same pileups with the rule switched off must give the same bits, and both the oracle's."""
    from floria_amd.pileup import Pileup
    rng = np.random.default_rng(5300 + seed)
    S = 500 + 100 * seed
    hap = rng.integers(0, 2, size=(3, S))
    reads = []
    for r in range(18 + 4 * seed):
        snps = np.sort(rng.choice(np.arange(1, S + 1), size=int(rng.integers(4, 14)), replace=False))
        al = hap[r % 3, snps - 1].copy()
        flip = rng.random(len(snps)) < 0.06
        al[flip] ^= 1
        reads.append((snps, al, rng.integers(8, 40, size=len(snps))))
    pile = Pileup.from_reads(reads)
    s, e = np.asarray([1, S // 4], np.uint32), np.asarray([S, S], np.uint32)
    eps = NON_DYADIC[seed % 3]
    ro, rg = both(arith, hip_lib, oracle_mod, pile, s, e, eps, P=4, B=6)
    assert_block_results_equal(ro, rg, f"seed {seed} eps {eps}")
    arith.set_option("arith_replay", 1)
    try:
        _, rr = both(arith, hip_lib, oracle_mod, pile, s, e, eps, P=4, B=6)
    finally:
        arith.set_option("arith_replay", 0)
    assert_block_results_equal(ro, rr, f"seed {seed} eps {eps}, every map replayed")


@pytest.mark.parametrize("full", (3, 7, 14, 28, 56, 112))
def test_a_position_map_that_is_full_when_a_position_is_touched_again(arith, hip_lib, oracle_mod, full):
    """a partition's position map is filled through `entry(pos).or_insert(..)` (utils_frags.rs:165): looked up first, room reserved only for a
    new position.  A block with exactly 3 / 7 / 14 / 28 / 56 / 112 positions — hashbrown's capacities — fills the map to the brim with its first read, and
    every later read touches positions that are there: the table must keep its bucket count (tests/test_order_emulation.py pins the oracle's side of it)."""
    from floria_amd.pileup import Pileup
    rng = np.random.default_rng(4200 + full)
    hap = rng.integers(0, 2, size=(2, full))
    reads = []
    for r in range(24):
        snps = np.arange(1, full + 1) if r < 2 else np.sort(rng.choice(np.arange(1, full + 1), size=int(rng.integers(1, full + 1)), replace=False))
        al = hap[r % 2, snps - 1].copy()
        flip = rng.random(len(snps)) < 0.08
        al[flip] ^= 1
        reads.append((snps, al, rng.integers(8, 40, size=len(snps))))
    pile = Pileup.from_reads(reads)
    s, e = np.asarray([1], np.uint32), np.asarray([full], np.uint32)
    for eps in NON_DYADIC:
        ro, rg = both(arith, hip_lib, oracle_mod, pile, s, e, eps, P=3, B=4)
        assert_block_results_equal(ro, rg, f"{full} positions, eps {eps}")


def test_blocks_spanning_more_than_1024_positions_in_reference_arithmetic(arith, hip_lib, oracle_mod):
    """Beyond 1024 positions per block the first-insertion order comes from the workgroup-wide sort instead of the per-wavefront counting sort, the position
    maps grow to 2048 buckets, and reads span several LDS tiles."""
    rng = np.random.default_rng(9300)
    pile = random_pileup(rng, 90, 1500, 3, max_len=1300, err=0.05, drop=0.1)
    S = int(pile.last.max())
    s, e = np.asarray([1, 200], np.uint32), np.asarray([S, min(S, 900)], np.uint32)
    ro, rg = both(arith, hip_lib, oracle_mod, pile, s, e, 0.04, P=3, B=5)
    assert int(pile.last.max() - pile.first.min()) + 1 > 1024
    assert_block_results_equal(ro, rg, "wide block")
    assert ro.min_prune_margin == rg.min_prune_margin


@pytest.mark.parametrize("eps", (0.04, 0.0437))
def test_config5_slice_wide_beam_in_reference_arithmetic(arith, hip_lib, oracle_mod, eps):
    """VERDICT r5 #3a: BASELINE config 5's shape (-p 8 -n 40: up to 320 states per job, the wide-beam path) in the reference's arithmetic against the
    oracle's mode 1 — a 5 % slice of the contig (8 strains, ~200 x), every block."""
    C = synth.CONFIGS[5]
    c = synth.make_config_contig(5, 0, 0.05)
    s, e = oracle_mod.block_ranges(c.snp_pos, C["block_length"])
    assert C["max_ploidy"] == 8 and C["beam"] == 40 and len(s) >= 20
    ro, rg = both(arith, hip_lib, oracle_mod, c.pileup, s, e, eps, P=C["max_ploidy"], B=C["beam"])
    assert_block_results_equal(ro, rg, f"config 5 slice eps {eps}")
    assert ro.min_prune_margin == rg.min_prune_margin


@pytest.mark.parametrize("seed", range(4))
def test_wide_beams_on_random_pileups_in_reference_arithmetic(arith, hip_lib, oracle_mod, seed):
    """-p up to 8 and -n up to 40 on random pileups (the fuzz of rounds 4-5 drew -p <= 7, -n <= 12)."""
    rng = np.random.default_rng(8800 + seed)
    pile = random_pileup(rng, int(rng.integers(60, 200)), int(rng.integers(20, 80)), int(rng.integers(3, 9)), max_len=int(rng.integers(8, 40)),
                         alleles=4 if seed == 3 else 2, q0_frac=0.1 if seed == 2 else 0.0, err=0.1)
    S = int(pile.last.max())
    s = np.asarray([1, max(1, S // 3)], np.uint32)
    e = np.asarray([S, min(S, S // 3 + 30)], np.uint32)
    P, B = ((8, 40), (8, 13), (6, 40), (7, 25))[seed]
    ro, rg = both(arith, hip_lib, oracle_mod, pile, s, e, NON_DYADIC[seed % 3], P=P, B=B)
    assert_block_results_equal(ro, rg, f"seed {seed} P {P} B {B}")
    assert ro.min_prune_margin == rg.min_prune_margin


def test_config4_full_size_in_reference_arithmetic(arith, hip_lib, oracle_mod):
    """VERDICT r5 #3b: the headline workload at full size (2000 contigs, ~14.5k blocks) in the mode the CLI runs at its default epsilon, through the call
    bench.py's second pass times: properties on every block, 128 random blocks against the oracle's mode 1 bit for bit."""
    from tests import test_gpu_fullsize as F
    eps = 0.04
    C = synth.CONFIGS[4]
    contigs = [synth.make_config_contig(4, i) for i in range(2000)]
    bc, bs, be = [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, C["block_length"])
        bc += [i] * len(s); bs += list(s); be += list(e)
    bc, bs, be = np.array(bc), np.array(bs), np.array(be)
    arena, parr, _ = hip_lib.pack_pileups([c.pileup for c in contigs])
    r = arith.phase_pileups_batch(parr, bc, bs, be, hip_lib.make_params(eps, C["max_ploidy"], C["beam"]))
    arena.free()
    assert 14000 < r.n_blocks < 15000 and r.min_prune_margin > 1e-9
    ne = np.diff(r.read_off.astype(np.int64)) > 0
    assert np.all(r.best_ploidy[ne] >= 1) and np.all(r.best_ploidy <= C["max_ploidy"]) and np.all(r.ploidies_tried >= r.best_ploidy)
    for p in range(C["max_ploidy"]):
        assert np.all(r.mec[r.ploidies_tried <= p, p] == 0.0)
    for blk in range(0, r.n_blocks, 97):
        ids, part = r.block(blk)
        assert np.array_equal(ids, F.reads_in_interval(contigs[bc[blk]].pileup, bs[blk], be[blk]))
        if len(ids):
            assert part.max() < r.best_ploidy[blk]
    rng = np.random.default_rng(4404)
    par = oracle_mod.make_params(eps, C["max_ploidy"], C["beam"])
    by_contig = {}
    for blk in rng.choice(r.n_blocks, size=128, replace=False):
        by_contig.setdefault(int(bc[blk]), []).append(int(blk))
    for ci, blks in by_contig.items():
        ro = oracle_mod.phase_blocks(contigs[ci].pileup, bs[blks], be[blks], par, threads=16)
        for k, blk in enumerate(blks):
            ids, part = r.block(blk)
            oid, opart = ro.block(k)
            assert ro.best_ploidy[k] == r.best_ploidy[blk] and ro.ploidies_tried[k] == r.ploidies_tried[blk], f"block {blk}"
            assert np.array_equal(oid, ids) and np.array_equal(opart, part), f"block {blk}"
            assert np.array_equal(ro.mec[k].view(np.uint64), r.mec[blk].view(np.uint64)), f"block {blk}"


@pytest.mark.parametrize("eps", (0.04, 0.0437))
def test_paired_fragments_with_host_given_set_orders(arith, hip_lib, oracle_mod, tmp_path, eps):
    """BASELINE config 3's shape (paired short reads) at a non-dyadic epsilon with the iteration order of every fragment's position set GIVEN
    (floria_pileup::set_order): the first mate's set extended by the second's, as combine_frags builds it (file_reader.rs:539-541).  S1 and S2 against the oracle's
    mode 1 with the same orders, through the resident, the CSR host-pileup and the packed host-pileup entry points; a set_order that is not a permutation is refused."""
    from floria_amd import synth_bam
    from tests.test_gpu_parity import groups_from_blocks
    for idx in (2, 5):
        c = synth.make_config_contig(3, idx, 0.2, keep_layout=True)
        ex = synth_bam.write_dataset(str(tmp_path / f"d{idx}"), [c], seed=3)[c.name]
        pile = ex["pileup"]
        pile.set_order = np.concatenate([oracle_mod.set_order_of(pile.read(i)[0], [np.asarray(x, np.uint32) for x in ex["segments"][i]]) for i in range(pile.n_reads)])
        s, e = hip_lib.get_range_with_lengths(ex["snp_pos0"], 500)
        ro, rg = both(arith, hip_lib, oracle_mod, pile, s, e, eps)
        assert_block_results_equal(ro, rg, f"contig {idx} eps {eps}")
        assert ro.min_prune_margin == rg.min_prune_margin
        bc = np.zeros(len(s), np.uint32)
        par = hip_lib.make_params(eps)
        arena, pinned = hip_lib.pin_pileups([pile])
        rp = arith.phase_pileups_batch(pinned, bc, s, e, par)
        arena.free()
        assert_block_results_equal(ro, rp, "CSR host pileups")
        arena, parr, _ = hip_lib.pack_pileups([pile])
        rk = arith.phase_pileups_batch(parr, bc, s, e, par)
        arena.free()
        assert_block_results_equal(ro, rk, "packed host pileups")
        groups, ranges = groups_from_blocks(rg, s, e)
        go = oracle_mod.reassign(pile, groups, ranges, eps)
        gg = arith.reassign(pile, groups, ranges, eps)
        assert go.n_groups == gg.n_groups and np.array_equal(go.range, gg.range) and np.array_equal(go.grp_off, gg.grp_off) and np.array_equal(go.grp_read, gg.grp_read)
    bad = pile.set_order.copy()
    two = int(np.nonzero(np.diff(pile.read_off) >= 2)[0][0])
    bad[int(pile.read_off[two])] = bad[int(pile.read_off[two]) + 1]
    pile.set_order = bad
    with pytest.raises(hip_lib.FloriaHipError, match="set_order of read"):
        arith.phase_blocks(pile, s, e, hip_lib.make_params(eps))
    pile.set_order = None
    arith.phase_blocks(pile, s, e, hip_lib.make_params(eps))          # (and the context is usable afterwards)


@pytest.mark.parametrize("wire", ("packed", "csr"))
def test_pipelined_upload_in_reference_arithmetic(arith, hip_lib, oracle_mod, wire):
    """VERDICT r5 #3d: the host-pileup entry points pipeline in this mode too - every chunk's cell orders are computed behind its flatten launch, on the chunk's
    stream, and the chunk's job group starts behind them.  Twelve config-4 contigs and two paired config-3 contigs (with host-given set orders) from pinned memory
    in 3 chunks against the oracle's mode 1, and against the same call in one piece."""
    C = synth.CONFIGS[4]
    piles, bc, bs, be = [], [], [], []
    for idx in range(12):
        c = synth.make_config_contig(4, 100 + idx, 1.0)
        s, e = oracle_mod.block_ranges(c.snp_pos, C["block_length"])
        piles.append(c.pileup); bc += [idx] * len(s); bs += list(s); be += list(e)
    bc, bs, be = np.asarray(bc, np.uint32), np.asarray(bs, np.uint32), np.asarray(be, np.uint32)
    par = hip_lib.make_params(0.04, C["max_ploidy"], C["beam"])
    if wire == "packed":
        arena, src, _ = hip_lib.pack_pileups(piles)
    else:
        arena, src = hip_lib.pin_pileups(piles)
    one = arith.phase_pileups_batch(src, bc, bs, be, par)
    arith.set_option("upload_chunks", 3)
    try:
        rg = arith.phase_pileups_batch(src, bc, bs, be, par)
        assert arith.timing()["upload_chunks"] == 3
    finally:
        arith.set_option("upload_chunks", 0)
    arena.free()
    assert_block_results_equal(one, rg, f"{wire}: 3 chunks against one piece")
    assert one.min_prune_margin == rg.min_prune_margin
    off = 0
    for idx in (0, 5, 11):
        sel = np.nonzero(bc == idx)[0]
        ro = oracle_mod.phase_blocks(piles[idx], bs[sel], be[sel], oracle_mod.make_params(0.04, C["max_ploidy"], C["beam"]), threads=8)
        for k, b in enumerate(sel):
            assert ro.best_ploidy[k] == rg.best_ploidy[b] and np.array_equal(ro.block(k)[1], rg.block(int(b))[1]), f"contig {idx} block {k}"
            assert np.array_equal(ro.mec[k].view(np.uint64), rg.mec[b].view(np.uint64))


def test_pipelined_upload_with_host_given_set_orders(arith, hip_lib, oracle_mod, tmp_path):
    from floria_amd import synth_bam
    piles, bc, bs, be, per = [], [], [], [], []
    for k, idx in enumerate((2, 5, 7)):
        c = synth.make_config_contig(3, idx, 0.2, keep_layout=True)
        ex = synth_bam.write_dataset(str(tmp_path / f"q{idx}"), [c], seed=3)[c.name]
        pile = ex["pileup"]
        if k != 1:              # (the middle contig carries none: its sets are emulated on the device, the others' are gathered)
            pile.set_order = np.concatenate([oracle_mod.set_order_of(pile.read(i)[0], [np.asarray(x, np.uint32) for x in ex["segments"][i]]) for i in range(pile.n_reads)])
        s, e = hip_lib.get_range_with_lengths(ex["snp_pos0"], 500)
        piles.append(pile); per.append((s, e)); bc += [k] * len(s); bs += list(s); be += list(e)
    bc, bs, be = np.asarray(bc, np.uint32), np.asarray(bs, np.uint32), np.asarray(be, np.uint32)
    par = hip_lib.make_params(0.0437)
    arena, pinned = hip_lib.pin_pileups(piles)
    arith.set_option("upload_chunks", 3)
    try:
        rg = arith.phase_pileups_batch(pinned, bc, bs, be, par)
        assert arith.timing()["upload_chunks"] == 3
        # a set order that is not a permutation is refused on this route as well
        keep = pinned[2].set_order.copy()
        two = int(np.nonzero(np.diff(pinned[2].read_off) >= 2)[0][0])
        pinned[2].set_order[int(pinned[2].read_off[two])] = pinned[2].set_order[int(pinned[2].read_off[two]) + 1]
        with pytest.raises(hip_lib.FloriaHipError, match="set_order of read .*contig 2"):
            arith.phase_pileups_batch(pinned, bc, bs, be, par)
        pinned[2].set_order[:] = keep
        again = arith.phase_pileups_batch(pinned, bc, bs, be, par)
    finally:
        arith.set_option("upload_chunks", 0)
    arena.free()
    assert_block_results_equal(rg, again, "after a refused call")
    for k in range(3):
        sel = np.nonzero(bc == k)[0]
        ro = oracle_mod.phase_blocks(piles[k], per[k][0], per[k][1], oracle_mod.make_params(0.0437), threads=8)
        for j, b in enumerate(sel):
            assert ro.best_ploidy[j] == rg.best_ploidy[b] and np.array_equal(ro.block(j)[1], rg.block(int(b))[1]), f"contig {k} block {j}"
            assert np.array_equal(ro.mec[j].view(np.uint64), rg.mec[b].view(np.uint64))


@pytest.mark.parametrize("seed", range(6))
def test_every_beam_path_in_reference_arithmetic(arith, hip_lib, oracle_mod, seed):
    """The three beam kernels carry the reference's running sums (generic: one lane per (state, partition); slab: terms folded per live slab; wide, round 6: one lane
    per live slab walks the cells): the same bits from each, and the oracle's - narrow and wide beams, 2 and 4 alleles, q = 0 cells, reads of several LDS tiles."""
    rng = np.random.default_rng(6600 + seed)
    long_reads = seed in (1, 4)
    pile = random_pileup(rng, 50 if long_reads else int(rng.integers(40, 160)), 700 if long_reads else int(rng.integers(20, 90)), int(rng.integers(2, 7)),
                         max_len=600 if long_reads else int(rng.integers(6, 40)), alleles=4 if seed % 3 == 2 else 2, q0_frac=0.1 if seed % 2 else 0.0, err=0.08, drop=0.05)
    S = int(pile.last.max())
    s = np.asarray([1, max(1, S // 4)], np.uint32)
    e = np.asarray([S, min(S, S // 4 + (400 if long_reads else 30))], np.uint32)
    P, B = ((4, 8), (3, 30), (8, 12), (6, 5), (5, 20), (7, 9))[seed]
    eps = NON_DYADIC[seed % 3]
    ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(eps, P, B), threads=8)
    for path in (1, 2, 3):
        arith.set_option("beam_path", path)
        try:
            rg = arith.phase_blocks(pile, s, e, hip_lib.make_params(eps, P, B))
        finally:
            arith.set_option("beam_path", 0)
        assert_block_results_equal(ro, rg, f"seed {seed} path {path} P {P} B {B}")
        assert ro.min_prune_margin == rg.min_prune_margin
