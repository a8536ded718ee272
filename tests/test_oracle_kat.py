"""Known-answer tests that pin the CPU oracle (oracle/floria_oracle.cpp).

The reference ships no tests or golden vectors for this path and cannot be built here,
so these KATs are derived BY HAND from the reference source (citations inline).  `-m "not gpu"`.
"""
import numpy as np
import pytest

from floria_amd.pileup import Pileup

EPS = 0.03125   # dyadic: every f64 sum on the path is exact


def test_weights_are_q24_and_match_f32_formula(oracle_mod):
    # utils_frags.rs:702-711: prob = 1f32 - 10f32.powf(q as f32 / -10.)
    w = oracle_mod.weight_q24()
    # numpy's own float32 pow is 1 ulp off for some q (e.g. q=2), so the expectation is the correctly rounded
    # powf: evaluate 10^x in float64 on the float32 argument and round once to float32 (what glibc powf returns)
    x = np.arange(256, dtype=np.float32) / np.float32(-10.0)
    pw = np.power(10.0, x.astype(np.float64)).astype(np.float32)
    expect = (np.float32(1.0) - pw).astype(np.float64) * 2.0 ** 24
    assert np.array_equal(expect, np.floor(expect)), "w(q) must be a multiple of 2^-24"
    assert np.array_equal(w.astype(np.float64), expect)
    assert w[0] == 0 and w[255] == 2 ** 24 and w[20] == int(round(0.99 * 2 ** 24))


def test_binary_heap_matches_hand_traces(oracle_mod):
    # std BinaryHeap, traced by hand:
    # two tied pushes keep insertion order in the array; into_sorted_vec swaps them
    h, s = oracle_mod.heap_trace([1.0, 1.0], 10)
    assert list(h) == [0, 1] and list(s) == [1, 0]
    # five ties with capacity 3: push0..2 -> [0,1,2]; push3 -> [0,1,2,3]; pop: last(3)->root, old root 0 evicted,
    # sift_down_to_bottom picks the RIGHT child on ties (`<=`) -> [2,1,3]; push4, pop evicts 2 -> [3,1,4]
    h, s = oracle_mod.heap_trace([2.0] * 5, 3)
    assert list(h) == [3, 1, 4] and list(s) == [1, 4, 3]
    # distinct scores: the 4 smallest survive, worst popped each time; sorted ascending
    h, s = oracle_mod.heap_trace([3., 1., 2., 1., 3., 2., 0.5], 4)
    assert sorted(h) == [1, 3, 5, 6] or sorted(h) == [1, 2, 3, 6]
    assert [float([3., 1., 2., 1., 3., 2., 0.5][i]) for i in s] == sorted([[3., 1., 2., 1., 3., 2., 0.5][i] for i in h])


def test_stable_binom_cdf_p_rev_values(oracle_mod):
    # utils_frags.rs:211-248 evaluated by hand with numpy (same libm)
    def ref(n, k, p, div=0.25):
        if n == 0:
            return 0.0
        a = k / n
        if a == 1.0:
            a = 0.9999999
        if a == 0.0:
            a = 0.0000001
        rel = a * np.log(a / p) + (1.0 - a) * np.log((1.0 - a) / (1.0 - p))
        if a < p:
            rel = -rel
        return -1.0 * n / div * rel
    for n, k in [(0, 0), (3, 3), (10, 1), (10, 0), (100, 3), (100, 4), (7, 7), (500, 250)]:
        assert oracle_mod.binom(n, k, EPS) == pytest.approx(ref(n, k, EPS), rel=1e-13, abs=1e-13)
    assert oracle_mod.binom(3, 3, EPS) == pytest.approx(-41.5888, abs=1e-3)      # SURVEY.md Appendix E value


def kat1_pileup():
    # 6 reads x 4 SNPs, read i carries allele i%2 everywhere, q=20
    return Pileup.from_reads([([1, 2, 3, 4], [i % 2] * 4, [20] * 4) for i in range(6)])


def test_kat1_two_clean_strains(oracle_mod):
    r = oracle_mod.phase_blocks(kat1_pileup(), [1], [4], oracle_mod.make_params(EPS, 5, 10, 2, 1))
    # ploidy 1: 4 sites x 3 minority reads = 12 (unit weights); expected = 24*eps = 0.75 -> continue
    # ploidy 2: two mirrored lineages tie at MEC 0.25; into_sorted_vec on the tied 2-heap picks the SECOND
    # lineage -> partition[0] = {1,3,5}, partition[1] = {0,2,4}; mec 0 < 0.75 -> best_ploidy = 2
    assert list(r.best_ploidy) == [2] and list(r.ploidies_tried) == [2]
    assert list(r.mec[0]) == [12.0, 0.0, 0.0, 0.0, 0.0]
    parts = r.partitions(0)
    assert list(parts[0]) == [1, 3, 5] and list(parts[1]) == [0, 2, 4]


def test_kat2_single_read_and_empty_block(oracle_mod):
    p = Pileup.from_reads([([3, 4, 5], [0, 1, 0], [30, 30, 30])])
    r = oracle_mod.phase_blocks(p, [1, 3], [2, 5], oracle_mod.make_params(EPS, 3, 10))
    # block (1,2): no read has last >= 1 and first <= 2 -> None (graph_processing.rs:129-131)
    assert r.best_ploidy[0] == 0 and r.read_off[1] == 0
    # block (3,5): one read; ploidy 1: every site has max count 1 <= 1 -> errors = 3*eps, bases 3,
    # expected = (3 + 3 eps) * eps; 3 eps = 0.09375 >= 0.0966..? 0.09375 < 0.09668 -> stop at ploidy 1
    assert r.best_ploidy[1] == 1 and r.mec[1, 0] == 3 * EPS
    assert list(r.block(1)[0]) == [0] and list(r.block(1)[1]) == [0]


def test_kat3_q0_observation_counts_as_present_but_weightless(oracle_mod):
    # a q=0 cell has w=0 (utils_frags.rs:706): it creates the allele key (set_to_seq_dict :166-168) so the
    # no_phred histogram counts it, but it never contributes to same/diff
    p = Pileup.from_reads([([1, 2], [0, 0], [0, 30]), ([1, 2], [1, 0], [30, 30]), ([1, 2], [1, 0], [30, 30])])
    rid, pb, po, mec, na, it = oracle_mod.one_ploidy(p, 1, 2, 1, EPS)
    # ploidy 1 unit counts: site 1 -> {0:1, 1:2}: bases 2, errors 1; site 2 -> {0:3}: bases 3; mec = 1
    assert mec == 1.0 and na == 6.0


def test_get_range_with_lengths_hand_example(oracle_mod):
    # utils_frags.rs:405-463 traced by hand: positions 100,200,...,1000; block 300, overlap 100
    pos = np.arange(1, 11) * 100
    s, e = oracle_mod.block_ranges(pos, 300, 100, 0.0005)
    # i=1..3: cum=100,200,300 (hit_new_left at cum>200 -> i=3); i=4: cum=400>300 -> push (0,3); left: g[3]+300<g[4]? no -> 4
    # i=5: cum=100 (from g[4]) ... i=7: cum=300 -> new_left=7; i=8: cum=400 -> push (4,7); left=8; i=9 last -> push (8,9)
    assert list(zip(s, e)) == [(1, 4), (5, 8), (9, 10)]


def test_get_range_with_lengths_on_reference_vcf_positions(oracle_mod):
    # block counts on tests/test.vcf's 954 SNP positions were computed independently in SURVEY.md §8d:
    # 17 at -l 10000, 33 at 5000, 176 at 500 (fixture: the POS column only)
    pos = np.load(__file__.rsplit("/", 1)[0] + "/golden/test_vcf_positions.npy")
    assert len(pos) == 954
    for L, n in ((10000, 17), (5000, 33), (500, 176)):
        s, e = oracle_mod.block_ranges(pos, L)
        assert len(s) == n
        assert s[0] == 1 and e[-1] == 954 and np.all(s <= e)


def test_dyadic_epsilon_mec_is_integer_plus_m_eps(oracle_mod):
    rng = np.random.default_rng(3)
    from tests.helpers import random_pileup
    p = random_pileup(rng, 60, 30, 3)
    r = oracle_mod.phase_blocks(p, [1, 11], [15, 30], oracle_mod.make_params(EPS, 4, 10))
    m = r.mec / EPS
    assert np.array_equal(m, np.round(m)), "mec_no_phred must be integer + m*eps"
    for b in range(2):
        ids, part = r.block(b)
        assert np.all(part < r.best_ploidy[b]) and np.array_equal(ids, np.sort(ids))


def test_kat4_hap_graph_edges_by_hand(oracle_mod):
    # update_hap_graph (graph_processing.rs:22-100) + HapNode::new (types_structs.rs:168-209), derived by hand:
    # reads 0,1 carry allele 0 and reads 2,3 allele 1 on SNPs 1..3 (q=30, w = 0.999); blocks (1,2) and (2,3), both phased
    # into {0,1} | {2,3}.  Node maps of block 2 hold SNPs 2,3 only.  Read 0 vs node 0: no diff; vs node 1: its allele is
    # absent at both SNPs -> diff = 2w = 1.998 -> rounds to 2: unambiguous, and read 0 sits in node 0 -> weight[0][0] += 1.
    from types import SimpleNamespace
    p = Pileup.from_reads([([1, 2, 3], [a] * 3, [30] * 3) for a in (0, 0, 1, 1)])
    res = SimpleNamespace(best_ploidy=np.array([2, 2], np.uint32), read_off=np.array([0, 4, 8], np.uint64),
                          read_id=np.array([0, 1, 2, 3, 0, 1, 2, 3], np.uint32), part=np.array([0, 0, 1, 1, 0, 0, 1, 1], np.uint8))
    cov, ew = oracle_mod.hap_graph(p, [1, 2], [2, 3], res)
    w = oracle_mod.weight_q24()[30] / 2.0 ** 24
    assert list(ew) == [2, 0, 0, 2]
    assert np.allclose(cov, 2 * w, rtol=0, atol=0) and len(cov) == 4      # sorted counts [2w, 2w], element [2*2/3]
    # ambiguity: with ONE node in the next block every contained read counts (:49-53)
    res1 = SimpleNamespace(best_ploidy=np.array([2, 1], np.uint32), read_off=res.read_off, read_id=res.read_id,
                           part=np.array([0, 0, 1, 1, 0, 0, 0, 0], np.uint8))
    _, ew1 = oracle_mod.hap_graph(p, [1, 2], [2, 3], res1)
    assert list(ew1) == [2, 2]


def test_kat5_haploset_stats_by_hand(oracle_mod):
    # utils_frags.rs:596-655 by hand: 3 reads over SNPs 1..3; SNP1 alleles {0:2, 1:1}, SNP2 {0:3}, SNP3 {1:1} (one read only)
    p = Pileup.from_reads([([1, 2], [0, 0], [30, 30]), ([1, 2, 3], [0, 0, 1], [30, 30, 30]), ([1, 2], [1, 0], [30, 30])])
    cov, err, total_err, total_cov = oracle_mod.haploset_stats(p, [0, 1, 2], 1, 4)
    # supports 3, 3, 1 (SNP 4 uncovered): total 7 over 3 covered SNPs; errors = (3-2) + 0 + 0 = 1
    assert total_cov == 7.0 and total_err == 1.0 and cov == 7.0 / 3.0 and err == 1.0 / 7.0


def test_kat6_hapq_by_hand(oracle_mod):
    # part_block_manip.rs:517-616 by hand.  SNPs 1000 bp apart, -l 1000.
    #  H0: 2 reads over SNPs 1..4, all allele 0            range (1,4)
    #  H1: 3 reads over SNPs 3..6: 0,1,1,1 — one of them reads allele 1 at SNP 3 (a minority)   range (3,6)
    #  H2: a single read over SNPs 8..9                     range (8,9)  -> HAPQ 0 (one read)
    #  H3: 2 reads over SNPs 11..12 that disagree at SNP 11 with equal quality (a TIE), H4: 2 clean reads 1,0 over 11..12
    q = 30
    reads = [([1, 2, 3, 4], [0, 0, 0, 0], [q] * 4), ([1, 2, 3, 4], [0, 0, 0, 0], [q] * 4),
             ([3, 4, 5, 6], [0, 1, 1, 1], [q] * 4), ([3, 4, 5, 6], [0, 1, 1, 1], [q] * 4), ([3, 4, 5, 6], [1, 1, 1, 1], [q] * 4),
             ([8, 9], [0, 0], [q] * 2),
             ([11, 12], [0, 0], [q] * 2), ([11, 12], [1, 0], [q] * 2),
             ([11, 12], [1, 0], [q] * 2), ([11, 12], [1, 0], [q] * 2)]
    p = Pileup.from_reads(reads)
    # ids after the Frag::cmp sort: find each group's reads by content
    def ids(first, alleles):
        out = [r for r in range(p.n_reads) if p.first[r] == first and list(p.allele[p.read_off[r]:p.read_off[r + 1]]) == alleles]
        return out
    h0 = ids(1, [0, 0, 0, 0]); h1 = ids(3, [0, 1, 1, 1]) + ids(3, [1, 1, 1, 1]); h2 = ids(8, [0, 0])
    h3 = ids(11, [0, 0]) + ids(11, [1, 0])[:1]; h4 = ids(11, [1, 0])[1:]
    assert [len(x) for x in (h0, h1, h2, h3, h4)] == [2, 3, 1, 2, 2]
    pos = 100 + 1000 * np.arange(12, dtype=np.uint64)
    hq, rel, avg = oracle_mod.hapq(p, [h0, h1, h2, h3, h4], [(1, 4), (3, 6), (8, 9), (11, 12), (11, 12)], pos, 1000)
    import math
    # H0 vs H1: overlap_percent = min(4-3+1, 6-1+1)/4 = 0.5; shared SNPs 3 (0 vs 0: same) and 4 (0 vs 1: diff) -> dist 0.5 -> penalty 0.25
    assert hq[0] == int(40 * 0.75 * (2 / 3) * math.log(3000 / 1000 + 1))         # 27
    assert hq[1] == int(40 * 0.75 * 1.0 * math.log(3000 / 1000 + 1))             # 41
    assert hq[2] == 0
    # the tie at SNP 11 of H3 resolves to allele 1 (last maximal entry of the inner map, ascending allele order): H3 == H4 on both SNPs,
    # dist 0, penalty 1 -> t1 = 0.  (Resolved to allele 0 it would be 40*0.5*(2/3)*ln 2 = 9.)
    assert hq[3] == 0 and hq[4] == 0
    # errors: H1 has one minority cell (SNP 3), H3 one (the tie: support 2, max 1); coverage 8 + 12 + 2 + 4 + 4 = 30
    assert avg == 2.0 / 30.0
    assert rel[0] == 0.0 and rel[1] == (1.0 / 12.0) / avg and rel[3] == (1.0 / 4.0) / avg and rel[4] == 0.0


def test_kat7_ploidy3_three_way_ties_through_the_heap(oracle_mod):
    """Hand trace of global_clustering.rs:49-150 with std::BinaryHeap at ploidy 3, -n 1 (limit = ploidy * 1 = 3 for the first 25
    reads), two SNPs, q = 20 (w = 0.99), eps = 2^-5.  Reads: r0 = (0,0), r1 = (1,1), r2 = (0,0).
      read 0: the three partitions are empty: (same, diff) = (0, 2 eps) each, n = (0.0625) as usize = 0 -> p-value 0 for all, nothing pruned; three children
              c0, c1, c2 (r0 in partition 0 / 1 / 2) tie at MEC 2 eps; pushes of equal elements never swap (`x <= parent` breaks): heap [c0, c1, c2].
      read 1: in every state the partition holding r0 gives diff = 2w = 1.98 -> (n, k) = (1, 1) -> -13.86, the empty ones 0 -> pruned (-13.86 - lse < ln 0.01);
              six children A=(c0,1) B=(c0,2) C=(c1,0) D=(c1,2) E=(c2,0) F=(c2,1), all at MEC 4 eps, pairwise different blocks.  Capacity 3: push D -> [A,B,C,D],
              pop: D to the root, A leaves, sift_down_to_bottom takes the RIGHT child on a tie (`data[child] <= data[child+1]`) -> [C,B,D]; push E, pop -> C
              leaves -> [D,B,E]; push F, pop -> D leaves -> [E,B,F].
      two reads only: into_sorted_vec of the tied [E,B,F]: swap(0,2) -> [F,B,E], nothing sifts; swap(0,1) -> [B,F,E]: the best state is B = (c0, 2):
              r0 in partition 0, r1 in partition 2.
      read 2 (states visited in array order E, B, F): the partition holding r0 gives same = 1.98 -> (1, 0) -> +0.127, the one holding r1 -13.86 (pruned), the
              empty one 0; children a=(E,1) .1875, b=(E,2) .125, c=(B,0) .125, d=(B,1) .1875, e=(F,0) .1875, f=(F,2) .125.  push a,b,c -> [a,b,c]; push d: sifts
              above b (d > b) and stops at a (d <= a) -> [a,d,c,b]; pop: b to the root, a leaves, the larger child d moves up -> [d,b,c]; push e -> [d,e,c,b],
              pop -> d leaves -> [e,b,c]; push f (f <= b stays) -> [e,b,c,f], pop: f to the root, e leaves, tie b <= c takes the right child c -> [c,b,f].
              into_sorted_vec of the tied [c,b,f] -> [b,f,c]: best = b = (E, 2): r2 joins r0 in partition 2, r1 sits in partition 0."""
    reads = [([1, 2], [0, 0], [20, 20]), ([1, 2], [1, 1], [20, 20]), ([1, 2], [0, 0], [20, 20])]
    rid, pb, po, mec, na, it = oracle_mod.one_ploidy(Pileup.from_reads(reads[:2]), 1, 2, 3, EPS, beam=1)
    assert list(rid) == [0, 1] and list(pb) == [0, 2]
    rid, pb, po, mec, na, it = oracle_mod.one_ploidy(Pileup.from_reads(reads), 1, 2, 3, EPS, beam=1)
    assert list(rid) == [0, 1, 2] and list(pb) == [2, 0, 2]
    # opt_iterate: only partition 2 has more than one read; moving r0 or r2 anywhere loses (own diff 0 against 2 eps / 2 w): no candidate, the first
    # round's score does not improve on itself (`new_score > prev_score` is false for equal scores) -> returned as it came
    assert list(po) == [2, 0, 2]
    # unit-weight MEC of that partition (local_clustering.rs:187-215): partition 0 = {r1}: every site has max count 1 <= 1 -> eps each; partition 2: none
    assert mec == 2 * EPS


def test_kat8_opt_iterate_equal_gains(oracle_mod):
    """Hand trace of local_clustering.rs:71-130, 292-358 at ploidy 2, two SNPs, q = 20.  Given partition: P0 = seven reads (0,0) [ids 0..6] and five reads (1,1)
    [ids 7..11], P1 = two reads (1,1) [ids 12, 13].
      round 1: in P0 allele 0 leads 7w : 5w; each (1,1) read of P0 has own diff 2w, diff 0 against P1 -> five candidates with the SAME gain 1.98 (no other read
               gains); stable sort keeps the enumeration order (ascending id here, the documented canonical order); number_of_moves = 5/10 = 0 -> 5/3 + 1 = 2; the
               loop breaks AFTER the move with mv_num = 3 > 2: reads 7, 8, 9, 10 move, read 11 stays.  Score -10w -> -2w: accepted.
      round 2: read 11 is the only candidate (gain 1.98); number_of_moves = 0 -> 1; it moves.  Score -2w -> -0: accepted.
      round 3: no candidate; -0 > -0 is false -> the partition of round 2 is returned.  Three rounds entered, two accepted."""
    reads = [([1, 2], [0, 0], [20, 20])] * 7 + [([1, 2], [1, 1], [20, 20])] * 7
    p = Pileup.from_reads(reads)                                  # ids follow the list order (equal spans: ties keep the input order)
    part = [0] * 7 + [0] * 5 + [1] * 2
    po, iters = oracle_mod.optimize_given(p, list(range(14)), part, 2, EPS)
    assert list(po) == [0] * 7 + [1] * 7
    assert iters in (2, 3)                                        # (how the restatement counts rounds is its own business; the partition is the known answer)
    # one round only: with 20 iterations capped at the first we cannot observe the intermediate state from outside, but a partition that needs just
    # the first round pins the `mv_num > number_of_moves` cut: four candidates -> number_of_moves 2 -> all four moved in ONE accepted round
    part4 = [0] * 7 + [0] * 4 + [1] * 3
    po4, it4 = oracle_mod.optimize_given(p, list(range(14)), part4, 2, EPS)
    assert list(po4) == [0] * 7 + [1] * 7 and it4 <= iters
