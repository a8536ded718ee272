"""A SECOND, independent restatement of floria's per-block phasing (seam S1) — TEST INFRASTRUCTURE ONLY.

Written straight from the Rust text with Python dicts, sets and lists, without consulting oracle/floria_oracle.cpp: its only purpose is
to catch a misreading that the C++ oracle and the HIP kernels (which were written against that oracle) could share.  It does not pin
parity (nothing here has met the Rust binary either); it is compared with the C++ oracle in tests/test_py_restatement.py.

Everything is kept as the reference keeps it: nested maps pos -> allele -> f64 count, deep clones per child, deep equality for the
duplicate test, RUNNING f64 sums (`diff += epsilon` between `diff += w`), std::collections::BinaryHeap restated from libstd.  The one
thing Python cannot reproduce is the iteration order of FxHashMap / FxHashSet; every such iteration here runs in ASCENDING key order
(positions, alleles, counter_ids).  For a dyadic epsilon no sum depends on it; elsewhere the C++ oracle has a mode with the same
orders (set_arith_mode(2)).

Reference text followed (all under /root/reference/src):
  types_structs.rs:87-112 (Frag order / identity), :114-153 (SearchNode), :216-251 (build_child_node), :253-268 (HapBlock),
                  :326-376 (build_truncated_hap_block)
  utils_frags.rs:32-75 (distance_read_haplo_epsilon_empty), :160-184 (set_to_seq_dict, hap_block_from_partition),
                 :211-258 (stable_binom_cdf_p_rev, log_sum_exp), :702-711 (phred_scale)
  global_clustering.rs:10-208 (beam_search_phasing, read_to_node_value)
  local_clustering.rs:12-59 (find_reads_in_interval), :71-130 (optimize_clustering), :187-260 (MEC statistics), :292-358 (opt_iterate)
  graph_processing.rs:103-252 (get_local_hap_blocks: the ploidy loop and its stop rule)
  constants.rs (NUM_ITER_OPTIMIZE = 20, DIV_FACTOR = 0.25, PROB_CUTOFF = 0.01)
"""
import math

import numpy as np

NUM_ITER_OPTIMIZE = 20
DIV_FACTOR = 0.25
PROB_CUTOFF = 0.01


class Frag:
    """types_structs.rs:68-112 — the fields the path reads."""
    __slots__ = ("counter_id", "seq_dict", "qual_dict", "positions", "first_position", "last_position", "_w")

    def __init__(self, counter_id, cells):
        """cells: iterable of (snp position, allele, quality)"""
        self.counter_id = counter_id
        self.seq_dict = {}
        self.qual_dict = {}
        for pos, al, q in cells:
            self.seq_dict[int(pos)] = int(al)
            self.qual_dict[int(pos)] = int(q)
        self.positions = sorted(self.seq_dict)                   # FxHashSet<SnpPosition>; iterated ascending here
        self.first_position = self.positions[0]
        self.last_position = self.positions[-1]
        self._w = {p: phred_weight(self.qual_dict[p]) for p in self.positions}

    def sort_key(self):
        # Ord for Frag (types_structs.rs:87-93): (self.first, OTHER.last, self.id) vs (other.first, SELF.last, other.id): first ascending, last DESCENDING, id
        return (self.first_position, -self.last_position, self.counter_id)


def phred_weight(q):
    # utils_frags.rs:702-711: 1. - 10_f32.powf(q as f32 / -10.), widened to f64
    return float(np.float32(1.0) - np.power(np.float32(10.0), np.float32(q) / np.float32(-10.0), dtype=np.float32))


def frags_from_pileup(pileup):
    """floria_amd.pileup.Pileup (CSR, already in Frag order with counter_id = index) -> [Frag]"""
    out = []
    for r in range(pileup.n_reads):
        b, e = int(pileup.read_off[r]), int(pileup.read_off[r + 1])
        out.append(Frag(r, zip(pileup.snp[b:e], pileup.allele[b:e], pileup.qual[b:e])))
    return out


# ---- utils_frags.rs -------------------------------------------------------------------------------------------------------------
def distance_read_haplo_epsilon_empty(r, hap, epsilon):
    diff = 0.0
    same = 0.0
    for pos in r.positions:
        empty_pos = True
        if pos in hap:
            for _key in sorted(hap[pos]):
                if hap[pos][_key] != 0.0:
                    empty_pos = False
                    break
        if empty_pos:
            diff += epsilon
            continue
        frag_var = r.seq_dict[pos]
        site = hap[pos]
        # max_by_key returns the LAST maximal element of the iteration
        consensus_var = None
        best = None
        for a in sorted(site):
            if best is None or site[a] >= best:
                best = site[a]
                consensus_var = a
        if frag_var == consensus_var:
            same += r._w[pos]
        else:
            count = site.get(frag_var)
            if count is not None and count == site[consensus_var]:
                same += r._w[pos]
                continue
            diff += r._w[pos]
    return same, diff


def set_to_seq_dict(frag_set, use_phred):
    hap_map = {}
    for frag in sorted(frag_set, key=lambda f: f.counter_id):
        for pos in frag.positions:
            var = frag.seq_dict[pos]
            sites = hap_map.setdefault(pos, {})
            if var not in sites:
                sites[var] = 0.0
            sites[var] += frag._w[pos] if use_phred else 1.0
    return hap_map


def hap_block_from_partition(part, use_qual):
    return [set_to_seq_dict(s, use_qual) for s in part]


def as_usize(x):
    # Rust `f64 as usize`: truncation toward zero, saturating, NaN -> 0
    if x != x or x <= 0.0:
        return 0
    if x >= 18446744073709551615.0:
        return 18446744073709551615
    return int(x)


def stable_binom_cdf_p_rev(n, k, p, div_factor):
    if n == 0:
        return 0.0
    n64 = float(n)
    k64 = float(k)
    a = k64 / n64
    if a == 1.0:
        a = 0.9999999
    if a == 0.0:
        a = 0.0000001
    rel_ent = a * math.log(a / p) + (1.0 - a) * math.log((1.0 - a) / (1.0 - p))
    if a < p:
        rel_ent = -rel_ent
    return -1.0 * n64 / div_factor * rel_ent


def log_sum_exp(probs):
    mx = float("nan")
    for v in probs:                                    # fold(f64::NAN, f64::max): max ignores a NaN operand
        if mx != mx:
            mx = v
        elif v == v and v > mx:
            mx = v
    s = 0.0
    for v in probs:
        s += math.exp(v - mx)
    return mx + math.log(s)


# ---- std::collections::BinaryHeap (library/alloc/src/collections/binary_heap/mod.rs), elements compared by key() only ------------
class BinaryHeap:
    """Max-heap; `le(a, b)` is the element type's `<=`.  For (Rc<SearchNode>, HapBlock) that is: score (partial_cmp), then
    HapBlock::cmp = blocks.len() (always equal) -> ties are Equal: `<=` and `>=` hold, `<` does not."""

    def __init__(self):
        self.data = []

    def __len__(self):
        return len(self.data)

    @staticmethod
    def _le(a, b):
        return a[0].score <= b[0].score

    @staticmethod
    def _lt(a, b):
        return a[0].score < b[0].score

    def push(self, item):
        old_len = len(self.data)
        self.data.append(item)
        self._sift_up(0, old_len)

    def _sift_up(self, start, pos):
        d = self.data
        elt = d[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if self._le(elt, d[parent]):
                break
            d[pos] = d[parent]
            pos = parent
        d[pos] = elt
        return pos

    def pop(self):
        d = self.data
        item = d.pop()
        if d:
            item, d[0] = d[0], item
            self._sift_down_to_bottom(0)
        return item

    def _sift_down_to_bottom(self, pos):
        d = self.data
        end = len(d)
        start = pos
        elt = d[pos]
        child = 2 * pos + 1
        while child <= max(end - 2, 0) and end >= 2:             # end.saturating_sub(2)
            if self._le(d[child], d[child + 1]):
                child += 1
            d[pos] = d[child]
            pos = child
            child = 2 * pos + 1
        if child == end - 1:
            d[pos] = d[child]
            pos = child
        d[pos] = elt
        self._sift_up(start, pos)

    def _sift_down_range(self, pos, end):
        d = self.data
        elt = d[pos]
        child = 2 * pos + 1
        while end >= 2 and child <= end - 2:
            if self._le(d[child], d[child + 1]):
                child += 1
            if not self._lt(elt, d[child]):                      # hole.element() >= hole.get(child)
                d[pos] = elt
                return
            d[pos] = d[child]
            pos = child
            child = 2 * pos + 1
        if child == end - 1 and self._lt(elt, d[child]):
            d[pos] = d[child]
            pos = child
        d[pos] = elt

    def into_sorted_vec(self):
        d = self.data
        end = len(d)
        while end > 1:
            end -= 1
            d[0], d[end] = d[end], d[0]
            self._sift_down_range(0, end)
        return d


# ---- types_structs.rs: SearchNode, build_child_node, build_truncated_hap_block ----------------------------------------------------
class SearchNode:
    __slots__ = ("read", "part", "score", "error_vec", "parent_node")

    def __init__(self, read, part, score, error_vec, parent_node):
        self.read = read
        self.part = part
        self.score = score
        self.error_vec = error_vec
        self.parent_node = parent_node


def build_truncated_hap_block(block, frag, part, current_startpos):
    block_vec = [{pos: dict(site) for pos, site in hap.items()} for hap in block]          # manual deepcopy
    for i in range(len(block)):
        for pos in block[i]:
            if pos < current_startpos:
                del block_vec[i][pos]
    # (num_after / num_before / blocks_broken feed break_positions, which only WEIRD_SPLIT = false code reads: constants.rs:18)
    for pos in frag.positions:                                   # frag.seq_dict.keys()
        var = frag.seq_dict[pos]
        sites = block_vec[part].setdefault(pos, {})
        if var not in sites:
            sites[var] = 0.0
        sites[var] += frag._w[pos]
    return block_vec


def read_to_node_value(node, frag, block, part_index, epsilon):
    same, diff = distance_read_haplo_epsilon_empty(frag, block[part_index], epsilon)
    new_error_vec = []
    for i in range(len(block)):
        if i == part_index:
            new_error_vec.append((node.error_vec[i][0] + same, node.error_vec[i][1] + diff))
        else:
            new_error_vec.append(node.error_vec[i])
    mec = 0.0
    for x in new_error_vec:
        mec += x[1]
    return -1.0 * mec, new_error_vec


# ---- global_clustering.rs:10-179 ------------------------------------------------------------------------------------------------------
def beam_search_phasing(ploidy, all_reads, epsilon, div_factor, cutoff_value, max_number_solns, margins=None):
    """-> partition: list of `ploidy` sets of Frag.  margins (a list) receives |p_k - lse - cutoff| of every pruning decision."""
    partition = [set() for _ in range(ploidy)]
    if not all_reads:
        return []
    first_block = hap_block_from_partition(partition, True)
    first_node = SearchNode(all_reads[0], None, 0.0, [(0.0, 0.0)] * ploidy, None)
    heap = BinaryHeap()
    heap.push((first_node, first_block))
    for i, frag in enumerate(all_reads):
        max_num_soln_mut = max_number_solns
        if i < 25:
            max_num_soln_mut = ploidy * max_number_solns
        heap_next = BinaryHeap()
        current_startpos = frag.first_position
        for node, block in heap.data:                            # search_node_heap.iter(): the underlying vector's order
            p_value_list = []
            for part_index in range(ploidy):
                same, diff = distance_read_haplo_epsilon_empty(frag, block[part_index], epsilon)
                p_value_list.append(1.0 * stable_binom_cdf_p_rev(as_usize(same + diff), as_usize(diff), epsilon, div_factor))
            lse = log_sum_exp(p_value_list)
            for j in range(ploidy):
                if margins is not None:
                    margins.append(abs((p_value_list[j] - lse) - cutoff_value))
                if p_value_list[j] - lse > cutoff_value:
                    score, new_error_vec = read_to_node_value(node, frag, block, j, epsilon)
                    new_node = SearchNode(frag, j, -score, new_error_vec, node)
                    new_block = build_truncated_hap_block(block, frag, j, current_startpos)
                    project_exists = False
                    for other in heap_next.data:
                        if other[1] == new_block and other[0].score >= new_node.score:
                            project_exists = True
                    if not project_exists:
                        heap_next.push((new_node, new_block))
                        if len(heap_next) > max_num_soln_mut:
                            heap_next.pop()
        heap = heap_next
    node_pointer = heap.into_sorted_vec()[0][0]
    while node_pointer.parent_node is not None:
        partition[node_pointer.part].add(node_pointer.read)
        node_pointer = node_pointer.parent_node
    return partition


# ---- local_clustering.rs ----------------------------------------------------------------------------------------------------------------
def find_reads_in_interval(start, end, all_frags):
    final_set = []
    for frag in all_frags:                                       # sorted by first_position
        if frag.last_position < start:
            continue
        if frag.first_position > end:
            break
        if frag.last_position - frag.first_position > 10000:
            continue
        final_set.append(frag)
    return final_set


def _mec_stats_of_block(hap_block, epsilon):
    binom_vec = []
    for hap in hap_block:
        errors = 0.0
        bases = 0.0
        for pos in sorted(hap):                                  # hap.values()
            seq_dict = hap[pos]
            allele_counts = [(a, seq_dict[a]) for a in sorted(seq_dict)]
            if not allele_counts:
                continue
            allele_counts.sort(key=lambda x: x[1])               # sort_by(|x, y| x.1.cmp(&y.1)): stable
            counts = [c for _, c in allele_counts]
            cons_bases = counts[-1]
            bases += cons_bases
            for i in range(len(counts) - 1):
                errors += counts[i]
            if cons_bases <= 1.0:
                errors += epsilon
        binom_vec.append((bases, errors))
    return binom_vec


def get_mec_stats_epsilon(hap_block, epsilon):
    return _mec_stats_of_block(hap_block, epsilon)               # use_gaps = true at the only call sites: nothing is removed


def get_mec_stats_epsilon_no_phred(read_part, epsilon):
    return _mec_stats_of_block(hap_block_from_partition(read_part, False), epsilon)


def opt_iterate(partition, hap_block, epsilon):
    ploidy = len(partition)
    best_moves = []
    for i in range(ploidy):
        if len(partition[i]) <= 1:
            continue
        for read in sorted(partition[i], key=lambda f: f.counter_id):
            _good, errors_read = distance_read_haplo_epsilon_empty(read, hap_block[i], epsilon)
            for j in range(ploidy):
                if j == i:
                    continue
                _g, read_errors_movej = distance_read_haplo_epsilon_empty(read, hap_block[j], epsilon)
                diff_score = errors_read - read_errors_movej
                if diff_score > 0.0:
                    best_moves.append((diff_score, (i, read, j)))
    moved_reads = set()
    new_part = [set(s) for s in partition]
    best_moves.sort(key=lambda m: -m[0])                         # sort_by(b.0.partial_cmp(a.0)): stable, descending gain
    number_of_moves = len(best_moves) // 10
    if number_of_moves == 0 and len(best_moves) > 0:
        number_of_moves = len(best_moves) // 3 + 1
    for mv_num, mv in enumerate(best_moves):
        i, read, j = mv[1]
        if read in moved_reads:
            continue
        if len(new_part[i]) == 1:
            continue
        new_part[j].add(read)
        new_part[i].discard(read)
        moved_reads.add(read)
        if mv_num > number_of_moves:
            break
    return new_part


def optimize_clustering(partition, epsilon, max_iters):
    if not any(len(p) > 0 for p in partition):
        return 0.0, partition, 0
    prev_hap_block = hap_block_from_partition(partition, True)
    s = 0.0
    for x in get_mec_stats_epsilon(prev_hap_block, epsilon):
        s += x[1]
    prev_score = s * -1.0
    best_part = partition
    ok_iters = 0
    for _ in range(max_iters):
        new_part = opt_iterate(best_part, prev_hap_block, epsilon)
        new_block = hap_block_from_partition(new_part, True)
        s = 0.0
        for x in get_mec_stats_epsilon(new_block, epsilon):
            s += x[1]
        new_score = s * -1.0
        if new_score > prev_score:
            prev_score = new_score
            best_part = new_part
            prev_hap_block = new_block
            ok_iters += 1
        else:
            return prev_score, best_part, ok_iters
    return prev_score, best_part, ok_iters


# ---- graph_processing.rs:103-252 ----------------------------------------------------------------------------------------------------
def get_local_hap_blocks(all_frags, start, end, epsilon, max_ploidy=5, max_number_solns=10, ploidy_sensitivity=2, stopping_heuristic=True):
    """One SNP block -> None (no reads) or dict(best_ploidy, tried, mec_vector [max_ploidy], reads [counter_id ascending],
    part [partition index of every read at best_ploidy], margins)."""
    reads = find_reads_in_interval(start, end, all_frags)
    if not reads:
        return None
    mec_vector = [0.0] * max_ploidy
    parts_vector = []
    expected_errors_ref = []
    margins = []
    best_ploidy = 1
    tried = 0
    for ploidy in range(1, max_ploidy + 1):
        best_ploidy = ploidy
        tried = ploidy
        num_alleles = 0.0
        vec_reads_own = sorted(reads, key=Frag.sort_key)
        part = beam_search_phasing(ploidy, vec_reads_own, epsilon, DIV_FACTOR, math.log(PROB_CUTOFF), max_number_solns, margins)
        _score, optimized_part, _it = optimize_clustering(part, epsilon, NUM_ITER_OPTIMIZE)
        for good, bad in get_mec_stats_epsilon_no_phred(optimized_part, epsilon):
            mec_vector[ploidy - 1] += bad
            num_alleles += good
            num_alleles += bad
        parts_vector.append(optimized_part)
        expected_errors_ref.append(num_alleles * epsilon)
        if ploidy > 1:
            if ploidy_sensitivity == 1:
                mec_threshold = 1.0 / (1.0 - epsilon) / (1.0 + 1.0 / (math.pow(float(ploidy), 0.50) + 1.00))
            elif ploidy_sensitivity == 2:
                mec_threshold = 1.0 / (1.0 - epsilon) / (1.0 + 1.0 / (math.pow(float(ploidy), 1.00) + 1.0 / 3.0))
            else:
                mec_threshold = 1.0 / (1.0 - epsilon) / (1.0 + 1.0 / (math.pow(float(ploidy), 1.00) + 1.00))
            prev = mec_vector[ploidy - 2]
            cur = mec_vector[ploidy - 1]
            if prev == 0.0:                                      # IEEE division: x/0 = inf (x > 0), 0/0 = NaN
                ratio = float("nan") if cur == 0.0 else float("inf")
            else:
                ratio = cur / prev
            if ratio < mec_threshold:
                pass
            elif stopping_heuristic:
                best_ploidy -= 1
                break
            if mec_vector[ploidy - 1] < expected_errors_ref[ploidy - 1]:
                break
        else:
            if mec_vector[0] < expected_errors_ref[0]:
                break
    best = parts_vector[best_ploidy - 1]
    ids = sorted(f.counter_id for f in reads)
    where = {}
    for k, s in enumerate(best):
        for f in s:
            where[f.counter_id] = k
    return dict(best_ploidy=best_ploidy, tried=tried, mec_vector=mec_vector, reads=ids, part=[where[i] for i in ids], margins=margins)
