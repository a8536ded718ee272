"""ctypes wrapper of oracle/libfloria_oracle.so — the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see floria_oracle.cpp header): imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never by anything under floria_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from floria_amd import _capi as capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libfloria_oracle.so")
    src = os.path.join(_HERE, "floria_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "floria_hip.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfloria_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.floria_oracle_last_error.restype = C.c_char_p
        L.floria_oracle_binom.restype = C.c_double
        L.floria_oracle_binom.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_double]
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError(f"oracle rc={rc}: {lib().floria_oracle_last_error().decode()}")


def weight_q24():
    out = np.zeros(256, np.uint32)
    _check(lib().floria_oracle_weight_q24(capi.ptr(out, C.c_uint32)))
    return out


def block_ranges(snp_pos, block_length, overlap_len=None, minimal_density=0.0005):
    snp_pos = np.ascontiguousarray(snp_pos, np.uint64)
    if overlap_len is None:
        overlap_len = block_length // 3            # graph_processing.rs:337
    out = C.POINTER(capi.CRanges)()
    _check(lib().floria_oracle_block_ranges(capi.ptr(snp_pos, C.c_uint64), C.c_uint32(len(snp_pos)), C.c_uint64(block_length),
                                            C.c_uint64(overlap_len), C.c_double(minimal_density), C.byref(out)))
    r = out.contents
    res = (capi.np_from(r.start, r.n, np.uint32), capi.np_from(r.end, r.n, np.uint32))
    lib().floria_oracle_ranges_free(out)
    return res


def make_params(epsilon, max_ploidy=5, beam=10, ploidy_sensitivity=2, stopping_heuristic=1):
    return capi.CParams(float(epsilon), int(max_ploidy), int(beam), int(ploidy_sensitivity), int(stopping_heuristic))


def phase_blocks(pileup, blk_start, blk_end, params, threads=1):
    cp = pileup.as_c()
    bs = np.ascontiguousarray(blk_start, np.uint32)
    be = np.ascontiguousarray(blk_end, np.uint32)
    out = C.POINTER(capi.CBlockResult)()
    _check(lib().floria_oracle_phase_blocks(C.byref(cp), capi.ptr(bs, C.c_uint32), capi.ptr(be, C.c_uint32), C.c_uint32(len(bs)),
                                            C.byref(params), C.c_uint32(threads), C.byref(out)))
    res = capi.BlockResult(out.contents)
    lib().floria_oracle_block_result_free(out)
    return res


def one_ploidy(pileup, start, end, ploidy, epsilon, beam=10):
    """(read_id, part_after_beam, part_after_optimize, mec_bad, num_alleles, iters) for one (block, ploidy) job."""
    cp = pileup.as_c()
    n = C.c_uint32(0)
    cap = pileup.n_reads + 1
    rid = np.zeros(cap, np.uint32)
    pb = np.zeros(cap, np.uint8)
    po = np.zeros(cap, np.uint8)
    mec = C.c_double(0)
    na = C.c_double(0)
    it = C.c_int(0)
    _check(lib().floria_oracle_one_ploidy(C.byref(cp), C.c_uint32(start), C.c_uint32(end), C.c_uint32(ploidy), C.c_double(epsilon),
                                          C.c_uint32(beam), C.byref(n), capi.ptr(rid, C.c_uint32), capi.ptr(pb, C.c_uint8),
                                          capi.ptr(po, C.c_uint8), C.byref(mec), C.byref(na), C.byref(it)))
    k = n.value
    return rid[:k].copy(), pb[:k].copy(), po[:k].copy(), mec.value, na.value, it.value


def optimize_given(pileup, read_ids, part, ploidy, epsilon):
    """optimize_clustering on a given partition of `read_ids` -> (part_after, successful iterations)."""
    cp = pileup.as_c()
    rid = np.ascontiguousarray(read_ids, np.uint32)
    pi = np.ascontiguousarray(part, np.uint8)
    po = np.zeros(len(rid), np.uint8)
    it = C.c_int(0)
    _check(lib().floria_oracle_optimize_given(C.byref(cp), capi.ptr(rid, C.c_uint32), capi.ptr(pi, C.c_uint8), C.c_uint32(len(rid)), C.c_uint32(ploidy),
                                              C.c_double(epsilon), capi.ptr(po, C.c_uint8), C.byref(it)))
    return po, it.value


def reassign(pileup, groups, ranges, epsilon, read_order=None):
    """groups: list of read-id arrays; ranges: [(start,end)] -> (list of arrays, [(start,end)])"""
    cp = pileup.as_c()
    off = np.zeros(len(groups) + 1, np.uint64)
    off[1:] = np.cumsum([len(g) for g in groups])
    reads = (np.concatenate([np.asarray(g, np.uint32) for g in groups]) if len(groups) else np.zeros(0, np.uint32)).astype(np.uint32)
    reads = np.ascontiguousarray(reads)
    rng = np.ascontiguousarray(np.asarray(ranges, np.uint32).reshape(-1))
    out = C.POINTER(capi.CGroups)()
    if read_order is None:
        po, no = None, 0
    else:
        ordv = np.ascontiguousarray(read_order, np.uint32)
        po, no = capi.ptr(ordv, C.c_uint32), len(ordv)
    _check(lib().floria_oracle_reassign_ordered(C.byref(cp), capi.ptr(off, C.c_uint64), capi.ptr(reads, C.c_uint32), capi.ptr(rng, C.c_uint32),
                                                C.c_uint32(len(groups)), po, C.c_uint32(no), C.c_double(epsilon), C.byref(out)))
    g = capi.Groups(out.contents)
    lib().floria_oracle_groups_free(out)
    return g


def heap_trace(scores, limit):
    scores = np.ascontiguousarray(scores, np.float64)
    n = len(scores)
    ids = np.zeros(n + 1, np.int32)
    srt = np.zeros(n + 1, np.int32)
    ln = C.c_uint32(0)
    _check(lib().floria_oracle_heap_trace(capi.ptr(scores, C.c_double), C.c_uint32(n), C.c_uint32(limit), capi.ptr(ids, C.c_int32),
                                          C.byref(ln), capi.ptr(srt, C.c_int32)))
    return ids[:ln.value].copy(), srt[:ln.value].copy()


def binom(n, k, p, div=0.25):
    return lib().floria_oracle_binom(n, k, p, div)


def hap_graph(pileup, blk_start, blk_end, res):
    """HapNode::new cov + update_hap_graph out_weights for one contig's blocks, given S1's result `res` (BlockResult) for
    exactly these blocks -> (node_cov [sum best_ploidy], edge_w [concatenated p1 x p2 matrices of consecutive non-empty blocks])."""
    cp = pileup.as_c()
    bs = np.ascontiguousarray(blk_start, np.uint32); be = np.ascontiguousarray(blk_end, np.uint32)
    bp = np.ascontiguousarray(res.best_ploidy, np.uint32)
    roff = np.ascontiguousarray(res.read_off, np.uint64); rid = np.ascontiguousarray(res.read_id, np.uint32); part = np.ascontiguousarray(res.part, np.uint8)
    ne_blocks = [int(p) for p in bp if p]
    cov = np.zeros(sum(ne_blocks) + 1, np.float64)
    ew = np.zeros(sum(a * b for a, b in zip(ne_blocks[:-1], ne_blocks[1:])) + 1, np.uint32)
    _check(lib().floria_oracle_hap_graph(C.byref(cp), capi.ptr(bs, C.c_uint32), capi.ptr(be, C.c_uint32), C.c_uint32(len(bs)), capi.ptr(bp, C.c_uint32),
                                         capi.ptr(roff, C.c_uint64), capi.ptr(rid, C.c_uint32), capi.ptr(part, C.c_uint8),
                                         capi.ptr(cov, C.c_double), capi.ptr(ew, C.c_uint32)))
    return cov[:-1], ew[:-1]


def set_order_mode(mode):
    """0 = canonical ascending counter_id at the iteration-order-dependent sites, 1 = descending (sensitivity tests only),
    2 = the order of an emulated FxHashSet (fxhash 0.2.1 + hashbrown as published; unverifiable here)."""
    lib().floria_oracle_set_order_mode(C.c_int(mode))


def set_a14_tie_mode(mode):
    """separate_broken_haplogroups (part_block_manip.rs:27-98): the order of the reads that share a first_position, which decides the read a split drops.
    0 = ascending counter_id (canonical), 1 = descending, 2 = an FxHashSet filled in the re-insertion order of S2 (an approximation, see the C++ comment);
    -1 = stop before the splits: the haplogroups as re-inserted, in input order (what floria_hip_set_option("s2_assign_only", 1) returns)."""
    lib().floria_oracle_set_a14_tie_mode(C.c_int(mode))


def a14_dropped(reset=True):
    """reads dropped by haplogroup splits since the last reset"""
    f = lib().floria_oracle_a14_dropped
    f.restype = C.c_uint64
    return int(f(C.c_int(1 if reset else 0)))


def set_arith_mode(mode):
    """0 = canonical arithmetic (every weighted sum an exact (Q24, #epsilon) pair turned into f64 once: what the HIP path computes),
    1 = the reference's running f64 sums, terms added in the iteration order of its (emulated) hash containers.  Identical for
    dyadic epsilon; mode 1 is for counting how often they part elsewhere (scripts/arith_sensitivity.py).
    2 = the same running f64 sums with every container iterated in ASCENDING key order (positions, alleles, counter_ids) instead of the emulated
    hash orders: the mode the independent Python restatement (oracle/py_restatement.py) is compared with at a non-dyadic epsilon."""
    lib().floria_oracle_set_arith_mode(C.c_int(mode))


def fxset_order(ops):
    """Iteration order of the emulated FxHashSet<&Frag> after `ops`: +k+1 inserts counter_id k, -(k+1) removes it."""
    ops = np.ascontiguousarray(ops, np.int64)
    out = np.zeros(len(ops) + 1, np.uint32)
    n = C.c_uint32(0)
    _check(lib().floria_oracle_fxset_order(capi.ptr(ops, C.c_int64), C.c_uint32(len(ops)), capi.ptr(out, C.c_uint32), C.byref(n)))
    return out[:n.value].copy()


def positions_order(segments, removed=()):
    """Iteration order (SNP positions) of Frag.positions for a fragment merged from `segments` — the ascending SNP positions of every alignment, in merge
    order (combine_frags, file_reader.rs:491-659) — after the positions in `removed` were taken out (--ignore-monomorphic, utils_frags.rs:745-755)."""
    keys = np.ascontiguousarray(np.concatenate([np.asarray(s, np.uint32) for s in segments]) if len(segments) else np.zeros(0, np.uint32), np.uint32)
    off = np.zeros(len(segments) + 1, np.uint32)
    off[1:] = np.cumsum([len(s) for s in segments])
    rem = np.ascontiguousarray(removed, np.uint32)
    out = np.zeros(max(1, len(keys)), np.uint32)
    n = C.c_uint32(0)
    _check(lib().floria_oracle_positions_order(capi.ptr(keys, C.c_uint32), capi.ptr(off, C.c_uint32), C.c_uint32(len(segments)), capi.ptr(rem, C.c_uint32), C.c_uint32(len(rem)),
                                              capi.ptr(out, C.c_uint32), C.byref(n)))
    return out[:n.value].copy()


def set_order_of(cell_snps, segments, removed=()):
    """the `set_order` entries of one read (include/floria_hip.h): index of every cell of `cell_snps` (ascending) in the iteration order of its position set"""
    order = positions_order(segments, removed)
    cell_snps = np.asarray(cell_snps, np.uint32)
    idx = np.searchsorted(cell_snps, order)
    assert len(order) == len(cell_snps) and np.array_equal(cell_snps[idx], order)
    return idx.astype(np.uint32)


def fxset_entry_order(keys):
    """iteration order and bucket count of an FxHashMap filled with `entry(key).or_insert(..)` per key (utils_frags.rs:165)"""
    keys = np.ascontiguousarray(keys, np.uint64)
    out = np.zeros(max(1, len(keys)), np.uint32)
    n, nb = C.c_uint32(0), C.c_uint32(0)
    _check(lib().floria_oracle_fxset_entry_order(capi.ptr(keys, C.c_uint64), C.c_uint32(len(keys)), capi.ptr(out, C.c_uint32), C.byref(n), C.byref(nb)))
    return out[:n.value].copy(), nb.value


def fxset_insert_order(keys):
    """Iteration order after inserting `keys` (counter_ids) in sequence into an empty set; repeated keys are no-ops."""
    return fxset_order(np.asarray(keys, np.int64) + 1)


def last_set_order(res):
    """After phase_blocks in order mode 2: per block, the reads partition by partition, each partition in its set's iteration order."""
    out = np.zeros(int(res.read_off[-1]), np.uint32)
    _check(lib().floria_oracle_last_set_order(capi.ptr(out, C.c_uint32), C.c_uint64(len(out))))
    return out


def s2_visit_order_emulated(res, set_order, paths):
    """The order in which process_reads_for_final_parts visits the reads under the emulated hash order (part_block_manip.rs:185-203):
    HapNode.frag_set (S1's final sets) -> joined_path_part (graph_processing.rs:690-692: a set filled node by node along the path, from
    the path's LAST node back to its first) -> read_to_parts_map (keys inserted part by part in each joined set's iteration order).
    paths: [[(block, row), ...]] per haplogroup, in the order the traceback walks them (end node first)."""
    node_order = {}
    for b in range(res.n_blocks):
        lo = int(res.read_off[b])
        ids, part = res.block(b)
        for r in range(int(res.best_ploidy[b])):
            k = int(np.count_nonzero(part == r))
            node_order[(b, r)] = set_order[lo:lo + k]
            lo += k
    keys = []
    for path in paths:
        joined_ops = np.concatenate([node_order[n] for n in path]) if path else np.zeros(0, np.uint32)
        keys.append(fxset_insert_order(joined_ops))
    return fxset_insert_order(np.concatenate(keys) if keys else np.zeros(0, np.uint32))


def haploset_stats(pileup, reads, lo, hi):
    """get_errors_cov_from_frags for one haploset -> (cov, err, total_err, total_cov)."""
    cp = pileup.as_c()
    r = np.ascontiguousarray(reads, np.uint32)
    out = np.zeros(4, np.float64)
    _check(lib().floria_oracle_haploset_stats(C.byref(cp), capi.ptr(r, C.c_uint32), C.c_uint32(len(r)), C.c_uint32(lo), C.c_uint32(hi), capi.ptr(out, C.c_double)))
    return out


def hapq(pileup, groups, ranges, snp_to_genome_pos, block_length):
    """get_hapq for one contig's haplosets -> (hapq uint8 [n], rel_err float64 [n], avg_err)."""
    cp = pileup.as_c()
    off = np.zeros(len(groups) + 1, np.uint64)
    off[1:] = np.cumsum([len(g) for g in groups])
    reads = np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint32) for g in groups]) if len(groups) else np.zeros(0, np.uint32), np.uint32)
    rng = np.ascontiguousarray(np.asarray(ranges, np.uint32).reshape(-1))
    pos = np.ascontiguousarray(snp_to_genome_pos, np.uint64)
    hq = np.zeros(len(groups) + 1, np.uint8)
    rel = np.zeros(len(groups) + 1, np.float64)
    avg = C.c_double(0)
    _check(lib().floria_oracle_hapq(C.byref(cp), capi.ptr(off, C.c_uint64), capi.ptr(reads, C.c_uint32), capi.ptr(rng, C.c_uint32),
                                    C.c_uint32(len(groups)), capi.ptr(pos, C.c_uint64), C.c_uint32(len(pos)), C.c_uint64(int(block_length)),
                                    capi.ptr(hq, C.c_uint8), capi.ptr(rel, C.c_double), C.byref(avg)))
    return hq[:-1], rel[:-1], avg.value
