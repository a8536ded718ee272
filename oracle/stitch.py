"""CPU restatement (plain Python / numpy / scipy) of the host stages between S1 and the files on disk.

TEST INFRASTRUCTURE ONLY, like the rest of oracle/: imported by tests/, never by anything under floria_amd/.  It checks the
C++ host code of floria_amd/host/ (stitch.cpp, writer.cpp) the way oracle/floria_oracle.cpp checks the kernels:

  build_hap_graph      process_chunks + update_hap_graph's >= 2.0 filter   graph_processing.rs:22-100,306-323
  lp_optimum           the LP of solve_lp_graph (solve_flow.rs:195-290) solved with scipy's HiGHS: the OPTIMAL VALUE a correct
                       solver must reach (the reference's minilp and the product's min-cost flow may stop at different optimal
                       vertices; parity of the flows themselves is unpinned)
  disjoint_paths       get_disjoint_paths_rewrite                          graph_processing.rs:462-750
  snpless_gap_frags    get_frags_in_snpless_gaps                           part_block_manip.rs:622-675
  expected_files       write_haplotypes / write_all_parts_file / write_fragset_haplotypes / write_nosnp_reads_parts and the
                       contig_ploidy_info.tsv row                          file_writer.rs:151-165,308-369,699-993
petgraph's StableGraph iteration orders (edge lists newest-first, toposort over node indices in reverse) are restated from
the crate as published (0.6.5); unverifiable here (no Rust toolchain), they only break exact ties.
"""
import math

import numpy as np

MIN_SHARED_READS_UNAMBIG = 2.0
F64_MAX = 1.7976931348623157e308


class Node:
    def __init__(self, col, row, nid, cov, ends, reads):
        self.col, self.row, self.id, self.cov, self.ends, self.reads = col, row, nid, cov, ends, list(reads)
        self.out_edges, self.in_edges, self.out_flows = [], [], []


def build_hap_graph(res, blk_start, blk_end, node_cov, edge_w):
    """columns of Nodes from an S1 BlockResult of ONE contig plus oracle.hap_graph's (cov, edge weights)."""
    cols, nid, k = [], 0, 0
    for b in range(res.n_blocks):
        p = int(res.best_ploidy[b])
        if p == 0:
            continue
        ids, part = res.block(b)
        col = []
        for r in range(p):
            col.append(Node(len(cols), r, nid, float(node_cov[k]), (int(blk_start[b]), int(blk_end[b])), ids[part == r].tolist()))
            nid += 1; k += 1
        cols.append(col)
    o = 0
    for c in range(len(cols) - 1):
        p1, p2 = len(cols[c]), len(cols[c + 1])
        for j in range(p1):
            for l in range(p2):
                w = float(edge_w[o + j * p2 + l])
                if w >= MIN_SHARED_READS_UNAMBIG:
                    cols[c][j].out_edges.append((l, w)); cols[c + 1][l].in_edges.append((j, w))
        o += p1 * p2
    return cols


def parse_debug_graph(path):
    """floria-hip --debug's debug_graph.txt -> (columns of Nodes, flows [((c,r),(c+1,r2),flow)], paths [(lo, hi, [read ids])])"""
    nodes, flows, paths, edges = {}, [], [], []
    for line in open(path):
        t = line.rstrip("\n").split("\t")
        if t[0] == "N":
            c, r = int(t[1]), int(t[2])
            nodes[(c, r)] = Node(c, r, int(t[3]), float(t[4]), (int(t[5]), int(t[6])), [int(x) for x in t[7:]])
        elif t[0] == "E":
            edges.append((int(t[1]), int(t[2]), int(t[3]), float(t[4])))
        elif t[0] == "F":
            flows.append(((int(t[1]), int(t[2])), (int(t[1]) + 1, int(t[3])), float(t[4])))
        elif t[0] == "P":
            paths.append((int(t[1]), int(t[2]), [int(x) for x in t[3:]]))
    ncol = 1 + max(c for c, _ in nodes) if nodes else 0
    cols = [[nodes[(c, r)] for r in range(sum(1 for k in nodes if k[0] == c))] for c in range(ncol)]
    for c, r, r2, w in edges:
        cols[c][r].out_edges.append((r2, w)); cols[c + 1][r2].in_edges.append((r, w))
    return cols, flows, paths


def lp_edges(cols):
    return [((n.col, n.row), (n.col + 1, l), w) for col in cols for n in col for l, w in n.out_edges]


def lp_optimum(cols):
    """min sum_e |x_e - a_e| s.t. x >= 0, inflow == outflow at interior nodes with both kinds of edges -> (value, x)"""
    from scipy.optimize import linprog
    edges = lp_edges(cols)
    E = len(edges)
    if E == 0:
        return 0.0, np.zeros(0)
    ix = {(e[0], e[1]): i for i, e in enumerate(edges)}
    a = np.array([e[2] for e in edges])
    A_eq, b_eq = [], []
    for c in range(1, len(cols) - 1):
        for n in cols[c]:
            if n.in_edges and n.out_edges:
                row = np.zeros(2 * E)
                for j, _ in n.in_edges:
                    row[ix[((c - 1, j), (c, n.row))]] = 1.0
                for l, _ in n.out_edges:
                    row[ix[((c, n.row), (c + 1, l))]] = -1.0
                A_eq.append(row); b_eq.append(0.0)
    A_ub = np.zeros((2 * E, 2 * E)); b_ub = np.zeros(2 * E)
    for i in range(E):                      # t_i >= x_i - a_i ; t_i >= a_i - x_i
        A_ub[2 * i, i] = 1.0; A_ub[2 * i, E + i] = -1.0; b_ub[2 * i] = a[i]
        A_ub[2 * i + 1, i] = -1.0; A_ub[2 * i + 1, E + i] = -1.0; b_ub[2 * i + 1] = -a[i]
    cost = np.concatenate([np.zeros(E), np.ones(E)])
    r = linprog(cost, A_ub=A_ub, b_ub=b_ub, A_eq=np.array(A_eq) if A_eq else None, b_eq=np.array(b_eq) if b_eq else None, bounds=[(0, None)] * (2 * E), method="highs")
    assert r.status == 0, r.message
    return float(r.fun), r.x[:E]


def check_flows(cols, flows, tol=1e-7):
    """the product's flows are feasible for the LP and reach its optimal value"""
    edges = lp_edges(cols)
    assert [(f[0], f[1]) for f in flows] == [(e[0], e[1]) for e in edges], "flow vector does not follow the reference's edge order"
    x = {(f[0], f[1]): f[2] for f in flows}
    assert all(v >= -tol for v in x.values())
    for c in range(1, len(cols) - 1):
        for n in cols[c]:
            if n.in_edges and n.out_edges:
                fin = sum(x[((c - 1, j), (c, n.row))] for j, _ in n.in_edges)
                fout = sum(x[((c, n.row), (c + 1, l))] for l, _ in n.out_edges)
                assert abs(fin - fout) <= tol, (c, n.row, fin, fout)
    value = sum(abs(x[(e[0], e[1])] - e[2]) for e in edges)
    opt, _ = lp_optimum(cols)
    assert abs(value - opt) <= 1e-6 * max(1.0, abs(opt)), (value, opt)
    return value


class _StableGraph:
    """petgraph StableGraph<(col,row), f64>, directed: edge lists newest-first, stable indices"""

    def __init__(self):
        self.nodes, self.edges = [], []          # node: [alive, col, row, head_out, head_in]; edge: [alive, src, dst, w, next_out, next_in]

    def add_node(self, col, row):
        self.nodes.append([True, col, row, -1, -1]); return len(self.nodes) - 1

    def add_edge(self, a, b, w):
        self.edges.append([True, a, b, w, self.nodes[a][3], self.nodes[b][4]])
        self.nodes[a][3] = self.nodes[b][4] = len(self.edges) - 1

    def out_edges(self, n):
        e = self.nodes[n][3]
        while e != -1:
            yield e
            e = self.edges[e][4]

    def in_edges(self, n):
        e = self.nodes[n][4]
        while e != -1:
            yield e
            e = self.edges[e][5]

    def remove_edge(self, ei):
        ed = self.edges[ei]
        if not ed[0]:
            return
        for node, head, nxt in ((ed[1], 3, 4), (ed[2], 4, 5)):
            if self.nodes[node][head] == ei:
                self.nodes[node][head] = ed[nxt]
            else:
                e = self.nodes[node][head]
                while self.edges[e][nxt] != ei:
                    e = self.edges[e][nxt]
                self.edges[e][nxt] = ed[nxt]
        ed[0] = False

    def remove_node(self, n):
        for e in list(self.out_edges(n)) + list(self.in_edges(n)):
            self.remove_edge(e)
        self.nodes[n][0] = False

    def alive(self):
        return [i for i, n in enumerate(self.nodes) if n[0]]

    def toposort(self):
        disc, fin, out = set(), set(), []
        for i in reversed(range(len(self.nodes))):
            if not self.nodes[i][0] or i in disc:
                continue
            stack = [i]
            while stack:
                nx = stack[-1]
                if nx not in disc:
                    disc.add(nx)
                    for e in self.out_edges(nx):
                        s = self.edges[e][2]
                        if s not in disc:
                            stack.append(s)
                else:
                    stack.pop()
                    if nx not in fin:
                        fin.add(nx); out.append(nx)
        return out[::-1]


def disjoint_paths(cols, flows, return_nodes=False):
    """get_disjoint_paths_rewrite -> [(lo, hi, sorted read ids)] in peeling order (with return_nodes also the (column, row) nodes
    of every path in traceback order, i.e. from the path's last node to its first)"""
    node_paths = []
    for col in cols:
        for n in col:
            n.out_flows = []
    for (c1, r1), (c2, r2), f in flows:
        if f < MIN_SHARED_READS_UNAMBIG:
            continue
        cols[c1][r1].out_flows.append((r2, f))
    g = _StableGraph()
    index = [[g.add_node(n.col, n.row) for n in col] for col in cols]
    for col in cols:
        for n in col:
            for r2, f in n.out_flows:
                g.add_edge(index[n.col][n.row], index[n.col + 1][r2], f)
    N = len(g.nodes)
    out = []
    while g.alive():
        score, prev, sink, source = [0.0] * N, [None] * N, [False] * N, [False] * N
        for i in g.alive():
            source[i] = next(g.in_edges(i), None) is None
            sink[i] = next(g.out_edges(i), None) is None
            score[i] = F64_MAX if source[i] else 0.0
        cut = []
        for node in g.toposort():
            for e in g.out_edges(node):
                _, s, t, flow, _, _ = g.edges[e]
                if min(score[s], flow) > score[t]:
                    if flow < score[s] * 0.33 and not source[s]:
                        if sum(1 for _ in g.in_edges(s)) == 1:
                            cut.append(e)
                        if sum(1 for _ in g.in_edges(t)) == 1:
                            score[t] = F64_MAX; source[t] = True
                    else:
                        score[t] = min(score[s], flow); prev[t] = s
        for e in cut:
            g.remove_edge(e)
        best, best_score = None, -F64_MAX
        for i in range(N):
            if score[i] > best_score and sink[i]:
                best, best_score = i, score[i]
        assert best is not None
        reads, lo, hi, path = set(), 2 ** 32 - 1, 0, []
        cur = best
        while cur is not None:
            n = cols[g.nodes[cur][1]][g.nodes[cur][2]]
            lo, hi = min(lo, n.ends[0]), max(hi, n.ends[1])
            reads.update(n.reads)
            path.append(cur)
            cur = prev[cur]
        for i in path:
            g.remove_node(i)
        node_paths.append([(g.nodes[i][1], g.nodes[i][2]) for i in path])
        out.append((lo, hi, sorted(reads)))
    return (out, node_paths) if return_nodes else out


def snpless_gap_frags(ranges, snp_pos0, snpless, final_spans, final_names, block_len):
    """get_frags_in_snpless_gaps; snpless = [(name, (first_pos_base, last_pos_base), seq_len)] (unpaired data) -> [(name, seq_len)]"""
    iv = [(int(snp_pos0[lo - 1]), int(snp_pos0[hi - 1]) + 1) for lo, hi in ranges]

    def count(start, stop):
        return sum(1 for s, e in iv if s < stop and e > start)
    out = [(nm, sl) for nm, sp, sl in snpless if count(sp[0], sp[1]) == 0]
    out += [(nm, None) for nm, sp in zip(final_names, final_spans) if count(sp[0], sp[1]) == 0]
    return out


def rust_fixed(v, prec):
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    return f"{v:.{prec}f}"


def _allele_iter(cnt):
    present = [a for a in range(4) if cnt[a]]
    return [0, 2, 1, 3] if len(present) == 4 else present


def expected_files(pileup, names, parts, ranges, stats, hapqs, rel_err, avg_err, snp_pos0, contig, contig_dir, contig_len, snpless_rows):
    """-> dict(file name -> text) of one contig's output directory plus 'ploidy_row' (the contig_ploidy_info.tsv line)"""
    S = len(snp_pos0)
    cnt_all, cov_all, c15, c30, c45 = np.zeros(S), np.zeros(S), np.zeros(S), np.zeros(S), np.zeros(S)
    vart, info, hset = [], [], []
    total_bases = 0
    for i, (reads, (lo, hi)) in enumerate(zip(parts, ranges)):
        if len(reads) == 0:
            continue
        lg, rg = int(snp_pos0[lo - 1]), int(snp_pos0[hi - 1])
        total_bases += rg - lg
        cov, err = stats[i][0], stats[i][1]
        q = int(hapqs[i])
        cnt_all[lo - 1:hi] += 1; cov_all[lo - 1:hi] += cov
        if q >= 15: c15[lo - 1:hi] += 1
        if q >= 30: c30[lo - 1:hi] += 1
        if q >= 45: c45[lo - 1:hi] += 1
        head = (f">HAP{i}.{contig_dir}\tCONTIG:{contig}\tSNPRANGE:{lo}-{hi}\tBASERANGE:{lg + 1}-{rg + 1}\tCOV:{rust_fixed(cov, 3)}\tERR:{rust_fixed(err, 4)}"
                f"\tHAPQ:{q}\tREL_ERR:{rust_fixed(rel_err[i], 3)}\n")
        hist = np.zeros((hi - lo + 1, 4), np.int64)
        for r in reads:
            s, a, _ = pileup.read(int(r))
            m = (s >= lo) & (s <= hi)
            np.add.at(hist, (s[m].astype(np.int64) - lo, a[m].astype(np.int64)), 1)
        info.append(f">HAP{i}.{contig_dir}\tSNPRANGE:{lo}-{hi}\n")
        alle = []
        for p in range(lo, hi + 1):
            c = hist[p - lo]
            order = _allele_iter(c)
            if not order:
                info.append(f"{p}:{int(snp_pos0[p - 1])}\t?\tNA\t\n"); alle.append("?"); continue
            best = order[0]
            for a in order[1:]:
                if c[a] >= c[best]:
                    best = a
            info.append(f"{p}:{int(snp_pos0[p - 1])}\t{best}\t" + "|".join(f"{a}:{int(c[a])}" for a in order) + "\t\n")
            alle.append(str(best))
        vart.append(head + "".join(alle) + "\n")
        hset.append(head + "".join(f"{names[int(r)]}\t{int(pileup.first[int(r)])}\t{int(pileup.last[int(r)])}\n" for r in sorted(int(x) for x in reads)))
    nz = int(np.count_nonzero(cnt_all > 0))
    rough = cov_all.sum() / nz if nz else float("nan")
    row = (f"{contig}\t{rust_fixed(cnt_all.sum() / S, 3)}\t{rust_fixed(total_bases / contig_len, 3)}\t{rust_fixed(rough, 3)}\t{total_bases}\t"
           f"{rust_fixed(c15.sum() / S, 3)}\t{rust_fixed(c30.sum() / S, 3)}\t{rust_fixed(c45.sum() / S, 3)}\t{rust_fixed(avg_err, 4)}\n")
    return {f"{contig}.vartigs": "".join(vart), "vartig_info.txt": "".join(info), f"{contig}.haplosets": "".join(hset),
            "reads_without_snps.tsv": "READ_NAME\tREAD_LENGTH_IN_BASES\n" + "".join(f"{nm}\t{sl}\n" for nm, sl in snpless_rows), "ploidy_row": row}


# ---- --output-reads (file_writer.rs:86-150, 168-217, 370-560), restated for the synthetic data sets of floria_amd/synth_bam.py ----------
def _fastq(name, seq, qual):
    return b"@" + name.encode() + b"\n" + seq + b"\n+\n" + qual + b"\n"


def _acgt(seq):
    return bytes(c if c in b"ACGT" else (c - 32 if c in b"acgt" else 65) for c in seq)          # DnaString::from_acgt_bytes


def _rc(seq):
    return bytes({65: 84, 67: 71, 71: 67, 84: 65}[c] for c in reversed(seq))


def read_records(names, pileup, alignments, snp_pos0, paired):
    """per read (pileup order): dict(name, paired, first, last, seq[2], qual[2], pos2seq {snp: (mate, offset in that mate's SEQ)}) —
    what frag_from_record + combine_frags leave in a Frag (file_reader.rs:505-560, 661-736)"""
    snp_pos0 = np.asarray(snp_pos0, np.int64)
    out = []
    for i, nm in enumerate(names):
        s, _, _ = pileup.read(i)
        called = set(int(x) for x in s)
        seqs, quals, pos2seq = [b"", b""], [b"", b""], {}
        for k, (pos, seq, cigar, qual) in enumerate(alignments[nm]):
            seqs[k] = _acgt(seq); quals[k] = bytes(min(255, q + 33) for q in qual)
            q, r = 0, pos
            for op, ln in cigar:
                if op == "M":
                    lo, hi = np.searchsorted(snp_pos0, r, "left"), np.searchsorted(snp_pos0, r + ln, "left")
                    for x in range(lo, hi):
                        if x + 1 in called:
                            pos2seq[x + 1] = (k, q + int(snp_pos0[x]) - r)
                if op in "MIS":
                    q += ln
                if op in "MDN":
                    r += ln
        out.append(dict(name=nm, paired=paired, first=int(pileup.first[i]), last=int(pileup.last[i]), seq=seqs, qual=quals, pos2seq=pos2seq))
    return out


def _paired_no_trim(rec):
    a = _fastq(rec["name"] + "/1", rec["seq"][0], rec["qual"][0]) if rec["seq"][0] else _fastq(rec["name"] + "/1", b"N", b"!")
    b = _fastq(rec["name"] + "/2", _rc(rec["seq"][1]), rec["qual"][1]) if rec["seq"][1] else _fastq(rec["name"] + "/2", b"N", b"!")
    return a, b


def expected_read_files(parts, ranges, recs, snpless_recs, extend_read_clipping=True, extension=25):
    """-> {relative path: bytes} of long_reads/ and short_reads/ (uncompressed content)"""
    files = {}
    for i, (reads, (lo, hi)) in enumerate(zip(parts, ranges)):
        if len(reads) == 0:
            continue
        single, p1, p2 = [], [], []
        for r in sorted(int(x) for x in reads):
            rec = recs[r]
            if not rec["seq"][0] and not rec["seq"][1]:
                continue
            if rec["first"] > hi or rec["last"] < lo:
                continue
            if rec["first"] > lo and extend_read_clipping:
                left = 0
            else:
                t = lo
                while t not in rec["pos2seq"]:
                    t += 1
                left = rec["pos2seq"][t][1]
            left = left - extension if left > extension else 0
            if rec["last"] < hi and extend_read_clipping:
                rp = 1 if rec["paired"] else 0
                right = len(rec["seq"][rp]) - 1 if rec["seq"][rp] else 0
            else:
                t = hi
                while t not in rec["pos2seq"]:
                    t -= 1
                rp, right = rec["pos2seq"][t]
            n = len(rec["seq"][rp])
            if n == 0:
                right = 0
            elif n > extension + 1 and right < n - extension - 1:
                right += extension
            else:
                right = n - 1
            if rec["paired"]:
                a, b = _paired_no_trim(rec); p1.append(a); p2.append(b)
            else:
                if left > right:
                    single.append(None); continue
                single.append(_fastq(rec["name"], rec["seq"][0][left:right + 1], rec["qual"][0][left:right + 1]))
        if p1:
            files[f"short_reads/{i}_part_paired1.fastq"] = b"".join(p1); files[f"short_reads/{i}_part_paired2.fastq"] = b"".join(p2)
        if single:
            files[f"long_reads/{i}_part.fastq"] = b"".join(x for x in single if x is not None)
    single, p1, p2 = [], [], []
    for rec in snpless_recs:
        if rec["paired"]:
            a, b = _paired_no_trim(rec); p1.append(a); p2.append(b)
        else:
            single.append(_fastq(rec["name"], rec["seq"][0], rec["qual"][0]) if rec["seq"][0] else _fastq(rec["name"], b"N", b"!"))
    if p1:
        files["short_reads/snpless_paired1.fastq"] = b"".join(p1); files["short_reads/snpless_paired2.fastq"] = b"".join(p2)
    if single:
        files["long_reads/snpless.fastq"] = b"".join(single)
    return files
