"""CPU tests of the C++ host code above the C ABI (floria_amd/host/): ingest and global stitching need no GPU.

  ingest.cpp   BAM (BGZF) / VCF / FASTA -> Frags, against the synthetic data sets of floria_amd/synth_bam.py whose pileup is known
  stitch.cpp   solve_lp_graph (exact min-cost flow) is feasible and OPTIMAL for the reference's LP (scipy / HiGHS), and
               get_disjoint_paths_rewrite equals the Python restatement (oracle/stitch.py) on the same flows
"""
import os
import subprocess

import numpy as np
import pytest

from floria_amd import synth, synth_bam
from oracle import stitch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "floria_amd", "host")


@pytest.fixture(scope="module")
def floria_hip(hip_lib):
    subprocess.check_call(["make", "-C", HOST, "floria-hip"], stdout=subprocess.DEVNULL)
    return os.path.join(HOST, "floria-hip")


def ingest(floria_hip, prefix, tmp_path, extra=()):
    from tests.test_gpu_cli import parse_frag_dump
    dump = prefix + ".frags"
    r = subprocess.run([floria_hip, "-b", prefix + ".bam", "-v", prefix + ".vcf", "-r", prefix + ".fa", "-o", str(tmp_path / "unused"), "-e", "0.03", "-l", "10000",
                        "--ingest-only", "--dump-frags", dump, *extra], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return parse_frag_dump(dump), r.stderr


@pytest.mark.parametrize("cfg,idx,scale", [(1, 0, 1.0), (4, 5, 0.5), (3, 2, 0.3)])
def test_ingest_reproduces_the_generators_pileup(floria_hip, tmp_path, cfg, idx, scale):
    c = synth.make_config_contig(cfg, idx, scale, keep_layout=True)
    prefix = str(tmp_path / "d")
    ex = synth_bam.write_dataset(prefix, [c], seed=11)[c.name]
    got, _ = ingest(floria_hip, prefix, tmp_path)
    reads, pile = got[c.name]["reads"], ex["pileup"]
    assert [g["name"] for g in reads] == ex["names"]
    assert [g["first"] for g in reads] == pile.first.tolist() and [g["last"] for g in reads] == pile.last.tolist()
    assert [g["span"] for g in reads] == [tuple(int(x) for x in sp) for sp in ex["spans"]]
    assert all(g["paired"] == (1 if cfg == 3 else 0) for g in reads)
    for i, g in enumerate(reads):
        s, a, q = pile.read(i)
        assert g["cells"] == list(zip(s.tolist(), a.tolist(), q.tolist())), f"read {i}"
    # the synthetic contig differs from its pileup only by the calls the deletions removed: the CIGAR walk was exercised
    if cfg != 3:
        assert pile.n_cells < c.pileup.n_cells


def test_vcf_filter_mapq_and_small_contigs(floria_hip, tmp_path):
    c = synth.make_config_contig(1, 0, keep_layout=True)
    small = synth.make_contig(np.random.SeedSequence([9, 9]), 40, 80, 2, "long", name="tiny", keep_layout=True)
    prefix = str(tmp_path / "d")
    ex = synth_bam.write_dataset(prefix, [c, small], seed=5)
    got, err = ingest(floria_hip, prefix, tmp_path)
    assert set(got) == {c.name} and "has < 100 variants" in err                      # floria.rs:233-247
    got, _ = ingest(floria_hip, prefix, tmp_path, extra=("--snp-count-filter", "10"))
    assert set(got) == {c.name, "tiny"} and len(got["tiny"]["reads"]) == ex["tiny"]["pileup"].n_reads
    got, _ = ingest(floria_hip, prefix, tmp_path, extra=("-m", "61"))                 # every synthetic alignment has MAPQ 60
    assert got == {}
    got, _ = ingest(floria_hip, prefix, tmp_path, extra=("-G", "tiny", "--snp-count-filter", "10"))
    assert set(got) == {"tiny"}
    # the VCF carries indel and '*' records between the SNPs (synth_bam.write_dataset): SNP numbering must skip them
    assert sum(1 for ln in open(prefix + ".vcf") if not ln.startswith("#")) > len(ex[c.name]["snp_pos0"]) + len(ex["tiny"]["snp_pos0"])


def random_graph(rng, n_cols, max_rows, n_reads=400):
    """a layered hap graph with integer edge weights >= 2, as update_hap_graph produces"""
    lines, rows, reads = [], [], iter(range(10 ** 6))
    nid = 0
    for c in range(n_cols):
        r = int(rng.integers(1, max_rows + 1))
        rows.append(r)
        for k in range(r):
            ids = sorted(set(int(x) for x in rng.integers(0, n_reads, size=int(rng.integers(1, 8)))))
            lines.append("N\t%d\t%d\t%d\t%.3f\t%d\t%d\t%s" % (c, k, nid, float(rng.random() * 30), 1 + 10 * c, 14 + 10 * c, "\t".join(map(str, ids))))
            nid += 1
    for c in range(n_cols - 1):
        for j in range(rows[c]):
            for l in range(rows[c + 1]):
                if rng.random() < 0.6:
                    lines.append("E\t%d\t%d\t%d\t%d" % (c, j, l, int(rng.integers(2, 60))))
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", range(40))
def test_lp_flows_are_optimal_and_paths_match_the_restatement(floria_hip, tmp_path, seed):
    rng = np.random.default_rng(300 + seed)
    path = str(tmp_path / "g.txt")
    open(path, "w").write(random_graph(rng, int(rng.integers(2, 12)), int(rng.integers(1, 5))))
    r = subprocess.run([floria_hip, "--stitch-graph", path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    open(path, "a").write(r.stdout)
    cols, flows, paths = stitch.parse_debug_graph(path)
    stitch.check_flows(cols, flows)
    assert all(abs(f[2] - round(f[2])) == 0 for f in flows)                      # a vertex of a network LP with integer data
    assert paths == stitch.disjoint_paths(cols, flows)
    # every node lies on exactly one path, so every read of the graph is in some haplogroup
    assert set(x for p in paths for x in p[2]) == set(x for col in cols for n in col for x in n.reads)


def test_lp_hand_case_with_a_unique_optimum(floria_hip, tmp_path):
    # a chain 0 -(7)-> 1 -(7)-> 2 -(4)-> 3: node 1 is balanced, node 2 needs inflow == outflow.  Lowering edges (0,1) and (1,2) to 4
    # costs 6, raising (2,3) to 7 costs 3 and any mix costs in between: the unique optimum is x = (7, 7, 7).
    g = "N\t0\t0\t0\t1\t1\t4\t0\nN\t1\t0\t1\t1\t5\t8\t1\nN\t2\t0\t2\t1\t9\t12\t2\nN\t3\t0\t3\t1\t13\t16\t3\nE\t0\t0\t0\t7\nE\t1\t0\t0\t7\nE\t2\t0\t0\t4\n"
    path = str(tmp_path / "h.txt")
    open(path, "w").write(g)
    r = subprocess.run([floria_hip, "--stitch-graph", path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fl = [ln.split("\t") for ln in r.stdout.splitlines() if ln.startswith("F")]
    assert [float(x[4]) for x in fl] == [7.0, 7.0, 7.0]
    ps = [ln.split("\t") for ln in r.stdout.splitlines() if ln.startswith("P")]
    assert ps == [["P", "1", "16", "0", "1", "2", "3"]]                          # one path through all four nodes


def _stitch(floria_hip, path, tie):
    r = subprocess.run([floria_hip, "--stitch-graph", path, "--lp-tie", tie], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [ln.split("\t") for ln in r.stdout.splitlines()]
    u = [x for x in rows if x[0] == "U"][0]
    return int(u[1]), int(u[2]), [float(x[4]) for x in rows if x[0] == "F"], [x for x in rows if x[0] == "P"], r.stdout


def test_lp_hand_case_with_two_optimal_vertices(floria_hip, tmp_path):
    # a chain 0 -(5)-> 1 -(3)-> 2: node 1 needs inflow == outflow, and every x in [3, 5] costs 2.  The reference's simplex stops at one of the
    # two vertices (3, 3) and (5, 5); which one cannot be known without the crate.  The flow solver returns either, on request, says that the
    # optimum is not unique, and both flows are >= 2 here so the single path through the three nodes is peeled either way.
    g = "N\t0\t0\t0\t1\t1\t4\t0\nN\t1\t0\t1\t1\t5\t8\t1\nN\t2\t0\t2\t1\t9\t12\t2\nE\t0\t0\t0\t5\nE\t1\t0\t0\t3\n"
    path = str(tmp_path / "t.txt")
    open(path, "w").write(g)
    c1, m1, f1, p1, _ = _stitch(floria_hip, path, "first")
    c2, m2, f2, p2, _ = _stitch(floria_hip, path, "last")
    assert c1 == c2 == 2 and m1 == m2 == 2                                       # both edges can move
    assert sorted([f1, f2]) == [[3.0, 3.0], [5.0, 5.0]]
    assert p1 == p2 == [["P", "1", "12", "0", "1", "2"]]


@pytest.mark.parametrize("seed", range(40))
def test_lp_both_tie_orders_are_optimal_and_uniqueness_is_reported(floria_hip, tmp_path, seed):
    """The two extremes of the solver's tie-breaking are both optimal for the reference's LP (HiGHS), have the same cost, and are identical
    exactly when the solver says the optimum is unique."""
    rng = np.random.default_rng(300 + seed)
    path = str(tmp_path / "g.txt")
    text = random_graph(rng, int(rng.integers(2, 12)), int(rng.integers(1, 5)))
    res = []
    for tie in ("first", "last"):
        open(path, "w").write(text)
        cost, movable, fl, ps, out = _stitch(floria_hip, path, tie)
        open(path, "a").write(out)
        cols, flows, paths = stitch.parse_debug_graph(path)
        stitch.check_flows(cols, flows)
        assert paths == stitch.disjoint_paths(cols, flows)
        res.append((cost, movable, fl, ps))
    assert res[0][0] == res[1][0]
    if res[0][1] == 0:
        assert res[1][1] == 0 and res[0][2] == res[1][2] and res[0][3] == res[1][3]
    if res[0][2] != res[1][2]:
        assert res[0][1] > 0 and res[1][1] > 0
        differing = sum(a != b for a, b in zip(res[0][2], res[1][2]))
        assert differing <= res[0][1]                                            # only edges reported as movable differ


def l_epsilon_restated(alignments):
    """l_epsilon_auto_detect (file_reader.rs:749-826) for gapless, unfiltered alignments [(pos, seq, cigar)] of ONE contig: every
    covered reference position is a pileup column; columns are counted and every 1000th is sampled (a sampled column with fewer
    than 5 bases does not advance the count, so the next one is sampled too)."""
    starts = np.array([a[0] for a in alignments]); ends = np.array([a[0] + len(a[1]) for a in alignments])
    count, errs, lens = 0, [], []
    pos, hi = int(starts.min()), int(ends.max())
    covered = np.zeros(hi + 1, np.int32)
    np.add.at(covered, starts, 1); np.add.at(covered, ends, -1)
    covered = np.cumsum(covered) > 0
    for p in np.nonzero(covered)[0]:
        if count % 1000 != 0:
            count += 1
            continue
        idx = np.nonzero((starts <= p) & (ends > p))[0]
        bases = {}
        for i in idx:
            lens.append(len(alignments[i][1]))
            b = alignments[i][1][p - alignments[i][0]]
            bases[b] = bases.get(b, 0) + 1
        tot, most = sum(bases.values()), max(bases.values())
        if tot < 5:
            continue
        errs.append((tot - most) / most)
        if len(errs) >= 1000 and lens:
            break
        count += 1
    lens.sort(); errs.sort()
    return max(lens[len(lens) * 66 // 100], 500), max(errs[len(errs) * 66 // 100], 0.01)


def test_auto_detect_matches_the_restatement(floria_hip, tmp_path):
    import re
    c = synth.make_config_contig(4, 1, keep_layout=True)
    prefix = str(tmp_path / "d")
    ex = synth_bam.write_dataset(prefix, [c], seed=2, edit_frac=0.0, realign=False)[c.name]
    r = subprocess.run([floria_hip, "-b", prefix + ".bam", "-v", prefix + ".vcf", "-r", prefix + ".fa", "-o", str(tmp_path / "u"), "--ingest-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = re.search(r"Estimated -l (\d+), -e ([0-9.eE+-]+)", r.stderr)
    l, e = l_epsilon_restated(ex["alignments"])
    assert m and int(m.group(1)) == l and abs(float(m.group(2)) - e) < 1e-5 * e, (m.group(0), l, e)


def nw_affine(q, r, match=1, mismatch=-1, gap_open=-2, gap_extend=-1):
    """global alignment score, gap of length n costs open + (n - 1) * extend (Gotoh, plain Python: independent of ingest.cpp)"""
    NEG = -10 ** 9
    n, m = len(q), len(r)
    M = [[NEG] * (m + 1) for _ in range(n + 1)]; X = [[NEG] * (m + 1) for _ in range(n + 1)]; Y = [[NEG] * (m + 1) for _ in range(n + 1)]
    M[0][0] = 0
    for i in range(1, n + 1): X[i][0] = gap_open + (i - 1) * gap_extend
    for j in range(1, m + 1): Y[0][j] = gap_open + (j - 1) * gap_extend
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            s = match if q[i - 1].upper() == r[j - 1].upper() else mismatch
            M[i][j] = max(M[i - 1][j - 1], X[i - 1][j - 1], Y[i - 1][j - 1]) + s
            X[i][j] = max(max(M[i - 1][j], Y[i - 1][j]) + gap_open, X[i - 1][j] + gap_extend)
            Y[i][j] = max(max(M[i][j - 1], X[i][j - 1]) + gap_open, Y[i][j - 1] + gap_extend)
    return max(M[n][m], X[n][m], Y[n][m])


def test_realign_changes_calls_next_to_unreported_indels(floria_hip, tmp_path):
    # alignment::realign (alignment.rs:7-64): a read that lost one base shortly before a SNP but was aligned without a gap shows the
    # NEXT reference base in the SNP column; re-aligning its 32 bases against the reference with each allele in turn recovers the allele
    # (or keeps the call).  Expected alleles come from a plain-Python Gotoh DP with the same scores.
    rng = np.random.default_rng(12)
    L = 4000
    ref = "".join("ACGT"[i] for i in rng.integers(0, 4, size=L))
    snps = list(range(200, 3800, 60))
    alt = {p: "ACGT"[("ACGT".index(ref[p]) + 1 + int(rng.integers(0, 3))) % 4] for p in snps}
    recs, truth = [], []
    for k in range(60):
        b = int(rng.integers(0, 1500)); e = b + 2000
        hap = k % 2
        seq = list(ref[b:e])
        for p in snps:
            if b <= p < e and hap:
                seq[p - b] = alt[p]
        cut = None
        if k % 3 == 0:                                  # drop one base 2..6 bases before some SNP, keep the all-match CIGAR
            p = [x for x in snps if b + 100 <= x < e - 100][k % 7]
            cut = p - b - int(rng.integers(2, 7))
            del seq[cut]
        seq = "".join(seq)
        recs.append((b, synth_bam.bam_record(0, b, f"r{k}", 0, 60, [("M", len(seq))], seq.encode(), np.full(len(seq), 30, np.uint8))))
        truth.append((b, seq))
    recs.sort(key=lambda t: t[0])
    prefix = str(tmp_path / "ra")
    synth_bam.write_bam(prefix + ".bam", [("ctg", L)], [r for _, r in recs])
    open(prefix + ".fa", "w").write(">ctg\n" + ref + "\n")
    with open(prefix + ".vcf", "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for p in snps:
            f.write(f"ctg\t{p + 1}\t.\t{ref[p]}\t{alt[p]}\t50\tPASS\t.\n")
    called, _ = ingest(floria_hip, prefix, tmp_path, extra=("--snp-count-filter", "10", "--no-realign"))
    realigned, _ = ingest(floria_hip, prefix, tmp_path, extra=("--snp-count-filter", "10"))
    seq_of = {f"r{k}": truth[k] for k in range(60)}
    changed = 0
    for rc, rr in zip(called["ctg"]["reads"], realigned["ctg"]["reads"]):
        assert rc["name"] == rr["name"] and [c[0] for c in rc["cells"]] == [c[0] for c in rr["cells"]]
        b, seq = seq_of[rc["name"]]
        for (snp, a_called, _), (_, a_re, _) in zip(rc["cells"], rr["cells"]):
            p = snps[snp - 1]
            qpos = p - b
            if qpos < 16 or qpos + 16 >= len(seq) or p < 16 or p + 16 >= L:
                assert a_re == a_called
                continue
            q = "".join(c if c in "ACGT" else "A" for c in seq[qpos - 16:qpos + 16])
            scores = [nw_affine(q, ref[p - 16:p] + al + ref[p + 1:p + 16]) for al in (ref[p], alt[p])]
            assert a_re == (0 if scores[0] >= scores[1] else 1), (rc["name"], snp)
            changed += a_re != a_called
    assert changed > 0


@pytest.mark.parametrize("sub_rate", [0.03, 0.12])
def test_realign_shortcut_is_exact_under_sequencing_errors(floria_hip, tmp_path, sub_rate):
    # realign decides most calls from the mismatch count of the two 32-base windows without running the DP (ingest.cpp, "Exact
    # shortcut"); with substitution errors at 3 % and 12 % the windows carry 0..8 mismatches, so both the shortcut and the DP
    # are taken, on either side of the bound.  Expected calls: the numpy Gotoh DP on every window (synth_bam.realign_dataset).
    c = synth.make_config_contig(1, 0, keep_layout=True)
    prefix = str(tmp_path / "d")
    ex = synth_bam.write_dataset(prefix, [c], seed=4, sub_rate=sub_rate)[c.name]
    got, _ = ingest(floria_hip, prefix, tmp_path)
    raw, _ = ingest(floria_hip, prefix, tmp_path, extra=("--no-realign",))
    pile = ex["pileup"]
    n_changed = 0
    for i, (g, g0) in enumerate(zip(got[c.name]["reads"], raw[c.name]["reads"])):
        s, a, q = pile.read(i)
        assert g["cells"] == list(zip(s.tolist(), a.tolist(), q.tolist())), f"read {i}"
        n_changed += sum(1 for x, y in zip(g["cells"], g0["cells"]) if x != y)
    assert n_changed > 0 if sub_rate > 0.1 else True                              # noise flips some calls, the same ones in both


@pytest.mark.parametrize("sub_rate,edit_frac", [(0.03, 0.1), (0.12, 0.5)])
def test_realign_calls_barely_depend_on_a_band(sub_rate, edit_frac):
    """The reference scores the 32 x 32 realignment windows with block-aligner at a fixed block size of 8 (alignment.rs:14, :51), a heuristic that
    walks an 8-wide block through the matrix; this library (and its numpy restatement) fills the whole matrix.  block-aligner's walk cannot be
    restated without the crate (Cargo.lock: block-aligner 0.4, not vendored), so what is measured here is the sensitivity of the CALL to any
    band at all: confine the DP to |i - j| <= b and count the calls that change.  A walk of 8-wide blocks that tracks the best cell covers at
    least the +-4 diagonals around the path; the windows are cut around a SNP the aligner placed on the diagonal, so the optimal path leaves the
    diagonal only by the indels inside the window.  Numbers recorded in DESIGN.md "Realignment"."""
    c = synth.make_config_contig(1, 0, keep_layout=True)
    d = synth_bam.contig_dataset(c, np.random.default_rng(4), edit_frac=edit_frac, sub_rate=sub_rate)
    Q, R0, R1, where = synth_bam.realign_windows(d)
    assert len(where) > 20000
    full = synth_bam.nw_affine_batch(Q, R0) >= synth_bam.nw_affine_batch(Q, R1)
    rates = {}
    for b in (8, 4, 2):
        banded = synth_bam.nw_affine_batch(Q, R0, band=b) >= synth_bam.nw_affine_batch(Q, R1, band=b)
        rates[b] = float((banded != full).mean())
    print("realign band sensitivity (sub_rate %.2f, edit_frac %.1f, %d windows): " % (sub_rate, edit_frac, len(where)) + ", ".join("+-%d: %.5f" % kv for kv in rates.items()))
    assert rates[8] == 0.0 and rates[4] <= 1e-3 and rates[2] <= 1e-2


def test_ignore_monomorphic(floria_hip, tmp_path):
    # --ignore-monomorphic (utils_frags.rs:713-772): SNPs whose second allele carries less than epsilon of the first's phred weight are
    # dropped from every read before phasing; reads are re-sorted (Frag::cmp) and renumbered, reads left without SNPs dropped.
    import collections
    eps = 0.03125
    c = synth.make_config_contig(1, 0, keep_layout=True)
    prefix = str(tmp_path / "d")
    synth_bam.write_dataset(prefix, [c], seed=5, realign=False)
    plain, _ = ingest(floria_hip, prefix, tmp_path, extra=("--no-realign", "-e", str(eps)))
    filt, _ = ingest(floria_hip, prefix, tmp_path, extra=("--no-realign", "-e", str(eps), "--ignore-monomorphic"))
    plain, filt = plain[c.name]["reads"], filt[c.name]["reads"]
    w = collections.defaultdict(lambda: collections.defaultdict(float))
    for g in plain:
        for snp, al, q in g["cells"]:
            w[snp][al] += float(np.float32(1.0) - np.float32(10.0) ** (np.float32(q) / np.float32(-10.0)))
    mono = set()
    for snp, m in w.items():
        v = sorted(m.values(), reverse=True)
        if len(v) == 1 or v[0] * eps > v[1]:
            mono.add(snp)
    assert 0 < len(mono) < len(w)
    want = []
    for k, g in enumerate(plain):
        cells = [x for x in g["cells"] if x[0] not in mono]
        if cells:
            want.append((cells[0][0], -cells[-1][0], k, g["name"], cells))
    want.sort()
    assert [(x[3], x[4]) for x in want] == [(g["name"], g["cells"]) for g in filt]
    assert [g["first"] for g in filt] == [x[4][0][0] for x in want] and [g["last"] for g in filt] == [x[4][-1][0] for x in want]


def test_position_sets_of_merged_and_cut_down_fragments_are_replayed(floria_hip, oracle_mod, tmp_path):
    """a pair's position set is the first mate's extended by the second's (file_reader.rs:539-541) and --ignore-monomorphic removes keys from the built
    set (utils_frags.rs:745-755): the host keeps what every such set went through and replays it (Frag::positions_order, floria_host.cpp) - compared here, read by
    read, with the oracle's emulation of the same sequence of container operations (which tests/test_order_emulation.py pins against an independent model)."""
    c = synth.make_config_contig(3, 2, 0.3, keep_layout=True)
    prefix = str(tmp_path / "p")
    ex = synth_bam.write_dataset(prefix, [c], seed=11)[c.name]
    got, _ = ingest(floria_hip, prefix, tmp_path)
    reads = got[c.name]["reads"]
    assert [g["name"] for g in reads] == ex["names"]
    n_merged = n_not_one_walk = 0
    for g, segs in zip(reads, ex["segments"]):
        keys = [x[0] for x in g["cells"]]
        if len(segs) == 2 and len(segs[1]):                     # the mate brought SNPs: `extend` ran
            n_merged += 1
            want = oracle_mod.positions_order([np.asarray(s, np.uint32) for s in segs]).tolist()
            assert g["order"] == want, g["name"]
            assert sorted(want) == keys
            n_not_one_walk += want != oracle_mod.positions_order([np.asarray(keys, np.uint32)]).tolist()
        else:
            assert "order" not in g
    assert n_merged > 500 and n_not_one_walk > 0
    # long reads under --ignore-monomorphic: the removed positions stay in the replay as removals
    c1 = synth.make_config_contig(1, 0, keep_layout=True)
    prefix = str(tmp_path / "m")
    synth_bam.write_dataset(prefix, [c1], seed=5, realign=False)
    plain, _ = ingest(floria_hip, prefix, tmp_path, extra=("--no-realign", "-e", "0.03125"))
    filt, _ = ingest(floria_hip, prefix, tmp_path, extra=("--no-realign", "-e", "0.03125", "--ignore-monomorphic"))
    before = {g["name"]: [x[0] for x in g["cells"]] for g in plain[c1.name]["reads"]}
    n_cut = 0
    for g in filt[c1.name]["reads"]:
        keys, orig = [x[0] for x in g["cells"]], before[g["name"]]
        if len(keys) == len(orig):
            assert "order" not in g
            continue
        n_cut += 1
        removed = sorted(set(orig) - set(keys))
        assert g["order"] == oracle_mod.positions_order([np.asarray(orig, np.uint32)], np.asarray(removed, np.uint32)).tolist(), g["name"]
    assert n_cut > 50


def test_hand_built_alignments_flags_cigar_ops_and_supplementary_merging(floria_hip, tmp_path):
    """Records the generator never writes, expectations derived by hand from file_reader.rs: alignment_passed_check (:184-235: paired
    or low-MAPQ supplementary, MAPQ, error flags, secondary), the CIGAR walk over = X N D I S H (:661-727), allele index of a multi-allelic
    record, combine_frags' supplementary branch (:566-655: merge when every gap between the pieces is <= --supp-aln-dist-cutoff, else
    the primary alone; only supplementary pieces -> dropped; last_pos_base = min), -X."""
    rng = np.random.default_rng(5)
    clen = 60000
    ref = synth_bam.BASES[rng.integers(0, 4, size=clen)].copy()
    snp_pos = 1000 + 500 * np.arange(100)                                   # SNP i+1 at 0-based position snp_pos[i]
    nxt = {65: 67, 67: 71, 71: 84, 84: 65}                                    # A->C->G->T->A
    alt = np.array([nxt[int(ref[q])] for q in snp_pos], np.uint8)
    alt2 = np.array([nxt[int(a)] for a in alt], np.uint8)                     # second ALT of the multi-allelic record (SNP 21)
    third = np.array([nxt[int(a)] for a in alt2], np.uint8)                   # a base that is no allele of a biallelic record
    prefix = str(tmp_path / "h")
    with open(prefix + ".fa", "w") as f:
        f.write(">c\n" + bytes(ref).decode() + "\n")
    with open(prefix + ".vcf", "w") as f:
        f.write("##fileformat=VCFv4.2\n##contig=<ID=c,length=%d>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts\n" % clen)
        for i, q in enumerate(snp_pos):
            a = chr(alt[i]) + ("," + chr(alt2[i]) if i == 20 else "")
            f.write(f"c\t{q + 1}\t.\t{chr(ref[q])}\t{a}\t50\tPASS\t.\tGT\t0/1\n")

    def seq_of(beg, end, calls):                                              # reference bases with `calls` {snp index: base}
        s = ref[beg:end].copy()
        for i, b in calls.items():
            s[snp_pos[i] - beg] = b
        return s
    recs = []

    def add(name, pos, flag, mapq, cigar, seq):
        recs.append((pos, synth_bam.bam_record(0, pos, name, flag, mapq, cigar, bytes(seq), np.full(len(seq), 30, np.uint8))))
    # A plain: SNPs 1..6 (positions 1000..3500), ALT everywhere
    add("A_plain", 900, 0, 60, [("M", 3000)], seq_of(900, 3900, {i: alt[i] for i in range(6)}))
    # B =/X ops and a soft clip: same span, REF at SNP 1, ALT at 2..6
    add("B_eqx", 900, 0, 60, [("S", 7), ("=", 100), ("X", 1), ("=", 2899)], np.concatenate([synth_bam.BASES[rng.integers(0, 4, size=7)], seq_of(900, 3900, {i: alt[i] for i in range(1, 6)})]))
    # C a 2000-base N skip: 5900..6899 aligned (SNPs 11, 12), 6900..8899 skipped (13..16), 8900..9899 aligned (SNPs 17, 18)
    add("C_skipN", 5900, 0, 60, [("M", 1000), ("N", 2000), ("M", 1000)], np.concatenate([seq_of(5900, 6900, {10: alt[10], 11: alt[11]}), seq_of(8900, 9900, {16: alt[16], 17: alt[17]})]))
    # D primary (SNPs 41..43) + supplementary 3000 bases on (SNPs 49, 50; hard-clipped): merged
    add("D_supp_near", 20900, 0, 60, [("M", 1200), ("S", 700)], np.concatenate([seq_of(20900, 22100, {40: alt[40], 41: alt[41], 42: alt[42]}), seq_of(24900, 25600, {})]))
    add("D_supp_near", 24900, 2048, 60, [("H", 1200), ("M", 700)], seq_of(24900, 25600, {48: alt[48], 49: alt[49]}))
    # E primary (SNPs 61, 62) + supplementary 29 500 bases before it (SNPs 1, 2): beyond --supp-aln-dist-cutoff 10000 -> the primary alone
    add("E_supp_far", 30900, 0, 60, [("M", 700)], seq_of(30900, 31600, {60: alt[60], 61: alt[61]}))
    add("E_supp_far", 900, 2048, 60, [("M", 700)], seq_of(900, 1600, {0: alt[0], 1: alt[1]}))
    # F only the supplementary piece survives (the primary has MAPQ 5): dropped
    add("F_supp_only", 40900, 0, 5, [("M", 700)], seq_of(40900, 41600, {80: alt[80]}))
    add("F_supp_only", 42900, 2048, 60, [("M", 700)], seq_of(42900, 43600, {84: alt[84]}))
    # G supplementary below MAPQ 60 is ignored: the primary alone
    add("G_supp_lowq", 44900, 0, 60, [("M", 700)], seq_of(44900, 45600, {88: alt[88], 89: alt[89]}))
    add("G_supp_lowq", 46900, 2048, 59, [("M", 700)], seq_of(46900, 47600, {92: alt[92]}))
    # H secondary, I duplicate, I2 unmapped flag: filtered
    add("H_secondary", 900, 256, 60, [("M", 700)], seq_of(900, 1600, {0: alt[0]}))
    add("I_duplicate", 900, 1024, 60, [("M", 700)], seq_of(900, 1600, {0: alt[0]}))
    add("I2_qcfail", 900, 512, 60, [("M", 700)], seq_of(900, 1600, {0: alt[0]}))
    # J multi-allelic record (SNP 21: REF, ALT1, ALT2): second ALT -> genotype 2; K a base that is no allele -> no call at SNP 23
    add("J_multi", 10900, 0, 60, [("M", 1200)], seq_of(10900, 12100, {20: alt2[20], 21: alt[21], 22: ref[snp_pos[22]]}))
    add("K_nocall", 11900, 0, 60, [("M", 700)], seq_of(11900, 12600, {22: third[22], 23: alt[23]}))
    # L covers no SNP: goes to the reads without SNPs
    add("L_snpless", 1100, 0, 60, [("M", 300)], seq_of(1100, 1400, {}))
    # M deletion across SNP 31 and an insertion before SNP 32: D removes the call, I shifts the read offset
    s_m = seq_of(15900, 17100, {30: alt[30], 31: alt[31], 32: alt[32]})
    s_m = np.concatenate([s_m[:100], s_m[101:300], synth_bam.BASES[rng.integers(0, 4, size=4)], s_m[300:]])       # delete offset 100 (= SNP 31 at 16000), insert 4 at offset 300
    add("M_indels", 15900, 0, 60, [("M", 100), ("D", 1), ("M", 199), ("I", 4), ("M", 900)], s_m)
    recs.sort(key=lambda t: t[0])
    synth_bam.write_bam(prefix + ".bam", [("c", clen)], [r for _, r in recs])

    def run(extra):
        got, _ = ingest(floria_hip, prefix, tmp_path, extra=("--snp-count-filter", "10", "--no-realign", "--supp-aln-dist-cutoff", "10000") + extra)
        return {g["name"]: g for g in got["c"]["reads"]}, [x[0] for x in got["c"]["snpless"]]
    got, snpless = run(())
    cells = {nm: [(c[0], c[1]) for c in g["cells"]] for nm, g in got.items()}
    assert sorted(got) == ["A_plain", "B_eqx", "C_skipN", "D_supp_near", "E_supp_far", "G_supp_lowq", "J_multi", "K_nocall", "M_indels"]
    assert snpless == ["L_snpless"]
    assert cells["A_plain"] == [(i, 1) for i in range(1, 7)]
    assert cells["B_eqx"] == [(1, 0)] + [(i, 1) for i in range(2, 7)]
    assert cells["C_skipN"] == [(11, 1), (12, 1), (17, 1), (18, 1)]
    assert cells["D_supp_near"] == [(41, 1), (42, 1), (43, 1), (49, 1), (50, 1)]
    assert got["D_supp_near"]["span"] == (20900, 22100)                       # first_pos_base = min, last_pos_base = min (:635-636)
    from oracle import oracle as _orc                                         # the merged set: the primary's positions extended by the supplementary piece's (:639)
    assert got["D_supp_near"]["order"] == _orc.positions_order([np.asarray([41, 42, 43], np.uint32), np.asarray([49, 50], np.uint32)]).tolist()
    assert "order" not in got["A_plain"] and "order" not in got["E_supp_far"]
    assert cells["E_supp_far"] == [(61, 1), (62, 1)]
    assert cells["G_supp_lowq"] == [(89, 1), (90, 1)]
    assert cells["J_multi"] == [(21, 2), (22, 1), (23, 0)]
    assert cells["K_nocall"] == [(24, 1)]
    assert cells["M_indels"] == [(32, 1), (33, 1)]
    assert all(c[2] == 30 for g in got.values() for c in g["cells"])
    # -X: supplementary alignments are not used at all
    got_x, _ = run(("-X",))
    assert [(c[0], c[1]) for c in got_x["D_supp_near"]["cells"]] == [(41, 1), (42, 1), (43, 1)] and got_x["D_supp_near"]["span"] == (20900, 22100)
    # default cutoff 40000: E's pieces are 29 500 bases apart -> merged
    got_d, _ = ingest(floria_hip, prefix, tmp_path, extra=("--snp-count-filter", "10", "--no-realign"))
    e = {g["name"]: g for g in got_d["c"]["reads"]}["E_supp_far"]
    assert [(c[0], c[1]) for c in e["cells"]] == [(1, 1), (2, 1), (61, 1), (62, 1)] and e["span"] == (900, 1600)


def test_flatmap_behaves_like_std_map():
    # the Frag maps are sorted vectors with std::map's interface (floria_host.hpp: FlatMap): random operations against std::map
    cpp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp")
    subprocess.check_call(["make", "-C", cpp, "-B", "flatmap_test"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(cpp, "flatmap_test")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "OK", r.stdout + r.stderr


def test_lpt_dealing_of_contigs_to_devices_matches_the_bench_queue(floria_hip):
    floria_hip_bin = floria_hip
    # floria-hip --devices deals the contigs of a batch with the rule bench.py broadcasts over RCCL (floria_amd/shard.py: longest first, least
    # loaded device, first minimum on ties); every item gets exactly one device and the loads are balanced to within the largest item
    import subprocess
    from floria_amd import shard
    rng = np.random.default_rng(3)
    for world, n in ((1, 5), (2, 7), (8, 100), (8, 3), (3, 0), (5, 64)):
        costs = rng.integers(1, 50, size=n).astype(float)
        if n > 10:
            costs[::7] = costs[0]                               # ties
        r = subprocess.run([floria_hip_bin, "--lpt-assign", str(world)] + [repr(float(c)) for c in costs], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.array([int(x) for x in r.stdout.split()], np.int32)
        want = shard.lpt_assign(costs, world)
        assert np.array_equal(got, want), (world, n)
        if n:
            load = np.bincount(got, weights=costs, minlength=world)
            assert load.max() - load.min() <= costs.max() + 1e-9 or n < world


def test_vcf_parser_on_the_reference_fixture(floria_hip, hip_lib):
    # tests/golden/test.vcf is the reference's own tests/test.vcf (Longshot 0.4.0 output for NZ_CP081897.1, long INFO fields, FORMAT/sample columns): a file
    # this repository did not write.  The C++ reader (ingest.cpp: get_vcf_profile, file_reader.rs:239-314) must find its 954 SNPs at the positions an
    # independent parse of the text gives (0-based, as rust-htslib's record.pos()), with REF + ALT as the allele list, and get_range_with_lengths must cut
    # them into 17 / 33 / 176 blocks at -l 10000 / 5000 / 500.
    import subprocess
    vcf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test.vcf")
    r = subprocess.run([floria_hip, "--vcf-profile", vcf, "NZ_CP081897.1", "some_other_contig"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().split("\n")
    assert lines[0] == "#NZ_CP081897.1\t954" and len(lines) == 955
    got_pos = np.array([int(l.split("\t")[0]) for l in lines[1:]], np.int64)
    got_al = [l.split("\t")[1] for l in lines[1:]]
    want_pos, want_al = [], []
    for l in open(vcf):
        if l.startswith("#"):
            continue
        f = l.rstrip("\n").split("\t")
        alts = f[4].split(",")
        if len(f[3]) == 1 and all(len(a) == 1 and a in "ACGT" for a in alts) and f[3] in "ACGT":          # the SNP filter of file_reader.rs:290-300
            want_pos.append(int(f[1]) - 1); want_al.append(f[3] + "".join(alts))
    assert np.array_equal(got_pos, np.array(want_pos)) and got_al == want_al
    assert np.array_equal(got_pos + 1, np.load(os.path.join(os.path.dirname(vcf), "test_vcf_positions.npy")))      # (the .npy holds the 1-based POS column)
    assert [len(hip_lib.get_range_with_lengths(got_pos, L)[0]) for L in (10000, 5000, 500)] == [17, 33, 176]


# ---- a BAM written byte by byte from the SAM/BAM specification (SAMv1 §4.1 BGZF, §4.2 BAM), by nothing of floria_amd/synth_bam.py -----------------------
def _bgzf_member(payload, level=6):
    import struct, zlib
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = co.compress(payload) + co.flush()
    bsize = 12 + 6 + len(cdata) + 8 - 1                                     # total block size minus 1
    assert bsize < 65536
    return (b"\x1f\x8b\x08\x04" + struct.pack("<IBBH", 0, 0, 255, 6) + b"BC" + struct.pack("<HH", 2, bsize) + cdata
            + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload)))


_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")        # the 28-byte EOF marker of the specification


def _bam_record(ref_id, pos, name, flag, mapq, cigar, seq, qual, tags=b"", real_cigar_in_cg=False):
    import struct
    ops = "MIDNSHP=X"
    enc = lambda cg: b"".join(struct.pack("<I", (n << 4) | ops.index(op)) for op, n in cg)
    l_seq = len(seq)
    ref_len = sum(n for op, n in cigar if op in "MDN=X")
    field_cigar = cigar
    if real_cigar_in_cg:                                                     # > 65535 operations: placeholder in the field, the real CIGAR in CG:B,I
        field_cigar = [("S", l_seq), ("N", ref_len)]
        tags = tags + b"CGBI" + struct.pack("<I", len(cigar)) + enc(cigar)
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    nib = [code[c] for c in seq] + ([0] if l_seq % 2 else [])
    packed = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(nib), 2))
    end = pos + max(ref_len, 1)
    def reg2bin(b, e):                                                       # SAMv1 §5.3
        e -= 1
        for sh, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
            if b >> sh == e >> sh:
                return off + (b >> sh)
        return 0
    body = struct.pack("<iiBBHHHIiii", ref_id, pos, len(name) + 1, mapq, reg2bin(max(pos, 0), max(end, 1)) if ref_id >= 0 else 4680, len(field_cigar), flag, l_seq, -1, -1, 0)
    body += name.encode() + b"\0" + enc(field_cigar) + packed + bytes(qual) + tags
    return struct.pack("<I", len(body)) + body


def test_bam_assembled_byte_by_byte_from_the_specification(floria_hip, tmp_path):
    """The ingest faces a file nothing of this repository's BAM writer produced: BGZF members cut at arbitrary places (inside the header, inside a
    record, a stored level-0 member, an EMPTY member in mid-file, the EOF marker), header text and two references, records with auxiliary tags (A, i,
    Z, B arrays) behind the qualities, the operations = X N P D I S H, an unmapped and a secondary record, an N base on a SNP, and an alignment with
    70 000 CIGAR operations (placeholder CIGAR + CG:B,I tag, SAMv1 §4.2.2).  Expectations are derived by hand from file_reader.rs:184-235, 661-736."""
    import struct
    L1, L2 = 120000, 3000
    rng = np.random.default_rng(77)
    ref1 = "".join("ACGT"[i] for i in rng.integers(0, 4, size=L1)); ref2 = "".join("ACGT"[i] for i in rng.integers(0, 4, size=L2))
    nxt = {"A": "C", "C": "G", "G": "T", "T": "A"}
    snps1 = [1000 + 400 * i for i in range(110)]                             # 0-based positions on c1: SNP k+1 at snps1[k] (1000 .. 44600)
    snps2 = [100 + 50 * i for i in range(40)]
    prefix = str(tmp_path / "spec")
    open(prefix + ".fa", "w").write(f">c1\n{ref1}\n>c2\n{ref2}\n")
    with open(prefix + ".vcf", "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
        for q in snps1:
            f.write(f"c1\t{q + 1}\t.\t{ref1[q]}\t{nxt[ref1[q]]}\t50\tPASS\tDP=10\n")
        for q in snps2:
            f.write(f"c2\t{q + 1}\t.\t{ref2[q]}\t{nxt[ref2[q]]}\t50\tPASS\tDP=10\n")

    def with_alt(ref, beg, end, alt_positions):
        s = list(ref[beg:end])
        for q in alt_positions:
            s[q - beg] = nxt[ref[q]]
        return "".join(s)
    tags = b"NMC\x03" + b"XAAx" + b"MDZ10A5\0" + b"ZBBs" + struct.pack("<Ihh", 2, -3, 7) + b"ASi" + struct.pack("<i", 1234)
    recs = []
    # R1  5S 150= 1X 100= 3P 200M : reference 900..1351; SNP 1 (1000) sits in the `=` run with the ALT base (the op letter is not what decides), SNP 2 (1400) not reached
    s1 = "ACGTA" + with_alt(ref1, 900, 1351, [1000])
    recs.append(_bam_record(0, 900, "R1_eqxp", 0, 60, [("S", 5), ("=", 150), ("X", 1), ("=", 100), ("P", 3), ("M", 200)], s1, [30] * len(s1), tags))
    # R2  300M 1200N 300M : 1300..1599 aligned (SNP 2 at 1400, REF), 1600..2799 skipped (SNPs 3..5), 2800..3099 aligned (SNP 6 at 3000, ALT)
    s2 = with_alt(ref1, 1300, 1600, []) + with_alt(ref1, 2800, 3100, [3000])
    recs.append(_bam_record(0, 1300, "R2_skip", 16, 60, [("M", 300), ("N", 1200), ("M", 300)], s2, [30] * len(s2)))
    # R3  deletion over SNP 8 (3800), insertion of 2 before SNP 9 (4200): 3700..4399; calls at SNP 9 only... and SNP 8 lost
    body = with_alt(ref1, 3700, 4400, [4200])
    s3 = body[:100] + body[101:300] + "GG" + body[300:]                       # D at offset 100 (= 3800), I after offset 300
    recs.append(_bam_record(0, 3700, "R3_indel", 0, 60, [("M", 100), ("D", 1), ("M", 199), ("I", 2), ("M", 400)], s3, [30] * len(s3), b"NMC\x03"))
    # R4  an N base on SNP 10 (4600) -> no call there; SNP 11 (5000) ALT
    s4 = list(with_alt(ref1, 4500, 5100, [5000])); s4[100] = "N"
    recs.append(_bam_record(0, 4500, "R4_nbase", 0, 60, [("M", 600)], "".join(s4), [30] * 600))
    # R5  secondary (filtered), R6 MAPQ 3 (filtered)
    recs.append(_bam_record(0, 4500, "R5_secondary", 256, 60, [("M", 600)], with_alt(ref1, 4500, 5100, [4600]), [30] * 600))
    recs.append(_bam_record(0, 4500, "R6_lowq", 0, 3, [("M", 600)], with_alt(ref1, 4500, 5100, [4600]), [30] * 600))
    # R7  70 000 operations: 35 000 x (1M 1I) from 9000: reference 9000..43999, read base of reference position p at offset 2 (p - 9000); ALT at SNPs 21 (9000) .. 108 (43800),
    #     every fourth one REF
    alt_at = [q for k, q in enumerate(snps1) if 9000 <= q < 44000 and k % 4]
    r7 = with_alt(ref1, 9000, 44000, alt_at)
    s7 = "".join(b + "T" for b in r7)
    recs.append(_bam_record(0, 9000, "R7_longcigar", 0, 60, [("M", 1), ("I", 1)] * 35000, s7, [30] * len(s7), b"NMC\x01", real_cigar_in_cg=True))
    # R8  on the second reference: SNPs 1..4 of c2 (100, 150, 200, 250), ALT at 150
    recs.append(_bam_record(1, 90, "R8_c2", 0, 60, [("M", 200)], with_alt(ref2, 90, 290, [150]), [30] * 200))
    # R9  unmapped (flag 4, no reference)
    recs.append(_bam_record(-1, -1, "R9_unmapped", 4, 0, [], "ACGTACGT", [20] * 8))
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:%d\n@SQ\tSN:c2\tLN:%d\n@PG\tID:byhand\n" % (L1, L2)
    stream = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 2)
    for nm, ln in ((b"c1", L1), (b"c2", L2)):
        stream += struct.pack("<I", len(nm) + 1) + nm + b"\0" + struct.pack("<I", ln)
    stream += b"".join(recs)
    cuts = [0, 3, 61, 200, 200, 5000, 5003, 60000, 120000, len(stream)]       # inside the magic, the header text, records; an empty member (200, 200)
    with open(prefix + ".bam", "wb") as f:
        for k in range(len(cuts) - 1):
            f.write(_bgzf_member(stream[cuts[k]:cuts[k + 1]], level=0 if k == 2 else 6))
        f.write(_BGZF_EOF)
    got, err = ingest(floria_hip, prefix, tmp_path, extra=("--snp-count-filter", "10", "--no-realign", "-t", "3"))
    c1 = {g["name"]: g for g in got["c1"]["reads"]}
    c2 = {g["name"]: g for g in got["c2"]["reads"]}
    cells = lambda g: [(c[0], c[1]) for c in g["cells"]]
    assert sorted(c1) == ["R1_eqxp", "R2_skip", "R3_indel", "R4_nbase", "R7_longcigar"] and sorted(c2) == ["R8_c2"]
    assert cells(c1["R1_eqxp"]) == [(1, 1)]
    assert cells(c1["R2_skip"]) == [(2, 0), (6, 1)]
    assert cells(c1["R3_indel"]) == [(9, 1)]
    assert cells(c1["R4_nbase"]) == [(11, 1)]
    want7 = [(k + 1, 1 if k % 4 else 0) for k, q in enumerate(snps1) if 9000 <= q < 44000]
    assert cells(c1["R7_longcigar"]) == want7 and len(want7) == 88
    assert c1["R7_longcigar"]["span"][0] == 9000
    assert cells(c2["R8_c2"]) == [(1, 0), (2, 1), (3, 0), (4, 0)]
    assert all(c[2] == 30 for g in list(c1.values()) + list(c2.values()) for c in g["cells"])


def test_bam_is_streamed_in_segments_of_complete_contigs(floria_hip, tmp_path):
    # BamStream: the file is mapped and inflated a window at a time; a segment holds every record of a run of complete contigs (the carry-over of a
    # contig that is still arriving goes to the next one).  Twelve contigs, one of them without reads in mid-file, windows from 16 KiB (several segments,
    # contigs larger than the window) to the default: identical Frags, every contig reported once, and the largest buffer shrinks with the window.
    import re
    cs = [synth.make_config_contig(4, 60 + i, 0.15 + 0.02 * i, keep_layout=True) for i in range(11)]
    prefix = str(tmp_path / "st")
    synth_bam.write_dataset(prefix, cs, seed=3)
    # a twelfth contig in the header and the VCF whose reads are missing: splice a reference without records between the others
    dumps, segs, peaks = [], [], []
    for kb in (16, 2000, 0):
        extra = ("--snp-count-filter", "20", "-t", "4") + (("--bam-window-kb", str(kb)) if kb else ())
        got, err = ingest(floria_hip, prefix, tmp_path, extra=extra)
        m = re.search(r"BAM: (\d+) records in (\d+) segments, .* largest inflated buffer (\d+) MiB", err)
        assert m, err
        dumps.append(got); segs.append(int(m.group(2))); peaks.append(int(m.group(3)))
        assert err.count("Number of reads passing filtering") == len(cs)
    assert segs[0] > 1 and segs[0] >= segs[1] >= segs[2] == 1 and segs[0] > segs[2]
    assert dumps[0] == dumps[2] and dumps[1] == dumps[2]
    assert peaks[0] <= peaks[2]
    assert sorted(dumps[2]) == sorted(c.name for c in cs)


def test_unsorted_bam_is_refused(floria_hip, tmp_path):
    # the reference fetches per contig through the .bai, i.e. needs a coordinate-sorted file; the streaming reader relies on the same order and says so
    cs = [synth.make_config_contig(4, 80 + i, 0.1, keep_layout=True) for i in range(2)]
    prefix = str(tmp_path / "us")
    ex = synth_bam.write_dataset(prefix, cs, seed=3)
    # rewrite the BAM with the two contigs' records swapped (contig 1's records first)
    import gzip, struct
    raw = gzip.open(prefix + ".bam", "rb").read()
    l_text = struct.unpack_from("<I", raw, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<I", raw, o)[0]; o += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<I", raw, o)[0]; o += 4 + ln + 4
    head, recs = raw[:o], []
    while o < len(raw):
        bs = struct.unpack_from("<I", raw, o)[0]
        recs.append((struct.unpack_from("<i", raw, o + 4)[0], raw[o:o + 4 + bs])); o += 4 + bs
    swapped = head + b"".join(r for t, r in recs if t == 1) + b"".join(r for t, r in recs if t == 0)
    with open(prefix + ".bam", "wb") as f:
        for k in range(0, len(swapped), 60000):
            f.write(_bgzf_member(swapped[k:k + 60000]))
        f.write(_BGZF_EOF)
    r = subprocess.run([floria_hip, "-b", prefix + ".bam", "-v", prefix + ".vcf", "-r", prefix + ".fa", "-o", str(tmp_path / "o"), "-e", "0.03", "-l", "10000", "--ingest-only"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "not sorted by reference sequence" in r.stderr


def test_unmapped_tail_larger_than_the_window_ends_the_stream(floria_hip, tmp_path):
    # a sorted BAM ends in its unplaced reads (tid = -1); when they outgrow the window BamStream::next used to hand the same empty segment
    # out for ever.  50 mapped + 3000 unplaced records with a 1-KiB window (main loop, and the -e / -l estimator loop when neither flag is given) must finish,
    # give the Frags of the same file without the tail, and the same with the default window.
    import struct
    rng = np.random.default_rng(5)
    L1 = 20000
    ref1 = "".join("ACGT"[i] for i in rng.integers(0, 4, size=L1))
    nxt = {"A": "C", "C": "G", "G": "T", "T": "A"}
    snps = [500 + 150 * i for i in range(120)]
    def files(prefix, n_unmapped):
        open(prefix + ".fa", "w").write(f">c1\n{ref1}\n")
        with open(prefix + ".vcf", "w") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
            for q in snps:
                f.write(f"c1\t{q + 1}\t.\t{ref1[q]}\t{nxt[ref1[q]]}\t50\tPASS\tDP=10\n")
        recs = []
        for r in range(50):
            beg = 200 * r
            s = list(ref1[beg:beg + 3000])
            for q in snps:
                if beg <= q < beg + 3000 and (r + q) % 3 == 0:
                    s[q - beg] = nxt[ref1[q]]
            recs.append(_bam_record(0, beg, f"m{r}", 0, 60, [("M", 3000)], "".join(s), [30] * 3000))
        for r in range(n_unmapped):
            recs.append(_bam_record(-1, -1, f"u{r}", 4, 0, [], "ACGTACGTAC" * 10, [20] * 100))
        text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:%d\n" % L1
        stream = b"BAM\1" + struct.pack("<I", len(text)) + text + struct.pack("<I", 1) + struct.pack("<I", 3) + b"c1\0" + struct.pack("<I", L1) + b"".join(recs)
        with open(prefix + ".bam", "wb") as f:
            for k in range(0, len(stream), 50000):
                f.write(_bgzf_member(stream[k:k + 50000]))
            f.write(_BGZF_EOF)
    pa, pb = str(tmp_path / "tail"), str(tmp_path / "notail")
    files(pa, 3000); files(pb, 0)
    def run(prefix, extra):
        dump = prefix + ".dump"
        r = subprocess.run([floria_hip, "-b", prefix + ".bam", "-v", prefix + ".vcf", "-r", prefix + ".fa", "-o", prefix + "_out", "--ingest-only", "--dump-frags", dump,
                            "--snp-count-filter", "10", "--no-realign", "-t", "2", *extra], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        return open(dump).read()
    base = run(pb, ("-e", "0.03", "-l", "10000"))
    assert base.count("m") >= 50
    assert run(pa, ("-e", "0.03", "-l", "10000", "--bam-window-kb", "1")) == base
    assert run(pa, ("-e", "0.03", "-l", "10000")) == base
    assert run(pa, ("--bam-window-kb", "1")) == run(pb, ())          # the estimator loop walks the same stream


def test_bcf2_input_assembled_from_the_specification(floria_hip, tmp_path):
    """rust-htslib's bcf::Reader (file_reader.rs:239-314) reads binary BCF2 as well as text VCF; so does ingest.cpp since round 5.  The file is assembled
    byte by byte from the VCF specification's BCF2 chapter inside this test (by nothing of floria_amd/): BGZF members cut inside the header text and inside a
    record, a contig dictionary given out of order with IDX=, typed strings with the 15+ length escape (a long ID, a long insertion allele), records with INFO
    and genotype blocks that must be skipped by their lengths, a multi-allelic SNP, an indel and a symbolic allele (dropped by the SNP filter), ALT '.'.
    Expected: exactly what the text VCF with the same records gives."""
    import struct
    recs = [("ctgB", 10, "rs1", "A", ["G"]), ("ctgB", 25, "x" * 40, "C", ["T", "G"]), ("ctgB", 31, ".", "AT", ["A"]), ("ctgB", 40, ".", "G", ["<DEL>"]),
            ("ctgB", 57, ".", "T", []), ("ctgA", 5, ".", "g", ["a"]), ("ctgA", 9, ".", "C", ["C" + "ACGT" * 8]), ("ctgA", 1200, "id2", "T", ["C"])]
    text = ("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##contig=<ID=ctgA,length=5000,IDX=1>\n##contig=<ID=ctgB,length=100,IDX=0>\n"
            "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\n")
    body = "".join(f"{c}\t{p}\t{i}\t{r}\t{','.join(a) if a else '.'}\t30\tPASS\tDP=7\tGT\t0/1\n" for c, p, i, r, a in recs)
    vcf = str(tmp_path / "x.vcf"); open(vcf, "w").write(text + body)

    def tstr(b):                                   # typed string: descriptor (len << 4 | 7), lengths >= 15 as 0xF7 + a typed int8 / int16
        b = b.encode()
        if len(b) < 15:
            return bytes([len(b) << 4 | 7]) + b
        return bytes([0xF7]) + (bytes([0x11, len(b)]) if len(b) < 128 else bytes([0x12]) + struct.pack("<h", len(b))) + b
    cid = {"ctgB": 0, "ctgA": 1}
    out = b"BCF\x02\x02" + struct.pack("<I", len(text) + 1) + text.encode() + b"\0"
    for c, p, i, r, a in recs:
        shared = struct.pack("<iiifII", cid[c], p - 1, len(r), 30.0, (1 + len(a)) << 16 | 1, 1 << 24 | 1)
        shared += (tstr(i) if i != "." else bytes([0x07])) + tstr(r) + b"".join(tstr(x) for x in a)
        shared += bytes([0x11, 0x00])                                   # FILTER: one int8, PASS = 0
        shared += bytes([0x11, 0x01, 0x11, 0x07])                       # INFO: key 1 (DP) = int8 7
        indiv = bytes([0x11, 0x02, 0x21, 0x02, 0x04])                   # FORMAT key 2 (GT): two int8 per sample: 0/1
        out += struct.pack("<II", len(shared), len(indiv)) + shared + indiv
    cuts = [0, 3, 200, len(out) - 37, len(out) - 11, len(out)]        # inside the magic, the header text, the last records
    bcf = str(tmp_path / "x.bcf")
    open(bcf, "wb").write(b"".join(_bgzf_member(out[a:b]) for a, b in zip(cuts[:-1], cuts[1:])) + _BGZF_EOF)
    got = {}
    for name, path in (("vcf", vcf), ("bcf", bcf)):
        r = subprocess.run([floria_hip, "--vcf-profile", path, "ctgA", "ctgB"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got[name] = r.stdout
    assert got["bcf"] == got["vcf"]
    assert got["vcf"].split("\n")[:4] == ["#ctgA\t2", "4\tga", "1199\tTC", "#ctgB\t3"] and "24\tCTG" in got["vcf"] and "56\tT" in got["vcf"]
    # a truncated file is an error, not a shorter list
    open(bcf, "wb").write(_bgzf_member(out[:len(out) - 9]) + _BGZF_EOF)
    r = subprocess.run([floria_hip, "--vcf-profile", bcf, "ctgA", "ctgB"], capture_output=True, text=True)
    assert r.returncode != 0 and "truncated" in r.stderr
