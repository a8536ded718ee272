"""Synthetic BAM + VCF + FASTA whose pileup is known exactly: the alignment-level counterpart of synth.py.

There is no htslib / pysam in this image and the reference's quick-start BAM is missing, so the ingest path of
`floria-hip` (floria_amd/host/ingest.cpp) is exercised with files written here: a contig of synth.make_contig(keep_layout=True)
is turned into

  * a random reference sequence, REF = its base at every SNP position, ALT = another base;
  * one alignment per read (two for a short-read pair) whose sequence is the reference with, at every covered SNP, the base
    the pileup cell says (allele 0 -> REF, 1 -> ALT) or a THIRD base where the synthetic read has no call;  base quality = the
    cell's quality;  a tenth of the long reads additionally carry a soft clip, an insertion and a deletion away from SNPs
    (the CIGAR walk must step over them) and a deletion across one SNP (that call disappears);
  * a coordinate-sorted BAM (BGZF blocks written with zlib), a VCF and a FASTA.

write_dataset returns the pileup a correct ingest must produce (reads in Frag::cmp order with the BAM record index as the
tie-break, floria.rs:289-293) together with read names and reference spans.
"""
import struct
import zlib

import numpy as np

from .pileup import Pileup

BASES = np.frombuffer(b"ACGT", np.uint8)


def _bgzf_block(data: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    bsize = len(cdata) + 25                                    # total block size - 1
    return (struct.pack("<4BI2BH", 31, 139, 8, 4, 0, 0, 255, 6) + struct.pack("<2BHH", 66, 67, 2, bsize) + cdata
            + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


_CIG = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
_NT16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_NT16_LUT = np.full(256, 15, np.uint8)
for _c, _i in _NT16.items():
    _NT16_LUT[ord(_c)] = _i


def bam_record(tid, pos, name, flag, mapq, cigar, seq, qual, next_pos=-1, tlen=0):
    """cigar: list of (op char, len); seq: bytes of ACGTN; qual: uint8 array."""
    ref_len = sum(n for op, n in cigar if op in "MDN=X")
    nm = name.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (n << 4) | _CIG[op]) for op, n in cigar)
    codes = _NT16_LUT[np.frombuffer(seq, np.uint8)]
    if len(codes) % 2:
        codes = np.append(codes, np.uint8(0))
    packed = ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)
    body = (struct.pack("<iiBBHHHIiii", tid, pos, len(nm), mapq, _reg2bin(pos, pos + max(1, ref_len)), len(cigar), flag, len(seq),
                        tid if next_pos >= 0 else -1, next_pos, tlen) + nm + cig + packed.tobytes() + bytes(np.asarray(qual, np.uint8)))
    return struct.pack("<I", len(body)) + body


def write_bam(path, targets, records):
    """targets: [(name, length)]; records: already coordinate-sorted list of bam_record() bytes."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{ln}\n" for n, ln in targets)
    head = b"BAM\1" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(targets))
    for n, ln in targets:
        head += struct.pack("<I", len(n) + 1) + n.encode() + b"\0" + struct.pack("<I", ln)
    raw = head + b"".join(records)
    with open(path, "wb") as f:
        for o in range(0, len(raw), 60000):
            f.write(_bgzf_block(raw[o:o + 60000]))
        f.write(_bgzf_block(b""))                               # EOF marker


def contig_dataset(contig, rng, edit_frac=0.1, mapq=60, sub_rate=0.0):
    """-> dict(ref=bytes, snps=[(pos0, REF, ALT)], records=[(pos, record bytes)], reads=[(name, [(snp, allele, qual)], span)])
    for one synth Contig made with keep_layout=True.  `reads` follows the BAM record order of the FIRST alignment of each read."""
    lay, p = contig.layout, contig.pileup
    clen = int(lay["contig_len"])
    snp_pos = contig.snp_pos.astype(np.int64)                   # 0-based genome position of SNP i (index i <-> SNP i + 1)
    ref = BASES[rng.integers(0, 4, size=clen)].copy()
    alt = np.array([BASES[(int(np.nonzero(BASES == ref[q])[0][0]) + 1 + int(rng.integers(0, 3))) % 4] for q in snp_pos], np.uint8)
    other = np.zeros(len(snp_pos), np.uint8)                    # a base that is neither REF nor ALT
    for i, q in enumerate(snp_pos):
        other[i] = [b for b in BASES if b != ref[q] and b != alt[i]][0]
    paired = lay["kind"] == "short"
    out_reads, recs, qual_by_name, seg_by_name = [], [], {}, {}
    for r in range(p.n_reads):
        snps, als, quals = p.read(r)
        cell = {int(s) - 1: (int(a), int(q)) for s, a, q in zip(snps, als, quals)}          # SNP index -> (allele, qual)
        segs = [(int(lay["start"][r]), min(int(lay["end"][r]), clen))]
        if paired:
            segs.append((int(lay["start2"][r]), min(int(lay["end2"][r]), clen)))
        segs = [(min(b, clen - 1), e) for b, e in segs]       # (a mate that starts beyond the contig's last base is one base at its end: a record with CIGAR 1M needs one base of SEQ)
        name = f"{contig.name}_r{r}"
        lost = set()
        recs_r, quals_r, seg_snps = [], [], []
        for k, (b, e) in enumerate(segs):
            e = max(e, b + 1)
            seq = ref[b:e].copy()
            qual = np.full(e - b, 30, np.uint8)
            lo, hi = np.searchsorted(snp_pos, b, "left"), np.searchsorted(snp_pos, e, "left")
            for i in range(lo, hi):
                o = int(snp_pos[i]) - b
                if i in cell:
                    seq[o] = ref[snp_pos[i]] if cell[i][0] == 0 else alt[i]
                    qual[o] = cell[i][1]
                else:
                    seq[o] = other[i]
            if sub_rate > 0:                                    # sequencing errors between the SNPs: what realign's windows have to absorb
                hit = np.nonzero(rng.random(e - b) < sub_rate)[0]
                hit = hit[~np.isin(hit + b, snp_pos[lo:hi])]
                seq[hit] = BASES[(np.searchsorted(BASES, seq[hit]) + rng.integers(1, 4, size=len(hit))) % 4]
            cigar = [("M", e - b)]
            pos = b
            if not paired and rng.random() < edit_frac and e - b > 400 and hi - lo >= 3:
                # soft clip (5 bases), a 2-base insertion and a 3-base deletion between SNPs, and a 1-base deletion ACROSS one SNP
                mid = lo + (hi - lo) // 2
                gaps = [i for i in range(lo, hi - 1) if snp_pos[i + 1] - snp_pos[i] > 12 and snp_pos[i] - b > 10]
                if len(gaps) >= 2 and mid not in (lo, hi - 1):
                    g_ins, g_del = gaps[0], gaps[-1]
                    if g_ins < mid - 1 and g_del > mid:
                        x_ins = int(snp_pos[g_ins]) - b + 4             # insertion after this reference offset
                        x_snp = int(snp_pos[mid]) - b                   # deleted reference base (the SNP itself)
                        x_del = int(snp_pos[g_del]) - b + 4             # 3-base deletion starting here
                        ins = BASES[rng.integers(0, 4, size=2)]
                        new_seq = np.concatenate([BASES[rng.integers(0, 4, size=5)], seq[:x_ins], ins, seq[x_ins:x_snp], seq[x_snp + 1:x_del], seq[x_del + 3:]])
                        new_q = np.concatenate([np.full(5, 20, np.uint8), qual[:x_ins], np.full(2, 20, np.uint8), qual[x_ins:x_snp], qual[x_snp + 1:x_del], qual[x_del + 3:]])
                        cigar = [("S", 5), ("M", x_ins), ("I", 2), ("M", x_snp - x_ins), ("D", 1), ("M", x_del - x_snp - 1), ("D", 3), ("M", (e - b) - x_del - 3)]
                        seq, qual = new_seq, new_q
                        lost.add(mid)
            flag = 0
            if paired:
                flag = 1 | 2 | (64 | 32 if k == 0 else 128 | 16)
            recs_r.append((pos, bam_record(0, pos, name, flag, mapq, cigar, bytes(seq), qual), bytes(seq), cigar))
            quals_r.append(bytes(qual))
            seg_snps.append([i + 1 for i in range(lo, hi) if i in cell and i not in lost])      # the SNPs THIS alignment calls (its own seq_dict, file_reader.rs:702-727)
        cells = [(i + 1, a, q) for i, (a, q) in sorted(cell.items()) if i not in lost]
        span = (min(b for b, _ in segs), min(max(e, b + 1) for b, e in segs)) if paired else (segs[0][0], max(segs[0][1], segs[0][0] + 1))
        out_reads.append((name, cells, span, recs_r, sum(max(e, b + 1) - b for b, e in segs)))
        qual_by_name[name] = quals_r
        seg_by_name[name] = seg_snps
    return dict(segments=seg_by_name, quals=qual_by_name, ref=bytes(ref), snps=[(int(q), chr(ref[q]), chr(alt[i])) for i, q in enumerate(snp_pos)], reads=out_reads, contig_len=clen)


def nw_affine_batch(Q, R, match=1, mismatch=-1, gap_open=-2, gap_extend=-1, band=None):
    """Global alignment scores of N sequence pairs at once (Q, R: uint8 [N, n] / [N, m]); a gap of length k costs open + (k - 1) * extend
    (Gotoh).  The scoring of alignment::realign (alignment.rs:15-18: NW1, Gaps { open: -2, extend: -1 }).  With `band` = b only cells with
    |i - j| <= b are computed (a static diagonal band): NOT block-aligner's adaptive block walk, which cannot be restated without the crate, but
    a lower bound on what any aligner confined to about 2b + 1 diagonals can find — used to measure how much the calls depend on the band at all
    (tests/test_host_cpu.py::test_realign_calls_barely_depend_on_a_band)."""
    N, n = Q.shape
    m = R.shape[1]
    NEG = -10 ** 6
    jj = np.arange(m + 1)
    M = np.full((N, m + 1), NEG, np.int32); X = np.full((N, m + 1), NEG, np.int32); Y = np.full((N, m + 1), NEG, np.int32)
    M[:, 0] = 0
    Y[:, 1:] = gap_open + gap_extend * np.arange(m)
    if band is not None:
        Y[:, jj > band] = NEG
    for i in range(1, n + 1):
        Mn = np.full((N, m + 1), NEG, np.int32); Xn = np.full((N, m + 1), NEG, np.int32); Yn = np.full((N, m + 1), NEG, np.int32)
        Xn[:, 0] = gap_open + (i - 1) * gap_extend
        sub = np.where(Q[:, i - 1:i] == R, match, mismatch).astype(np.int32)
        Mn[:, 1:] = np.maximum(np.maximum(M[:, :-1], X[:, :-1]), Y[:, :-1]) + sub
        Xn[:, 1:] = np.maximum(np.maximum(M[:, 1:], Y[:, 1:]) + gap_open, X[:, 1:] + gap_extend)
        for j in range(1, m + 1):
            Yn[:, j] = np.maximum(np.maximum(Mn[:, j - 1], Xn[:, j - 1]) + gap_open, Yn[:, j - 1] + gap_extend)
            if band is not None and abs(i - j) > band:
                Mn[:, j] = NEG; Xn[:, j] = NEG; Yn[:, j] = NEG
        if band is not None and i > band:
            Xn[:, 0] = NEG
        M, X, Y = Mn, Xn, Yn
    return np.maximum(np.maximum(M[:, m], X[:, m]), Y[:, m])


def realign_windows(d, flank=16):
    """The (read window, REF window, ALT window) triples alignment::realign (alignment.rs:21-37) scores for contig_dataset `d`, as uint8 arrays
    [N, 2 * flank], and for each the (read index, cell index) it decides."""
    ref = np.frombuffer(d["ref"], np.uint8)
    snp_pos = np.array([q for q, _, _ in d["snps"]], np.int64)
    alleles = [(ord(r), ord(a)) for _, r, a in d["snps"]]
    wq, wr0, wr1, where = [], [], [], []
    for ri, (name, cells, span, recs_r, slen) in enumerate(d["reads"]):
        cell_ix = {c[0] - 1: k for k, c in enumerate(cells)}
        for pos, _rec, seq, cigar in recs_r:                      # a later alignment of the read overwrites an earlier call (extend)
            sq = np.frombuffer(seq, np.uint8)
            q, r = 0, pos
            for op, ln in cigar:
                if op == "M":
                    lo, hi = np.searchsorted(snp_pos, r, "left"), np.searchsorted(snp_pos, r + ln, "left")
                    for i in range(lo, hi):
                        if i not in cell_ix:
                            continue
                        gp = int(snp_pos[i]); qp = q + (gp - r)
                        if flank > gp or flank + gp >= len(ref) or flank > qp or flank + qp >= len(sq):
                            continue
                        w = ref[gp - flank:gp + flank].copy()
                        w0 = w.copy(); w0[flank] = alleles[i][0]
                        w1 = w.copy(); w1[flank] = alleles[i][1]
                        wq.append(sq[qp - flank:qp + flank]); wr0.append(w0); wr1.append(w1); where.append((ri, cell_ix[i]))
                if op in "MIS":
                    q += ln
                if op in "MDN":
                    r += ln
    return np.array(wq, np.uint8).reshape(-1, 2 * flank), np.array(wr0, np.uint8).reshape(-1, 2 * flank), np.array(wr1, np.uint8).reshape(-1, 2 * flank), where


def realign_dataset(d, flank=16):
    """What alignment::realign (alignment.rs:7-64) makes of the calls of contig_dataset `d`: every call whose 2 x 16-base windows fit is
    replaced by the allele whose reference window aligns best to the read's window (first best).  Updates d["reads"][...] cells in place."""
    Q, R0, R1, where = realign_windows(d, flank)
    if not len(where):
        return 0
    s0 = nw_affine_batch(Q, R0); s1 = nw_affine_batch(Q, R1)
    changed = 0
    for (ri, k), a0, a1 in zip(where, s0, s1):
        new = 0 if a0 >= a1 else 1
        name, cells, span, recs_r, slen = d["reads"][ri]
        if cells[k][1] != new:
            changed += 1
        cells[k] = (cells[k][0], new, cells[k][2])
    return changed


def write_dataset(prefix, contigs, seed=0, extra_vcf_lines=True, edit_frac=0.1, realign=True, sub_rate=0.0):
    """Write {prefix}.bam / .vcf / .fa for a list of synth Contigs (keep_layout=True).  With realign=False the returned pileups hold the
    calls as sequenced (floria-hip --no-realign).  Returns, per contig name,
    dict(pileup=Pileup in the order a correct ingest produces, names=[read name], spans=[(first_pos_base, last_pos_base)], segments=[[SNPs of alignment k]],
         snp_pos0=[0-based genome position of every SNP], contig_len, seq_len=[bases of every read])."""
    rng = np.random.default_rng(seed)
    targets, all_recs, expect = [], [], {}
    fa = open(prefix + ".fa", "w")
    vcf = open(prefix + ".vcf", "w")
    vcf.write("##fileformat=VCFv4.2\n")
    datasets = []
    for tid, c in enumerate(contigs):
        d = contig_dataset(c, rng, edit_frac=edit_frac, sub_rate=sub_rate)
        if realign:                # the pileup floria makes of these files when a reference FASTA is given (it always is): calls realigned
            d["realigned_calls"] = realign_dataset(d)
        datasets.append(d)
        targets.append((c.name, d["contig_len"]))
        vcf.write(f"##contig=<ID={c.name},length={d['contig_len']}>\n")
    vcf.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tsample\n")
    for tid, (c, d) in enumerate(zip(contigs, datasets)):
        fa.write(f">{c.name} synthetic\n")
        s = d["ref"].decode()
        for o in range(0, len(s), 60):
            fa.write(s[o:o + 60] + "\n")
        for k, (q, r, a) in enumerate(d["snps"]):
            vcf.write(f"{c.name}\t{q + 1}\t.\t{r}\t{a}\t50\tPASS\t.\tGT\t0/1\n")
            if extra_vcf_lines and k % 97 == 5 and k + 1 < len(d["snps"]) and d["snps"][k + 1][0] > q + 3:
                # records the SNP filter must skip (file_reader.rs:290-300): an indel, a symbolic / non-ACGT allele
                vcf.write(f"{c.name}\t{q + 2}\t.\t{s[q + 1]}{s[q + 2]}\t{s[q + 1]}\t50\tPASS\t.\tGT\t0/1\n")
                vcf.write(f"{c.name}\t{q + 3}\t.\t{s[q + 2]}\t*\t50\tPASS\t.\tGT\t0/1\n")
        # BAM records of this contig, coordinate-sorted (stable: ties keep read order); record index = position in this list
        flat = []
        for ri, (name, cells, span, recs_r, slen) in enumerate(d["reads"]):
            for k, (pos, rec, _seq, _cig) in enumerate(recs_r):
                flat.append((pos, ri, k, rec))
        flat.sort(key=lambda t: (t[0], t[1], t[2]))
        first_index = {}
        for idx, (pos, ri, k, rec) in enumerate(flat):
            if k == 0:
                first_index[ri] = idx                                   # counter_id of the merged Frag = record index of the first-in-pair / only alignment
            all_recs.append(struct.pack("<I", len(rec) - 4) + struct.pack("<i", tid) + rec[8:])   # patch refID
        order = sorted(range(len(d["reads"])), key=lambda ri: first_index[ri])
        reads = [d["reads"][ri] for ri in order]
        pile_reads = [([x[0] for x in cells], [x[1] for x in cells], [x[2] for x in cells]) for _, cells, _, _, _ in reads if cells]
        names = [nm for nm, cells, _, _, _ in reads if cells]
        spans = [sp for _, cells, sp, _, _ in reads if cells]
        slens = [sl for _, cells, _, _, sl in reads if cells]
        pile = Pileup.from_reads(pile_reads)                            # Frag::cmp with the input (= record) order as the tie-break
        # Pileup.from_reads sorted the reads: recover the permutation to carry names / spans along
        first = np.array([r[0][0] for r in pile_reads], np.int64)
        last = np.array([r[0][-1] for r in pile_reads], np.int64)
        perm = np.lexsort((np.arange(len(pile_reads)), -last, first))
        by_name = {nm: (recs_r, d["quals"][nm]) for nm, _, _, recs_r, _ in reads}
        expect[c.name] = dict(pileup=pile, names=[names[i] for i in perm], spans=[spans[i] for i in perm], seq_len=[slens[i] for i in perm],
                              paired=c.layout["kind"] == "short",
                              # per read (pileup order): the SNPs of every alignment in the order combine_frags merges them (first in pair, then its mate,
                              # file_reader.rs:504-541) - what Frag.positions is built from; a single alignment's list for unpaired reads
                              segments=[d["segments"][nm] for nm in (names[i] for i in perm)],
                              read_alignments={nm: [(pos, seq, cig, q) for (pos, _rec, seq, cig), q in zip(*by_name[nm])] for nm in by_name},
                              snp_pos0=np.array([q for q, _, _ in d["snps"]], np.uint64), contig_len=d["contig_len"],
                              snpless=[(nm, sp, sl) for nm, cells, sp, _, sl in reads if not cells],
                              alignments=sorted([(pos, seq, cig) for _, _, _, recs_r, _ in d["reads"] for pos, _, seq, cig in recs_r], key=lambda t: t[0]))
    fa.close(); vcf.close()
    write_bam(prefix + ".bam", targets, all_recs)
    return expect
