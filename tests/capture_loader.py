"""Parsers for everything a run of the REAL floria binary leaves behind (docs/golden.md) — test infrastructure, like oracle/.

    trace.log                              `MEC vector [..], error_thresh .., SNPs interval a b` lines   (graph_processing.rs:258-266)
    <contig>/local_parts/J-L-S-P.haplosets  the partition get_local_hap_blocks chose for block J         (graph_processing.rs:289-300,
                                            file_writer.rs:919-993 with an empty range list: `#i` headers, `name first last` rows)
    <contig>/<contig>.haplosets             the final haplosets                                          (file_writer.rs:919-993)
    <contig>/<contig>.vartigs               header + allele string per haploset                          (file_writer.rs:699-899)
    contig_ploidy_info.tsv                  one row per contig                                           (file_writer.rs:901-914)

No capture exists in this repository (no Rust toolchain in the build image: parity is unpinned); every parser is exercised on files of the
same formats in tests/test_reference_capture.py, so that a capture dropped into tests/golden/reference_capture/ is compared at once.
"""
import glob
import os
import re

import numpy as np

MEC_RE = re.compile(r"MEC vector\s*\[([^\]]*)\]\s*,\s*error_thresh\s*([^,]+),\s*SNPs interval\s+(\d+)\s+(\d+)")
MEC_BARE_RE = re.compile(r"MEC vector[^\[]*\[([^\]]*)\]")
HEAD_RE = re.compile(r"^>HAP(\d+)\.(\S*)\tCONTIG:(\S*)\tSNPRANGE:(\d+)-(\d+)\tBASERANGE:(\d+)-(\d+)\tCOV:(\S+)\tERR:(\S+)\tHAPQ:(\d+)\tREL_ERR:(\S+)$")


def _floats(txt):
    return np.array([float(x) for x in txt.split(",") if x.strip()], np.float64)


def parse_mec_vectors(log_text):
    """-> list of float64 arrays in log order.  Rust's {:?} prints the shortest decimal that round-trips, so float() recovers the bits."""
    return [_floats(m.group(1)) for m in MEC_BARE_RE.finditer(log_text)]


def parse_mec_trace(log_text):
    """-> list of dict(mec, error_thresh, snp_start, snp_end) in log order (the interval identifies the block whatever the thread order was)"""
    out = []
    for m in MEC_RE.finditer(log_text):
        out.append(dict(mec=_floats(m.group(1)), error_thresh=float(m.group(2)), snp_start=int(m.group(3)), snp_end=int(m.group(4))))
    return out


def parse_local_part_file(path):
    """one `J-L-S-P.haplosets` file -> dict(block, sub, snp_start, best_ploidy, parts = {partition index: [(read name, first, last), ..]})"""
    j, sub, s, p = (int(x) for x in os.path.basename(path)[:-len(".haplosets")].split("-"))
    parts, cur = {}, None
    for line in open(path):
        line = line.rstrip("\n")
        if not line:
            continue
        if line.startswith("#"):
            cur = int(line[1:]); parts[cur] = []
        else:
            name, first, last = line.rsplit("\t", 2)
            parts[cur].append((name, int(first), int(last)))
    return dict(block=j, sub=sub, snp_start=s, best_ploidy=p, parts=parts)


def parse_local_parts(directory):
    """every block of a contig's local_parts/ directory, by block index"""
    return {d["block"]: d for d in (parse_local_part_file(f) for f in sorted(glob.glob(os.path.join(directory, "*.haplosets"))))}


def parse_haplosets(path):
    """the final <contig>.haplosets -> list of dict(index, dir, contig, snp_range, base_range, cov, err, hapq, rel_err (strings as printed), reads)"""
    out = []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            m = HEAD_RE.match(line)
            assert m, f"{path}: header does not parse: {line!r}"
            out.append(dict(index=int(m.group(1)), dir=m.group(2), contig=m.group(3), snp_range=(int(m.group(4)), int(m.group(5))),
                            base_range=(int(m.group(6)), int(m.group(7))), cov=m.group(8), err=m.group(9), hapq=int(m.group(10)), rel_err=m.group(11), reads=[]))
        elif line:
            name, first, last = line.rsplit("\t", 2)
            out[-1]["reads"].append((name, int(first), int(last)))
    return out


def parse_vartigs(path):
    """<contig>.vartigs -> list of dict(header fields as parse_haplosets, alleles = the 0/1/2/? string)"""
    out = []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            m = HEAD_RE.match(line)
            assert m, f"{path}: header does not parse: {line!r}"
            out.append(dict(index=int(m.group(1)), dir=m.group(2), contig=m.group(3), snp_range=(int(m.group(4)), int(m.group(5))),
                            base_range=(int(m.group(6)), int(m.group(7))), cov=m.group(8), err=m.group(9), hapq=int(m.group(10)), rel_err=m.group(11), alleles=None))
        elif line:
            assert out and out[-1]["alleles"] is None, f"{path}: allele line without a header"
            out[-1]["alleles"] = line
    for v in out:
        assert v["alleles"] is not None and len(v["alleles"]) == v["snp_range"][1] - v["snp_range"][0] + 1, f"{path}: HAP{v['index']}: allele string does not cover its SNP range"
    return out


def parse_ploidy_info(path):
    """contig_ploidy_info.tsv -> (header fields, {contig: [fields as printed]})"""
    lines = [l.rstrip("\n").split("\t") for l in open(path) if l.strip()]
    return lines[0], {l[0]: l[1:] for l in lines[1:]}


def partition_of_block(local, names):
    """a parsed local_parts block -> list over partitions 0..best_ploidy-1 of sorted read ids (`names`: read name -> id of the sorted pileup);
    partitions the writer skipped (empty sets) come back empty"""
    idx = {n: i for i, n in enumerate(names)}
    return [sorted(idx[n] for n, _, _ in local["parts"].get(k, [])) for k in range(local["best_ploidy"])]


def partition_from_result(res, b):
    """floria_block_result (oracle or HIP, same layout) block b -> the same shape as partition_of_block"""
    ids, part = res.block(b)
    return [sorted(int(r) for r in ids[part == k]) for k in range(int(res.best_ploidy[b]))]


def haploset_diff(ref, got):
    """two parsed .haplosets / .vartigs files -> list of human-readable differences (empty = identical content)"""
    out = []
    if len(ref) != len(got):
        out.append(f"{len(ref)} haplosets in the capture, {len(got)} here")
    for a, b in zip(ref, got):
        for f in ("index", "contig", "snp_range", "base_range", "cov", "err", "hapq", "rel_err", "reads", "alleles"):
            if f in a and a[f] != b.get(f):
                va, vb = a[f], b.get(f)
                if f == "reads":
                    sa, sb = set(va), set(vb)
                    va, vb = f"{len(sa)} reads", f"{len(sb)} reads, {len(sa ^ sb)} not shared"
                out.append(f"HAP{a['index']} {f}: capture {va} | here {vb}")
    return out
