#!/usr/bin/env python3
"""How much of floria-hip's output depends on WHICH optimal vertex of the stitching LP is used?

The reference solves the LP with minilp (a simplex); floria-hip solves it exactly as a min-cost flow (floria_amd/host/stitch.cpp).  The
optimum value is the same; where the optimum is not unique the reference's flows are whatever vertex its simplex stops at, which
cannot be reproduced here.  This script runs the tool on a set of synthetic contigs with the two extremes of the flow solver's
tie-breaking (--lp-tie first|last) and reports, per contig and in total: whether the optimum is unique, whether the flows differ,
whether the .haplosets files differ, and the fraction of reads whose haploset changes.  Needs a GPU (the tool's device stages).

    python scripts/lp_tie_rate.py [--contigs 40] [--out gpurun_out/lp_tie_rate.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from floria_amd import synth, synth_bam  # noqa: E402


def read_haplosets(path):
    """{read name: haploset index} and the list of haploset read-name sets"""
    sets, cur = [], None
    for line in open(path):
        if line.startswith(">"):
            cur = set(); sets.append(cur)
        elif line.strip():
            cur.add(line.split("\t")[0])
    return sets


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=40)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "lp_tie_rate.json"))
    a = ap.parse_args()
    tool = os.path.join(ROOT, "floria_amd", "host", "floria-hip")
    subprocess.check_call(["make", "-C", os.path.dirname(tool), "floria-hip"], stdout=subprocess.DEVNULL)
    cs = []
    for i in range(a.contigs):
        cfg = (4, 4, 4, 3, 1)[i % 5]                          # mostly config-4-shaped contigs, some config 3 and config 1
        cs.append(synth.make_config_contig(cfg, 100 + i, 0.25 + 0.015 * (i % 20), keep_layout=True))
    tmp = tempfile.mkdtemp(prefix="lptie")
    prefix = os.path.join(tmp, "data")
    synth_bam.write_dataset(prefix, cs, seed=13)
    res = {}
    for tie in ("first", "last"):
        out = os.path.join(tmp, "o_" + tie)
        r = subprocess.run([tool, "-b", prefix + ".bam", "-v", prefix + ".vcf", "-r", prefix + ".fa", "-o", out, "-e", "0.03125", "-l", "5000", "--debug",
                            "--snp-count-filter", "50", "--lp-report", "--lp-tie", tie], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr); sys.exit(1)
        m = re.search(r"LP: the optimum is not unique for (\d+) of (\d+) contigs \((\d+) of (\d+) edge flows", r.stderr)
        res[tie] = dict(out=out, lp=[int(x) for x in m.groups()])
    rows, tot_reads, moved_reads = [], 0, 0
    for c in cs:
        f = [os.path.join(res[t]["out"], c.name, c.name + ".haplosets") for t in ("first", "last")]
        g = [[ln for ln in open(os.path.join(res[t]["out"], c.name, "debug_graph.txt")) if ln.startswith("F")] for t in ("first", "last")]
        h = [read_haplosets(x) for x in f]
        # a read "moves" when the set of reads it shares a haploset with changes
        where = [{n: frozenset(s) for s in hs for n in s} for hs in h]
        names = set(where[0]) | set(where[1])
        moved = sum(1 for n in names if where[0].get(n) != where[1].get(n))
        tot_reads += len(names); moved_reads += moved
        rows.append(dict(contig=c.name, flows_differ=g[0] != g[1], n_edges=len(g[0]), edges_differ=sum(x != y for x, y in zip(g[0], g[1])),
                         haplosets=[len(h[0]), len(h[1])], haplosets_differ=set(map(frozenset, h[0])) != set(map(frozenset, h[1])), reads=len(names), reads_moved=moved))
    summary = dict(contigs=len(cs), lp_not_unique_contigs=res["first"]["lp"][0], lp_movable_edges=res["first"]["lp"][2], lp_edges=res["first"]["lp"][3],
                   contigs_flows_differ=sum(r["flows_differ"] for r in rows), contigs_haplosets_differ=sum(r["haplosets_differ"] for r in rows),
                   reads=tot_reads, reads_moved=moved_reads, reads_moved_frac=moved_reads / max(1, tot_reads))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(dict(summary=summary, contigs=rows), open(a.out, "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
