"""Loader for golden outputs captured from the REAL floria binary (docs/golden.md).  No capture can be made in this image (no Rust
toolchain), so the comparison test skips until tests/golden/reference_capture/ exists — but every parser it needs (tests/capture_loader.py)
is exercised here on files of the reference's formats, so that a capture is compared the moment it is dropped in:

  S1          MEC vector of every block (by SNP interval) and the per-block partitions of local_parts/ against the HIP path and the oracle
  end to end  .haplosets / .vartigs / contig_ploidy_info.tsv against what floria-hip writes for the same inputs
"""
import glob
import os
import subprocess

import numpy as np
import pytest

from tests import capture_loader as cl

CAP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_capture")
HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "floria_amd", "host")


def test_mec_trace_parser_round_trips_f64():
    vals = np.array([12.0, 0.09375, 1.0000000000000002, 3.5e-07], np.float64)
    line = "2024-01-01 TRACE [floria::graph_processing] MEC vector [" + ", ".join(repr(float(v)) for v in vals) + "]\n"
    got = cl.parse_mec_vectors(line * 2)
    assert len(got) == 2 and np.array_equal(got[0].view(np.uint64), vals.view(np.uint64))
    # the full line of graph_processing.rs:260-266: `MEC vector {:?}, error_thresh {:?}, SNPs interval  {} {}`
    full = "TRACE [floria::graph_processing] MEC vector [12.0, 0.1875, 0.0, 0.0, 0.0], error_thresh 0.75, SNPs interval  17 1042\n"
    tr = cl.parse_mec_trace(full + full.replace("17 1042", "700 1800"))
    assert [(t["snp_start"], t["snp_end"]) for t in tr] == [(17, 1042), (700, 1800)]
    assert tr[0]["error_thresh"] == 0.75 and tr[0]["mec"].tolist() == [12.0, 0.1875, 0.0, 0.0, 0.0]


def write_local_part(directory, j, snp_start, best_ploidy, parts_by_name):
    """file_writer.rs:919-993 with an empty range list, as get_local_hap_blocks calls it (graph_processing.rs:289-300): `#i` per non-empty set, then
    `name\\tfirst\\tlast` rows sorted by Frag order (here: the order given)"""
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, f"{j}-0-{snp_start}-{best_ploidy}.haplosets"), "w") as f:
        for i, rows in enumerate(parts_by_name):
            if not rows:
                continue
            f.write(f"#{i}\n")
            for name, first, last in rows:
                f.write(f"{name}\t{first}\t{last}\n")


def test_local_parts_loader_on_the_reference_format(oracle_mod, tmp_path):
    """Blocks phased by the oracle, written the way the reference dumps them under --trace (read NAMES, one file per block, empty sets skipped,
    names with tabs-free but otherwise arbitrary characters), parsed back and compared the way the capture test compares: identical partitions."""
    from floria_amd import synth
    c = synth.make_config_contig(4, 3, 0.5)
    pile = c.pileup
    s, e = oracle_mod.block_ranges(c.snp_pos, 10000)
    ro = oracle_mod.phase_blocks(pile, s, e, oracle_mod.make_params(0.03125), threads=8)
    names = [f"read/{i}:x y" for i in range(pile.n_reads)]
    d = str(tmp_path / "local_parts")
    for b in range(ro.n_blocks):
        if ro.best_ploidy[b] == 0:
            continue
        want = cl.partition_from_result(ro, b)
        write_local_part(d, b, int(s[b]), int(ro.best_ploidy[b]), [[(names[r], int(pile.first[r]), int(pile.last[r])) for r in part] for part in want])
    loc = cl.parse_local_parts(d)
    assert sorted(loc) == [b for b in range(ro.n_blocks) if ro.best_ploidy[b]]
    for b, blk in loc.items():
        assert blk["snp_start"] == int(s[b]) and blk["best_ploidy"] == int(ro.best_ploidy[b]) and blk["sub"] == 0
        assert cl.partition_of_block(blk, names) == cl.partition_from_result(ro, b)
    # a partition the optimiser emptied is skipped by the writer and comes back empty
    write_local_part(str(tmp_path / "lp2"), 0, 5, 3, [[("a", 1, 9)], [], [("b", 2, 8), ("c", 3, 7)]])
    blk = cl.parse_local_parts(str(tmp_path / "lp2"))[0]
    assert cl.partition_of_block(blk, ["a", "b", "c"]) == [[0], [], [1, 2]]


HAPLOSETS = (">HAP0.out/ctg\tCONTIG:ctg\tSNPRANGE:1-4\tBASERANGE:101-977\tCOV:12.500\tERR:0.0312\tHAPQ:60\tREL_ERR:1.000\n"
             "r1\t1\t4\nr2 with space\t2\t3\n"
             ">HAP2.out/ctg\tCONTIG:ctg\tSNPRANGE:3-6\tBASERANGE:640-1501\tCOV:3.000\tERR:NaN\tHAPQ:0\tREL_ERR:inf\n"
             "r9\t3\t6\n")
VARTIGS = (">HAP0.out/ctg\tCONTIG:ctg\tSNPRANGE:1-4\tBASERANGE:101-977\tCOV:12.500\tERR:0.0312\tHAPQ:60\tREL_ERR:1.000\n01?1\n"
           ">HAP2.out/ctg\tCONTIG:ctg\tSNPRANGE:3-6\tBASERANGE:640-1501\tCOV:3.000\tERR:NaN\tHAPQ:0\tREL_ERR:inf\n2?10\n")


def test_final_file_parsers_on_the_reference_formats(tmp_path):
    """.haplosets (file_writer.rs:945-985), .vartigs (:801-830) and contig_ploidy_info.tsv (:901-914) as the reference prints them"""
    hp, vp, pp = (str(tmp_path / n) for n in ("c.haplosets", "c.vartigs", "contig_ploidy_info.tsv"))
    open(hp, "w").write(HAPLOSETS); open(vp, "w").write(VARTIGS)
    open(pp, "w").write("contig\taverage_straincount\twhole_contig_multiplicity\tapproximate_coverage_ignoring_indels\ttotal_vartig_bases_covered\t"
                        "average_straincount_min15hapq\taverage_straincount_min30hapq\taverage_straincount_min45hapq\tavg_err\n"
                        "ctg\t2.000\t1.950\t41.250\t9750\t2.000\t1.800\t1.500\t0.0312\n")
    hs = cl.parse_haplosets(hp)
    assert [h["index"] for h in hs] == [0, 2] and hs[0]["reads"] == [("r1", 1, 4), ("r2 with space", 2, 3)] and hs[1]["err"] == "NaN" and hs[1]["rel_err"] == "inf"
    assert hs[0]["snp_range"] == (1, 4) and hs[0]["base_range"] == (101, 977) and hs[0]["hapq"] == 60 and hs[0]["dir"] == "out/ctg"
    vs = cl.parse_vartigs(vp)
    assert [v["alleles"] for v in vs] == ["01?1", "2?10"]
    head, rows = cl.parse_ploidy_info(pp)
    assert head[0] == "contig" and rows["ctg"][0] == "2.000" and rows["ctg"][-1] == "0.0312"
    assert cl.haploset_diff(hs, cl.parse_haplosets(hp)) == []
    open(hp, "w").write(HAPLOSETS.replace("r9\t3\t6", "r8\t3\t6").replace("HAPQ:60", "HAPQ:59"))
    d = cl.haploset_diff(hs, cl.parse_haplosets(hp))
    assert len(d) == 2 and "hapq" in d[0] and "reads" in d[1]


DATASETS = {            # scripts/capture_reference.sh: name -> (contigs, -l)
    "long": (lambda synth: [synth.make_config_contig(1, 0, keep_layout=True), synth.make_config_contig(4, 3, 0.5, keep_layout=True)], 10000),
    "short": (lambda synth: [synth.make_config_contig(3, 2, 0.3, keep_layout=True)], 500),
}


def _dataset(name, tmp_path, oracle_mod):
    """the inputs of a captured run, regenerated, with the iteration orders of the merged fragments' position sets (arithmetic mode 1 needs them)"""
    from floria_amd import synth, synth_bam
    make, block_length = DATASETS[name]
    cs = make(synth)
    prefix = str(tmp_path / f"golden_{name}")
    expect = synth_bam.write_dataset(prefix, cs, seed=7)
    for c in cs:
        ex = expect[c.name]
        if ex["paired"]:
            pile = ex["pileup"]
            pile.set_order = np.concatenate([oracle_mod.set_order_of(pile.read(i)[0], [np.asarray(x, np.uint32) for x in ex["segments"][i]]) for i in range(pile.n_reads)])
    return cs, expect, prefix, block_length


def _compare_s1(run, c, ex, s, e, res, trace, local, who, n_seen):
    for b in range(res.n_blocks):
        if res.best_ploidy[b] == 0:
            continue
        ref = trace[(int(s[b]), int(e[b]))]["mec"]; n_seen[0] += 1
        tried = int(res.ploidies_tried[b])
        assert np.array_equal(ref[:tried].view(np.uint64), res.mec[b, :tried].view(np.uint64)), (run, c.name, b, f"MEC vector: {who} vs reference")
        # the partition the reference chose: the first true check of the heap's tie-breaking and of opt_iterate
        assert b in local, (run, c.name, b, "no local_parts file for this block")
        assert local[b]["best_ploidy"] == int(res.best_ploidy[b]), (run, c.name, b, f"chosen ploidy: {who} vs reference")
        assert cl.partition_of_block(local[b], ex["names"]) == cl.partition_from_result(res, b), (run, c.name, b, f"partition: {who} vs reference")


def _runs():
    return sorted(glob.glob(os.path.join(CAP, "*", "e*")))


@pytest.mark.skipif(not os.path.isdir(CAP), reason="no capture of the real floria binary (docs/golden.md, scripts/capture_reference.sh): parity stays unpinned")
def test_oracle_against_the_captured_reference(oracle_mod, tmp_path):
    """The CPU half (no GPU needed): S1 of the oracle against the capture.  A dyadic epsilon in arithmetic mode 0 (every sum exact, S1 independent of the hash orders:
    any mismatch is a bug of the restatement or of SURVEY's Appendix A); any other epsilon in mode 1 with the emulated containers (order mode 2)."""
    for run in _runs():
        name, eps = os.path.basename(os.path.dirname(run)), float(os.path.basename(run)[1:])
        cs, expect, _, bl = _dataset(name, tmp_path, oracle_mod)
        trace = {(t["snp_start"], t["snp_end"]): t for t in cl.parse_mec_trace(open(os.path.join(run, "trace.log")).read())}
        dyadic = float(eps * 2 ** 20).is_integer()
        n_seen = [0]
        for c in cs:
            ex = expect[c.name]
            s, e = oracle_mod.block_ranges(ex["snp_pos0"], bl)
            if not dyadic:
                oracle_mod.set_arith_mode(1); oracle_mod.set_order_mode(2)
            try:
                ro = oracle_mod.phase_blocks(ex["pileup"], s, e, oracle_mod.make_params(eps), threads=8 if dyadic else 1)
            finally:
                oracle_mod.set_arith_mode(0); oracle_mod.set_order_mode(0)
            _compare_s1(run, c, ex, s, e, ro, trace, cl.parse_local_parts(os.path.join(run, c.name, "local_parts")), "oracle", n_seen)
        assert n_seen[0] == len(trace)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(CAP), reason="no capture of the real floria binary (docs/golden.md, scripts/capture_reference.sh): parity stays unpinned")
def test_against_the_captured_reference(gpu_ctx, hip_lib, oracle_mod, tmp_path):
    """The product against the capture: S1 through the C ABI — canonical arithmetic at a dyadic epsilon, `arith = 1` (the reference's running sums, what floria-hip runs
    there) at any other — and, end to end, the files floria-hip writes for the same inputs."""
    subprocess.check_call(["make", "-C", HOST, "floria-hip"], stdout=subprocess.DEVNULL)
    for run in _runs():
        name, eps = os.path.basename(os.path.dirname(run)), float(os.path.basename(run)[1:])
        cs, expect, prefix, bl = _dataset(name, tmp_path, oracle_mod)
        trace = {(t["snp_start"], t["snp_end"]): t for t in cl.parse_mec_trace(open(os.path.join(run, "trace.log")).read())}
        dyadic = float(eps * 2 ** 20).is_integer()
        n_seen = [0]
        gpu_ctx.set_option("arith", 0 if dyadic else 1)
        try:
            for c in cs:
                ex = expect[c.name]
                s, e = hip_lib.get_range_with_lengths(ex["snp_pos0"], bl)
                rg = gpu_ctx.phase_blocks(ex["pileup"], s, e, hip_lib.make_params(eps))
                _compare_s1(run, c, ex, s, e, rg, trace, cl.parse_local_parts(os.path.join(run, c.name, "local_parts")), "HIP", n_seen)
        finally:
            gpu_ctx.set_option("arith", 0)
        assert n_seen[0] == len(trace)
        # ---- end to end: the files floria-hip writes for the same inputs (differences that S1 does not show localise to the LP vertex, petgraph's
        # tie-breaking or the S2 visiting order: DESIGN.md §7)
        out = str(tmp_path / f"out_{name}_{os.path.basename(run)}")
        subprocess.check_call([os.path.join(HOST, "floria-hip"), "-b", prefix + ".bam", "-v", prefix + ".vcf", "-r", prefix + ".fa", "-o", out, "-e", repr(eps), "-l", str(bl)],
                              stderr=subprocess.DEVNULL)
        _, ref_rows = cl.parse_ploidy_info(os.path.join(run, "contig_ploidy_info.tsv"))
        _, got_rows = cl.parse_ploidy_info(os.path.join(out, "contig_ploidy_info.tsv"))
        assert ref_rows == got_rows, (run, "contig_ploidy_info.tsv")
        for c in cs:
            d = cl.haploset_diff(cl.parse_haplosets(os.path.join(run, c.name, c.name + ".haplosets")), cl.parse_haplosets(os.path.join(out, c.name, c.name + ".haplosets")))
            # (the directory in the header is whatever -o was in the two runs)
            d = [x for x in d if " dir:" not in x]
            assert not d, (run, c.name, ".haplosets", d[:10])
            d = cl.haploset_diff(cl.parse_vartigs(os.path.join(run, c.name, c.name + ".vartigs")), cl.parse_vartigs(os.path.join(out, c.name, c.name + ".vartigs")))
            assert not d, (run, c.name, ".vartigs", d[:10])


def test_capture_comparison_on_a_stand_in_capture(oracle_mod, tmp_path, monkeypatch):
    """The whole comparison path on a STAND-IN capture: the oracle's own S1 results for the paired short-read data set at -e 0.04 (arithmetic mode 1, emulated containers,
    merged fragments' set orders), written in the formats the reference leaves behind - `MEC vector` trace lines, local_parts/ files with read names - into the capture
    layout of scripts/capture_reference.sh, then compared by test_oracle_against_the_captured_reference's code.  Proves nothing about parity (it is the oracle against
    itself); it proves that a real capture dropped into tests/golden/reference_capture/ is read, matched block by block and compared without further work."""
    import sys
    mod = sys.modules[__name__]
    cap = tmp_path / "reference_capture"
    run = cap / "short" / "e0.04"
    cs, expect, _, bl = _dataset("short", tmp_path, oracle_mod)
    lines = []
    for c in cs:
        ex = expect[c.name]
        s, e = oracle_mod.block_ranges(ex["snp_pos0"], bl)
        oracle_mod.set_arith_mode(1); oracle_mod.set_order_mode(2)
        try:
            ro = oracle_mod.phase_blocks(ex["pileup"], s, e, oracle_mod.make_params(0.04), threads=1)
        finally:
            oracle_mod.set_arith_mode(0); oracle_mod.set_order_mode(0)
        pile = ex["pileup"]
        for b in range(ro.n_blocks):
            if ro.best_ploidy[b] == 0:
                continue
            lines.append("TRACE [floria::graph_processing] MEC vector [" + ", ".join(repr(float(v)) for v in ro.mec[b]) + f"], error_thresh 0.5, SNPs interval  {int(s[b])} {int(e[b])}\n")
            write_local_part(str(run / c.name / "local_parts"), b, int(s[b]), int(ro.best_ploidy[b]),
                             [[(ex["names"][r], int(pile.first[r]), int(pile.last[r])) for r in part] for part in cl.partition_from_result(ro, b)])
    os.makedirs(run, exist_ok=True)
    (run / "trace.log").write_text("".join(lines))
    monkeypatch.setattr(mod, "CAP", str(cap))
    assert _runs() == [str(run)]
    test_oracle_against_the_captured_reference.__wrapped__(oracle_mod, tmp_path) if hasattr(test_oracle_against_the_captured_reference, "__wrapped__") else test_oracle_against_the_captured_reference(oracle_mod, tmp_path)
    # ... and a capture that disagrees is reported with the block it disagrees at
    bad = (run / "trace.log").read_text().replace("MEC vector [", "MEC vector [1e-9, ", 1)
    (run / "trace.log").write_text(bad)
    with pytest.raises(AssertionError, match="MEC vector: oracle vs reference"):
        test_oracle_against_the_captured_reference(oracle_mod, tmp_path)
