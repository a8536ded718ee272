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


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(CAP), reason="no capture of the real floria binary (docs/golden.md): parity stays unpinned")
def test_against_the_captured_reference(gpu_ctx, hip_lib, oracle_mod, tmp_path):
    from floria_amd import synth, synth_bam
    cs = [synth.make_config_contig(1, 0, keep_layout=True), synth.make_config_contig(4, 3, 0.5, keep_layout=True)]
    prefix = str(tmp_path / "golden_in")
    expect = synth_bam.write_dataset(prefix, cs, seed=7)
    subprocess.check_call(["make", "-C", HOST, "floria-hip"], stdout=subprocess.DEVNULL)
    for run in sorted(glob.glob(os.path.join(CAP, "e*"))):
        eps = float(os.path.basename(run)[1:])
        trace = {(t["snp_start"], t["snp_end"]): t for t in cl.parse_mec_trace(open(os.path.join(run, "trace.log")).read())}
        # a dyadic epsilon pins the PRODUCT (both arithmetics exact, S1 independent of the hash orders); any other epsilon checks the
        # oracle's restatement of the reference's running sums in emulated hash order and nothing else
        dyadic = float(eps * 2 ** 20).is_integer()
        n_seen = 0
        for c in cs:
            ex = expect[c.name]
            s, e = hip_lib.get_range_with_lengths(ex["snp_pos0"], 10000)
            rg = gpu_ctx.phase_blocks(ex["pileup"], s, e, hip_lib.make_params(eps)) if dyadic else None
            if not dyadic:
                oracle_mod.set_arith_mode(1); oracle_mod.set_order_mode(2)
            try:
                ro = oracle_mod.phase_blocks(ex["pileup"], s, e, oracle_mod.make_params(eps), threads=8 if dyadic else 1)
            finally:
                oracle_mod.set_arith_mode(0); oracle_mod.set_order_mode(0)
            local = cl.parse_local_parts(os.path.join(run, c.name, "local_parts"))
            for b in range(ro.n_blocks):
                if ro.best_ploidy[b] == 0:
                    continue
                ref = trace[(int(s[b]), int(e[b]))]["mec"]; n_seen += 1
                tried = int(ro.ploidies_tried[b])
                if dyadic:
                    assert np.array_equal(ref[:tried].view(np.uint64), rg.mec[b, :tried].view(np.uint64)), (run, c.name, b, "MEC vector: HIP vs reference")
                assert np.array_equal(ref[:tried].view(np.uint64), ro.mec[b, :tried].view(np.uint64)), (run, c.name, b, "MEC vector: oracle vs reference")
                # the partition the reference chose: the first true check of the heap's tie-breaking and of opt_iterate
                assert b in local, (run, c.name, b, "no local_parts file for this block")
                assert local[b]["best_ploidy"] == int(ro.best_ploidy[b]), (run, c.name, b, "chosen ploidy: oracle vs reference")
                want = cl.partition_of_block(local[b], ex["names"])
                assert want == cl.partition_from_result(ro, b), (run, c.name, b, "partition: oracle vs reference")
                if dyadic:
                    assert want == cl.partition_from_result(rg, b), (run, c.name, b, "partition: HIP vs reference")
        assert n_seen == len(trace)
        # ---- end to end: the files floria-hip writes for the same inputs (differences that S1 does not show localise to the LP vertex, petgraph's
        # tie-breaking or the S2 visiting order: DESIGN.md §7)
        if not dyadic:
            continue
        out = str(tmp_path / f"out_{os.path.basename(run)}")
        subprocess.check_call([os.path.join(HOST, "floria-hip"), "-b", prefix + ".bam", "-v", prefix + ".vcf", "-r", prefix + ".fa", "-o", out, "-e", repr(eps), "-l", "10000"],
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
