"""Loader for golden outputs captured from the REAL floria binary (docs/golden.md).  No capture can be made in this image (no
Rust toolchain), so these tests skip until tests/golden/reference_capture/ exists; the parser is exercised on a synthetic line."""
import glob
import os
import re

import numpy as np
import pytest

CAP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_capture")
MEC_RE = re.compile(r"MEC vector[^\[]*\[([^\]]*)\]")


def parse_mec_vectors(log_text):
    """`log::trace!("MEC vector {:?}", mec_vector)` lines (graph_processing.rs:258-266) -> list of float64 arrays, in log order.
    Rust's {:?} prints the shortest decimal that round-trips, so float() recovers the bits."""
    return [np.array([float(x) for x in m.group(1).split(",") if x.strip()], np.float64) for m in MEC_RE.finditer(log_text)]


def test_mec_trace_parser_round_trips_f64():
    vals = np.array([12.0, 0.09375, 1.0000000000000002, 3.5e-07], np.float64)
    line = "2024-01-01 TRACE [floria::graph_processing] MEC vector [" + ", ".join(repr(float(v)) for v in vals) + "]\n"
    got = parse_mec_vectors(line * 2)
    assert len(got) == 2 and np.array_equal(got[0].view(np.uint64), vals.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(CAP), reason="no capture of the real floria binary (docs/golden.md): parity stays unpinned")
def test_s1_against_the_captured_reference(gpu_ctx, hip_lib, oracle_mod, tmp_path):
    from floria_amd import synth, synth_bam
    cs = [synth.make_config_contig(1, 0, keep_layout=True), synth.make_config_contig(4, 3, 0.5, keep_layout=True)]
    expect = synth_bam.write_dataset(str(tmp_path / "golden_in"), cs, seed=7)
    for run in sorted(glob.glob(os.path.join(CAP, "e*"))):
        eps = float(os.path.basename(run)[1:])
        mecs = parse_mec_vectors(open(os.path.join(run, "trace.log")).read())
        k = 0
        for c in cs:
            ex = expect[c.name]
            s, e = hip_lib.get_range_with_lengths(ex["snp_pos0"], 10000)
            # a dyadic epsilon pins the PRODUCT (both arithmetics exact, S1 independent of the hash orders); any other epsilon checks the
            # oracle's restatement of the reference's running sums in emulated hash order and nothing else
            dyadic = float(eps * 2 ** 20).is_integer()
            rg = gpu_ctx.phase_blocks(ex["pileup"], s, e, hip_lib.make_params(eps)) if dyadic else None
            if not dyadic:
                oracle_mod.set_arith_mode(1); oracle_mod.set_order_mode(2)
            try:
                ro = oracle_mod.phase_blocks(ex["pileup"], s, e, oracle_mod.make_params(eps), threads=8 if dyadic else 1)
            finally:
                oracle_mod.set_arith_mode(0); oracle_mod.set_order_mode(0)
            for b in range(ro.n_blocks):
                if ro.best_ploidy[b] == 0:
                    continue
                ref = mecs[k]; k += 1
                tried = int(ro.ploidies_tried[b])
                if dyadic:
                    assert np.array_equal(ref[:tried].view(np.uint64), rg.mec[b, :tried].view(np.uint64)), (run, c.name, b, "HIP vs reference")
                assert np.array_equal(ref[:tried].view(np.uint64), ro.mec[b, :tried].view(np.uint64)), (run, c.name, b, "oracle vs reference")
        assert k == len(mecs)
