#!/usr/bin/env python
"""(GPU) floria-hip end to end on seeded synthetic BAM / VCF / FASTA datasets against the oracle chain (tests/test_gpu_cli.py: run_and_check — ingest, hap graph,
LP flows, joined paths, S2, every output file byte for byte), more datasets than the test suite holds, at a dyadic and at a non-dyadic epsilon.
usage: scripts/cli_fuzz.py [first seed = 0] [count = 30]"""
import os, subprocess, sys, tempfile, pathlib, traceback
sys.path.insert(0, ".")
import numpy as np
from floria_amd import synth
from oracle import oracle
from tests import test_gpu_cli as T

oracle.build()
subprocess.check_call(["make", "-C", T.HOST, "floria-hip"], stdout=subprocess.DEVNULL)
exe = os.path.join(T.HOST, "floria-hip")
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
for seed in range(s0, s0 + cnt):
    rng = np.random.default_rng(8800 + seed)
    kind = int(rng.integers(0, 3))
    if kind == 0:
        contigs, bl = [synth.make_config_contig(1, 10 + seed, float(rng.choice([0.5, 1.0])), keep_layout=True)], 10000
    elif kind == 1:
        contigs, bl = [synth.make_config_contig(4, 20 + seed, 0.5, keep_layout=True), synth.make_config_contig(4, 500 + seed, 0.4, keep_layout=True)], 10000
    else:
        contigs, bl = [synth.make_config_contig(3, 30 + seed, 0.15, keep_layout=True)], 500
    eps = float(rng.choice([0.03125, 0.04, 0.0437]))
    extra = ("--output-reads",) if rng.random() < 0.3 else ()
    sub = 0.08 if (kind == 0 and rng.random() < 0.3) else 0.0
    with tempfile.TemporaryDirectory() as d:
        try:
            # (paired short reads: their fragments are merged from two mates; since round 6 floria-hip replays their position sets and phases them in the reference's
            # arithmetic at a non-dyadic epsilon like everything else - run_and_check hands the oracle the same set orders from ITS emulation)
            T.run_and_check(exe, oracle, pathlib.Path(d), contigs, bl, extra=extra, sub_rate=sub, eps=eps)
        except Exception as ex:
            bad += 1
            print(f"FAILED seed {seed} kind {kind} eps {eps} extra {extra} sub {sub}")
            traceback.print_exc(limit=3)
            tb = ex.__traceback__
            while tb is not None:                                  # a file that differs: show how
                loc = tb.tb_frame.f_locals
                if tb.tb_frame.f_code.co_name == "_run_and_check" and "want" in loc and "fn" in loc:
                    import difflib
                    got = open(os.path.join(loc["cdir"], loc["fn"])).read().splitlines()
                    exp = loc["want"][loc["fn"]].splitlines()
                    for ln in list(difflib.unified_diff(exp, got, "oracle chain", "floria-hip", lineterm="", n=1))[:40]:
                        print("   ", ln)
                tb = tb.tb_next
print(f"seeds {s0}..{s0 + cnt - 1}: {bad} of {cnt} end-to-end runs differ from the oracle chain")
