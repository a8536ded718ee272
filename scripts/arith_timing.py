#!/usr/bin/env python
"""(GPU) where the time of the reference-arithmetic mode goes: kernel-family sums of one resident S1 call, canonical against arith = 1.
usage: scripts/arith_timing.py [contigs = 250] [epsilon = 0.04] [speculate = -1] [opt_threads = 0] [knob=value ..]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.04
C = synth.CONFIGS[4]
contigs = [synth.make_config_contig(4, i, 1.0) for i in range(n)]
ctx = lib.FloriaHip(0)
hs = ctx.upload_batch([c.pileup for c in contigs])
bc, bs, be = [], [], []
for i, c in enumerate(contigs):
    s, e = lib.get_range_with_lengths(c.snp_pos, C["block_length"])
    bc += [i] * len(s); bs += list(s); be += list(e)
bc, bs, be = (np.asarray(x, np.uint32) for x in (bc, bs, be))
par = lib.make_params(eps, C["max_ploidy"], C["beam"])
if len(sys.argv) > 3:
    ctx.set_option("speculate", int(sys.argv[3]))
if len(sys.argv) > 4:
    ctx.set_option("opt_threads", int(sys.argv[4]))
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
import os
for mode in [int(x) for x in os.environ.get("ARITH_MODES", "0,1").split(",")]:      # (ARITH_MODES=0: the canonical call only, e.g. under scripts/arith_trace.sh)
    ctx.set_option("arith", mode)
    ctx.phase_blocks_batch(hs, bc, bs, be, par, copy_out=False)
    t = time.perf_counter()
    ctx.phase_blocks_batch(hs, bc, bs, be, par, copy_out=False)
    dt = (time.perf_counter() - t) * 1e3
    tm = ctx.timing()
    print(f"arith {mode}: {len(bs)} blocks, {dt:.1f} ms wall; beam {tm['beam_ms']:.1f} optimise {tm['optimize_ms']:.1f} select+order {tm['select_ms']:.1f} total {tm['total_ms']:.1f} ms; beam steps {tm['beam_steps']}")
