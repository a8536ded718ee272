#!/usr/bin/env python
"""(GPU) BASELINE config 5 (1 contig, 50k SNPs, 100k long reads, 8 strains, -p 8 -n 40) at full size: ms per resident S1 call in both arithmetics and the kernels behind it.
usage: scripts/config5_timing.py [scale = 1.0] [eps = 0.04]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib, synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.04
C = synth.CONFIGS[5]
c = synth.make_config_contig(5, 0, scale)
s, e = lib.get_range_with_lengths(c.snp_pos, C["block_length"])
ctx = lib.FloriaHip(0)
h = ctx.upload_batch([c.pileup])
bc = np.zeros(len(s), np.uint32)
par = lib.make_params(eps, C["max_ploidy"], C["beam"])
for mode in (0, 1):
    ctx.set_option("arith", mode)
    ctx.phase_blocks_batch(h, bc, s, e, par, copy_out=False)
    t = time.perf_counter()
    r = ctx.phase_blocks_batch(h, bc, s, e, par)
    dt = (time.perf_counter() - t) * 1e3
    tm = ctx.timing()
    print(f"arith {mode}: {len(s)} blocks, {dt:.1f} ms per call; beam {tm['beam_ms']:.1f} optimise {tm['optimize_ms']:.1f} ms; beam steps {tm['beam_steps']}; best ploidy histogram {np.bincount(r.best_ploidy, minlength=9).tolist()}", flush=True)
