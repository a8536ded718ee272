#!/usr/bin/env python
"""(GPU) the reference-arithmetic mode's one-time pass per upload: the cells of every read in the iteration order of its position set (cell_order kernels, arith_kernel.h).
First S1 call after an upload (orders computed) against the second (orders kept), with the home-bucket rule and with every read emulated table by table.
usage: scripts/cell_order_timing.py [contigs = 2000]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
C = synth.CONFIGS[4]
contigs = [synth.make_config_contig(4, i, 1.0) for i in range(n)]
ctx = lib.FloriaHip(0)
ctx.set_option("arith", 1)
bc, bs, be = [], [], []
for i, c in enumerate(contigs):
    s, e = lib.get_range_with_lengths(c.snp_pos, C["block_length"])
    bc += [i] * len(s); bs += list(s); be += list(e)
bc, bs, be = (np.asarray(x, np.uint32) for x in (bc, bs, be))
par = lib.make_params(0.04, C["max_ploidy"], C["beam"])
hs = ctx.upload_batch([c.pileup for c in contigs])
ctx.phase_blocks_batch(hs, bc, bs, be, par, copy_out=False)      # warm-up (allocations)
for replay in (0, 1, 0):
    ctx.set_option("arith_replay", replay)                       # (drops the kept orders)
    t = time.perf_counter(); ctx.phase_blocks_batch(hs, bc, bs, be, par, copy_out=False); t1 = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); ctx.phase_blocks_batch(hs, bc, bs, be, par, copy_out=False); t2 = (time.perf_counter() - t) * 1e3
    print(f"every read emulated = {replay}: first call {t1:.1f} ms, second {t2:.1f} ms -> cell orders {t1 - t2:.1f} ms")
