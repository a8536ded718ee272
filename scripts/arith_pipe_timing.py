#!/usr/bin/env python
"""(GPU) the host-pileup entry point in the reference's arithmetic: ms per call of floria_hip_phase_pileups_batch_packed over BASELINE config 4 for several chunk counts
(1 = the whole batch uploaded first, as before round 6), canonical beside it; FLORIA_HIP_TRACE=1 prints the host-side marks of one call.
usage: scripts/arith_pipe_timing.py [contigs = 2000] [eps = 0.04]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.04
C = synth.CONFIGS[4]
contigs = [synth.make_config_contig(4, i) for i in range(n)]
piles = [c.pileup for c in contigs]
arena, parr, nbytes = lib.pack_pileups(piles)
bc, bs, be = [], [], []
for i, c in enumerate(contigs):
    s, e = lib.get_range_with_lengths(c.snp_pos, C["block_length"])
    bc += [i] * len(s); bs += list(s); be += list(e)
bc, bs, be = (np.asarray(x, np.uint32) for x in (bc, bs, be))
ctx = lib.FloriaHip(0)
par = lib.make_params(eps, C["max_ploidy"], C["beam"])
for mode in (0, 1):
    ctx.set_option("arith", mode)
    for chunks in (1, 2, 3, 5):
        ctx.set_option("upload_chunks", chunks)
        ctx.phase_pileups_batch(parr, bc, bs, be, par, copy_out=False)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            ctx.phase_pileups_batch(parr, bc, bs, be, par, copy_out=False)
            ts.append((time.perf_counter() - t) * 1e3)
        tm = ctx.timing()
        print(f"arith {mode} chunks {chunks} (used {tm['upload_chunks']}): {min(ts):.1f} / {sorted(ts)[1]:.1f} / {max(ts):.1f} ms per call; beam {tm['beam_ms']:.1f} optimise {tm['optimize_ms']:.1f} select+order {tm['select_ms']:.1f} h2d {tm['h2d_ms']:.1f}", flush=True)
