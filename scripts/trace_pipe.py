import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib, synth
C = synth.CONFIGS[4]
contigs = [synth.make_config_contig(4, i) for i in range(2000)]
arena, parr, nbytes = lib.pack_pileups([c.pileup for c in contigs])
bc, bs, be = [], [], []
for i, c in enumerate(contigs):
    s, e = lib.get_range_with_lengths(c.snp_pos, C["block_length"])
    bc += [i] * len(s); bs += list(s); be += list(e)
bc, bs, be = (np.asarray(x, np.uint32) for x in (bc, bs, be))
ctx = lib.FloriaHip(0)
par = lib.make_params(0.03125, 5, 10)
for k in range(3):
    t = time.perf_counter()
    ctx.phase_pileups_batch(parr, bc, bs, be, par, copy_out=False)
    print("call", k, (time.perf_counter() - t) * 1e3, "ms", flush=True)
