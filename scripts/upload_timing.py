import sys, time
sys.path.insert(0, ".")
import numpy as np
from floria_amd import lib, synth
ctx = lib.FloriaHip(0)
cs = [synth.make_config_contig(4, i) for i in range(200)]
ctx.upload(cs[0].pileup).free()
t = time.perf_counter(); res = [ctx.upload(c.pileup) for c in cs]; dt = time.perf_counter() - t
cells = sum(int(c.pileup.read_off[-1]) for c in cs)
print(f"upload of 200 contigs ({cells/1e6:.1f} M cells, {cells*9/1e6:.0f} MB host CSR): {dt*1e3:.0f} ms -> {cells*9/dt/1e9:.2f} GB/s of host pileup bytes, validated + flattened + copied")
