"""Dev helper: S2 on ONE big contig (BASELINE config 5 at full size: 100k reads) with haplogroups as stitching leaves them (every read in
its strain's group, 2.5 % of the reads in a second group too): the workgroup-parallel kernel vs the one-wavefront chain kernel."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from floria_amd import synth, lib
c = synth.make_config_contig(5, 0, keep_truth=True)
p = c.pileup
ctx = lib.FloriaHip(0)
rc = ctx.upload(p)
rng = np.random.default_rng(1)
G = 8
groups = [np.nonzero(c.strain == k)[0].astype(np.uint32) for k in range(G)]
for r in rng.choice(p.n_reads, size=p.n_reads // 40, replace=False):
    k = int(rng.integers(0, G))
    if c.strain[r] != k: groups[k] = np.append(groups[k], np.uint32(r))
groups = [np.sort(g) for g in groups]
ranges = [(1, int(p.last.max()))] * G
out = {}
for path, name in ((1, "parallel"), (2, "chain")):
    ctx.set_option("reassign_path", path)
    for it in range(3):
        t = time.perf_counter(); g = ctx.reassign(rc, groups, ranges, 0.03125); dt = time.perf_counter() - t
    tm = ctx.timing()
    out[name] = g
    print(f"{name}: reads {p.n_reads}, with a choice {tm['jobs']}, kernel {tm['reassign_ms']:.2f} ms, call {dt*1e3:.1f} ms")
assert np.array_equal(out["parallel"].grp_read, out["chain"].grp_read) and np.array_equal(out["parallel"].grp_off, out["chain"].grp_off)
print("identical results")
