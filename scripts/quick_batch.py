"""Dev helper: throughput of a batch of config-4 contigs on the GPU (+ optional oracle parity)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from floria_amd import synth, lib
from oracle import oracle

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
check = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 4
slots = int(sys.argv[4]) if len(sys.argv) > 4 else 0
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
C = synth.CONFIGS[cfg]
t = time.time()
contigs = [synth.make_config_contig(cfg, i, scale) for i in range(nc)]
print("gen", time.time() - t)
ctx = lib.FloriaHip(0)
if slots: ctx.set_slots(slots)
res = [ctx.upload(c.pileup) for c in contigs]
bc, bs, be = [], [], []
for i, c in enumerate(contigs):
    s, e = lib.get_range_with_lengths(c.snp_pos, C["block_length"])
    bc += [i] * len(s); bs += list(s); be += list(e)
par = lib.make_params(0.03125, C["max_ploidy"], C["beam"])
for it in range(3):
    t = time.time(); r = ctx.phase_blocks_batch(res, bc, bs, be, par); dt = time.time() - t
    tm = ctx.timing()
    print(f"iter {it}: blocks={len(bs)} wall={dt:.3f}s -> {len(bs)/dt:.0f} blocks/s | beam {tm['beam_ms']:.1f} opt {tm['optimize_ms']:.1f} sel {tm['select_ms']:.2f} h2d {tm['h2d_ms']:.1f} d2h {tm['d2h_ms']:.1f} total {tm['total_ms']:.1f} ms steps={tm['beam_steps']} bytes={tm['algorithmic_bytes']}")
print("best ploidy hist", np.bincount(r.best_ploidy), "tried", np.bincount(r.ploidies_tried), "margin", r.min_prune_margin, "mean reads/block", float(np.diff(r.read_off).mean()))
off = 0
for i in range(min(check, nc)):
    s, e = oracle.block_ranges(contigs[i].snp_pos, C["block_length"])
    t = time.time(); ro = oracle.phase_blocks(contigs[i].pileup, s, e, oracle.make_params(0.03125, C["max_ploidy"], C["beam"]), threads=8); dt = time.time() - t
    nb = len(s)
    lo, hi = int(r.read_off[off]), int(r.read_off[off + nb])
    ok = (np.array_equal(ro.best_ploidy, r.best_ploidy[off:off + nb]) and np.array_equal(ro.part, r.part[lo:hi]) and np.array_equal(ro.read_id, r.read_id[lo:hi])
          and np.array_equal(ro.mec, r.mec[off:off + nb]) and np.array_equal(ro.ploidies_tried, r.ploidies_tried[off:off + nb]))
    print(f"contig {i}: oracle {dt:.2f}s ({nb/dt:.1f} blocks/s, 8 thr) parity={ok} truth_ploidy={contigs[i].ploidy_truth} nblocks={nb}")
    off += nb
