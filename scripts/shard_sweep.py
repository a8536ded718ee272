"""Dev helper: S1 time of config-4 shards of 250/500/1000/2000 contigs under the ploidy-stage modes, and upload rates
(pageable through the staging ring vs pinned DMA).  Prints one JSON line per measurement."""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from floria_amd import synth, lib

sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "250,500,1000,2000".split(","))]
specs = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2".split(","))]
nmax = max(sizes)
t = time.time()
contigs = [synth.make_config_contig(4, i) for i in range(nmax)]
print("gen %.1fs" % (time.time() - t), flush=True)
ctx = lib.FloriaHip(0)
par = lib.make_params(0.03125)
piles = [c.pileup for c in contigs]
tot_bytes = sum(4 * (p.n_reads + 1) + 8 * p.n_reads + 6 * p.n_cells for p in piles)
for thr in (1, 4, 8, 12):
    ctx.set_option("stage_threads", thr)
    for it in range(2):
        t = time.perf_counter(); res = ctx.upload_batch(piles); dt = time.perf_counter() - t
        tm = ctx.timing()
        for r in res: r.free()
    print(json.dumps({"upload": "pageable", "threads": thr, "contigs": nmax, "MB": tot_bytes / 1e6, "ms": dt * 1e3, "GBps": tot_bytes / dt / 1e9, "flatten_ms": tm["select_ms"], "h2d_ms": tm["h2d_ms"]}), flush=True)
arena, pinned = lib.pin_pileups(piles)
for it in range(3):
    t = time.perf_counter(); res = ctx.upload_batch(pinned); dt = time.perf_counter() - t
    tm = ctx.timing()
    if it < 2:
        for r in res: r.free()
print(json.dumps({"upload": "pinned", "contigs": nmax, "MB": tot_bytes / 1e6, "ms": dt * 1e3, "GBps": tot_bytes / dt / 1e9, "flatten_ms": tm["select_ms"], "h2d_ms": tm["h2d_ms"]}), flush=True)
blocks = [lib.get_range_with_lengths(c.snp_pos, 10000) for c in contigs]
ref = {}
for n in sizes:
    bc, bs, be = [], [], []
    for i in range(n):
        s, e = blocks[i]
        bc += [i] * len(s); bs += list(s); be += list(e)
    bc, bs, be = (np.ascontiguousarray(x, np.uint32) for x in (bc, bs, be))
    for spec in specs:
        ctx.set_option("speculate", spec)
        best = 1e9
        for it in range(4):
            t = time.perf_counter(); r = ctx.phase_blocks_batch(res[:n], bc, bs, be, par); dt = time.perf_counter() - t
            if it: best = min(best, dt)
        tm = ctx.timing()
        key = (n,)
        if key not in ref: ref[key] = r
        same = all(np.array_equal(getattr(ref[key], f), getattr(r, f)) for f in ("best_ploidy", "ploidies_tried", "part", "read_id")) and np.array_equal(ref[key].mec.view(np.uint64), r.mec.view(np.uint64)) and ref[key].min_prune_margin == r.min_prune_margin
        print(json.dumps({"contigs": n, "blocks": len(bs), "speculate": spec, "ms": best * 1e3, "blocks_per_s": len(bs) / best, "beam_ms": tm["beam_ms"], "opt_ms": tm["optimize_ms"], "phase_ms": tm["phase_ms"], "groups": tm["streams"], "stage_width": tm["stage_width"], "jobs": tm["jobs"], "beam_steps": tm["beam_steps"], "same_as_first": bool(same)}), flush=True)
