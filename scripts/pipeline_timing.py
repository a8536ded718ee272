#!/usr/bin/env python
"""Wall time of every device entry point on one bench-shaped batch (BASELINE config 4, N contigs): S1, hap graph, S2, haploset stats, HAPQ.
usage: scripts/pipeline_timing.py [n_contigs=200]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from floria_amd import lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = lib.FloriaHip(0)
contigs = [synth.make_config_contig(4, i) for i in range(n)]
res = [ctx.upload(c.pileup) for c in contigs]
par = lib.make_params(0.03125)
bc, bs, be = [], [], []
for i, c in enumerate(contigs):
    s, e = lib.get_range_with_lengths(c.snp_pos, 10000)
    bc += [i] * len(s); bs += list(s); be += list(e)

def timed(f, reps=3):
    f(); t = time.perf_counter()
    for _ in range(reps): out = f()
    return (time.perf_counter() - t) / reps * 1e3, out

t_s1, r = timed(lambda: ctx.phase_blocks_batch(res, bc, bs, be, par))
r = ctx.phase_blocks_batch(res, bc, bs, be, par)
t_hg, hg = timed(lambda: ctx.hap_graph(r), 1)
groups, ranges, gc = [], [], []
for b in range(r.n_blocks):
    for part in r.partitions(b):
        if len(part): groups.append(part); ranges.append((int(bs[b]), int(be[b]))); gc.append(bc[b])
t_s2, out = timed(lambda: ctx.reassign_batch(res, gc, groups, ranges, 0.03125))
tm_s2 = ctx.timing()
fg, fr, fc = [], [], []
for ci, g in enumerate(out):
    for k in range(g.n_groups): fg.append(g.group(k)); fr.append(tuple(int(x) for x in g.range[k])); fc.append(ci)
t_st, st = timed(lambda: ctx.haploset_stats(res, fc, fg, fr))
def all_hapq():
    o = 0; tot = []
    for ci in range(n):
        idx = [k for k in range(o, len(fc)) if fc[k] == ci]
        if not idx: continue
        o = idx[-1] + 1
        tot.append(ctx.hapq(res[ci], [fg[k] for k in idx], [fr[k] for k in idx], contigs[ci].snp_pos, 10000)[0])
    return tot
t_hq, hq = timed(all_hapq, 1)
t_hqb, hqb = timed(lambda: ctx.hapq_batch(res, fc, fg, fr, [c.snp_pos for c in contigs], 10000), 2)
assert np.array_equal(np.concatenate(hq), hqb[0])
print(f"{n} contigs, {r.n_blocks} blocks, {len(groups)} block haplosets -> {len(fg)} final haplosets")
print(f"S1 phase_blocks_batch {t_s1:.1f} ms | hap_graph {t_hg:.1f} ms | S2 reassign_batch {t_s2:.1f} ms | haploset_stats {t_st:.1f} ms | hapq (per contig, {n} calls) {t_hq:.1f} ms | hapq_batch {t_hqb:.1f} ms")
print("S2 device timing:", {k: round(v, 2) for k, v in tm_s2.items() if k in ("reassign_ms", "h2d_ms", "d2h_ms", "total_ms")})
print("HAPQ histogram:", np.bincount(np.concatenate(hq), minlength=61)[[0, 10, 20, 30, 40, 50, 60]].tolist(), "...")
# the C entry point alone (arrays prebuilt): how much of S2 is libfloria_hip.so and how much the Python list handling
import ctypes as C
from floria_amd import _capi as capi
arr = (C.c_void_p * len(res))(*[c._h for c in res])
gcv = np.ascontiguousarray(gc, np.uint32)
off = np.zeros(len(groups) + 1, np.uint64); off[1:] = np.cumsum([len(g) for g in groups])
reads = np.ascontiguousarray(np.concatenate(groups), np.uint32)
rng = np.ascontiguousarray(np.asarray(ranges, np.uint32).reshape(-1))
L = lib.load()
def c_only():
    out = C.POINTER(C.POINTER(capi.CGroups))()
    rc = L.floria_hip_reassign_batch(ctx._h, arr, C.c_uint32(len(res)), capi.ptr(gcv, C.c_uint32), capi.ptr(off, C.c_uint64), capi.ptr(reads, C.c_uint32),
                                     capi.ptr(rng, C.c_uint32), C.c_uint32(len(groups)), None, None, C.c_double(0.03125), C.byref(out))
    assert rc == 0
    L.floria_hip_groups_array_free(out, C.c_uint32(len(res)))
t_c, _ = timed(c_only)
print(f"floria_hip_reassign_batch alone: {t_c:.1f} ms")
