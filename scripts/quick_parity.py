"""Dev helper: GPU vs oracle on a few small workloads (prints mismatches)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from floria_amd import synth, lib
from floria_amd.pileup import Pileup
from oracle import oracle

def compare(name, pile, s, e, eps, P=5, B=10, threads=8):
    par = lib.make_params(eps, P, B)
    t = time.time(); ro = oracle.phase_blocks(pile, s, e, oracle.make_params(eps, P, B), threads=threads); to = time.time() - t
    ctx = lib.FloriaHip(0)
    t = time.time(); rg = ctx.phase_blocks(pile, s, e, par); tg = time.time() - t
    ok = True
    for f in ("best_ploidy", "ploidies_tried", "read_off", "read_id", "part"):
        a, b = getattr(ro, f), getattr(rg, f)
        if a.shape != b.shape or not np.array_equal(a, b):
            ok = False
            bad = np.nonzero(a != b)[0] if a.shape == b.shape else []
            print(f"  MISMATCH {f}: {len(bad)} of {a.size}", bad[:10])
    if not np.array_equal(ro.mec, rg.mec):
        ok = False; print("  MISMATCH mec\n", ro.mec[:4], "\n", rg.mec[:4])
    print(f"{name}: blocks={len(s)} reads/blk={np.diff(ro.read_off)[:6]} best={ro.best_ploidy[:8]} ok={ok} oracle={to:.2f}s gpu={tg:.2f}s margin o={ro.min_prune_margin:.3g} g={rg.min_prune_margin:.3g}")
    print("   timing", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in ctx.timing().items()})
    ctx.close()
    return ok

if __name__ == "__main__":
    ok = True
    reads = [([1, 2, 3, 4], [i % 2] * 4, [20] * 4) for i in range(6)]
    ok &= compare("KAT-1", Pileup.from_reads(reads), [1], [4], 0.03125)
    c = synth.make_config_contig(1, 0)
    s, e = oracle.block_ranges(c.snp_pos, 10000)
    ok &= compare("cfg1", c.pileup, s, e, 0.03125)
    c = synth.make_config_contig(4, 0, scale=0.3)
    s, e = oracle.block_ranges(c.snp_pos, 10000)
    ok &= compare("cfg4x0.3", c.pileup, s, e, 0.03125)
    ok &= compare("cfg4x0.3 eps.04", c.pileup, s, e, 0.04)
    c = synth.make_config_contig(3, 0, scale=0.2)
    s, e = oracle.block_ranges(c.snp_pos, 500)
    ok &= compare("cfg3x0.2", c.pileup, s, e, 0.03125)
    if len(sys.argv) > 1:
        c = synth.make_config_contig(4, 1)
        s, e = oracle.block_ranges(c.snp_pos, 10000)
        ok &= compare("cfg4 full contig", c.pileup, s, e, 0.03125)
    print("ALL OK" if ok else "FAILURES")
