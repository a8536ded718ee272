#!/usr/bin/env python
"""(CPU) VERDICT r5 #9: how much can alignment::realign's CALL depend on block-aligner's fixed-block walk (alignment.rs:44-52: block size 8 = min = max) instead of the exact
affine-gap DP this repository scores (realign_kernel.h, synth_bam.nw_affine_batch)?  The crate is not vendored and not available offline, so its direction / tie / step rules
cannot be restated with certainty; instead EVERY plausible fixed-block walk is run (direction by the maxima or by the sums of the block's right and bottom borders, ties right
or down, steps of 1, 2, 4 or 8 cells: scripts/probes/block_walk.c) on noisy, indel-rich 32 x 32 windows, and the calls (argmax over the two alleles, first best) are compared
with the exact DP's.  If no member of the family changes calls beyond a rate r, neither does the crate's walk, whichever member it is.
usage: scripts/block_walk_family.py [windows = 40000] [seed = 1]"""
import ctypes as C, os, subprocess, sys
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "probes", "block_walk.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(here, "probes", "block_walk.c")])
L = C.CDLL(so)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BASES = np.frombuffer(b"ACGT", np.uint8)
FL = 16


def score(Q, R, B, step=8, rule=0, tie=0):
    out = np.zeros(len(Q), np.int32)
    L.batch(Q.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p), C.c_int(len(Q)), C.c_int(Q.shape[1]), C.c_int(R.shape[1]), C.c_int(B), C.c_int(step), C.c_int(rule), C.c_int(tie),
            out.ctypes.data_as(C.c_void_p))
    return out


def windows(n, sub, n_indel):
    """reference windows R0 / R1 (the two alleles at the centre) and a read window: the true allele's window with substitutions at rate `sub` and `n_indel` indels of 1-3 bases,
    cut back to 32 bases around the SNP the way realign cuts the read (fixed offsets: the read position of the SNP +- 16)"""
    Q = np.zeros((n, 2 * FL), np.uint8); R0 = np.zeros((n, 2 * FL), np.uint8); R1 = np.zeros((n, 2 * FL), np.uint8); truth = np.zeros(n, np.int8)
    for x in range(n):
        ref = BASES[rng.integers(0, 4, size=6 * FL)].copy()
        c = 3 * FL
        alt = BASES[(int(np.searchsorted(BASES, ref[c])) + 1 + int(rng.integers(0, 3))) % 4]
        r0 = ref[c - FL:c + FL].copy(); r1 = r0.copy(); r1[FL] = alt
        t = int(rng.integers(0, 2)); truth[x] = t
        read = ref.copy(); read[c] = alt if t else ref[c]
        pos = list(range(len(read)))             # read offset -> reference offset bookkeeping through the edits
        seq = list(read)
        snp_at = c
        for _ in range(n_indel):
            where = int(rng.integers(c - FL + 1, c + FL - 1))
            k = int(rng.integers(1, 4))
            if where == snp_at:
                continue
            if rng.random() < 0.5:               # insertion into the read
                ins = list(BASES[rng.integers(0, 4, size=k)])
                seq[where:where] = ins
                if where <= snp_at:
                    snp_at += k
            else:                                 # deletion from the read (never the SNP itself)
                lo, hi = where, min(where + k, len(seq))
                if lo <= snp_at < hi:
                    continue
                del seq[lo:hi]
                if hi <= snp_at:
                    snp_at -= hi - lo
        seq = np.array(seq, np.uint8)
        hit = np.nonzero(rng.random(len(seq)) < sub)[0]
        hit = hit[hit != snp_at]
        seq[hit] = BASES[(np.searchsorted(BASES, seq[hit]) + rng.integers(1, 4, size=len(hit))) % 4]
        Q[x] = seq[snp_at - FL:snp_at + FL]
        R0[x] = r0; R1[x] = r1
    return Q, R0, R1, truth


print(f"# realign's call under every plausible fixed-block walk (block 8) against the exact DP; {N} windows per regime; call = allele 0 unless allele 1 scores strictly higher (first best, alignment.rs:52-56)")
for sub, n_indel in ((0.0, 0), (0.05, 1), (0.10, 2), (0.15, 3), (0.10, 5)):
    Q, R0, R1, truth = windows(N, sub, n_indel)
    e0, e1 = score(Q, R0, 0), score(Q, R1, 0)
    exact = (e1 > e0).astype(np.int8)
    line = [f"substitutions {sub:.2f}, {n_indel} indels of 1-3 bases per window: exact DP calls the sequenced allele in {100 * float((exact == truth).mean()):.2f} %"]
    worst = worst_overlap = 0.0
    for rule in (0, 1):
        for tie in (0, 1):
            for B, step in ((8, 1), (8, 2), (8, 4), (8, 8), (16, 4), (16, 8)):       # (16: the crate raises a block size below its SIMD width to that width, 16 lanes of i16 under AVX2)
                w0, w1 = score(Q, R0, B, step, rule, tie), score(Q, R1, B, step, rule, tie)
                call = (w1 > w0).astype(np.int8)
                diff = float((call != exact).mean())
                worst = max(worst, diff)
                if step < B:
                    worst_overlap = max(worst_overlap, diff)
                lost = float(((w0 != e0) | (w1 != e1)).mean())
                line.append(f"  block {B} rule {'max' if rule == 0 else 'sum'} tie {'right' if tie == 0 else 'down'} step {step}: calls differing from the exact DP {100 * diff:.3f} %, windows where a walk's SCORE differs {100 * lost:.2f} %, walk calls the sequenced allele in {100 * float((call == truth).mean()):.2f} %")
    print("\n".join(line))
    print(f"  -> worst member of the family: {100 * worst:.3f} % of the calls differ; worst member whose consecutive blocks overlap (step < block, as in the published design): {100 * worst_overlap:.3f} %")
