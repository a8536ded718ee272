"""Shared helpers for the parity tests."""
import numpy as np

from floria_amd.pileup import Pileup


def assert_block_results_equal(ro, rg, ctx=""):
    for f in ("best_ploidy", "ploidies_tried", "read_off", "read_id", "part"):
        a, b = getattr(ro, f), getattr(rg, f)
        assert a.shape == b.shape, f"{ctx}: {f} shape {a.shape} vs {b.shape}"
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError(f"{ctx}: {f} differs at {len(bad)} of {a.size} entries, first {bad[:8]}: oracle {a[bad[:8]]} hip {b[bad[:8]]}")
    # mec_vector entries are f64 built from exact integer sums + m*eps: bit-identical, not approximately equal
    assert np.array_equal(ro.mec.view(np.uint64), rg.mec.view(np.uint64)), f"{ctx}: mec_vector differs\n{ro.mec}\n{rg.mec}"


def random_pileup(rng, n_reads, n_snps, ploidy, max_len=12, alleles=2, qlo=5, qhi=40, err=0.05, drop=0.1, q0_frac=0.0):
    """Small ragged random pileup with `ploidy` planted haplotypes."""
    hap = rng.integers(0, alleles, size=(ploidy, n_snps))
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(1, max_len + 1))
        s = int(rng.integers(1, n_snps + 1))
        e = min(n_snps, s + L - 1)
        snps = np.arange(s, e + 1)
        keep = rng.random(len(snps)) >= drop
        keep[0] = True
        snps = snps[keep]
        st = int(rng.integers(0, ploidy))
        al = hap[st, snps - 1].copy()
        flip = rng.random(len(snps)) < err
        al[flip] = rng.integers(0, alleles, size=int(flip.sum()))
        q = rng.integers(qlo, qhi + 1, size=len(snps))
        if q0_frac:
            q[rng.random(len(snps)) < q0_frac] = 0
        reads.append((snps, al, q))
    return Pileup.from_reads(reads)
