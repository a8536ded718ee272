#!/usr/bin/env python
"""(CPU) the C++ oracle against the independent Python restatement of seam S1 (oracle/py_restatement.py) on many more random pileups than tests/test_py_restatement.py holds:
dyadic epsilon in the canonical arithmetic, 0.04 / 0.05 / 0.0437 in the running-sum mode with ascending orders.   usage: scripts/restatement_sweep.py [first seed = 100000] [count = 2000] [workers = 6] [big]"""
import sys
sys.path.insert(0, ".")
from multiprocessing import Pool


BIG = False


def big_case(seed):      # larger than the test suite's generator: up to 400 reads, 200 SNPs, reads of up to 80 cells, -p up to 8, -n up to 13
    import numpy as np
    from tests.helpers import random_pileup
    rng = np.random.default_rng(seed)
    ploidy = int(rng.integers(1, 7)); n_reads = int(rng.integers(50, 400)); n_snps = int(rng.integers(30, 200))
    p = random_pileup(rng, n_reads, n_snps, ploidy, max_len=int(rng.integers(4, 80)), alleles=2 if seed % 4 else 4, q0_frac=0.1 if seed % 3 == 0 else 0.0,
                      qlo=5 if seed % 5 else 20, qhi=40 if seed % 5 else 20, err=float(rng.choice([0.0, 0.02, 0.1, 0.25])))
    return p, n_snps, int(rng.integers(1, 9)), int(rng.integers(1, 14))


def run(seed):
    from tests.test_py_restatement import case, compare
    try:
        p, n_snps, P, B = big_case(seed) if BIG else case(seed)
        compare(p, n_snps, P, B, 0.03125 if seed % 2 else 0.0625, 0)
        compare(p, n_snps, P, B, (0.04, 0.05, 0.0437)[seed % 3], 2)
        return None
    except AssertionError as e:
        return f"seed {seed}: {str(e)[:300]}"


if __name__ == "__main__":
    from oracle import oracle
    oracle.build()
    s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    BIG = len(sys.argv) > 4 and sys.argv[4] == "big"          # (set before the pool forks)
    with Pool(workers) as pool:
        bad = [r for r in pool.imap_unordered(run, range(s0, s0 + cnt), chunksize=8) if r]
    for b in bad:
        print("MISMATCH", b)
    print(f"seeds {s0}..{s0 + cnt - 1}{' (large cases)' if BIG else ''}, each in both arithmetics (dyadic epsilon: canonical; 0.04 / 0.05 / 0.0437: running sums, ascending orders): {len(bad)} disagreements between the two restatements")
