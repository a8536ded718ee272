"""The C++ oracle against a SECOND restatement of seam S1 (oracle/py_restatement.py: Python dicts / sets / lists written straight from the Rust, without
consulting the C++): the only available defence against a misreading shared by the oracle and the kernels built against it.  It does not pin
parity — neither side has met the Rust binary.

  * dyadic epsilon (every f64 sum exact in any order): the oracle's canonical (Q24, #eps) arithmetic == the Python running sums, bit for bit;
  * epsilon = 0.04 / 0.05 / 0.0437: the oracle's running-sum mode with every container iterated in ascending key order (set_arith_mode(2)) == the Python
    restatement, whose dict / set iterations are sorted the same way — and the canonical mode really is a different function there.
Cases: ploidy 1-5, -n 1/2/3/10, -p 1..5, biallelic and 4-allele sites, q = 0 cells (zero-weight keys), constant qualities (exact ties), 5-90 reads."""
import numpy as np
import pytest

from oracle import oracle, py_restatement as pr
from tests.helpers import random_pileup


def case(seed):
    rng = np.random.default_rng(seed)
    ploidy = int(rng.integers(1, 5)); n_reads = int(rng.integers(5, 90)); n_snps = int(rng.integers(4, 40))
    p = random_pileup(rng, n_reads, n_snps, ploidy, max_len=int(rng.integers(1, 16)), alleles=2 if seed % 4 else 4, q0_frac=0.1 if seed % 3 == 0 else 0.0,
                      qlo=5 if seed % 5 else 20, qhi=40 if seed % 5 else 20, err=0.02 if seed % 2 else 0.1)
    return p, n_snps, int(rng.integers(1, 6)), int(rng.choice([1, 2, 3, 10]))


def compare(p, n_snps, P, B, eps, arith):
    s = np.array([1], np.uint32); e = np.array([n_snps], np.uint32)
    oracle.set_arith_mode(arith)
    try:
        ro = oracle.phase_blocks(p, s, e, oracle.make_params(eps, P, B), threads=1)
    finally:
        oracle.set_arith_mode(0)
    g = pr.get_local_hap_blocks(pr.frags_from_pileup(p), 1, n_snps, eps, max_ploidy=P, max_number_solns=B)
    ids, part = ro.block(0)
    assert g["reads"] == ids.tolist()
    assert (g["best_ploidy"], g["tried"]) == (int(ro.best_ploidy[0]), int(ro.ploidies_tried[0]))
    assert g["part"] == part.tolist()
    assert np.array_equal(np.array(g["mec_vector"]).view(np.uint64), ro.mec[0].view(np.uint64)), (g["mec_vector"], ro.mec[0])     # bit for bit
    assert min(g["margins"]) == ro.min_prune_margin
    return ro


@pytest.mark.parametrize("chunk", range(4))
def test_dyadic_epsilon_python_running_sums_equal_the_oracle(chunk):
    best = []
    for seed in range(1000 + 60 * chunk, 1000 + 60 * (chunk + 1)):
        p, n_snps, P, B = case(seed)
        best.append(int(compare(p, n_snps, P, B, 0.03125 if seed % 2 else 0.0625, 0).best_ploidy[0]))
    assert len(set(best)) >= 3                      # (the cases are not all trivial)


@pytest.mark.parametrize("chunk", range(4))
def test_decimal_epsilon_python_running_sums_equal_the_oracle_in_running_mode(chunk):
    differs = 0
    for seed in range(5000 + 60 * chunk, 5000 + 60 * (chunk + 1)):
        p, n_snps, P, B = case(seed)
        eps = (0.04, 0.05, 0.0437)[seed % 3]
        r2 = compare(p, n_snps, P, B, eps, 2)
        r0 = oracle.phase_blocks(p, np.array([1], np.uint32), np.array([n_snps], np.uint32), oracle.make_params(eps, P, B), threads=1)
        differs += not (np.array_equal(r0.part, r2.part) and np.array_equal(r0.mec.view(np.uint64), r2.mec.view(np.uint64)))
    assert differs > 10                             # the canonical form is a different function at a non-dyadic epsilon


def test_binary_heap_restatement_against_the_oracle_heap():
    # the two std::collections::BinaryHeap restatements (C++: oracle.heap_trace, Python: pr.BinaryHeap) on score sequences full of ties
    rng = np.random.default_rng(11)
    for _ in range(200):
        n = int(rng.integers(1, 40)); limit = int(rng.integers(1, 12))
        scores = rng.integers(0, 4, size=n).astype(np.float64) * 0.5
        ids, srt = oracle.heap_trace(scores, limit)
        h = pr.BinaryHeap()
        for i, sc in enumerate(scores):
            h.push((pr.SearchNode(None, i, float(sc), None, None), None))
            if len(h) > limit:
                h.pop()
        assert [x[0].part for x in h.data] == ids.tolist()
        assert [x[0].part for x in h.into_sorted_vec()] == srt.tolist()
