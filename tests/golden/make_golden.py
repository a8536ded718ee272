"""Regenerates tests/golden/*.npz.

The reference is Rust and cannot be built or imported in this environment, so these vectors are
NOT outputs of the reference: they are (a) the hand-derived KAT of SURVEY.md Appendix E and (b) regression
vectors produced by the CPU oracle (oracle/floria_oracle.cpp) on seeded synthetic pileups.  They pin the oracle
against silent drift and give the GPU tests inputs/outputs that do not need the oracle at run time.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from floria_amd import synth  # noqa: E402
from floria_amd.pileup import Pileup  # noqa: E402
from oracle import oracle  # noqa: E402
from tests.helpers import random_pileup  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(name, pile, s, e, eps, P, B, sens=2, stop=1):
    r = oracle.phase_blocks(pile, s, e, oracle.make_params(eps, P, B, sens, stop))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), read_off=pile.read_off, snp=pile.snp, allele=pile.allele, qual=pile.qual,
                        first=pile.first, last=pile.last, blk_start=np.asarray(s, np.uint32), blk_end=np.asarray(e, np.uint32),
                        eps=eps, max_ploidy=P, beam=B, sens=sens, stop=stop, best_ploidy=r.best_ploidy, ploidies_tried=r.ploidies_tried,
                        out_read_off=r.read_off, out_read_id=r.read_id, out_part=r.part, mec=r.mec)
    print(name, "blocks", len(s), "best", r.best_ploidy[:10])


if __name__ == "__main__":
    dump("kat1", Pileup.from_reads([([1, 2, 3, 4], [i % 2] * 4, [20] * 4) for i in range(6)]), [1], [4], 0.03125, 5, 10)
    c = synth.make_config_contig(1, 0)
    s, e = oracle.block_ranges(c.snp_pos, 10000)
    dump("cfg1_substitute", c.pileup, s, e, 0.03125, 5, 10)
    c = synth.make_config_contig(3, 0, scale=0.1)
    s, e = oracle.block_ranges(c.snp_pos, 500)
    dump("cfg3_x0.1", c.pileup, s, e, 0.03125, 5, 10)
    c = synth.make_config_contig(4, 3, scale=0.25)
    s, e = oracle.block_ranges(c.snp_pos, 10000)
    dump("cfg4_x0.25_eps04", c.pileup, s, e, 0.04, 5, 10)
    rng = np.random.default_rng(11)
    p = random_pileup(rng, 120, 40, 3, alleles=4, q0_frac=0.05)
    dump("random_4allele_q0", p, [1, 10, 25], [15, 30, 40], 0.03125, 4, 6, sens=1)
