"""`-m "not gpu"`: the oracle reproduces the committed fixtures (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

from floria_amd.pileup import Pileup

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load_golden(path):
    z = np.load(path)
    pile = Pileup(z["read_off"], z["snp"], z["allele"], z["qual"], z["first"], z["last"])
    return z, pile


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_fixture(oracle_mod, path):
    z, pile = load_golden(path)
    r = oracle_mod.phase_blocks(pile, z["blk_start"], z["blk_end"],
                                oracle_mod.make_params(float(z["eps"]), int(z["max_ploidy"]), int(z["beam"]), int(z["sens"]), int(z["stop"])), threads=2)
    assert np.array_equal(r.best_ploidy, z["best_ploidy"]) and np.array_equal(r.ploidies_tried, z["ploidies_tried"])
    assert np.array_equal(r.read_off, z["out_read_off"]) and np.array_equal(r.read_id, z["out_read_id"])
    assert np.array_equal(r.part, z["out_part"])
    assert np.array_equal(r.mec.view(np.uint64), z["mec"].view(np.uint64))


def test_oracle_is_thread_count_invariant(oracle_mod):
    z, pile = load_golden(GOLD[0])
    p = oracle_mod.make_params(float(z["eps"]), int(z["max_ploidy"]), int(z["beam"]))
    a = oracle_mod.phase_blocks(pile, z["blk_start"], z["blk_end"], p, threads=1)
    b = oracle_mod.phase_blocks(pile, z["blk_start"], z["blk_end"], p, threads=4)
    assert np.array_equal(a.part, b.part) and np.array_equal(a.mec, b.mec)
