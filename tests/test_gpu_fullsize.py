"""`-m gpu`: the BASELINE.json configurations at FULL size (the driver must see them, not a dev script).

At these sizes the oracle cannot phase everything in seconds, so every test combines
  * size-independent properties on ALL blocks: the read list of every block equals an independent numpy interval query, every read gets
    one haplotype < best_ploidy, ploidies_tried / best_ploidy are consistent, the pruning-margin certificate is > 1e-9, and the reported
    MEC of a block sample equals an oracle-independent numpy recomputation from the returned partition;
  * a random block sample against the oracle, bit for bit (partitions, chosen ploidy, f64 MEC vector).
"""
import numpy as np
import pytest

from floria_amd import synth
from floria_amd.pileup import reads_in_interval
from tests.test_gpu_parity import numpy_mec_no_phred

pytestmark = pytest.mark.gpu
EPS = 0.03125


def phase_config(gpu_ctx, hip_lib, cfg, n_contigs, first=0, packed=False):
    C = synth.CONFIGS[cfg]
    contigs = [synth.make_config_contig(cfg, first + i) for i in range(n_contigs)]
    bc, bs, be = [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, C["block_length"])
        bc += [i] * len(s); bs += list(s); be += list(e)
    par = hip_lib.make_params(EPS, C["max_ploidy"], C["beam"])
    if packed:                                                         # the bench's path: compact wire form, pipelined upload + expansion + S1
        arena, parr, _ = hip_lib.pack_pileups([c.pileup for c in contigs])
        r = gpu_ctx.phase_pileups_batch(parr, bc, bs, be, par)
    else:                                                              # the same from CSR arrays in pinned memory
        arena, pinned = hip_lib.pin_pileups([c.pileup for c in contigs])
        r = gpu_ctx.phase_pileups_batch(pinned, bc, bs, be, par)
    arena.free()
    return C, contigs, np.array(bc), np.array(bs), np.array(be), r


def check_properties(r, contigs, bc, bs, be, C, mec_sample, rng):
    assert r.n_blocks == len(bs) and r.min_prune_margin > 1e-9
    ne = np.diff(r.read_off.astype(np.int64)) > 0
    assert np.all(r.best_ploidy[ne] >= 1) and np.all(r.best_ploidy[~ne] == 0) and np.all(r.best_ploidy <= C["max_ploidy"])
    assert np.all(r.ploidies_tried >= r.best_ploidy) and np.all(r.ploidies_tried <= C["max_ploidy"])
    # untried ploidies report 0; tried ones a finite MEC
    for p in range(C["max_ploidy"]):
        assert np.all(r.mec[r.ploidies_tried <= p, p] == 0.0)
    for blk in range(r.n_blocks):
        ids, part = r.block(blk)
        if blk % 97 == 0 or r.n_blocks < 2000:                            # read lists vs the interval query
            assert np.array_equal(ids, reads_in_interval(contigs[bc[blk]].pileup, bs[blk], be[blk]))
        if len(ids):
            assert part.max() < r.best_ploidy[blk]
    for blk in rng.choice(r.n_blocks, size=min(mec_sample, r.n_blocks), replace=False):
        ids, part = r.block(int(blk))
        if len(ids) == 0:
            continue
        bp = int(r.best_ploidy[blk])
        assert r.mec[blk, bp - 1] == numpy_mec_no_phred(contigs[bc[blk]].pileup, ids.astype(np.int64), part, bp, EPS), f"block {blk}"


def check_sample_vs_oracle(oracle_mod, r, contigs, bc, bs, be, C, sample):
    par = oracle_mod.make_params(EPS, C["max_ploidy"], C["beam"])
    by_contig = {}
    for blk in sample:
        by_contig.setdefault(int(bc[blk]), []).append(int(blk))
    for ci, blks in by_contig.items():
        ro = oracle_mod.phase_blocks(contigs[ci].pileup, bs[blks], be[blks], par, threads=16)
        for k, blk in enumerate(blks):
            ids, part = r.block(blk)
            oid, opart = ro.block(k)
            assert ro.best_ploidy[k] == r.best_ploidy[blk] and ro.ploidies_tried[k] == r.ploidies_tried[blk], f"block {blk}"
            assert np.array_equal(oid, ids) and np.array_equal(opart, part), f"block {blk}"
            assert np.array_equal(ro.mec[k].view(np.uint64), r.mec[blk].view(np.uint64)), f"block {blk}"


def test_config2_full_size_every_block_vs_oracle(gpu_ctx, hip_lib, oracle_mod):
    # 1 contig, 10k SNPs, 20k long reads, 3 strains: all ~146 blocks against the oracle
    C, contigs, bc, bs, be, r = phase_config(gpu_ctx, hip_lib, 2, 1)
    assert 120 <= r.n_blocks <= 180
    rng = np.random.default_rng(2)
    check_properties(r, contigs, bc, bs, be, C, 64, rng)
    check_sample_vs_oracle(oracle_mod, r, contigs, bc, bs, be, C, range(r.n_blocks))


def test_config3_full_size_properties_and_oracle_sample(gpu_ctx, hip_lib, oracle_mod):
    # 200 contigs, 500k SNPs, 2M paired short reads: ~93k blocks; properties on all of them, 256 random blocks against the oracle
    C, contigs, bc, bs, be, r = phase_config(gpu_ctx, hip_lib, 3, 200)
    assert r.n_blocks > 80000
    rng = np.random.default_rng(3)
    check_properties(r, contigs, bc, bs, be, C, 512, rng)
    check_sample_vs_oracle(oracle_mod, r, contigs, bc, bs, be, C, rng.choice(r.n_blocks, size=256, replace=False))


def test_config4_shard_of_250_contigs(gpu_ctx, hip_lib, oracle_mod):
    # the per-GPU share of the 8-GPU strong-scaling point (2000 contigs / 8): ~1.8k blocks, fewer jobs than wave slots, so the batch
    # runs its ploidies as speculative stages; 48 random blocks against the oracle
    C, contigs, bc, bs, be, r = phase_config(gpu_ctx, hip_lib, 4, 250, first=1000)
    assert 1500 < r.n_blocks < 2200
    rng = np.random.default_rng(4)
    check_properties(r, contigs, bc, bs, be, C, 128, rng)
    check_sample_vs_oracle(oracle_mod, r, contigs, bc, bs, be, C, rng.choice(r.n_blocks, size=48, replace=False))


def test_config5_full_size_wide_beam(gpu_ctx, hip_lib, oracle_mod):
    # 1 contig, 50k SNPs, 100k long reads, 8 strains, -p 8 -n 40 (beam_wide_kernel: up to 320 states per job): properties on all ~725
    # blocks, 64 random blocks against the oracle
    C, contigs, bc, bs, be, r = phase_config(gpu_ctx, hip_lib, 5, 1)
    assert 600 <= r.n_blocks <= 850
    rng = np.random.default_rng(5)
    check_properties(r, contigs, bc, bs, be, C, 64, rng)
    check_sample_vs_oracle(oracle_mod, r, contigs, bc, bs, be, C, rng.choice(r.n_blocks, size=64, replace=False))


def test_config4_full_size_2000_contigs(gpu_ctx, hip_lib, oracle_mod):
    # THE headline workload at full size, through the call bench.py times (floria_hip_phase_pileups_batch_packed: 2000 contigs, 1M SNPs, 5M long
    # reads, ~14.5k blocks, 2.65 GB of CSR pileup travelling as 0.68 GB): properties on every block, 256 random blocks against the oracle bit for bit
    C, contigs, bc, bs, be, r = phase_config(gpu_ctx, hip_lib, 4, 2000, packed=True)
    assert 14000 < r.n_blocks < 15000
    rng = np.random.default_rng(44)
    check_properties(r, contigs, bc, bs, be, C, 256, rng)
    check_sample_vs_oracle(oracle_mod, r, contigs, bc, bs, be, C, rng.choice(r.n_blocks, size=256, replace=False))
