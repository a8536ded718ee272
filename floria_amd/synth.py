"""Seeded synthetic reads x SNPs pileups for the BASELINE.json configs.

BASELINE.json fixes contigs / SNPs / reads / ploidy only; everything else is chosen here and is part
of every reported number:

  per contig : `ploidy` strain haplotypes over S biallelic SNPs (allele ~ Bernoulli(0.5), columns that
               are monomorphic across strains are redrawn); SNP genome positions = cumulative
               Geometric(1/100 bp) gaps; strain abundances ~ Dirichlet(1)
  long reads : start ~ U[0, contig_len), length ~ LogNormal(mean 10 kb, sigma 0.4) clipped to [1 kb, 50 kb]
  short pairs: 2 x 150 bp, insert ~ Normal(500, 50)
  per covered SNP: allele flipped w.p. 0.03, dropped (no call) w.p. 0.02, qual ~ U{5..40}
  reads without SNPs are discarded; reads are sorted by Frag::cmp

RNG: numpy Generator(PCG64(SeedSequence([1577 + config, contig_index]))) — contigs are independent
streams, so any rank can generate exactly its own shard.
"""
from dataclasses import dataclass, field

import numpy as np

from .pileup import Pileup

BASE_SEED = 1577


@dataclass
class Contig:
    name: str
    pileup: Pileup
    snp_pos: np.ndarray          # uint64 [S] genome position (bp) of SNP i (0-based index i <-> SNP i+1)
    ploidy_truth: int
    strain: np.ndarray = field(default=None, repr=False)   # true strain of every read (for sanity checks)
    layout: dict = field(default=None, repr=False)         # keep_layout: genome interval(s) of every read, contig length (synth_bam.py)


@dataclass
class Workload:
    name: str
    contigs: list
    epsilon: float
    max_ploidy: int
    beam: int
    block_length: int
    ploidy_sensitivity: int = 2
    stopping_heuristic: int = 1
    snp_density: float = 0.0005


def make_contig(seed_seq, n_snps, n_reads, ploidy, kind="long", name="ctg", snp_gap_mean=100.0,
                flip=0.03, drop=0.02, qlo=5, qhi=40, keep_truth=False, keep_layout=False):
    rng = np.random.Generator(np.random.PCG64(seed_seq))
    hap = rng.integers(0, 2, size=(ploidy, n_snps), dtype=np.uint8)
    if ploidy >= 2:
        for _ in range(64):
            mono = (hap == hap[0]).all(axis=0)
            if not mono.any():
                break
            hap[:, mono] = rng.integers(0, 2, size=(ploidy, int(mono.sum())), dtype=np.uint8)
    pos = np.cumsum(rng.geometric(1.0 / snp_gap_mean, size=n_snps)).astype(np.int64)
    contig_len = int(pos[-1] + snp_gap_mean)
    abund = rng.dirichlet(np.ones(ploidy))
    strain = rng.choice(ploidy, size=n_reads, p=abund)
    start = rng.integers(0, contig_len, size=n_reads)
    insert = None
    if kind == "long":
        sigma = 0.4
        length = np.clip(rng.lognormal(np.log(10000.0) - sigma * sigma / 2, sigma, size=n_reads), 1000, 50000).astype(np.int64)
        lo1 = np.searchsorted(pos, start, "left")
        hi1 = np.searchsorted(pos, start + length, "left")
        lo2 = hi2 = hi1
    elif kind == "short":
        insert = np.maximum(rng.normal(500.0, 50.0, size=n_reads), 150).astype(np.int64)
        lo1 = np.searchsorted(pos, start, "left")
        hi1 = np.searchsorted(pos, start + 150, "left")
        lo2 = np.maximum(np.searchsorted(pos, start + insert - 150, "left"), hi1)
        hi2 = np.maximum(np.searchsorted(pos, start + insert, "left"), lo2)
    else:
        raise ValueError(kind)
    c1 = hi1 - lo1
    c2 = hi2 - lo2
    cnt = c1 + c2
    tot = int(cnt.sum())
    rid = np.repeat(np.arange(n_reads), cnt)
    off = np.zeros(n_reads + 1, np.int64)
    off[1:] = np.cumsum(cnt)
    k = np.arange(tot) - off[rid]                     # index of the cell inside its read
    snp0 = np.where(k < c1[rid], lo1[rid] + k, lo2[rid] + (k - c1[rid]))   # 0-based SNP index
    keep = rng.random(tot) >= drop
    flipm = rng.random(tot) < flip
    qual = rng.integers(qlo, qhi + 1, size=tot).astype(np.uint8)
    allele = hap[strain[rid], snp0] ^ flipm.astype(np.uint8)
    rid, snp0, allele, qual = rid[keep], snp0[keep], allele[keep], qual[keep]
    # per-read first/last, discard empty reads
    cnt2 = np.bincount(rid, minlength=n_reads)
    alive = np.nonzero(cnt2 > 0)[0]
    off2 = np.zeros(n_reads + 1, np.int64)
    off2[1:] = np.cumsum(cnt2)
    first = snp0[off2[alive]] + 1
    last = snp0[off2[alive + 1] - 1] + 1
    order = np.lexsort((alive, -last, first))          # Frag::cmp; tie-break = pre-sort index
    src = alive[order]
    lens = cnt2[src]
    new_off = np.zeros(len(src) + 1, np.int64)
    new_off[1:] = np.cumsum(lens)
    gather = np.repeat(off2[src] - new_off[:-1], lens) + np.arange(int(new_off[-1]))
    pile = Pileup(new_off.astype(np.uint32), (snp0[gather] + 1).astype(np.uint32), allele[gather].astype(np.uint8),
                  qual[gather].astype(np.uint8), first[order].astype(np.uint32), last[order].astype(np.uint32))
    layout = None
    if keep_layout:           # genome intervals [start, end) of the kept reads, in pileup order (second mate for short pairs)
        if kind == "long":
            layout = dict(kind=kind, contig_len=contig_len, start=start[src], end=(start + length)[src])
        else:
            layout = dict(kind=kind, contig_len=contig_len, start=start[src], end=(start + 150)[src], start2=(start + insert - 150)[src], end2=(start + insert)[src])
    return Contig(name, pile, pos.astype(np.uint64), ploidy, strain[src] if keep_truth else None, layout)


# (n_contigs, snps/contig, reads/contig, kind, ploidy spec, max_ploidy, beam, block_length)
CONFIGS = {
    # config 1 substitute (the quick-start BAM is missing): 3 strains, ~30x long reads over
    # 954 SNPs spaced like tests/test.vcf (median gap ~51 bp -> gap mean 124 bp reproduces ~17 blocks)
    1: dict(n_contigs=1, snps=954, reads=360, kind="long", ploidy=3, max_ploidy=5, beam=10, block_length=10000),
    2: dict(n_contigs=1, snps=10000, reads=20000, kind="long", ploidy=3, max_ploidy=5, beam=10, block_length=10000),
    3: dict(n_contigs=200, snps=2500, reads=10000, kind="short", ploidy=(2, 4), max_ploidy=5, beam=10, block_length=500),
    4: dict(n_contigs=2000, snps=500, reads=2500, kind="long", ploidy=(2, 4), max_ploidy=5, beam=10, block_length=10000),
    5: dict(n_contigs=1, snps=50000, reads=100000, kind="long", ploidy=8, max_ploidy=8, beam=40, block_length=10000),
}


def contig_ploidy(config, idx):
    spec = CONFIGS[config]["ploidy"]
    if isinstance(spec, tuple):
        lo, hi = spec
        return lo + int(np.random.Generator(np.random.PCG64(np.random.SeedSequence([BASE_SEED + config, idx, 7]))).integers(0, hi - lo + 1))
    return spec


def make_config_contig(config, idx, scale=1.0, keep_truth=False, keep_layout=False):
    """Contig `idx` of BASELINE config `config`.  scale<1 shrinks SNPs and reads per contig together
    (parity-test sizes); scale=1 is the BASELINE size."""
    c = CONFIGS[config]
    snps = max(8, int(round(c["snps"] * scale)))
    reads = max(4, int(round(c["reads"] * scale)))
    ss = np.random.SeedSequence([BASE_SEED + config, idx])
    return make_contig(ss, snps, reads, contig_ploidy(config, idx), c["kind"], name=f"cfg{config}_ctg{idx}", keep_truth=keep_truth, keep_layout=keep_layout)


def make_workload(config, contig_ids=None, scale=1.0, epsilon=0.03125, n_contigs=None):
    c = CONFIGS[config]
    if contig_ids is None:
        contig_ids = range(c["n_contigs"] if n_contigs is None else n_contigs)
    contigs = [make_config_contig(config, i, scale) for i in contig_ids]
    return Workload(f"config{config}", contigs, epsilon, c["max_ploidy"], c["beam"], c["block_length"])
