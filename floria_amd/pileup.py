"""Flat reads x SNPs pileup: the CSR form of floria's `Vec<Frag>` (types_structs.rs:68-85).

One `Pileup` is one contig's reads after `all_frags.sort(); counter_id = index`
(floria.rs:289-293): sorted by `Frag::cmp` (types_structs.rs:87-93 — first_position ascending,
last_position DEscending, counter_id ascending) with read id == row index.
"""
from dataclasses import dataclass

import numpy as np

from . import _capi as capi


@dataclass
class Pileup:
    read_off: np.ndarray  # uint32 [n_reads+1]
    snp: np.ndarray       # uint32 [n_cells], 1-based SNP index, ascending within a read
    allele: np.ndarray    # uint8  [n_cells], 0..3
    qual: np.ndarray      # uint8  [n_cells]
    first: np.ndarray     # uint32 [n_reads]
    last: np.ndarray      # uint32 [n_reads]
    set_order: np.ndarray = None   # optional uint32 [n_cells]: per read, the indices of its cells in the iteration order of Frag.positions (include/floria_hip.h)

    @property
    def n_reads(self):
        return int(self.first.shape[0])

    @property
    def n_cells(self):
        return int(self.snp.shape[0])

    def as_c(self):
        """CPileup view; keep `self` alive while it is in use."""
        for name, dt in (("read_off", np.uint32), ("snp", np.uint32), ("allele", np.uint8),
                         ("qual", np.uint8), ("first", np.uint32), ("last", np.uint32)):
            a = np.ascontiguousarray(getattr(self, name), dtype=dt)
            setattr(self, name, a)
        so = None
        if self.set_order is not None:
            self.set_order = np.ascontiguousarray(self.set_order, dtype=np.uint32)
            so = capi.ptr(self.set_order, capi.C.c_uint32)
        return capi.CPileup(capi.ptr(self.read_off, capi.C.c_uint32), capi.ptr(self.snp, capi.C.c_uint32),
                            capi.ptr(self.allele, capi.C.c_uint8), capi.ptr(self.qual, capi.C.c_uint8),
                            capi.ptr(self.first, capi.C.c_uint32), capi.ptr(self.last, capi.C.c_uint32),
                            self.n_reads, so)

    def read(self, r):
        lo, hi = int(self.read_off[r]), int(self.read_off[r + 1])
        return self.snp[lo:hi], self.allele[lo:hi], self.qual[lo:hi]

    @staticmethod
    def from_reads(reads):
        """reads: iterable of (snps, alleles, quals) with snps strictly ascending 1-based SNP indices.
        Sorts by Frag::cmp with the input order as the pre-sort tie-break (the reference ties on the
        pre-sort counter_id)."""
        reads = [(np.asarray(s, np.uint32), np.asarray(a, np.uint8), np.asarray(q, np.uint8)) for s, a, q in reads]
        reads = [r for r in reads if len(r[0]) > 0]
        first = np.array([r[0][0] for r in reads], np.int64)
        last = np.array([r[0][-1] for r in reads], np.int64)
        order = np.lexsort((np.arange(len(reads)), -last, first))
        reads = [reads[i] for i in order]
        lens = np.array([len(r[0]) for r in reads], np.int64)
        off = np.zeros(len(reads) + 1, np.uint32)
        off[1:] = np.cumsum(lens)
        cat = (lambda i, dt: np.concatenate([r[i] for r in reads]).astype(dt) if reads else np.zeros(0, dt))
        return Pileup(off, cat(0, np.uint32), cat(1, np.uint8), cat(2, np.uint8),
                      first[order].astype(np.uint32), last[order].astype(np.uint32))

    def algorithmic_bytes(self, read_ids, max_ploidy):
        """SURVEY.md §8(d) bytes(block) for a block holding `read_ids`."""
        L = (self.read_off[1:][read_ids].astype(np.int64) - self.read_off[:-1][read_ids].astype(np.int64))
        n = len(read_ids)
        return int(16 + np.sum(8 + (L + 3) // 4 + (L + 7) // 8 + L) + n + 8 * max_ploidy + 4)


def reads_in_interval(p: Pileup, start, end):
    """local_clustering.rs:12-59 (find_reads_in_interval), vectorised; ascending read id."""
    m = (p.last >= start) & (p.first <= end) & ((p.last.astype(np.int64) - p.first.astype(np.int64)) <= 10000)
    return np.nonzero(m)[0].astype(np.uint32)
