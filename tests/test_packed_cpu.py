"""The compact wire form of a pileup (include/floria_hip.h: floria_pileup_packed) — host side: floria_hip_pack_pileup against an
independent numpy unpacking, the byte budget SURVEY.md §8(d) counts, and the inputs the packer must refuse.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from floria_amd import _capi as capi
from floria_amd.pileup import Pileup
from tests.helpers import random_pileup


def unpack(q, n_cells):
    """floria_pileup_packed -> (read_off, snp, allele, qual, first, last) with numpy only."""
    R = q.n_reads
    ro = capi.np_from(q.read_off, R + 1, np.uint32); bo = capi.np_from(q.bit_off, R + 1, np.uint32)
    first = capi.np_from(q.first, R, np.uint32); last = capi.np_from(q.last, R, np.uint32)
    nbits = int(bo[R])
    bits = np.unpackbits(capi.np_from(q.present, (nbits + 7) // 8, np.uint8), bitorder="little")[:nbits]
    pos = np.nonzero(bits)[0]
    rid = np.searchsorted(bo[1:], pos, side="right")
    snp = (first[rid].astype(np.int64) + (pos - bo[rid].astype(np.int64))).astype(np.uint32)
    c = np.arange(n_cells)
    al = ((capi.np_from(q.allele2, (n_cells + 3) // 4, np.uint8)[c >> 2] >> (2 * (c & 3))) & 3).astype(np.uint8) if n_cells else np.zeros(0, np.uint8)
    return ro, snp, al, capi.np_from(q.qual, n_cells, np.uint8), first, last, bo


@pytest.mark.parametrize("seed,alleles,max_len,drop", [(1, 2, 12, 0.1), (2, 4, 40, 0.3), (3, 2, 1, 0.0), (4, 3, 300, 0.02)])
def test_pack_round_trip(hip_lib, seed, alleles, max_len, drop):
    rng = np.random.default_rng(seed)
    p = random_pileup(rng, 300, 400, 3, max_len=max_len, alleles=alleles, drop=drop, qlo=0, qhi=93)
    arena, arr, nbytes = hip_lib.pack_pileups([p], pinned=False)
    ro, snp, al, qu, first, last, bo = unpack(arr[0], p.n_cells)
    assert np.array_equal(ro, p.read_off) and np.array_equal(first, p.first) and np.array_equal(last, p.last)
    assert np.array_equal(snp, p.snp) and np.array_equal(al, p.allele) and np.array_equal(qu, p.qual)
    assert np.array_equal(np.diff(bo.astype(np.int64)), p.last.astype(np.int64) - p.first.astype(np.int64) + 1)
    # the payload is what SURVEY.md §8(d) counts per read (2-bit alleles, presence bits over the span, quality bytes) + 16 B of offsets
    spans = p.last.astype(np.int64) - p.first.astype(np.int64) + 1
    payload = p.n_cells + (p.n_cells + 3) // 4 + (int(spans.sum()) + 7) // 8 + 16 * p.n_reads
    assert payload <= nbytes <= payload + 7 * 64 + 16


def test_pack_several_contigs_and_an_empty_one(hip_lib):
    rng = np.random.default_rng(9)
    ps = [random_pileup(rng, 50, 80, 2), Pileup.from_reads([]), random_pileup(rng, 7, 30, 2, max_len=30)]
    arena, arr, _ = hip_lib.pack_pileups(ps, pinned=False)
    for q, p in zip(arr, ps):
        assert q.n_reads == p.n_reads
        if p.n_reads:
            ro, snp, al, qu, *_ = unpack(q, p.n_cells)
            assert np.array_equal(snp, p.snp) and np.array_equal(al, p.allele) and np.array_equal(qu, p.qual) and np.array_equal(ro, p.read_off)


def test_packer_refuses_what_it_cannot_represent(hip_lib):
    L = hip_lib.load()
    rng = np.random.default_rng(5)
    p = random_pileup(rng, 20, 40, 2)
    bad = Pileup(p.read_off.copy(), p.snp.copy(), p.allele.copy(), p.qual.copy(), p.first.copy(), p.last.copy())
    bad.allele[3] = 4                                             # 2-bit alleles: index > 3 is outside the envelope
    with pytest.raises(hip_lib.FloriaHipError) as e:
        hip_lib.pack_pileups([bad], pinned=False)
    assert e.value.code == capi.FLORIA_E_UNSUPPORTED
    bad = Pileup(p.read_off.copy(), p.snp.copy(), p.allele.copy(), p.qual.copy(), p.first.copy(), p.last.copy())
    bad.snp[int(bad.read_off[5])] = bad.last[5] + 1              # a cell beyond the read's last_position has no presence bit
    with pytest.raises(hip_lib.FloriaHipError) as e:
        hip_lib.pack_pileups([bad], pinned=False)
    assert e.value.code == capi.FLORIA_E_INVALID
    bad = Pileup(p.read_off.copy(), p.snp.copy(), p.allele.copy(), p.qual.copy(), p.first.copy(), p.last.copy())
    bad.last[2] = bad.first[2] - 1 if bad.first[2] > 1 else 0    # last < first
    c = bad.as_c()
    assert L.floria_hip_pack_bytes(C.byref(c)) == 0
    # read_off sizes the buffers through read_off[R] but the packing loop walks [read_off[r], read_off[r+1]) — an offset array that is not
    # strictly ascending from 0 must be refused before anything is written (it used to read and write out of bounds)
    for mangle in ("nonmonotone", "overshoot", "nonzero_start", "empty_read"):
        bad = Pileup(p.read_off.copy(), p.snp.copy(), p.allele.copy(), p.qual.copy(), p.first.copy(), p.last.copy())
        if mangle == "nonmonotone":
            bad.read_off[1], bad.read_off[2] = bad.read_off[2] + 5, bad.read_off[1]
        elif mangle == "overshoot":
            bad.read_off[3] = bad.read_off[-1] + 1000
        elif mangle == "nonzero_start":
            bad.read_off[0] = 1
        else:
            bad.read_off[4] = bad.read_off[5]
        c = bad.as_c()
        assert L.floria_hip_pack_bytes(C.byref(c)) == 0, mangle
        buf = (C.c_char * (1 << 20))()
        out = capi.CPileupPacked()
        assert L.floria_hip_pack_pileup(C.byref(c), buf, len(buf), C.byref(out)) == capi.FLORIA_E_INVALID, mangle
