"""`-m gpu`: the upload path — raw CSR pileups by DMA, validation + flatten on the device (upload_kernel.h) — and the launch
plan of S1 (job groups, ploidy stages): none of it may change a result."""
import numpy as np
import pytest

from floria_amd import synth
from floria_amd.pileup import Pileup
from tests.helpers import assert_block_results_equal, random_pileup

pytestmark = pytest.mark.gpu
EPS = 0.03125
M64 = (1 << 64) - 1


def w24_table():
    """phred_scale (utils_frags.rs:702-711): (1f32 - 10f32.powf(q as f32 / -10.)) as f64, times 2^24 (exact).  powf is the
    platform libm's, as for the Rust binary (numpy's vectorised float32 power differs from it by an ulp for some q)."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    t = np.zeros(256, np.float64)
    for q in range(256):
        x = np.float32(q) / np.float32(-10.0)
        w = np.float32(1.0) - np.float32(libm.powf(10.0, float(x)))
        t[q] = float(w) * 16777216.0
    assert np.all(t == np.floor(t))
    return t.astype(np.uint32)


def hash_tables(length=4 * 65536):
    """The library's Rq1 / Rq2 multiplier tables (splitmix64 from a fixed seed, floria_hip.hip: ensure_hash)."""
    s = 0x1577f10a1a
    out = np.zeros((length, 4), np.uint64)
    for i in range(length):
        for k in range(4):
            s = (s + 0x9e3779b97f4a7c15) & M64
            z = s
            z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M64
            z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M64
            out[i, k] = z ^ (z >> 31)
    return out[:, 0] | np.uint64(1), out[:, 2] | np.uint64(1)


def host_flatten(p: Pileup, w24, rq1=None, rq2=None):
    aw = (p.allele.astype(np.uint32) << 28) | w24[p.qual]
    n = p.n_reads
    meta = np.zeros((n, 8), np.uint32)
    meta[:, 0] = p.read_off[:-1]; meta[:, 1] = np.diff(p.read_off.astype(np.int64)); meta[:, 2] = p.first; meta[:, 3] = p.last
    tw = None
    if rq1 is not None:
        idx = ((p.snp.astype(np.int64) & 65535) << 2) | p.allele.astype(np.int64)
        w = w24[p.qual].astype(np.uint64)
        with np.errstate(over="ignore"):
            t1 = rq1[idx] * w; t2 = rq2[idx] * w
            tw = np.zeros((n, 2), np.uint64)
            tw[:, 0] = np.add.reduceat(t1, p.read_off[:-1].astype(np.int64)); tw[:, 1] = np.add.reduceat(t2, p.read_off[:-1].astype(np.int64))
        meta[:, 4] = tw[:, 0] & np.uint64(0xffffffff); meta[:, 5] = tw[:, 0] >> np.uint64(32)
        meta[:, 6] = tw[:, 1] & np.uint64(0xffffffff); meta[:, 7] = tw[:, 1] >> np.uint64(32)
    return aw, tw, meta


def check_resident(rc, p, w24, rq=None):
    aw, tw, meta = host_flatten(p, w24, *(rq or (None, None)))
    assert np.array_equal(rc.download("read_off", p.n_reads + 1), p.read_off)
    assert np.array_equal(rc.download("first", p.n_reads), p.first) and np.array_equal(rc.download("last", p.n_reads), p.last)
    assert np.array_equal(rc.download("snp", p.n_cells), p.snp)
    assert np.array_equal(rc.download("cell_aw", p.n_cells), aw)
    m = rc.download("meta", 8 * p.n_reads).reshape(-1, 8)
    assert np.array_equal(m[:, :4], meta[:, :4])
    if tw is not None:
        assert np.array_equal(rc.download("tw", 2 * p.n_reads).reshape(-1, 2), tw)
        assert np.array_equal(m, meta)
    else:                                        # the packed record carries the same hash constants as the tw array
        t = rc.download("tw", 2 * p.n_reads).reshape(-1, 2)
        assert np.array_equal(m[:, 4].astype(np.uint64) | (m[:, 5].astype(np.uint64) << np.uint64(32)), t[:, 0])
        assert np.array_equal(m[:, 6].astype(np.uint64) | (m[:, 7].astype(np.uint64) << np.uint64(32)), t[:, 1])


def test_device_flatten_equals_the_host_formulas(gpu_ctx, hip_lib):
    # what floria_hip_contig_upload computed on the host in round 1 (allele | Q24 weight word, per-read hash constants, packed
    # metadata) now comes out of flatten_kernel: compare every resident array, bit for bit, with a numpy restatement
    rng = np.random.default_rng(99)
    piles = [random_pileup(rng, 400, 300, 3, max_len=70, alleles=4, q0_frac=0.05, qlo=0, qhi=93),
             random_pileup(rng, 1, 5, 1, max_len=1),
             synth.make_config_contig(4, 3, 0.5).pileup,
             synth.make_config_contig(3, 0, 0.1).pileup]
    w24 = w24_table()
    rq = hash_tables()
    res = gpu_ctx.upload_batch(piles)
    for rc, p in zip(res, piles):
        check_resident(rc, p, w24, rq)
    one = gpu_ctx.upload(piles[0])                # the single-contig entry point is the same path
    check_resident(one, piles[0], w24, rq)
    one.free()
    for r in res:
        r.free()


def test_pinned_staged_and_mixed_uploads_agree(gpu_ctx, hip_lib, oracle_mod):
    # 128 config-4 contigs (~170 MB of pileup): pageable arrays go through the pinned staging ring (more 16-MB segments than the
    # ring's 8 buffers, several filler threads), floria_hip_host_alloc arrays go by DMA as they are; the resident bytes and
    # the phasing results must not depend on the route
    contigs = [synth.make_config_contig(4, 100 + i) for i in range(128)]
    piles = [c.pileup for c in contigs]
    w24 = w24_table()
    nbytes = sum(4 * (p.n_reads + 1) + 8 * p.n_reads + 6 * p.n_cells for p in piles)
    a = gpu_ctx.upload_batch(piles)
    t = gpu_ctx.timing()
    assert t["upload_staged_bytes"] == nbytes > 9 * (16 << 20) and t["upload_pinned_bytes"] == 0
    arena, pinned = hip_lib.pin_pileups(piles)
    b = gpu_ctx.upload_batch(pinned)
    t = gpu_ctx.timing()
    assert t["upload_staged_bytes"] == 0 and t["upload_pinned_bytes"] == nbytes
    mixed = [pinned[i] if i % 2 else piles[i] for i in range(len(piles))]
    gpu_ctx.set_option("stage_threads", 3)
    c = gpu_ctx.upload_batch(mixed)
    gpu_ctx.set_option("stage_threads", 8)
    for i in (0, 1, 63, 127):
        for rc in (a[i], b[i], c[i]):
            check_resident(rc, piles[i], w24)
    par = hip_lib.make_params(EPS)
    bc, bs, be = [], [], []
    for i, cg in enumerate(contigs[:6]):
        s, e = hip_lib.get_range_with_lengths(cg.snp_pos, 10000)
        bc += [i] * len(s); bs += list(s); be += list(e)
    ra = gpu_ctx.phase_blocks_batch(a[:6], bc, bs, be, par)
    rb = gpu_ctx.phase_blocks_batch(b[:6], bc, bs, be, par)
    rc_ = gpu_ctx.phase_blocks_batch(c[:6], bc, bs, be, par)
    assert_block_results_equal(ra, rb, "pageable vs pinned")
    assert_block_results_equal(ra, rc_, "pageable vs mixed")
    ro = oracle_mod.phase_blocks(piles[0], bs[:bc.count(0)], be[:bc.count(0)], oracle_mod.make_params(EPS), threads=8)
    assert np.array_equal(ro.part, ra.part[:int(ra.read_off[bc.count(0)])]) and np.array_equal(ro.best_ploidy, ra.best_ploidy[:bc.count(0)])
    for r in a + b + c:
        r.free()
    arena.free()


def test_device_validation_reports_what_the_contract_says(gpu_ctx, hip_lib):
    u32, u8 = np.uint32, np.uint8

    def pile(off, snp, al, q, first, last):
        return Pileup(np.array(off, u32), np.array(snp, u32), np.array(al, u8), np.array(q, u8), np.array(first, u32), np.array(last, u32))

    good = Pileup.from_reads([([1, 2, 3], [0, 1, 0], [20, 20, 20]), ([2, 3], [1, 1], [30, 30])])
    cases = [
        (pile([0, 2, 2], [1, 2], [0, 0], [9, 9], [1, 2], [2, 2]), -1, "no cells"),
        (pile([0, 2, 3], [1, 2, 2], [0, 0, 0], [9, 9, 9], [1, 2], [3, 2]), -1, "first/last"),
        (pile([0, 2], [0, 2], [0, 0], [9, 9], [0], [2]), -1, "1-based"),
        (pile([0, 3], [1, 3, 2], [0, 0, 0], [9, 9, 9], [1], [2]), -1, "ascending"),
        (pile([0, 2], [1, 2], [0, 4], [9, 9], [1], [2]), -4, "allele index > 3"),
        (pile([0, 1, 2], [5, 3], [0, 0], [20, 20], [5, 3], [5, 3]), -1, "Frag::cmp"),
        (pile([0, 2, 4], [1, 4, 1, 5], [0, 0, 0, 0], [9, 9, 9, 9], [1, 1], [4, 5]), -1, "Frag::cmp"),       # equal first: last must not ascend
        (pile([0, 3, 2], [1, 2, 3], [0, 0, 0], [9, 9, 9], [1, 3], [3, 3]), -1, "no cells"),                  # read_off not monotone: a read runs past n_cells = read_off[n_reads]
    ]
    for bad, code, what in cases:
        with pytest.raises(hip_lib.FloriaHipError) as ei:
            gpu_ctx.upload(bad)
        assert ei.value.code == code and what in str(ei.value), (what, str(ei.value))
        with pytest.raises(hip_lib.FloriaHipError) as ei:                 # inside a batch the contig is named; no handle leaks out
            gpu_ctx.upload_batch([good, bad, good])
        assert ei.value.code == code and "contig 1" in str(ei.value)
    ok = gpu_ctx.upload_batch([good, good])                              # the context is still usable
    r = gpu_ctx.phase_blocks_batch(ok, [0, 1], [1, 1], [3, 3], hip_lib.make_params(EPS))
    assert r.n_blocks == 2 and np.array_equal(r.read_id, [0, 1, 0, 1])
    for x in ok:
        x.free()
    empty = gpu_ctx.upload_batch([Pileup(np.zeros(1, u32), np.zeros(0, u32), np.zeros(0, u8), np.zeros(0, u8), np.zeros(0, u32), np.zeros(0, u32)), good])
    r = gpu_ctx.phase_blocks_batch(empty, [0, 1], [1, 1], [3, 3], hip_lib.make_params(EPS))
    assert r.best_ploidy[0] == 0 and r.best_ploidy[1] >= 1
    for x in empty:
        x.free()


def test_job_groups_on_warm_pools_with_few_slots(gpu_ctx, hip_lib):
    # two job groups run a ploidy apart on separate streams; every launch lane must own its scratch slice for the
    # whole call.  Few slots (jobs >> resident workgroups, persistent grids) and warm pools (no allocation, hence no implicit
    # device synchronisation, between the launches) is the regime where an aliasing slice would corrupt results.
    n_contigs = 300
    contigs = [synth.make_config_contig(4, 500 + i) for i in range(n_contigs)]
    res = gpu_ctx.upload_batch([c.pileup for c in contigs])
    par = hip_lib.make_params(EPS)
    bc, bs, be = [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
        bc += [i] * len(s); bs += list(s); be += list(e)
    assert len(bc) >= 2048
    gpu_ctx.set_option("speculate", 0)
    runs = {}
    try:
        for slots in (0, 96):
            gpu_ctx.set_option("slots", slots)
            for groups in (1, 2, 2, 3, 1):                               # the repeated entries run on warm pools
                gpu_ctx.set_option("groups", groups)
                r = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
                assert gpu_ctx.timing()["streams"] == min(groups, len(bc) // 1024)
                runs.setdefault("ref", r)
                assert_block_results_equal(runs["ref"], r, f"slots {slots} groups {groups}")
                assert r.min_prune_margin == runs["ref"].min_prune_margin
    finally:
        gpu_ctx.set_option("slots", 0); gpu_ctx.set_option("groups", 0); gpu_ctx.set_option("speculate", -1)
    for r in res:
        r.free()


@pytest.mark.parametrize("cfg,n_contigs,scale", [(4, 24, 1.0), (3, 3, 0.3), (2, 1, 0.1)])
def test_ploidy_stages_do_not_change_results(gpu_ctx, hip_lib, oracle_mod, cfg, n_contigs, scale):
    # running several ploidies of a block at once (speculative stages) must give exactly the sequential loop's result: the same
    # stop decisions, partitions, MEC vectors (zeros beyond `tried`) and the same pruning-margin certificate
    C = synth.CONFIGS[cfg]
    contigs = [synth.make_config_contig(cfg, 40 + i, scale) for i in range(n_contigs)]
    res = gpu_ctx.upload_batch([c.pileup for c in contigs])
    par = hip_lib.make_params(EPS, C["max_ploidy"], C["beam"])
    bc, bs, be = [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, C["block_length"])
        bc += [i] * len(s); bs += list(s); be += list(e)
    out = {}
    try:
        for spec, width in ((0, 1), (1, C["max_ploidy"]), (2, 3), (3, max(1, C["max_ploidy"] - 3) if C["max_ploidy"] >= 5 else 1)):
            gpu_ctx.set_option("speculate", spec)
            out[spec] = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
            assert gpu_ctx.timing()["stage_width"] == width
    finally:
        gpu_ctx.set_option("speculate", -1)
    for spec in (1, 2, 3):
        assert_block_results_equal(out[0], out[spec], f"speculate {spec}")
        assert out[0].min_prune_margin == out[spec].min_prune_margin
    # the same with the chip over-subscribed (few wave slots: most jobs of the higher ploidies are dequeued after the stop rule of their
    # block is known and are dropped, others are dropped in mid-run), with the pruning switched off, and with full-size gated grids
    try:
        gpu_ctx.set_option("speculate", 1)
        for slots, div in ((48, 2), (16, 1), (0, 4)):
            gpu_ctx.set_option("slots", slots); gpu_ctx.set_option("spec_gate_div", div)
            for rep in range(2):
                r = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
                assert_block_results_equal(out[0], r, f"speculate 1, slots {slots}, gate_div {div}, rep {rep}")
                assert out[0].min_prune_margin == r.min_prune_margin
    finally:
        gpu_ctx.set_option("speculate", -1); gpu_ctx.set_option("slots", 0); gpu_ctx.set_option("spec_gate_div", 2)
    # a host that initialised HIP with the default 4 hardware queues: the plan must stay within two job groups and one ploidy per stage
    # (lanes that share a queue and wait on each other's events serialise), and give the same results
    try:
        gpu_ctx.set_option("hw_queues", 4)
        r = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
        t = gpu_ctx.timing()
        assert t["stage_width"] == 1 and t["streams"] <= 2
        assert_block_results_equal(out[0], r, "plan for 4 hardware queues")
    finally:
        gpu_ctx.set_option("hw_queues", 12)
    n0 = bc.count(0)
    ro = oracle_mod.phase_blocks(contigs[0].pileup, bs[:n0], be[:n0], oracle_mod.make_params(EPS, C["max_ploidy"], C["beam"]), threads=8)
    assert np.array_equal(ro.mec.view(np.uint64), out[1].mec[:n0].view(np.uint64)) and np.array_equal(ro.ploidies_tried, out[1].ploidies_tried[:n0])
    for r in res:
        r.free()


def test_pipelined_upload_and_phase_equals_upload_then_phase(gpu_ctx, hip_lib, oracle_mod):
    # floria_hip_phase_pileups_batch streams the cell arrays in chunks and starts a chunk's blocks when it has landed; job groups,
    # chunk count, pinned or pageable sources and the optimistic biallelic plan must not change a single result
    contigs = [synth.make_config_contig(4, 700 + i) for i in range(96)]
    piles = [c.pileup for c in contigs]
    par = hip_lib.make_params(EPS)
    bc, bs, be = [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, 10000)
        bc += [i] * len(s); bs += list(s); be += list(e)
    res = gpu_ctx.upload_batch(piles)
    ref = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
    for r in res:
        r.free()
    arena, pinned = hip_lib.pin_pileups(piles)
    try:
        for chunks in (0, 1, 2, 3, 4):
            gpu_ctx.set_option("upload_chunks", chunks)
            got = gpu_ctx.phase_pileups_batch(pinned, bc, bs, be, par)
            t = gpu_ctx.timing()
            assert t["upload_chunks"] == (chunks if chunks else t["upload_chunks"]) and t["upload_staged_bytes"] == 0
            if chunks > 1:
                assert t["streams"] == chunks
            assert_block_results_equal(ref, got, f"pipelined, {chunks} chunks")
            assert got.min_prune_margin == ref.min_prune_margin
        gpu_ctx.set_option("upload_chunks", 3)
        got = gpu_ctx.phase_pileups_batch(piles, bc, bs, be, par)                 # pageable: uploaded first, then phased
        assert gpu_ctx.timing()["upload_chunks"] == 1 and gpu_ctx.timing()["upload_staged_bytes"] > 0
        assert_block_results_equal(ref, got, "pageable sources")
        # resident handles come back on request and serve the later stages (hap graph on the still-resident batch)
        got, kept = gpu_ctx.phase_pileups_batch(pinned, bc, bs, be, par, keep=True)
        assert_block_results_equal(ref, got, "keep")
        hg = gpu_ctx.hap_graph(got)
        again = gpu_ctx.phase_blocks_batch(kept, bc, bs, be, par)
        assert_block_results_equal(ref, again, "kept handles")
        hg2 = gpu_ctx.hap_graph(again)
        assert np.array_equal(hg.edge_w, hg2.edge_w) and np.array_equal(hg.node_cov.view(np.uint64), hg2.node_cov.view(np.uint64))
        kept.free()
        # a batch that breaks the optimistic plan (a 4-allele contig and one with q = 0 cells among biallelic ones) is phased again
        rng = np.random.default_rng(5)
        odd = [piles[0], random_pileup(rng, 300, 80, 3, max_len=40, alleles=4), piles[1], random_pileup(rng, 200, 50, 2, max_len=25, q0_frac=0.1)]
        obc, obs, obe = [], [], []
        for i, p in enumerate(odd):
            if i in (0, 2):
                s, e = hip_lib.get_range_with_lengths(contigs[i // 2].snp_pos, 10000)
            else:
                S = int(p.last.max()); s, e = [1, S // 2], [S // 2 + 3, S]
            obc += [i] * len(s); obs += list(s); obe += list(e)
        oa, op = hip_lib.pin_pileups(odd)
        gpu_ctx.set_option("upload_chunks", 2)
        got = gpu_ctx.phase_pileups_batch(op, obc, obs, obe, par)
        for i, p in enumerate(odd):
            sel = [k for k in range(len(obc)) if obc[k] == i]
            ro = oracle_mod.phase_blocks(p, [obs[k] for k in sel], [obe[k] for k in sel], oracle_mod.make_params(EPS), threads=8)
            lo, hi = int(got.read_off[sel[0]]), int(got.read_off[sel[-1] + 1])
            assert np.array_equal(ro.best_ploidy, got.best_ploidy[sel[0]:sel[-1] + 1]) and np.array_equal(ro.part, got.part[lo:hi])
            assert np.array_equal(ro.mec.view(np.uint64), got.mec[sel[0]:sel[-1] + 1].view(np.uint64))
        # an invalid contig in a late chunk: the call fails as a whole and the context stays usable
        bad = Pileup(odd[1].read_off.copy(), odd[1].snp.copy(), odd[1].allele.copy(), odd[1].qual.copy(), odd[1].first.copy(), odd[1].last.copy())
        bad.snp[5:7] = bad.snp[5:7][::-1]
        ba, bp = hip_lib.pin_pileups([piles[0], piles[1], piles[2], bad])
        with pytest.raises(hip_lib.FloriaHipError) as ei:
            gpu_ctx.phase_pileups_batch(bp, [0, 3], [1, 1], [50, 20], par)
        assert ei.value.code == -1 and "contig 3" in str(ei.value)
        got = gpu_ctx.phase_pileups_batch(pinned, bc, bs, be, par)
        assert_block_results_equal(ref, got, "after a failed call")
        oa.free(); ba.free()
    finally:
        gpu_ctx.set_option("upload_chunks", 0)
    arena.free()


def _batch_contigs(hip_lib, gpu_ctx, batch, piles):
    return [hip_lib.ResidentContig(gpu_ctx, handle=hip_lib.C.c_void_p(batch._arr[j]), n_reads=piles[j].n_reads) for j in range(len(piles))]


def test_packed_upload_equals_csr_upload(gpu_ctx, hip_lib):
    # the compact wire form (floria_pileup_packed: presence bits, 2-bit alleles, quality bytes) is expanded on the device and must leave
    # exactly the resident bytes of a CSR upload: every array of every contig, and S1 straight from packed host pileups == from CSR ones
    rng = np.random.default_rng(404)
    contigs = [synth.make_config_contig(4, 300 + i, 0.6) for i in range(24)]
    piles = [random_pileup(rng, 400, 300, 3, max_len=70, alleles=4, q0_frac=0.05, qlo=0, qhi=93, drop=0.3),
             random_pileup(rng, 1, 5, 1, max_len=1),
             random_pileup(rng, 900, 60, 2, max_len=3, drop=0.0),                 # short reads: 1-3 cells, spans of a few bits
             synth.make_config_contig(3, 0, 0.1).pileup] + [c.pileup for c in contigs]
    w24 = w24_table()
    rq = hash_tables()
    arena, parr, nbytes = hip_lib.pack_pileups(piles)
    csr_bytes = sum(4 * (p.n_reads + 1) + 8 * p.n_reads + 6 * p.n_cells for p in piles)
    assert nbytes < 0.45 * csr_bytes
    pb = gpu_ctx.upload_batch_packed(parr)
    t = gpu_ctx.timing()
    assert t["upload_pinned_bytes"] + t["upload_staged_bytes"] < 0.45 * csr_bytes
    for rc, p in zip(_batch_contigs(hip_lib, gpu_ctx, pb, piles), piles):
        check_resident(rc, p, w24, rq)
        rc._h = None                                   # (views: the batch owns the handles)
    # S1: one pipelined call from packed host pileups, one from the CSR arrays, and the resident batch
    par = hip_lib.make_params(EPS)
    bc, bs, be = [], [], []
    for i, cg in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(cg.snp_pos, 10000)
        bc += [4 + i] * len(s); bs += list(s); be += list(e)
    r_res = gpu_ctx.phase_blocks_batch(pb, bc, bs, be, par)
    r_pk = gpu_ctx.phase_pileups_batch(parr, bc, bs, be, par)
    carena, pinned = hip_lib.pin_pileups(piles)
    gpu_ctx.set_option("upload_chunks", 3)
    r_pk3 = gpu_ctx.phase_pileups_batch(parr, bc, bs, be, par)
    assert gpu_ctx.timing()["upload_chunks"] == 3
    r_csr = gpu_ctx.phase_pileups_batch(hip_lib.c_pileups(pinned), bc, bs, be, par)
    # five chunks, one ploidy per stage (the plan of a full-size batch): a packed call runs them as three job groups, first | middle | last chunk
    # (the biallelic contigs alone: a batch with four alleles or q = 0 cells abandons the optimistic pipelined plan and phases again from the resident arrays)
    arena2, parr2, _ = hip_lib.pack_pileups([c.pileup for c in contigs])
    gpu_ctx.set_option("speculate", 0)
    gpu_ctx.set_option("upload_chunks", 5)
    r_pk5 = gpu_ctx.phase_pileups_batch(parr2, [x - 4 for x in bc], bs, be, par)
    t5 = gpu_ctx.timing()
    assert t5["upload_chunks"] == 5 and t5["streams"] == 3 and t5["stage_width"] == 1
    gpu_ctx.set_option("speculate", -1)
    gpu_ctx.set_option("upload_chunks", 0)
    assert_block_results_equal(r_res, r_pk5, "resident vs packed host pileups in 5 chunks / 3 job groups")
    arena2.free()
    assert_block_results_equal(r_res, r_pk, "resident vs packed host pileups")
    assert_block_results_equal(r_res, r_pk3, "resident vs packed host pileups in 3 chunks")
    assert_block_results_equal(r_res, r_csr, "resident vs CSR host pileups")
    assert r_pk.min_prune_margin == r_csr.min_prune_margin
    pb.free(); arena.free(); carena.free()


def test_packed_upload_is_validated(gpu_ctx, hip_lib):
    rng = np.random.default_rng(77)
    good = random_pileup(rng, 40, 60, 2, max_len=20, drop=0.2)
    par = hip_lib.make_params(EPS)

    def corrupted(edit):
        arena, parr, _ = hip_lib.pack_pileups([good, good, good])
        q = parr[1]
        edit(q)
        return arena, parr

    def clear_first_bit(q):                      # the first presence bit of read 3 cleared: its cells no longer add up
        bo = capi_np(q.bit_off, q.n_reads + 1, np.uint32)
        pr = capi_np(q.present, (int(bo[-1]) + 7) // 8, np.uint8)
        b = int(bo[3]); pr[b >> 3] &= np.uint8(~(1 << (b & 7)) & 0xff)

    def extra_bit(q):                            # a presence bit too many in read 5 (a gap position of the read set)
        bo = capi_np(q.bit_off, q.n_reads + 1, np.uint32)
        pr = capi_np(q.present, (int(bo[-1]) + 7) // 8, np.uint8)
        bits = np.unpackbits(pr, bitorder="little")
        for r in range(q.n_reads):
            z = np.nonzero(bits[int(bo[r]):int(bo[r + 1])] == 0)[0]
            if len(z):
                b = int(bo[r]) + int(z[0]); pr[b >> 3] |= np.uint8(1 << (b & 7)); return
        raise AssertionError("no gap in the test pileup")

    def span_mismatch(q):                        # bit_off disagrees with last - first + 1
        bo = capi_np(q.bit_off, q.n_reads + 1, np.uint32)
        bo[7:] += 1

    capi_np = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(n,))      # (a VIEW of the packed buffer: the edits land in what is uploaded)
    for edit in (clear_first_bit, extra_bit, span_mismatch):
        arena, parr = corrupted(edit)
        with pytest.raises(hip_lib.FloriaHipError) as ei:
            gpu_ctx.upload_batch_packed(parr)
        assert ei.value.code == -1 and "contig 1" in str(ei.value), str(ei.value)
        with pytest.raises(hip_lib.FloriaHipError):
            gpu_ctx.phase_pileups_batch(parr, [0, 1, 2], [1, 1, 1], [60, 60, 60], par)
        arena.free()
    arena, parr, _ = hip_lib.pack_pileups([good])          # the context is still usable
    r = gpu_ctx.phase_pileups_batch(parr, [0], [1], [60], par)
    assert r.n_blocks == 1 and r.best_ploidy[0] >= 1
    arena.free()


def test_last_ploidy_beside_the_optimise_launch_below_it(gpu_ctx, hip_lib):
    # one ploidy per stage with the LAST ploidy's beam launch running beside the optimise launch of the ploidy below, every job waiting for its block's stop rule
    # (tail_overlap, run_phase): a 120-contig config-4 shard with 1-3 job groups, few and many wave slots, small and large waiting grids; every field of every
    # call equals the plain stage-after-stage run
    contigs = [synth.make_config_contig(4, 1700 + i) for i in range(120)]
    C = synth.CONFIGS[4]
    res = gpu_ctx.upload_batch([c.pileup for c in contigs])
    par = hip_lib.make_params(EPS, C["max_ploidy"], C["beam"])
    bc, bs, be = [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, C["block_length"])
        bc += [i] * len(s); bs += list(s); be += list(e)
    try:
        gpu_ctx.set_option("speculate", 0); gpu_ctx.set_option("tail_overlap", 0)
        ref = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
        assert int(ref.ploidies_tried.max()) == C["max_ploidy"], "the shard must reach the last ploidy for this test to mean anything"
        n_calls = 0
        gpu_ctx.set_option("tail_overlap", 1)
        for slots in (0, 37, 512):
            for groups in (1, 2, 3):
                for waves in (1, 2, 8):
                    gpu_ctx.set_option("slots", slots); gpu_ctx.set_option("groups", groups); gpu_ctx.set_option("tail_waves", waves)
                    for rep in range(2):
                        r = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
                        assert_block_results_equal(ref, r, f"tail_overlap, slots {slots}, groups {groups}, waves {waves}, rep {rep}")
                        assert r.min_prune_margin == ref.min_prune_margin
                        n_calls += 1
        assert n_calls == 54
        # a smaller max_ploidy: the last stage is one many blocks reach
        par3 = hip_lib.make_params(EPS, 3, C["beam"])
        gpu_ctx.set_option("slots", 0); gpu_ctx.set_option("groups", 0); gpu_ctx.set_option("tail_waves", 2); gpu_ctx.set_option("tail_overlap", 0)
        ref3 = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par3)
        gpu_ctx.set_option("tail_overlap", 1)
        assert_block_results_equal(ref3, gpu_ctx.phase_blocks_batch(res, bc, bs, be, par3), "tail_overlap at max_ploidy 3")
    finally:
        gpu_ctx.set_option("speculate", -1); gpu_ctx.set_option("slots", 0); gpu_ctx.set_option("groups", 0); gpu_ctx.set_option("tail_waves", 2); gpu_ctx.set_option("tail_overlap", 0)
        for r in res:
            r.free()


def test_speculative_stages_under_stress(gpu_ctx, hip_lib):
    # race hunter for the speculative ploidy stages (the path every multi-GPU shard takes): a 250-contig config-4 shard — the per-GPU share of the
    # 8-GPU job — phased again and again with all ploidies at once / {1,2,3}{4,5}, few and many wave slots (over-subscription: jobs are dropped at dequeue
    # and in mid-run by stop flags that other workgroups publish meanwhile) and 1-3 job groups; EVERY field of EVERY call must equal the sequential stages'
    contigs = [synth.make_config_contig(4, 700 + i) for i in range(250)]
    C = synth.CONFIGS[4]
    res = gpu_ctx.upload_batch([c.pileup for c in contigs])
    par = hip_lib.make_params(EPS, C["max_ploidy"], C["beam"])
    bc, bs, be = [], [], []
    for i, c in enumerate(contigs):
        s, e = hip_lib.get_range_with_lengths(c.snp_pos, C["block_length"])
        bc += [i] * len(s); bs += list(s); be += list(e)
    try:
        gpu_ctx.set_option("speculate", 0)
        ref = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
        n_calls = 0
        for spec in (1, 2):
            for slots in (37, 96, 512):
                for groups in (1, 2, 3):
                    gpu_ctx.set_option("speculate", spec); gpu_ctx.set_option("slots", slots); gpu_ctx.set_option("groups", groups)
                    for rep in range(4 if slots == 37 else 6):
                        r = gpu_ctx.phase_blocks_batch(res, bc, bs, be, par)
                        assert_block_results_equal(ref, r, f"speculate {spec}, slots {slots}, groups {groups}, rep {rep}")
                        assert r.min_prune_margin == ref.min_prune_margin
                        n_calls += 1
        assert n_calls >= 90
    finally:
        gpu_ctx.set_option("speculate", -1); gpu_ctx.set_option("slots", 0); gpu_ctx.set_option("groups", 0)
        for r in res:
            r.free()
