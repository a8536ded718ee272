"""ctypes binding of libfloria_hip.so (include/floria_hip.h) — the Python mirror of the two reference
seams: `generate_hap_graph`'s per-block loop (graph_processing.rs:325-372) and
`process_reads_for_final_parts` (part_block_manip.rs:174-274).

No fallback: if the shared library is missing or no HIP device is usable, everything here raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _capi as capi
from .pileup import Pileup

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_SO = os.path.join(_CSRC, "libfloria_hip.so")
_LIB = None

# every symbol include/floria_hip.h declares
SYMBOLS = [
    "floria_hip_create", "floria_hip_destroy", "floria_hip_last_error", "floria_hip_version", "floria_hip_init_env", "floria_hip_realign", "floria_hip_selftest",
    "floria_hip_block_ranges", "floria_hip_ranges_free", "floria_hip_contig_upload", "floria_hip_contig_free",
    "floria_hip_phase_blocks_resident", "floria_hip_phase_blocks", "floria_hip_block_result_free",
    "floria_hip_phase_blocks_batch", "floria_hip_reassign", "floria_hip_groups_free", "floria_hip_last_timing",
    "floria_hip_set_slots", "floria_hip_reassign_batch", "floria_hip_groups_array_free",
    "floria_hip_hap_graph", "floria_hip_hap_graph_free", "floria_hip_reassign_ordered", "floria_hip_haploset_stats",
    "floria_hip_hapq", "floria_hip_hapq_batch",
    "floria_hip_contig_upload_batch", "floria_hip_host_alloc", "floria_hip_host_free", "floria_hip_set_option",
    "floria_hip_contig_download", "floria_hip_phase_pileups_batch",
    "floria_hip_pack_bytes", "floria_hip_pack_pileup", "floria_hip_pack_bytes_batch", "floria_hip_pack_pileups_batch", "floria_hip_contig_upload_batch_packed", "floria_hip_phase_pileups_batch_packed",
]


class FloriaHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfloria_hip rc={code}: {msg}")
        self.code = code


def build(force=False):
    """Compile libfloria_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).  make decides whether the library is
    stale (every *.h of csrc/ and include/floria_hip.h are prerequisites); where no hipcc exists (the GPU box gets the prebuilt
    library with the snapshot) an existing library is used as it is."""
    import shutil
    have_hipcc = os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")
    if not have_hipcc and os.path.exists(_SO):
        return _SO
    subprocess.check_call(["make", "-C", _CSRC] + (["-B"] if force else []) + ["libfloria_hip.so"], stdout=subprocess.DEVNULL)
    return _SO


def load():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")      # streams of the job groups / upload pipeline on separate hardware queues (read when HIP initialises)
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise FloriaHipError(-2, f"{_SO} is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(_SO)
        L.floria_hip_last_error.restype = C.c_char_p
        L.floria_hip_version.restype = C.c_char_p
        for s in ("floria_hip_destroy", "floria_hip_ranges_free", "floria_hip_contig_free", "floria_hip_block_result_free", "floria_hip_groups_free",
                  "floria_hip_groups_array_free", "floria_hip_hap_graph_free"):
            getattr(L, s).restype = None
        L.floria_hip_destroy.argtypes = [C.c_void_p]
        L.floria_hip_contig_free.argtypes = [C.c_void_p]
        L.floria_hip_host_alloc.restype = C.c_void_p
        L.floria_hip_host_alloc.argtypes = [C.c_size_t]
        L.floria_hip_host_free.restype = None
        L.floria_hip_host_free.argtypes = [C.c_void_p]
        L.floria_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.floria_hip_pack_bytes.restype = C.c_size_t
        L.floria_hip_pack_bytes.argtypes = [C.c_void_p]
        L.floria_hip_pack_bytes_batch.restype = C.c_size_t
        L.floria_hip_pack_bytes_batch.argtypes = [C.c_void_p, C.c_uint32]
        L.floria_hip_pack_pileups_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        L.floria_hip_pack_pileup.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise FloriaHipError(rc, load().floria_hip_last_error().decode())


def make_params(epsilon, max_ploidy=5, beam=10, ploidy_sensitivity=2, stopping_heuristic=1):
    """The hot-path fields of `Options` (types_structs.rs:20-51) with the CLI defaults
    (parse_cmd_line.rs: -p 5, -n 10, -s 2, stopping heuristic on)."""
    return capi.CParams(float(epsilon), int(max_ploidy), int(beam), int(ploidy_sensitivity), int(stopping_heuristic))


def get_range_with_lengths(snp_to_genome_pos, block_length, overlap_len=None, minimal_density=0.0005):
    """utils_frags::get_range_with_lengths (utils_frags.rs:405-463); overlap defaults to block_length/3
    as in generate_hap_graph (graph_processing.rs:334-339).  Returns (start, end) uint32 arrays, 1-based inclusive."""
    g = np.ascontiguousarray(snp_to_genome_pos, np.uint64)
    if overlap_len is None:
        overlap_len = block_length // 3
    out = C.POINTER(capi.CRanges)()
    _check(load().floria_hip_block_ranges(capi.ptr(g, C.c_uint64), C.c_uint32(len(g)), C.c_uint64(block_length), C.c_uint64(overlap_len),
                                          C.c_double(minimal_density), C.byref(out)))
    r = out.contents
    res = (capi.np_from(r.start, r.n, np.uint32), capi.np_from(r.end, r.n, np.uint32))
    load().floria_hip_ranges_free(out)
    return res


class PinnedArena:
    """Pinned host memory (floria_hip_host_alloc) handed out as numpy arrays: pileups whose arrays live here upload by DMA
    with no staging copy, and arrays carved consecutively are back to back, so a batch travels as one transfer per field."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self._p = load().floria_hip_host_alloc(C.c_size_t(self.nbytes))
        if not self._p:
            raise FloriaHipError(-3, load().floria_hip_last_error().decode())
        self._buf = (C.c_uint8 * self.nbytes).from_address(self._p)
        self._cursor = 0

    def take(self, count, dtype):
        """Next `count` elements of `dtype` (element-aligned, NOT padded: consecutive takes of one dtype are contiguous)."""
        dt = np.dtype(dtype)
        off = (self._cursor + dt.itemsize - 1) // dt.itemsize * dt.itemsize
        end = off + int(count) * dt.itemsize
        if end > self.nbytes:
            raise MemoryError("PinnedArena exhausted")
        self._cursor = end
        return np.frombuffer(self._buf, dtype=dt, count=int(count), offset=off)

    def free(self):
        if self._p:
            self._buf = None
            load().floria_hip_host_free(C.c_void_p(self._p))
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pin_pileups(pileups):
    """Copy pileups into ONE pinned arena, field by field (all read_off arrays back to back, then all first, ...):
    what a host that marshals `Vec<Frag>` into upload buffers writes directly.  Returns (arena, [Pileup views])."""
    nr = sum(p.n_reads for p in pileups)
    nc = sum(p.n_cells for p in pileups)
    with_order = any(p.set_order is not None for p in pileups)
    arena = PinnedArena(4 * (nr + len(pileups)) + 8 * nr + (10 if with_order else 6) * nc + 256)
    fields = {}
    for name, dt in (("read_off", np.uint32), ("first", np.uint32), ("last", np.uint32), ("snp", np.uint32), ("allele", np.uint8), ("qual", np.uint8)):
        views = []
        for p in pileups:
            src = np.ascontiguousarray(getattr(p, name), dt)
            v = arena.take(len(src), dt)
            v[:] = src
            views.append(v)
        fields[name] = views
    out = [Pileup(fields["read_off"][i], fields["snp"][i], fields["allele"][i], fields["qual"][i], fields["first"][i], fields["last"][i]) for i in range(len(pileups))]
    if with_order:
        for p, o in zip(pileups, out):
            if p.set_order is not None:
                v = arena.take(p.n_cells, np.uint32)
                v[:] = np.ascontiguousarray(p.set_order, np.uint32)
                o.set_order = v
    return arena, out


class _PlainArena:
    """Pageable stand-in for PinnedArena (tests of the host-side packer on machines without a GPU)."""

    def __init__(self, nbytes):
        self._arr = np.zeros(int(nbytes) + 64, np.uint8)
        self._p = self._arr.ctypes.data

    def free(self):
        pass


def pack_pileups(pileups, pinned=True):
    """The compact wire form (floria_pileup_packed) of a list of Pileups, packed by floria_hip_pack_pileup into ONE pinned arena:
    what a host marshals its Vec<Frag> into for the PCIe link.  Returns (arena, ctypes array of floria_pileup_packed, packed bytes)."""
    L = load()
    n = len(pileups)
    carr = c_pileups(pileups)
    size = int(L.floria_hip_pack_bytes_batch(carr, C.c_uint32(n)))
    if n and size == 0:
        raise FloriaHipError(-1, "pileups cannot be packed (null field, last < first, or spans of 2^32 bits and more)")
    arena = (PinnedArena if pinned else _PlainArena)(size + 128)
    arr = (capi.CPileupPacked * n)()
    base = arena._p + (64 - arena._p % 64) % 64
    _check(L.floria_hip_pack_pileups_batch(carr, C.c_uint32(n), C.c_void_p(base), C.c_size_t(size), arr))
    return arena, arr, size


class ResidentContig:
    def __init__(self, ctx, pileup: Pileup = None, handle=None, n_reads=None):
        self.ctx = ctx
        if handle is not None:
            self._h = handle
            self.n_reads = n_reads
            return
        self.n_reads = pileup.n_reads
        cp = pileup.as_c()
        h = C.c_void_p()
        _check(load().floria_hip_contig_upload(ctx._h, C.byref(cp), C.byref(h)))
        self._h = h

    FIELDS = {"read_off": (0, np.uint32), "first": (1, np.uint32), "last": (2, np.uint32), "snp": (3, np.uint32),
              "cell_aw": (4, np.uint32), "tw": (5, np.uint64), "meta": (6, np.uint32)}

    def download(self, field, count):
        """Diagnostic (floria_hip_contig_download): `count` elements of a resident array."""
        fid, dt = self.FIELDS[field]
        out = np.zeros(int(count), dt)
        _check(load().floria_hip_contig_download(self._h, C.c_int(fid), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
        return out

    def free(self):
        if self._h:
            load().floria_hip_contig_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ContigBatch:
    """The handles of one floria_hip_contig_upload_batch call, kept as the ctypes array the C entry points take (no per-contig
    Python objects: a host loop that uploads and phases batch after batch pays only the C calls)."""

    def __init__(self, ctx, handles, n):
        self.ctx, self._arr, self.n = ctx, handles, n

    def __len__(self):
        return self.n

    def free(self):
        if self._arr is not None:
            L = load()
            for i in range(self.n):
                L.floria_hip_contig_free(C.c_void_p(self._arr[i]))
            self._arr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def c_pileups(pileups):
    """ctypes array of floria_pileup for a list of Pileups (keep the Pileups alive while it is in use)."""
    return (capi.CPileup * len(pileups))(*[p.as_c() for p in pileups])


class FloriaHip:
    """One context per device (floria_hip_create)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(load().floria_hip_create(C.c_int(device), C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            load().floria_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_slots(self, n):
        _check(load().floria_hip_set_slots(self._h, C.c_uint32(n)))

    def set_option(self, key, value):
        """floria_hip_set_option.  "arith" = 1 selects the reference's running-sum arithmetic (a different function at an epsilon that is not a
        multiple of 2^-10, slower kernels; include/floria_hip.h); every other key is a tuning / test knob that changes no result."""
        _check(load().floria_hip_set_option(self._h, key.encode(), C.c_int64(int(value))))

    def selftest(self, epsilon, n_max=1024):
        """floria_hip_selftest: max |f32 screen - host table| / n of the binomial p-value on this device."""
        out = C.c_double(0.0)
        fn = load().floria_hip_selftest
        fn.argtypes = [C.c_void_p, C.c_double, C.c_uint32, C.POINTER(C.c_double)]
        _check(fn(self._h, C.c_double(epsilon), C.c_uint32(n_max), C.byref(out)))
        return out.value

    def upload(self, pileup: Pileup):
        return ResidentContig(self, pileup)

    def upload_batch(self, pileups):
        """floria_hip_contig_upload_batch: one allocation, DMA of the raw arrays, validate + flatten on the device."""
        n = len(pileups)
        arr = c_pileups(pileups)
        hs = (C.c_void_p * n)()
        _check(load().floria_hip_contig_upload_batch(self._h, arr, C.c_uint32(n), hs))
        return [ResidentContig(self, handle=C.c_void_p(hs[i]), n_reads=pileups[i].n_reads) for i in range(n)]

    def upload_batch_c(self, carr, n=None):
        """The same for a prebuilt ctypes floria_pileup array -> ContigBatch."""
        n = len(carr) if n is None else n
        hs = (C.c_void_p * n)()
        _check(load().floria_hip_contig_upload_batch(self._h, carr, C.c_uint32(n), hs))
        return ContigBatch(self, hs, n)

    def upload_batch_packed(self, parr, n=None):
        """floria_hip_contig_upload_batch_packed for a ctypes floria_pileup_packed array (pack_pileups) -> ContigBatch."""
        n = len(parr) if n is None else n
        hs = (C.c_void_p * n)()
        _check(load().floria_hip_contig_upload_batch_packed(self._h, parr, C.c_uint32(n), hs))
        return ContigBatch(self, hs, n)

    def timing(self):
        t = capi.CTiming()
        _check(load().floria_hip_last_timing(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in capi.CTiming._fields_}

    # S1 --------------------------------------------------------------------------------------------
    def phase_blocks(self, contig, blk_start, blk_end, params):
        """get_local_hap_blocks for every (start,end) SNP range of one contig (graph_processing.rs:345-362)."""
        if isinstance(contig, Pileup):
            rc = self.upload(contig)
            try:
                return self.phase_blocks(rc, blk_start, blk_end, params)
            finally:
                rc.free()
        bs = np.ascontiguousarray(blk_start, np.uint32)
        be = np.ascontiguousarray(blk_end, np.uint32)
        out = C.POINTER(capi.CBlockResult)()
        _check(load().floria_hip_phase_blocks_resident(self._h, contig._h, capi.ptr(bs, C.c_uint32), capi.ptr(be, C.c_uint32),
                                                       C.c_uint32(len(bs)), C.byref(params), C.byref(out)))
        res = capi.BlockResult(out.contents)
        load().floria_hip_block_result_free(out)
        return res

    def phase_blocks_batch(self, contigs, blk_contig, blk_start, blk_end, params, copy_out=True):
        arr = contigs._arr if isinstance(contigs, ContigBatch) else (C.c_void_p * len(contigs))(*[c._h for c in contigs])
        bc = np.ascontiguousarray(blk_contig, np.uint32)
        bs = np.ascontiguousarray(blk_start, np.uint32)
        be = np.ascontiguousarray(blk_end, np.uint32)
        out = C.POINTER(capi.CBlockResult)()
        _check(load().floria_hip_phase_blocks_batch(self._h, arr, C.c_uint32(len(contigs)), capi.ptr(bc, C.c_uint32), capi.ptr(bs, C.c_uint32),
                                                    capi.ptr(be, C.c_uint32), C.c_uint32(len(bs)), C.byref(params), C.byref(out)))
        res = capi.BlockResult(out.contents) if copy_out else None
        load().floria_hip_block_result_free(out)
        return res

    def phase_pileups_batch(self, pileups, blk_contig, blk_start, blk_end, params, keep=False, copy_out=True):
        """floria_hip_phase_pileups_batch: S1 straight from host pileups (a list of Pileups or a prebuilt c_pileups array),
        transfers pipelined with the kernels.  Returns the BlockResult, or (BlockResult, ContigBatch) with keep=True."""
        carr = pileups if isinstance(pileups, C.Array) else c_pileups(pileups)
        n = len(carr)
        packed = isinstance(carr, C.Array) and carr._type_ is capi.CPileupPacked          # (pack_pileups: the compact wire form)
        bc = np.ascontiguousarray(blk_contig, np.uint32)
        bs = np.ascontiguousarray(blk_start, np.uint32)
        be = np.ascontiguousarray(blk_end, np.uint32)
        out = C.POINTER(capi.CBlockResult)()
        hs = (C.c_void_p * n)() if keep else None
        fn = load().floria_hip_phase_pileups_batch_packed if packed else load().floria_hip_phase_pileups_batch
        _check(fn(self._h, carr, C.c_uint32(n), capi.ptr(bc, C.c_uint32), capi.ptr(bs, C.c_uint32),
                  capi.ptr(be, C.c_uint32), C.c_uint32(len(bs)), C.byref(params), C.byref(out), hs))
        res = capi.BlockResult(out.contents) if copy_out else None
        load().floria_hip_block_result_free(out)
        return (res, ContigBatch(self, hs, n)) if keep else res

    def hap_graph(self, res):
        """HapNode::new coverage + update_hap_graph out_weights (graph_processing.rs:22-100) for the batch that produced
        `res`; must directly follow the phase_blocks* call on this context (the batch is still resident in HBM)."""
        bp = np.ascontiguousarray(res.best_ploidy, np.uint32)
        c = capi.CBlockResult()
        c.n_blocks = res.n_blocks; c.max_ploidy = res.max_ploidy; c.best_ploidy = capi.ptr(bp, C.c_uint32); c.batch_token = res.batch_token
        out = C.POINTER(capi.CHapGraph)()
        _check(load().floria_hip_hap_graph(self._h, C.byref(c), C.byref(out)))
        g = capi.HapGraph(out.contents)
        load().floria_hip_hap_graph_free(out)
        return g

    def haploset_stats(self, contigs, grp_contig, groups, ranges):
        """get_errors_cov_from_frags (utils_frags.rs:596-655) per haploset -> float64 [n_groups, 4] = cov, err, total_err, total_cov."""
        arr = (C.c_void_p * len(contigs))(*[c._h for c in contigs])
        gc = np.ascontiguousarray(grp_contig, np.uint32)
        off = np.zeros(len(groups) + 1, np.uint64)
        off[1:] = np.cumsum([len(g) for g in groups])
        reads = np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint32) for g in groups]) if len(groups) else np.zeros(0, np.uint32), np.uint32)
        rng = np.ascontiguousarray(np.asarray(ranges, np.uint32).reshape(-1))
        out = np.zeros((len(groups), 4), np.float64)
        _check(load().floria_hip_haploset_stats(self._h, arr, C.c_uint32(len(contigs)), capi.ptr(gc, C.c_uint32), capi.ptr(off, C.c_uint64),
                                                capi.ptr(reads, C.c_uint32), capi.ptr(rng, C.c_uint32), C.c_uint32(len(groups)), capi.ptr(out, C.c_double)))
        return out

    def hapq(self, contig, groups, ranges, snp_to_genome_pos, block_length):
        """get_hapq (part_block_manip.rs:517-616) for one contig's haplosets -> (hapq uint8 [n], rel_err float64 [n], avg_err)."""
        off = np.zeros(len(groups) + 1, np.uint64)
        off[1:] = np.cumsum([len(g) for g in groups])
        reads = np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint32) for g in groups]) if len(groups) else np.zeros(0, np.uint32), np.uint32)
        rng = np.ascontiguousarray(np.asarray(ranges, np.uint32).reshape(-1))
        pos = np.ascontiguousarray(snp_to_genome_pos, np.uint64)
        hq = np.zeros(len(groups), np.uint8)
        rel = np.zeros(len(groups), np.float64)
        avg = C.c_double(0)
        _check(load().floria_hip_hapq(self._h, contig._h, capi.ptr(off, C.c_uint64), capi.ptr(reads, C.c_uint32), capi.ptr(rng, C.c_uint32),
                                      C.c_uint32(len(groups)), capi.ptr(pos, C.c_uint64), C.c_uint32(len(pos)), C.c_uint64(int(block_length)),
                                      capi.ptr(hq, C.c_uint8), capi.ptr(rel, C.c_double), C.byref(avg)))
        return hq, rel, avg.value

    def realign(self, read_windows, ref_windows, alleles, n_alleles, want_scores=False):
        """alignment::realign for n SNP calls: read_windows / ref_windows uint8 [n, 32], alleles uint8 [n, 4], n_alleles uint8 [n]
        -> best allele index uint8 [n] (and the best score int32 [n])."""
        q = np.ascontiguousarray(read_windows, np.uint8); r = np.ascontiguousarray(ref_windows, np.uint8)
        al = np.ascontiguousarray(alleles, np.uint8); na = np.ascontiguousarray(n_alleles, np.uint8)
        n = len(na)
        assert q.shape == (n, 32) and r.shape == (n, 32) and al.shape == (n, 4)
        best = np.zeros(n, np.uint8)
        score = np.zeros(n, np.int32) if want_scores else None
        _check(load().floria_hip_realign(self._h, capi.ptr(q, C.c_uint8), capi.ptr(r, C.c_uint8), capi.ptr(al, C.c_uint8), capi.ptr(na, C.c_uint8), C.c_uint64(n),
                                         capi.ptr(best, C.c_uint8), capi.ptr(score, C.c_int32) if want_scores else None))
        return (best, score) if want_scores else best

    def hapq_batch(self, contigs, grp_contig, groups, ranges, snp_positions, block_length):
        """get_hapq for the haplosets of many contigs in one call -> (hapq uint8 [n], rel_err float64 [n], avg_err float64 [n_contigs])."""
        arr = (C.c_void_p * len(contigs))(*[c._h for c in contigs])
        gc = np.ascontiguousarray(grp_contig, np.uint32)
        off = np.zeros(len(groups) + 1, np.uint64)
        off[1:] = np.cumsum([len(g) for g in groups])
        reads = np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint32) for g in groups]) if len(groups) else np.zeros(0, np.uint32), np.uint32)
        rng = np.ascontiguousarray(np.asarray(ranges, np.uint32).reshape(-1))
        pos = [np.ascontiguousarray(p, np.uint64) for p in snp_positions]
        pp = (C.POINTER(C.c_uint64) * len(pos))(*[capi.ptr(p, C.c_uint64) for p in pos])
        ns = np.ascontiguousarray([len(p) for p in pos], np.uint32)
        hq = np.zeros(len(groups), np.uint8)
        rel = np.zeros(len(groups), np.float64)
        avg = np.zeros(len(contigs), np.float64)
        _check(load().floria_hip_hapq_batch(self._h, arr, C.c_uint32(len(contigs)), capi.ptr(gc, C.c_uint32), capi.ptr(off, C.c_uint64),
                                            capi.ptr(reads, C.c_uint32), capi.ptr(rng, C.c_uint32), C.c_uint32(len(groups)), pp, capi.ptr(ns, C.c_uint32),
                                            C.c_uint64(int(block_length)), capi.ptr(hq, C.c_uint8), capi.ptr(rel, C.c_double), capi.ptr(avg, C.c_double)))
        return hq, rel, avg

    # S2 --------------------------------------------------------------------------------------------
    def reassign_batch(self, contigs, grp_contig, groups, ranges, epsilon, read_orders=None):
        """process_reads_for_final_parts for many resident contigs in one launch -> list of _capi.Groups (contig order)."""
        arr = (C.c_void_p * len(contigs))(*[c._h for c in contigs])
        gc = np.ascontiguousarray(grp_contig, np.uint32)
        off = np.zeros(len(groups) + 1, np.uint64)
        off[1:] = np.cumsum([len(g) for g in groups])
        reads = np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint32) for g in groups]) if len(groups) else np.zeros(0, np.uint32), np.uint32)
        rng = np.ascontiguousarray(np.asarray(ranges, np.uint32).reshape(-1))
        out = C.POINTER(C.POINTER(capi.CGroups))()
        if read_orders is None:
            po, poo = None, None
        else:                                        # one visiting order (array of read ids) per contig
            oo = np.zeros(len(contigs) + 1, np.uint64)
            oo[1:] = np.cumsum([len(o) for o in read_orders])
            ordv = np.ascontiguousarray(np.concatenate([np.asarray(o, np.uint32) for o in read_orders]) if len(read_orders) else np.zeros(0, np.uint32), np.uint32)
            po, poo = capi.ptr(ordv, C.c_uint32), capi.ptr(oo, C.c_uint64)
        _check(load().floria_hip_reassign_batch(self._h, arr, C.c_uint32(len(contigs)), capi.ptr(gc, C.c_uint32), capi.ptr(off, C.c_uint64),
                                                capi.ptr(reads, C.c_uint32), capi.ptr(rng, C.c_uint32), C.c_uint32(len(groups)), po, poo,
                                                C.c_double(epsilon), C.byref(out)))
        res = [capi.Groups(out[i].contents) for i in range(len(contigs))]
        load().floria_hip_groups_array_free(out, C.c_uint32(len(contigs)))
        return res

    def reassign(self, contig, groups, ranges, epsilon, read_order=None):
        """process_reads_for_final_parts (part_block_manip.rs:174-274): groups = list of read-id arrays,
        ranges = [(start,end)] -> _capi.Groups"""
        if isinstance(contig, Pileup):
            rc = self.upload(contig)
            try:
                return self.reassign(rc, groups, ranges, epsilon, read_order)
            finally:
                rc.free()
        off = np.zeros(len(groups) + 1, np.uint64)
        off[1:] = np.cumsum([len(g) for g in groups])
        reads = np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint32) for g in groups]) if len(groups) else np.zeros(0, np.uint32), np.uint32)
        rng = np.ascontiguousarray(np.asarray(ranges, np.uint32).reshape(-1))
        out = C.POINTER(capi.CGroups)()
        if read_order is None:
            po, no = None, 0
        else:
            ordv = np.ascontiguousarray(read_order, np.uint32)
            po, no = capi.ptr(ordv, C.c_uint32), len(ordv)
        _check(load().floria_hip_reassign_ordered(self._h, contig._h, capi.ptr(off, C.c_uint64), capi.ptr(reads, C.c_uint32), capi.ptr(rng, C.c_uint32),
                                                  C.c_uint32(len(groups)), po, C.c_uint32(no), C.c_double(epsilon), C.byref(out)))
        g = capi.Groups(out.contents)
        load().floria_hip_groups_free(out)
        return g
