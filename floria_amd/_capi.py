"""ctypes mirror of include/floria_hip.h (struct layouts only; no library is loaded here)."""
import ctypes as C

import numpy as np

FLORIA_OK = 0
FLORIA_E_INVALID = -1
FLORIA_E_DEVICE = -2
FLORIA_E_NOMEM = -3
FLORIA_E_UNSUPPORTED = -4
FLORIA_MAX_ALLELES = 4
FLORIA_MAX_PLOIDY = 16

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


class CPileup(C.Structure):
    _fields_ = [("read_off", u32p), ("snp", u32p), ("allele", u8p), ("qual", u8p),
                ("first", u32p), ("last", u32p), ("n_reads", C.c_uint32), ("set_order", u32p)]


class CPileupPacked(C.Structure):
    _fields_ = [("read_off", u32p), ("first", u32p), ("last", u32p), ("bit_off", u32p),
                ("present", u8p), ("allele2", u8p), ("qual", u8p), ("n_reads", C.c_uint32), ("set_order", u32p)]


class CParams(C.Structure):
    _fields_ = [("epsilon", C.c_double), ("max_ploidy", C.c_uint32), ("beam", C.c_uint32),
                ("ploidy_sensitivity", C.c_uint32), ("stopping_heuristic", C.c_int32)]


class CBlockResult(C.Structure):
    _fields_ = [("n_blocks", C.c_uint32), ("max_ploidy", C.c_uint32), ("best_ploidy", u32p),
                ("ploidies_tried", u32p), ("read_off", u64p), ("read_id", u32p), ("part", u8p),
                ("mec", f64p), ("min_prune_margin", C.c_double), ("batch_token", C.c_uint64)]


i32p = C.POINTER(C.c_int32)


class CHapGraph(C.Structure):
    _fields_ = [("n_blocks", C.c_uint32), ("node_off", u64p), ("node_cov", f64p), ("pred", i32p), ("edge_off", u64p), ("edge_w", u32p)]


class CGroups(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("grp_off", u64p), ("grp_read", u32p), ("range", u32p)]


class CRanges(C.Structure):
    _fields_ = [("n", C.c_uint32), ("start", u32p), ("end", u32p)]


class CTiming(C.Structure):
    _fields_ = [("beam_ms", C.c_double), ("optimize_ms", C.c_double), ("select_ms", C.c_double),
                ("reassign_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("total_ms", C.c_double), ("beam_launches", C.c_uint32), ("optimize_launches", C.c_uint32),
                ("algorithmic_bytes", C.c_uint64), ("beam_steps", C.c_uint64),
                ("beam_launch_bytes", C.c_uint64), ("jobs", C.c_uint64),
                ("streams", C.c_uint32), ("stage_width", C.c_uint32), ("phase_ms", C.c_double),
                ("upload_pinned_bytes", C.c_uint64), ("upload_staged_bytes", C.c_uint64),
                ("upload_chunks", C.c_uint32), ("reserved", C.c_uint32),
                ("beam_union_ms", C.c_double), ("optimize_union_ms", C.c_double)]


def ptr(a, ctype):
    """Pointer to a C-contiguous numpy array (the array must outlive the call)."""
    return a.ctypes.data_as(C.POINTER(ctype))


def np_from(p, n, dtype):
    """Copy n elements from a ctypes pointer into a fresh numpy array."""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True)


class BlockResult:
    """Python view of floria_block_result (copied out of library-owned memory)."""

    def __init__(self, c: CBlockResult):
        nb, mp = c.n_blocks, c.max_ploidy
        self.n_blocks, self.max_ploidy = nb, mp
        self.best_ploidy = np_from(c.best_ploidy, nb, np.uint32)
        self.ploidies_tried = np_from(c.ploidies_tried, nb, np.uint32)
        self.read_off = np_from(c.read_off, nb + 1, np.uint64)
        tot = int(self.read_off[nb]) if nb else 0
        self.read_id = np_from(c.read_id, tot, np.uint32)
        self.part = np_from(c.part, tot, np.uint8)
        self.mec = np_from(c.mec, nb * mp, np.float64).reshape(nb, mp)
        self.min_prune_margin = float(c.min_prune_margin)
        self.batch_token = int(c.batch_token)

    def block(self, b):
        lo, hi = int(self.read_off[b]), int(self.read_off[b + 1])
        return self.read_id[lo:hi], self.part[lo:hi]

    def partitions(self, b):
        """List of read-id arrays, one per haplotype (the `Vec<FxHashSet<&Frag>>` of the reference)."""
        ids, part = self.block(b)
        return [ids[part == k] for k in range(int(self.best_ploidy[b]))]


class HapGraph:
    def __init__(self, c):
        nb = c.n_blocks
        self.n_blocks = nb
        self.node_off = np_from(c.node_off, nb + 1, np.uint64)
        self.node_cov = np_from(c.node_cov, int(self.node_off[nb]) if nb else 0, np.float64)
        self.pred = np_from(c.pred, nb, np.int32)
        self.edge_off = np_from(c.edge_off, nb + 1, np.uint64)
        self.edge_w = np_from(c.edge_w, int(self.edge_off[nb]) if nb else 0, np.uint32)


class Groups:
    def __init__(self, c: CGroups):
        n = c.n_groups
        self.n_groups = n
        self.grp_off = np_from(c.grp_off, n + 1, np.uint64)
        tot = int(self.grp_off[n]) if n else 0
        self.grp_read = np_from(c.grp_read, tot, np.uint32)
        self.range = np_from(c.range, 2 * n, np.uint32).reshape(n, 2)

    def group(self, g):
        return self.grp_read[int(self.grp_off[g]):int(self.grp_off[g + 1])]
