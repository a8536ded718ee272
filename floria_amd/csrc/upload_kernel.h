// upload_kernel.h — device side of floria_hip_contig_upload_batch: validate a batch of raw CSR pileups (the invariants of
// include/floria_hip.h: `Frag` ordering types_structs.rs:87-93, cells strictly ascending, first/last == end cells, 1-based SNP
// indices, allele index < 4) and flatten them into the resident form the clustering kernels read:
//
//   cell_aw[c] = allele << 28 | w24[qual]      phred_scale (utils_frags.rs:702-711) folded into the cell, no LUT in any kernel
//   tw[2r..]   = sum over the read's cells of Rq{1,2}[hash_idx(snp, allele)] * w   (the read's term of the linear state hash)
//   meta[8r..] = {cell offset, cell count, first, last, tw1 lo/hi, tw2 lo/hi}       one 32-B record per beam step
//
// The raw arrays arrive by DMA straight from the caller's (pinned) memory; nothing is computed per cell on the host.
// 16 lanes per read, 16 reads per one-wave workgroup (four at a time); per-contig status words are reduced with atomics.
#pragma once
#include "common.h"

namespace fl {

// error codes, in the order the host-side contract lists them; the smallest (read, code) pair of a contig is reported
enum : uint32_t { UP_OK = 0, UP_NO_CELLS = 1, UP_FIRST_LAST = 2, UP_ONE_BASED = 3, UP_NOT_ASCENDING = 4, UP_ALLELE = 5, UP_ORDER = 6 };
constexpr uint32_t UP_PACKED = 0;      // (status word only: expand_kernel's report, ahead of whatever flatten_kernel makes of the same read)

struct UploadContig {              // raw inputs (device copies) and flattened outputs of one contig
    const uint32_t* read_off;      // [n_reads+1]
    const uint32_t* first;
    const uint32_t* last;
    const uint32_t* snp;           // [n_cells]
    const uint8_t*  allele;        // [n_cells] raw, transient
    const uint8_t*  qual;          // [n_cells] raw, transient
    uint32_t* cell_aw;
    uint64_t* tw;
    uint32_t* meta;
    uint32_t  n_reads, n_cells;    // n_cells = read_off[n_reads] as the host read it (bounds every cell access)
};
struct UploadStatus {              // per contig, zero-initialised except err
    unsigned long long err;        // min over failing reads of (read << 8 | code); ~0 = valid
    uint32_t max_len, max_allele, has_q0, pad;
};
struct UploadArgs {
    const UploadContig* contigs;
    const uint64_t* read_prefix;   // [n_contigs+1] global read index of each contig's first read
    UploadStatus* status;
    const uint32_t* w24;           // [256]
    const uint64_t *Rq1, *Rq2;     // [4*HASH_M]
    uint32_t n_contigs, pad;
    uint64_t read_base;            // global index of the launch's first read (read_prefix holds global indices)
    uint64_t n_reads_total;        // reads of this launch
};

// one contig of a PACKED upload (include/floria_hip.h: floria_pileup_packed): device copies of the compact arrays and the CSR arrays expand_kernel writes
struct PackedContig {
    const uint32_t* bit_off;       // [n_reads+1]
    const uint8_t*  present;
    const uint8_t*  allele2;
    uint32_t* snp;                 // out: [n_cells]  (= UploadContig::snp)
    uint8_t*  allele;              // out: [n_cells]  (= UploadContig::allele)
    uint64_t  present_bytes;
};

template <int CTRL> __device__ __forceinline__ uint64_t up_dpp64(uint64_t x) {
    const uint32_t l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, CTRL, 0xf, 0xf, false);
    const uint32_t h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), CTRL, 0xf, 0xf, false);
    return ((uint64_t)h << 32) | l;
}

constexpr int UP_READS_PER_WG = 16;     // one 64-lane workgroup = 4 reads x 16 lanes at a time, 4 times

// Contig of global read gr0: the last c in [0, n) with prefix[c] <= gr0.  All 64 lanes sample the prefix array at once: one memory round trip per
// level, two levels for up to 4096 contigs (a binary search is 12 dependent round trips — most of what a read's flattening used to cost).
__device__ __forceinline__ uint32_t wave_find_contig(const uint64_t* prefix, uint32_t n, uint64_t gr0, uint32_t lane) {
    uint32_t lo = 0, span = n;
    while (span > 1) {
        const uint32_t stride = (span + 63) / 64;
        const bool in = lane * stride < span;
        const uint64_t v = in ? prefix[lo + lane * stride] : ~0ull;
        const uint32_t cnt = (uint32_t)__popcll(__ballot(in && v <= gr0));          // >= 1: prefix[lo] <= gr0
        lo += (cnt - 1) * stride;
        span = min(stride, span - (cnt - 1) * stride);
    }
    return lo;
}

// Compact wire form -> CSR, on the device: 16 lanes per read walk the read's presence bits 16 at a time; a set bit j is cell
// read_off[r] + (set bits before j) with SNP first + j and the 2-bit allele of that cell.  The CSR arrays land where a plain upload
// would have put them and flatten_kernel then validates them like any other upload; this kernel only has to catch what would corrupt
// memory or go unnoticed: presence bits that do not add up to the read's cell count (UP_PACKED), spans that disagree with first / last.
// One-wave workgroups (they fit into any free wave slot while persistent clustering kernels own most of the chip), loads in batches of four.
__global__ __launch_bounds__(64) void expand_kernel(UploadArgs g, const PackedContig* pk) {
    const uint32_t lane = threadIdx.x, sub = lane & 15, grp = lane >> 4;
    const uint64_t lr0 = (uint64_t)blockIdx.x * UP_READS_PER_WG;
    uint32_t ci = wave_find_contig(g.read_prefix, g.n_contigs, g.read_base + lr0, lane);
    for (int it = 0; it < UP_READS_PER_WG / 4; ++it) {
        const uint64_t lr = lr0 + (uint64_t)it * 4 + grp, gr = g.read_base + lr;
        const bool live = lr < g.n_reads_total;
        if (live) while (g.read_prefix[ci + 1] <= gr) ++ci;
        uint32_t b = 0, ncell = 0, F = 0, span = 0, r = 0;
        uint64_t bit0 = 0;
        UploadContig cd{};
        PackedContig pc{};
        bool bad = false;
        if (live) {
            cd = g.contigs[ci]; pc = pk[ci];
            r = (uint32_t)(gr - g.read_prefix[ci]);
            b = G(cd.read_off)[r];
            const uint32_t e = G(cd.read_off)[r + 1];
            F = G(cd.first)[r];
            const uint32_t L = G(cd.last)[r];
            bit0 = G(pc.bit_off)[r];
            const uint64_t bit1 = G(pc.bit_off)[r + 1];
            if (e <= b || e > cd.n_cells || L < F || bit1 <= bit0 || bit1 - bit0 != (uint64_t)(L - F) + 1 || (bit1 + 7) / 8 > pc.present_bytes) bad = true;   // (flatten reports e <= b as UP_NO_CELLS)
            else { ncell = e - b; span = L - F + 1; }
        }
        // the four reads of a wavefront walk together: the longest span sets the trip count
        uint32_t smax = span;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_xor(smax, o); smax = v > smax ? v : smax; }
        uint32_t seen = 0;
        for (uint32_t j0 = 0; j0 < smax; j0 += 64) {
            bool set[4]; uint32_t rank[4], a2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t j = j0 + 16 * u + sub;
                const uint64_t bi = bit0 + j;
                set[u] = j < span ? ((G(pc.present)[bi >> 3] >> (bi & 7)) & 1) != 0 : false;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t m16 = (uint32_t)(__ballot(set[u]) >> (16 * grp)) & 0xffffu;
                rank[u] = seen + (uint32_t)__popc(m16 & ((1u << sub) - 1u));
                seen += (uint32_t)__popc(m16);
                set[u] = set[u] && rank[u] < ncell;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) a2[u] = set[u] ? G(pc.allele2)[(b + rank[u]) >> 2] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (set[u]) {
                    const uint32_t c = b + rank[u];
                    pc.snp[c] = F + j0 + 16 * u + sub;
                    pc.allele[c] = (uint8_t)((a2[u] >> (2 * (c & 3))) & 3u);
                }
        }
        if (live && sub == 0 && (bad ? G(cd.read_off)[r + 1] > b : seen != ncell))
            atomicMin(&g.status[ci].err, ((unsigned long long)r << 8) | UP_PACKED);
    }
}

__global__ __launch_bounds__(64) void flatten_kernel(UploadArgs g) {
    const uint32_t lane = threadIdx.x, sub = lane & 15, grp = lane >> 4;
    const uint64_t lr0 = (uint64_t)blockIdx.x * UP_READS_PER_WG;
    uint32_t ci = wave_find_contig(g.read_prefix, g.n_contigs, g.read_base + lr0, lane);
    // row (16-lane) reductions: DPP quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
    auto row_sum = [](uint64_t v) { v += up_dpp64<0xB1>(v); v += up_dpp64<0x4E>(v); v += up_dpp64<0x141>(v); v += up_dpp64<0x140>(v); return v; };
    for (int it = 0; it < UP_READS_PER_WG / 4; ++it) {
        const uint64_t lr = lr0 + (uint64_t)it * 4 + grp, gr = g.read_base + lr;
        const bool live = lr < g.n_reads_total;
        if (live) while (g.read_prefix[ci + 1] <= gr) ++ci;
        uint64_t t1 = 0, t2 = 0;
        uint32_t r = 0, b = 0, e = 0, code = UP_OK, ma = 0, q0 = 0, F = 0, L = 0;
        UploadContig cd{};
        if (live) {
            cd = g.contigs[ci];
            r = (uint32_t)(gr - g.read_prefix[ci]);
            b = G(cd.read_off)[r]; e = G(cd.read_off)[r + 1];
            F = G(cd.first)[r]; L = G(cd.last)[r];
            auto flag = [&](uint32_t c) { code = (code == UP_OK || c < code) ? c : code; };
            if (e <= b || e > cd.n_cells) { flag(UP_NO_CELLS); e = b; }
            else {
                if (sub == 0) {
                    if (G(cd.snp)[b] != F || G(cd.snp)[e - 1] != L) flag(UP_FIRST_LAST);
                    if (G(cd.snp)[b] == 0) flag(UP_ONE_BASED);
                    if (r > 0) {                                     // Frag::cmp (types_structs.rs:87-93)
                        const uint32_t Fp = G(cd.first)[r - 1], Lp = G(cd.last)[r - 1];
                        if (!(Fp < F || (Fp == F && Lp >= L))) flag(UP_ORDER);
                    }
                }
                // four cells per lane at a time: their raw fields are requested together, then the table entries they select
                for (uint32_t c0 = b + sub; c0 < e; c0 += 64) {
                    uint32_t sn[4], sp[4], al[4], q[4], w[4];
                    uint64_t r1[4], r2[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t c = c0 + 16 * u;
                        const bool v = c < e;
                        sn[u] = v ? G(cd.snp)[c] : 0u; sp[u] = v && c > b ? G(cd.snp)[c - 1] : 0u;
                        al[u] = v ? (uint32_t)G(cd.allele)[c] : 0u; q[u] = v ? (uint32_t)G(cd.qual)[c] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (al[u] > 3) { flag(UP_ALLELE); al[u] = 3; }
                        w[u] = g.w24[q[u]];
                        const uint32_t idx = hash_idx(sn[u], al[u]);
                        r1[u] = g.Rq1[idx]; r2[u] = g.Rq2[idx];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t c = c0 + 16 * u;
                        if (c >= e) break;
                        if (c > b && sn[u] <= sp[u]) flag(UP_NOT_ASCENDING);
                        cd.cell_aw[c] = (al[u] << 28) | w[u];
                        ma = al[u] > ma ? al[u] : ma; q0 |= q[u] == 0 ? 1u : 0u;
                        t1 += r1[u] * (uint64_t)w[u]; t2 += r2[u] * (uint64_t)w[u];
                    }
                }
            }
        }
        t1 = row_sum(t1); t2 = row_sum(t2);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t oc = __shfl_xor(code, o), om = __shfl_xor(ma, o), oq = __shfl_xor(q0, o);
            code = (oc != UP_OK && (code == UP_OK || oc < code)) ? oc : code;
            ma = om > ma ? om : ma; q0 |= oq;
        }
        if (live && sub == 0) {
            cd.tw[2 * (uint64_t)r] = t1; cd.tw[2 * (uint64_t)r + 1] = t2;
            uint32_t* mr = cd.meta + 8 * (uint64_t)r;
            *(uint4*)mr = make_uint4(b, e - b, F, L);
            *(uint4*)(mr + 4) = make_uint4((uint32_t)t1, (uint32_t)(t1 >> 32), (uint32_t)t2, (uint32_t)(t2 >> 32));
            UploadStatus* st = g.status + ci;
            atomicMax(&st->max_len, e - b);
            if (ma) atomicMax(&st->max_allele, ma);
            if (q0) atomicOr(&st->has_q0, 1u);
            if (code != UP_OK) atomicMin(&st->err, ((unsigned long long)r << 8) | code);
        }
    }
}

}  // namespace fl
