// upload_kernel.h — device side of floria_hip_contig_upload_batch: validate a batch of raw CSR pileups (the invariants of
// include/floria_hip.h: `Frag` ordering types_structs.rs:87-93, cells strictly ascending, first/last == end cells, 1-based SNP
// indices, allele index < 4) and flatten them into the resident form the clustering kernels read:
//
//   cell_aw[c] = allele << 28 | w24[qual]      phred_scale (utils_frags.rs:702-711) folded into the cell, no LUT in any kernel
//   tw[2r..]   = sum over the read's cells of Rq{1,2}[hash_idx(snp, allele)] * w   (the read's term of the linear state hash)
//   meta[8r..] = {cell offset, cell count, first, last, tw1 lo/hi, tw2 lo/hi}       one 32-B record per beam step
//
// The raw arrays arrive by DMA straight from the caller's (pinned) memory; nothing is computed per cell on the host.
// 16 lanes per read, 16 reads per 256-thread workgroup; per-contig status words are reduced with atomics.
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

// Compact wire form -> CSR, on the device: 16 lanes per read walk the read's presence bits 16 at a time; a set bit j is cell
// read_off[r] + (set bits before j) with SNP first + j and the 2-bit allele of that cell.  The CSR arrays land where a plain upload
// would have put them and flatten_kernel then validates them like any other upload; this kernel only has to catch what would corrupt
// memory or go unnoticed: presence bits that do not add up to the read's cell count (UP_PACKED), spans that disagree with first / last.
__global__ __launch_bounds__(256) void expand_kernel(UploadArgs g, const PackedContig* pk) {
    const uint32_t sub = threadIdx.x & 15, grp = (threadIdx.x & 63) >> 4;
    const uint64_t lr = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4), gr = g.read_base + lr;
    const bool live = lr < g.n_reads_total;
    uint32_t lo = 0, hi = g.n_contigs;
    while (live && hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g.read_prefix[mid] <= gr) lo = mid; else hi = mid; }
    uint32_t b = 0, ncell = 0, F = 0, span = 0, r = 0;
    uint64_t bit0 = 0;
    UploadContig cd{};
    PackedContig pc{};
    bool bad = false;
    if (live) {
        cd = g.contigs[lo]; pc = pk[lo];
        r = (uint32_t)(gr - g.read_prefix[lo]);
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
    for (uint32_t j0 = 0; j0 < smax; j0 += 16) {
        const uint32_t j = j0 + sub;
        bool set = false;
        if (j < span) { const uint64_t bi = bit0 + j; set = (G(pc.present)[bi >> 3] >> (bi & 7)) & 1; }
        const uint32_t m16 = (uint32_t)(__ballot(set) >> (16 * grp)) & 0xffffu;
        const uint32_t rank = seen + (uint32_t)__popc(m16 & ((1u << sub) - 1u));
        if (set && rank < ncell) {
            const uint32_t c = b + rank;
            pc.snp[c] = F + j;
            pc.allele[c] = (uint8_t)((G(pc.allele2)[c >> 2] >> (2 * (c & 3))) & 3u);
        }
        seen += (uint32_t)__popc(m16);
    }
    if (live && sub == 0 && (bad ? G(cd.read_off)[r + 1] > b : seen != ncell))
        atomicMin(&g.status[lo].err, ((unsigned long long)r << 8) | UP_PACKED);
}

__global__ __launch_bounds__(256) void flatten_kernel(UploadArgs g) {
    __shared__ uint32_t s_w24[256];
    s_w24[threadIdx.x] = g.w24[threadIdx.x];
    __syncthreads();
    const uint32_t sub = threadIdx.x & 15;
    const uint64_t lr = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4), gr = g.read_base + lr;
    const bool live = lr < g.n_reads_total;
    // contig of this read: last prefix <= gr  (the 16 lanes of a read search identically; n_contigs is a few thousand at most)
    uint32_t lo = 0, hi = g.n_contigs;
    while (live && hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g.read_prefix[mid] <= gr) lo = mid; else hi = mid; }
    uint64_t t1 = 0, t2 = 0;
    uint32_t r = 0, b = 0, e = 0, code = UP_OK, ma = 0, q0 = 0;
    UploadContig cd{};
    if (live) {
        cd = g.contigs[lo];
        r = (uint32_t)(gr - g.read_prefix[lo]);
        b = G(cd.read_off)[r]; e = G(cd.read_off)[r + 1];
        auto flag = [&](uint32_t c) { code = (code == UP_OK || c < code) ? c : code; };
        if (e <= b || e > cd.n_cells) { flag(UP_NO_CELLS); e = b; }
        else {
            const uint32_t F = G(cd.first)[r], L = G(cd.last)[r];
            if (sub == 0) {
                if (G(cd.snp)[b] != F || G(cd.snp)[e - 1] != L) flag(UP_FIRST_LAST);
                if (G(cd.snp)[b] == 0) flag(UP_ONE_BASED);
                if (r > 0) {                                     // Frag::cmp (types_structs.rs:87-93)
                    const uint32_t Fp = G(cd.first)[r - 1], Lp = G(cd.last)[r - 1];
                    if (!(Fp < F || (Fp == F && Lp >= L))) flag(UP_ORDER);
                }
            }
            for (uint32_t c = b + sub; c < e; c += 16) {
                const uint32_t s = G(cd.snp)[c];
                if (c > b && s <= G(cd.snp)[c - 1]) flag(UP_NOT_ASCENDING);
                uint32_t al = G(cd.allele)[c];
                const uint32_t q = G(cd.qual)[c];
                if (al > 3) { flag(UP_ALLELE); al = 3; }
                const uint32_t w = s_w24[q];
                cd.cell_aw[c] = (al << 28) | w;
                ma = al > ma ? al : ma; q0 |= q == 0 ? 1u : 0u;
                const uint32_t idx = hash_idx(s, al);
                t1 += g.Rq1[idx] * (uint64_t)w; t2 += g.Rq2[idx] * (uint64_t)w;
            }
        }
    }
    // row (16-lane) reductions: DPP quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
    auto row_sum = [](uint64_t v) { v += up_dpp64<0xB1>(v); v += up_dpp64<0x4E>(v); v += up_dpp64<0x141>(v); v += up_dpp64<0x140>(v); return v; };
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
        *(uint4*)mr = make_uint4(b, e - b, G(cd.first)[r], G(cd.last)[r]);
        *(uint4*)(mr + 4) = make_uint4((uint32_t)t1, (uint32_t)(t1 >> 32), (uint32_t)t2, (uint32_t)(t2 >> 32));
        UploadStatus* st = g.status + lo;
        atomicMax(&st->max_len, e - b);
        if (ma) atomicMax(&st->max_allele, ma);
        if (q0) atomicOr(&st->has_q0, 1u);
        if (code != UP_OK) atomicMin(&st->err, ((unsigned long long)r << 8) | code);
    }
}

}  // namespace fl
