// stats_kernel.h — per-haploset coverage and error statistics (first half):
// utils_frags::get_errors_cov_from_frags (utils_frags.rs:596-655), the COV / ERR fields of the vartig and haploset headers
// (file_writer.rs:801-815,952-967).  Unit-count histogram (set_to_seq_dict(.., false), :606) of the haploset's reads over its
// SNP range, then per position: support = sum of allele counts, max_count = the running-sum rule of :619-625 evaluated
// in ascending allele order (for biallelic sites this is the larger count in any order), errors += support - max_count.
// One 256-thread workgroup per haploset; the histogram lives in a per-call HBM slab (u32 counts, atomics).
#pragma once
#include "common.h"

namespace fl {

struct StatsArgs {
    const ContigDev* contigs;
    const uint32_t* grp_contig;     // [n_groups]
    const uint64_t* grp_off;        // [n_groups+1] into grp_read
    const uint32_t* grp_read;
    const uint32_t* grp_range;      // [2*n_groups] inclusive SNP range
    const uint64_t* hist_off;       // [n_groups+1] u32 cells (range * A) per group
    uint32_t* hist;                 // zero-initialised
    double*   out;                  // [4*n_groups]: cov, err, total_err, total_cov
    uint32_t  n_groups;
};

template <int A>
__global__ __launch_bounds__(256) void stats_kernel(StatsArgs g) {
    __shared__ unsigned long long s_sup, s_err;
    __shared__ uint32_t s_nz;
    const uint32_t gi = blockIdx.x, tid = threadIdx.x;
    if (gi >= g.n_groups) return;
    const ContigDev cd = g.contigs[g.grp_contig[gi]];
    const uint32_t lo = g.grp_range[2 * gi], hi = g.grp_range[2 * gi + 1];
    uint32_t* hist = g.hist + g.hist_off[gi];
    if (tid == 0) { s_sup = 0; s_err = 0; s_nz = 0; }
    __syncthreads();
    if (hi >= lo) {
        const uint32_t grp = tid >> 4, sub = tid & 15;
        const uint64_t r0 = g.grp_off[gi], r1 = g.grp_off[gi + 1];
        for (uint64_t i = r0 + grp; i < r1; i += 16) {                    // 16 lanes per read
            const uint32_t r = g.grp_read[i];
            const uint32_t cb = G(cd.read_off)[r], ce = G(cd.read_off)[r + 1];
            for (uint32_t c = cb + sub; c < ce; c += 16) {
                const uint32_t sn = G(cd.cell_snp)[c];
                if (sn >= lo && sn <= hi) atomicAdd(&hist[(uint64_t)(sn - lo) * A + (G(cd.cell_aw)[c] >> 28)], 1u);
            }
        }
        __syncthreads();
        unsigned long long sup = 0, err = 0;
        uint32_t nz = 0;
        for (uint32_t pr = tid; pr <= hi - lo; pr += 256) {
            uint32_t support = 0, mx = 0;
            bool any = false;
            uint32_t cn[A];
            bool all = true;
#pragma unroll
            for (int a = 0; a < A; ++a) { cn[a] = hist[(uint64_t)pr * A + a]; all = all && cn[a] != 0; }
#pragma unroll
            for (int x = 0; x < A; ++x) {            // the inner map's iteration order: ascending, except 0,2,1,3 with all four alleles (hapq_kernel.h)
                const int a = (A == 4 && all) ? (x == 1 ? 2 : x == 2 ? 1 : x) : x;
                const uint32_t cnt = cn[a];
                if (cnt) { any = true; if (cnt > support) mx = cnt; support += cnt; }      // :619-625
            }
            if (any) { nz++; sup += support; err += support - mx; }
        }
        sup = wave_sum_u64(sup); err = wave_sum_u64(err); nz = wave_sum_u32(nz);
        if ((tid & 63) == 0) { atomicAdd(&s_sup, sup); atomicAdd(&s_err, err); atomicAdd(&s_nz, nz); }
    }
    __syncthreads();
    if (tid == 0) {
        const double total_support = (double)s_sup, errors = (double)s_err;
        g.out[4 * (uint64_t)gi + 0] = s_nz ? total_support / (double)s_nz : 0.0;            // mean over the covered SNPs (:641-647)
        g.out[4 * (uint64_t)gi + 1] = errors / total_support;                               // NaN for an empty haploset, as in the reference
        g.out[4 * (uint64_t)gi + 2] = errors;
        g.out[4 * (uint64_t)gi + 3] = total_support;
    }
}

}  // namespace fl
