// reassign_kernel.h — final read -> haplogroup reassignment, one contig per workgroup.
//
// Follows part_block_manip.rs:184-222 (process_reads_for_final_parts): every read that sits in >= 1
// stitched haplogroup is removed from all of them (:195-200 — after which every histogram is empty) and
// greedily re-inserted into the candidate with minimal (diff + 1, id, same) (:203-222), the histograms
// being updated after every insertion (add_read_to_block, utils_frags.rs:465-474).  The chain is
// sequentially dependent and strongly order-sensitive (DESIGN.md "Iteration order"): reads are visited in the order the
// caller passes (a Rust host passes its own read_to_parts_map iteration order and gets the reference's result), or in
// ascending counter_id when none is given.
// separate_broken_haplogroups / sort_parts (:27-98, :276-288) are integer bookkeeping done by the host.
#pragma once
#include "beam_fast_kernel.h"     // rl32 / rl64 / uni

namespace fl {

struct ReassignArgs {
    const ContigDev* contigs;
    uint32_t n_contigs;
    // per contig (CSR over contigs)
    const uint64_t* r2g_off_base;   // [n_contigs]   offset of the contig's read->groups offsets array
    const uint64_t* r2g_off;        // per contig: [n_reads+1] offsets into r2g (relative to the contig's r2g base)
    const uint64_t* r2g_base;       // [n_contigs]
    const uint32_t* r2g;            // candidate group ids (contig-local, ascending) of every read
    const uint64_t* grp_base;       // [n_contigs]   first group slot of the contig
    const uint64_t* grp_hist_off;   // [total groups] offset (u64 cells) of the group's histogram window
    const uint32_t* grp_pos0;       // [total groups] first SNP position of the window
    uint64_t* hist;                 // zero-initialised, [sum window*A]
    int32_t*  assign;               // per contig reads: chosen contig-local group id or -1
    const uint64_t* assign_base;    // [n_contigs]
    const uint32_t* order;          // optional caller-given visiting order of the reads (nullptr = ascending counter_id)
    const uint64_t* order_off;      // [n_contigs+1] into order
    const uint32_t* multi;          // visiting indices (ascending) of the reads with MORE THAN ONE candidate group, per contig
    const uint64_t* multi_off;      // [n_contigs+1] into multi
    double eps;
    uint32_t* queue_head;
};

constexpr int REASSIGN_THREADS = 256;

// One WORKGROUP per contig.  A read with a single candidate group has no choice to make (part_block_manip.rs:203-222 picks the
// minimum over one element): its assignment is known up front and its add_read_to_block only matters to a LATER read that has a
// choice.  So the sequential chain runs over the multi-candidate reads only, and between two of them every single-candidate read
// of the visiting order is added to its histogram by all lanes at once (integer atomics: the sums do not depend on the order of
// the adds, and every read that chooses sees exactly the contributions of the reads visited before it — the result equals the
// serial loop's for ANY visiting order).  Contigs without a multi-candidate read need no histogram at all.
template <int A>
__global__ __launch_bounds__(REASSIGN_THREADS) void reassign_kernel(ReassignArgs g) {
    __shared__ uint32_t s_ci;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_ci = atomicAdd(g.queue_head, 1u);
        __syncthreads();
        const uint32_t ci = s_ci;
        if (ci >= g.n_contigs) break;
        const ContigDev cd = g.contigs[ci];
        const uint64_t* roff = g.r2g_off + g.r2g_off_base[ci];
        const uint32_t* r2g = g.r2g + g.r2g_base[ci];
        const uint64_t gb = g.grp_base[ci];
        int32_t* assign = g.assign + g.assign_base[ci];
        const uint32_t* ord = g.order ? g.order + g.order_off[ci] : nullptr;
        const uint32_t n_visit = g.order ? (uint32_t)(g.order_off[ci + 1] - g.order_off[ci]) : cd.n_reads;
        const uint32_t* multi = g.multi + g.multi_off[ci];
        const uint32_t n_multi = (uint32_t)(g.multi_off[ci + 1] - g.multi_off[ci]);
        // every single-candidate read: assigned to its candidate
        for (uint32_t v = tid; v < n_visit; v += REASSIGN_THREADS) {
            const uint32_t r = ord ? ord[v] : v;
            const uint64_t c0 = roff[r], c1 = roff[r + 1];
            if (c1 - c0 == 1) assign[r] = (int32_t)r2g[c0];
        }
        // add_read_to_block (utils_frags.rs:465-474) for the single-candidate reads of the visiting range [v0, v1): 16 lanes per read
        auto bulk_add = [&](uint32_t v0, uint32_t v1) {
            const uint32_t grp = tid >> 4, sub = tid & 15;
            for (uint32_t v = v0 + grp; v < v1; v += REASSIGN_THREADS / 16) {
                const uint32_t r = ord ? ord[v] : v;
                const uint64_t c0 = roff[r];
                if (roff[r + 1] - c0 != 1) continue;
                const uint32_t gid = r2g[c0];
                unsigned long long* h = (unsigned long long*)(g.hist + g.grp_hist_off[gb + gid]);
                const uint32_t p0 = g.grp_pos0[gb + gid];
                const uint32_t cb = G(cd.read_off)[r], ce = G(cd.read_off)[r + 1];
                for (uint32_t c = cb + sub; c < ce; c += 16) {
                    const uint32_t aq = G(cd.cell_aw)[c];
                    atomicAdd(h + (uint64_t)(G(cd.cell_snp)[c] - p0) * A + (aq >> 28), (unsigned long long)(aq & 0x0fffffffu));
                }
            }
        };
        uint32_t prev = 0;
        for (uint32_t k = 0; k < n_multi; ++k) {
            const uint32_t m = multi[k];
            bulk_add(prev, m);
            __threadfence();
            __syncthreads();
            if (wid == 0) {                                             // the read that has a choice: one wavefront, lanes over its cells
                const uint32_t r = ord ? ord[m] : m;
                const uint64_t c0 = roff[r];
                const uint32_t nc = (uint32_t)(roff[r + 1] - c0);
                const uint32_t cb = G(cd.read_off)[r], ce = G(cd.read_off)[r + 1];
                uint32_t best = r2g[c0];
                double bd = 0.0, bsame = 0.0;
                bool have = false;
                for (uint32_t x = 0; x < nc; ++x) {
                    const uint32_t gid = r2g[c0 + x];
                    const uint64_t* h = g.hist + g.grp_hist_off[gb + gid];
                    const uint32_t p0 = g.grp_pos0[gb + gid];
                    uint64_t qs = 0, qd = 0; uint32_t mm = 0;
                    for (uint32_t c = cb + lane; c < ce; c += 64) {               // distance_read_haplo_epsilon_empty, utils_frags.rs:32-75
                        const uint32_t aq = G(cd.cell_aw)[c], al = aq >> 28;
                        const uint64_t* cp = h + (uint64_t)(G(cd.cell_snp)[c] - p0) * A;
                        uint64_t mx = 0, va = 0;
#pragma unroll
                        for (int a = 0; a < A; ++a) {                             // (coherent loads: the cells were written by atomics of other wavefronts)
                            const uint64_t q = __hip_atomic_load(cp + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            mx = q > mx ? q : mx; va = (a == (int)al) ? q : va;
                        }
                        if (mx == 0) mm += 1;
                        else if (va == mx) qs += (aq & 0x0fffffffu);
                        else qd += (aq & 0x0fffffffu);
                    }
                    qs = wave_sum_u64(qs); qd = wave_sum_u64(qd); mm = wave_sum_u32(mm);
                    const double kd = qm_to_f64(qd, mm, g.eps) + 1.;              // (diff + 1., id, same) :211
                    const double ks = qm_to_f64(qs, 0, g.eps);
                    const bool less = !have || kd < bd || (kd == bd && (gid < best || (gid == best && ks < bsame)));
                    if (less) { have = true; bd = kd; bsame = ks; best = gid; }
                }
                unsigned long long* h = (unsigned long long*)(g.hist + g.grp_hist_off[gb + best]);
                const uint32_t p0 = g.grp_pos0[gb + best];
                for (uint32_t c = cb + lane; c < ce; c += 64) {
                    const uint32_t aq = G(cd.cell_aw)[c];
                    atomicAdd(h + (uint64_t)(G(cd.cell_snp)[c] - p0) * A + (aq >> 28), (unsigned long long)(aq & 0x0fffffffu));
                }
                if (lane == 0) assign[r] = (int32_t)best;
            }
            __threadfence();
            __syncthreads();
            prev = m + 1;
        }
        // (the single-candidate reads after the last choice change histograms nobody reads any more)
    }
}

}  // namespace fl
