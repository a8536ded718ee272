// reassign_kernel.h — final read -> haplogroup reassignment, one contig per wavefront.
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
    double eps;
    uint32_t* queue_head;
};

template <int A>
__global__ __launch_bounds__(64) void reassign_kernel(ReassignArgs g) {
    const uint32_t lane = threadIdx.x;
    constexpr int K = 4;                                   // cells per lane held in registers: reads of up to 256 cells take the fast path
    for (;;) {
        uint32_t ci = 0;
        if (lane == 0) ci = atomicAdd(g.queue_head, 1u);
        ci = __shfl(ci, 0);
        if (ci >= g.n_contigs) break;
        const ContigDev cd = g.contigs[ci];
        const uint64_t* roff = g.r2g_off + g.r2g_off_base[ci];
        const uint32_t* r2g = g.r2g + g.r2g_base[ci];
        const uint64_t gb = g.grp_base[ci];
        int32_t* assign = g.assign + g.assign_base[ci];
        const uint32_t* ord = g.order ? g.order + g.order_off[ci] : nullptr;
        const uint32_t n_visit = g.order ? (uint32_t)(g.order_off[ci + 1] - g.order_off[ci]) : cd.n_reads;

        // The chain read -> candidate groups -> their histogram windows -> cells -> histogram cells is five dependent loads; every
        // one of them would be a full memory round trip per read.  Software pipeline over the visiting order, three reads deep:
        //   S1 (read v+3): candidate range [c0, c1) and cell range [cb, ce)      S2 (read v+2): candidate ids (lane x = x-th) and cells
        //   S3 (read v+1): histogram offset / first position of lane x's candidate
        // so that read v starts with everything but the histogram cells in registers.
        struct Rd {
            uint32_t r; uint64_t c0, c1; uint32_t cb, ce;           // S1 (wave-uniform values, loaded by every lane)
            uint32_t sn[K], aw[K], gid;                             // S2
            uint64_t hoff; uint32_t p0;                             // S3
        };
        auto s1 = [&](uint32_t v, Rd& d) {
            const bool in = v < n_visit;
            d.r = in ? (ord ? ord[v] : v) : 0;
            d.c0 = in ? roff[d.r] : 0; d.c1 = in ? roff[d.r + 1] : 0;
            d.cb = G(cd.read_off)[d.r]; d.ce = G(cd.read_off)[d.r + 1];
        };
        auto s2 = [&](Rd& d) {
            const uint32_t nc = (uint32_t)(d.c1 - d.c0), L = d.ce - d.cb;
            d.gid = lane < nc ? r2g[d.c0 + lane] : 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint32_t c = lane + 64 * k; const bool v = c < L && nc != 0;
                d.sn[k] = v ? G(cd.cell_snp)[d.cb + c] : 0; d.aw[k] = v ? G(cd.cell_aw)[d.cb + c] : 0;
            }
        };
        auto s3 = [&](Rd& d) {
            const uint32_t nc = (uint32_t)(d.c1 - d.c0);
            d.hoff = lane < nc ? g.grp_hist_off[gb + d.gid] : 0; d.p0 = lane < nc ? g.grp_pos0[gb + d.gid] : 0;
        };
        Rd cur, n1, n2, n3;
        s1(0, cur); s1(1, n1); s1(2, n2);
        s2(cur); s2(n1);
        s3(cur);
        for (uint32_t v = 0; v < n_visit; ++v) {
            s1(v + 3, n3); s2(n2); s3(n1);                           // requests for the following reads, behind nothing that this read waits for first
            const uint32_t r = uni(cur.r);
            const uint32_t nc = uni((uint32_t)(cur.c1 - cur.c0));
            const uint32_t cb = uni(cur.cb), ce = uni(cur.ce), L = ce - cb;
            if (nc != 0) {
                const bool fast = L <= 64u * K && nc <= 64u;
                uint32_t best = fast ? rl32(cur.gid, 0) : r2g[cur.c0];
                uint32_t best_x = 0;
                if (nc > 1) {
                    double bd = 0.0, bsame = 0.0;
                    bool have = false;
                    for (uint32_t x = 0; x < nc; ++x) {
                        uint32_t gid; const uint64_t* h; uint32_t p0;
                        if (fast) { gid = rl32(cur.gid, x); h = g.hist + rl64(cur.hoff, x); p0 = rl32(cur.p0, x); }
                        else { gid = r2g[cur.c0 + x]; h = g.hist + g.grp_hist_off[gb + gid]; p0 = g.grp_pos0[gb + gid]; }
                        uint64_t qs = 0, qd = 0; uint32_t m = 0;
                        auto classify = [&](uint32_t snp, uint32_t aq) {                      // utils_frags.rs:32-75
                            const uint32_t al = aq >> 28;
                            const uint64_t* cp = h + (uint64_t)(snp - p0) * A;
                            uint64_t mx = 0, va = 0;
#pragma unroll
                            for (int a = 0; a < A; ++a) { const uint64_t q = cp[a]; mx = q > mx ? q : mx; va = (a == (int)al) ? q : va; }
                            if (mx == 0) m += 1;
                            else if (va == mx) qs += (aq & 0x0fffffffu);
                            else qd += (aq & 0x0fffffffu);
                        };
                        if (fast) {
#pragma unroll
                            for (int k = 0; k < K; ++k) if (lane + 64 * k < L) classify(cur.sn[k], cur.aw[k]);
                        } else {
                            for (uint32_t c = cb + lane; c < ce; c += 64) classify(G(cd.cell_snp)[c], G(cd.cell_aw)[c]);
                        }
                        qs = wave_sum_u64(qs); qd = wave_sum_u64(qd); m = wave_sum_u32(m);
                        const double kd = qm_to_f64(qd, m, g.eps) + 1.;              // (diff + 1., id, same) :211
                        const double ks = qm_to_f64(qs, 0, g.eps);
                        const bool less = !have || kd < bd || (kd == bd && (gid < best || (gid == best && ks < bsame)));
                        if (less) { have = true; bd = kd; bsame = ks; best = gid; best_x = x; }
                    }
                }
                // add_read_to_block (utils_frags.rs:465-474)
                {
                    uint64_t* h; uint32_t p0;
                    if (fast) { h = g.hist + rl64(cur.hoff, best_x); p0 = rl32(cur.p0, best_x); }
                    else { h = g.hist + g.grp_hist_off[gb + best]; p0 = g.grp_pos0[gb + best]; }
                    if (fast) {
                        uint64_t* ptr[K]; uint64_t val[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) ptr[k] = h + (uint64_t)(cur.sn[k] - p0) * A + (cur.aw[k] >> 28);
#pragma unroll
                        for (int k = 0; k < K; ++k) if (lane + 64 * k < L) val[k] = *ptr[k];
#pragma unroll
                        for (int k = 0; k < K; ++k) if (lane + 64 * k < L) *ptr[k] = val[k] + (cur.aw[k] & 0x0fffffffu);
                    } else {
                        for (uint32_t c = cb + lane; c < ce; c += 64) {
                            const uint32_t aq = G(cd.cell_aw)[c];
                            h[(uint64_t)(G(cd.cell_snp)[c] - p0) * A + (aq >> 28)] += (aq & 0x0fffffffu);
                        }
                    }
                }
                if (lane == 0) assign[r] = (int32_t)best;
            }
            __syncthreads();
            cur = n1; n1 = n2; n2 = n3;
        }
    }
}

}  // namespace fl
