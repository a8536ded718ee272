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
#include "wave_util.h"     // rl32 / rl64 / uni

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
    const uint32_t* list;           // the contigs this launch handles
    uint32_t n_list;
    double eps;
    uint32_t* queue_head;
    // reference-arithmetic mode (reassign_kernel<A, true>, arith_kernel.h): the reads' cells in the iteration order of Frag.positions
    const uint2*    cell_ord;
    const uint64_t* cell_ord_off;
};

constexpr int REASSIGN_THREADS = 256;

// Contigs in which few reads have a choice: one WORKGROUP per contig.  A read with a single candidate group has no choice to make (part_block_manip.rs:203-222 picks the
// minimum over one element): its assignment is known up front and its add_read_to_block only matters to a LATER read that has a
// choice.  So the sequential chain runs over the multi-candidate reads only, and between two of them every single-candidate read
// of the visiting order is added to its histogram by all lanes at once (integer atomics: the sums do not depend on the order of
// the adds, and every read that chooses sees exactly the contributions of the reads visited before it — the result equals the
// serial loop's for ANY visiting order).  Contigs without a multi-candidate read need no histogram at all.
// ARITH: the reference's own f64 arithmetic (floria_hip_set_option("arith", 1)): the distance is the running sum over the read's cells in the order
// of its position set (utils_frags.rs:33-72) — lane x folds the x-th candidate group, cell by cell.
template <int A, bool ARITH = false>
__global__ __launch_bounds__(REASSIGN_THREADS) void reassign_kernel(ReassignArgs g) {
    __shared__ uint32_t s_ci;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_ci = atomicAdd(g.queue_head, 1u);
        __syncthreads();
        if (s_ci >= g.n_list) break;
        const uint32_t ci = g.list[s_ci];
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
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // the histogram atomics (performed at L2) have completed; the loads below are agent-scope loads of the same L2.  (An agent-scope fence would write the XCD's whole L2 back and invalidate it, twice per read with a choice.)
            __syncthreads();
            if (wid == 0) {                                             // the read that has a choice: one wavefront, lanes over its cells
                const uint32_t r = ord ? ord[m] : m;
                const uint64_t c0 = roff[r];
                const uint32_t nc = (uint32_t)(roff[r + 1] - c0);
                const uint32_t cb = G(cd.read_off)[r], ce = G(cd.read_off)[r + 1];
                uint32_t best = r2g[c0];
                double bd = 0.0, bsame = 0.0;
                bool have = false;
                if constexpr (ARITH) {
                    const uint2* co = g.cell_ord + g.cell_ord_off[ci];
                    for (uint32_t x0 = 0; x0 < nc; x0 += 64) {
                        const uint32_t x = x0 + lane;
                        const bool act = x < nc;
                        const uint32_t gid = act ? r2g[c0 + x] : 0;
                        const uint64_t* h = g.hist + g.grp_hist_off[gb + gid];
                        const uint32_t p0 = g.grp_pos0[gb + gid];
                        double df = 0.0; uint64_t qs = 0;
                        if (act)
                            for (uint32_t c = cb; c < ce; ++c) {
                                const uint2 ca = co[c];
                                const uint32_t aq = ca.y, al = aq >> 28;
                                const uint64_t* cp = h + (uint64_t)(ca.x - p0) * A;
                                uint64_t mx = 0, va = 0;
#pragma unroll
                                for (int a = 0; a < A; ++a) { const uint64_t q = __hip_atomic_load(cp + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mx = q > mx ? q : mx; va = (a == (int)al) ? q : va; }
                                if (mx == 0) df += g.eps;
                                else if (va == mx) qs += (aq & 0x0fffffffu);
                                else df += (double)(aq & 0x0fffffffu) * 0x1p-24;
                            }
                        const double kd_l = df + 1., ks_l = (double)qs * 0x1p-24;
                        const uint32_t cnt = nc - x0 < 64u ? nc - x0 : 64u;
                        for (uint32_t y = 0; y < cnt; ++y) {                          // candidates in ascending id, as the reference's min_by visits them
                            const double kd = shfl_f64(kd_l, (int)y), ks = shfl_f64(ks_l, (int)y);
                            const uint32_t gy = (uint32_t)__shfl((int)gid, (int)y);
                            const bool less = !have || kd < bd || (kd == bd && (gy < best || (gy == best && ks < bsame)));
                            if (less) { have = true; bd = kd; bsame = ks; best = gy; }
                        }
                    }
                } else
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
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // the histogram atomics (performed at L2) have completed; the loads below are agent-scope loads of the same L2.  (An agent-scope fence would write the XCD's whole L2 back and invalidate it, twice per read with a choice.)
            __syncthreads();
            prev = m + 1;
        }
        // (the single-candidate reads after the last choice change histograms nobody reads any more)
    }
}

// Contigs in which MOST reads have a choice (e.g. haplosets of overlapping blocks fed in unstitched): the chain is as long as the
// read list, so what matters is the latency of one link.  One wavefront per contig, software-pipelined three reads deep.
template <int A>
__global__ __launch_bounds__(64) void reassign_chain_kernel(ReassignArgs g) {
    const uint32_t lane = threadIdx.x;
    constexpr int K = 4;                                   // cells per lane held in registers: reads of up to 256 cells take the fast path
    for (;;) {
        uint32_t qi = 0;
        if (lane == 0) qi = atomicAdd(g.queue_head, 1u);
        qi = __shfl(qi, 0);
        if (qi >= g.n_list) break;
        const uint32_t ci = g.list[qi];
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
