// beam_kernel.h — beam-search phasing of one (SNP block, ploidy) job per wavefront.
//
// Follows global_clustering.rs:10-179 (beam_search_phasing) + :181-208 (read_to_node_value) +
// types_structs.rs:326-376 (build_truncated_hap_block) + utils_frags.rs:32-75
// (distance_read_haplo_epsilon_empty) + :211-258 (stable_binom_cdf_p_rev, log_sum_exp), including the
// std::collections::BinaryHeap push/pop/into_sorted_vec tie-breaking.
//
// MI355X design (DESIGN.md §Beam kernel):
//  * one 64-lane wavefront = one job; a persistent grid of `slots` waves pulls jobs from an atomic queue,
//    so thousands of sequential beam searches are in flight and their dependent steps interleave;
//  * a beam state's histograms live in a per-slot slab in HBM/L2, layout [pos][partition][allele] u64,
//    so (a) the p partitions x A alleles of one position are one contiguous 16..128 B piece,
//    (b) cloning a state on a branch is a contiguous, fully coalesced copy of the live SNP window only,
//    (c) "truncate positions < read.first" (types_structs.rs:356-358) is free: a window lower bound;
//  * lane <-> (state, partition) pair: the read's cells are wave-uniform (staged in LDS, read as
//    broadcasts), every lane walks them against its own partition's cells -> no cross-lane reduction;
//  * children are NOT materialised: the reference's deep-clone + deep-equality duplicate test
//    (global_clustering.rs:118-127) becomes a 128-bit linear hash of the truncated histograms that is
//    updated incrementally; only the <= limit survivors are materialised (in place for the first child of
//    a parent, one windowed copy for each further child);
//  * the order-defining heap work runs on lane 0 over LDS; the duplicate scan is lane-parallel (ballot).
#pragma once
#include "common.h"

namespace fl {

constexpr int BEAM_TILE = 256;     // read cells staged per LDS tile
constexpr int EARLY_READS = 25;    // global_clustering.rs:50-53

struct BeamArgs {
    BlockSet bs;
    const uint32_t* job_block;     // [n_jobs] block index of job (cost-descending)
    uint32_t  n_jobs;
    uint32_t  ploidy, beam;
    uint32_t  span_max;            // max blk_span over the batch (slab addressing)
    uint32_t  n_max;               // max reads per block
    uint32_t* queue_head;
    const uint8_t* blk_done;       // stop rule already fired for this block (graph_processing.rs:198-251)
    const uint32_t* stop_at;       // speculative stages: smallest ploidy at which the stop rule is known to break (optimize_kernel.h), else null
    uint64_t* state_pool;          // [slots][nbuf][span_max][ploidy][A]
    uint64_t  state_stride;        // bytes per slot (beam_slab_kernel: the u64 slabs followed by their code bytes)
    uint32_t* hist_pool;           // [slots][hist_stride] traceback records
    uint64_t  hist_stride;
    const double* binom_tab;       // host libm table of stable_binom_cdf_p_rev(n,k), tri-indexed, n <= binom_nmax
    uint32_t  binom_nmax;
    double    eps, div_factor, cutoff;
    float     ln_eps, ln_1meps;    // ln(eps), ln(1 - eps) rounded to f32: the level-1 screen of beam_slab_kernel's pruning test
    const uint64_t *Rq1, *Rp1, *Rq2, *Rp2;   // [span_max*A] random multipliers of the linear state hash
    uint8_t*  part_out;            // [blk_read_off[n_blocks]] partition of every read of every block
    double*   job_margin;          // [n_blocks*max_ploidy] min |p_k - lse - ln(PROB_CUTOFF)| over the pruning decisions of the (block, ploidy) job
    uint32_t  max_ploidy;
    uint32_t* diag;                // [0] = count of binom evaluations beyond the table, [1] = free-list underflow
    unsigned long long* steps_done;
    unsigned long long* prof;      // [32] phase cycle counters (-DFLORIA_PROF)
    uint32_t  no_bulk;             // (tests) beam_slab_kernel: no bulk-insert shortcut, every child through the entry table and the duplicate test
    // reference-arithmetic mode (beam_kernel<A, true>, arith_kernel.h): the reads' cells in the iteration order of Frag.positions
    const uint2*    cell_ord;      // [cells of the call's contigs] {SNP, allele << 28 | weight}, every read's cells in set order
    const uint64_t* cell_ord_off;  // [n_contigs] where a contig's part of cell_ord starts
    // per-block dataflow for the last ploidy stage (beam_slab_kernel<.., SPEC = true> only): the launch runs BESIDE the optimise launch of ploidy
    // wait_tried and a job starts when that launch has decided its block — tried[b] >= wait_tried, published with release order after blk_done[b]
    // (optimize_kernel.h) — graph_processing.rs:132-252 runs a block's ploidies back to back.  0 = the launch follows the stop rule of every block.
    const uint32_t* tried;
    uint32_t  wait_tried;
    uint32_t  wait_ticks;          // give up waiting after this many ticks of the 100-MHz wall clock and run the job whether it is needed or not
};

__host__ __device__ inline uint32_t beam_hist_off(uint32_t i, uint32_t LM, uint32_t B) {
    return i < (uint32_t)EARLY_READS ? i * LM : (uint32_t)EARLY_READS * LM + (i - EARLY_READS) * B;
}

// LDS carve-up (bytes) for limit_max = LM states
struct BeamLds {
    uint32_t off_cpos, off_caw, off_crp1, off_crp2;
    uint32_t off_st[2];            // state arrays x2 (current / next)
    uint32_t off_ent, off_heap, off_efree, off_bfree, off_flag, off_ev[2], total;
};
// per-state record
struct __align__(8) StateRec { uint64_t q; uint64_t h1, h2; double score; uint32_t m; uint16_t buf; uint16_t src; uint8_t k; uint8_t pad[3]; uint32_t pad2; };
// per heap entry (a child that is currently in the next heap)
struct __align__(8) EntryRec { double score; uint64_t h1, h2; uint64_t q; uint32_t m; uint16_t parent; uint8_t k; uint8_t pad; double df; };   // df: the child's running-sum distance (ARITH)

// p_arith: ploidy when the kernel keeps the reference's per-partition running sums (error_vec of SearchNode, types_structs.rs:114-125), else 0
__host__ __device__ inline BeamLds beam_lds_layout(uint32_t LM, uint32_t p_arith = 0) {
    BeamLds L;
    uint32_t o = 0;
    L.off_cpos = o; o += BEAM_TILE * 4;
    L.off_caw = o;  o += BEAM_TILE * 4;
    L.off_crp1 = o; o += BEAM_TILE * 8;
    L.off_crp2 = o; o += BEAM_TILE * 8;
    L.off_st[0] = o; o += LM * sizeof(StateRec);
    L.off_st[1] = o; o += LM * sizeof(StateRec);
    L.off_ent = o;  o += (LM + 1) * sizeof(EntryRec);
    L.off_heap = o; o += ((LM + 1) * 2 + 7) & ~7u;
    L.off_efree = o; o += ((LM + 1) * 2 + 7) & ~7u;
    L.off_bfree = o; o += (LM * 2 + 7) & ~7u;
    L.off_flag = o; o += (LM + 7) & ~7u;
    L.off_ev[0] = o; o += LM * p_arith * 8;
    L.off_ev[1] = o; o += LM * p_arith * 8;
    L.total = o;
    return L;
}

// ---- std BinaryHeap on entry ids, ordered by EntryRec::score (ties Equal) — lane 0 only --------------------
__device__ inline void heap_push(uint16_t* hid, const EntryRec* ent, uint32_t& len, uint16_t x) {
    uint32_t pos = len++;
    const double xs = ent[x].score;
    while (pos > 0) {                                   // sift_up(0, old_len)
        uint32_t par = (pos - 1) >> 1;
        uint16_t pid = hid[par];
        if (xs <= ent[pid].score) break;
        hid[pos] = pid;
        pos = par;
    }
    hid[pos] = x;
}
__device__ inline uint16_t heap_pop(uint16_t* hid, const EntryRec* ent, uint32_t& len) {
    uint16_t item = hid[--len];                         // data.pop()
    if (len == 0) return item;
    uint16_t root = hid[0];                             // swap(item, data[0]); item <- root
    const double xs = ent[item].score;
    const uint32_t end = len;
    uint32_t pos = 0, child = 1;
    const uint32_t lim = end >= 2 ? end - 2 : 0;        // end.saturating_sub(2)
    while (child <= lim) {                              // sift_down_to_bottom(0)
        if (ent[hid[child]].score <= ent[hid[child + 1]].score) child++;
        hid[pos] = hid[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) { hid[pos] = hid[child]; pos = child; }
    while (pos > 0) {                                   // sift_up(0, pos)
        uint32_t par = (pos - 1) >> 1;
        if (xs <= ent[hid[par]].score) break;
        hid[pos] = hid[par];
        pos = par;
    }
    hid[pos] = item;
    return root;
}
// into_sorted_vec()[0] on an array of state indices ordered by score[] — returns the id at position 0
__device__ inline uint16_t heap_sorted_first(uint16_t* hid, const StateRec* st, uint32_t len) {
    uint32_t end = len;
    while (end > 1) {
        --end;
        uint16_t t = hid[0]; hid[0] = hid[end]; hid[end] = t;          // swap(0, end)
        uint16_t hole = hid[0];                                        // sift_down_range(0, end)
        const double hs = st[hole].score;
        uint32_t pos = 0, child = 1;
        const uint32_t lim = end >= 2 ? end - 2 : 0;
        bool placed = false;
        while (child <= lim) {
            if (st[hid[child]].score <= st[hid[child + 1]].score) child++;
            if (hs >= st[hid[child]].score) { hid[pos] = hole; placed = true; break; }
            hid[pos] = hid[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (!placed) {
            if (child == end - 1 && hs < st[hid[child]].score) { hid[pos] = hid[child]; pos = child; }
            hid[pos] = hole;
        }
    }
    return hid[0];
}

// ARITH: the reference's own f64 arithmetic (floria_hip_set_option("arith", 1)) — the distance of a read to a partition is the running sum
// `diff += w` / `diff += epsilon` over the read's cells in the order of its position set (utils_frags.rs:33-72), a node carries the
// running sum of every partition and its score is their sum in partition order (global_clustering.rs:196-202).  For a dyadic epsilon
// both forms give the same bits; for any other epsilon they differ in the last bits and, through `as usize`, sometimes by one.
template <int A, bool ARITH = false>
__global__ __launch_bounds__(64) void beam_kernel(BeamArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const uint32_t p = g.ploidy, B = g.beam, LM = p * B;
    const BeamLds LY = beam_lds_layout(LM, ARITH ? p : 0u);
    uint32_t* c_pos = (uint32_t*)(smem + LY.off_cpos);
    uint32_t* c_aw  = (uint32_t*)(smem + LY.off_caw);
    uint64_t* c_rp1 = (uint64_t*)(smem + LY.off_crp1);
    uint64_t* c_rp2 = (uint64_t*)(smem + LY.off_crp2);
    EntryRec* ent   = (EntryRec*)(smem + LY.off_ent);
    uint16_t* hid   = (uint16_t*)(smem + LY.off_heap);
    uint16_t* efree = (uint16_t*)(smem + LY.off_efree);
    uint16_t* bfree = (uint16_t*)(smem + LY.off_bfree);
    uint8_t*  flag  = (uint8_t*)(smem + LY.off_flag);
    __shared__ uint32_t s_heap_len, s_efree_n, s_bfree_n;

    const uint32_t PA = p * A;                                  // u64 cells per SNP position of one state
    const uint64_t state_stride = (uint64_t)g.span_max * PA;     // u64 per state slab
    uint64_t* slot_states = g.state_pool + (uint64_t)blockIdx.x * LM * state_stride;
    uint32_t* slot_hist = g.hist_pool + (uint64_t)blockIdx.x * g.hist_stride;

    const uint32_t S = 64 / p;                                   // states per lane chunk
    const uint32_t my_sl = lane / p, my_k = lane % p;            // lane <-> (state-in-chunk, partition)
    const bool lane_pair = my_sl < S;
    const uint64_t rk1 = c_rk1[my_k], rk2 = c_rk2[my_k];
    double min_margin = 1e300;
    uint32_t n_fallback = 0;

    for (;;) {
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(g.queue_head, 1u);
        job = __shfl(job, 0);
        if (job >= g.n_jobs) break;
        const uint32_t b = g.job_block[job];
        if (g.blk_done[b]) continue;
        if (g.stop_at && (uint32_t)__shfl((int)__hip_atomic_load(&g.stop_at[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0) < g.ploidy) continue;      // (speculative stages, see optimize_kernel.h; lane 0's reading for the whole wave)
        min_margin = 1e300;                                 // per (block, ploidy) job: the host keeps the jobs the stop rule reached
        const ContigDev cd = g.bs.contigs[g.bs.blk_contig[b]];
        const uint64_t roff = g.bs.blk_read_off[b];
        const uint32_t n = (uint32_t)(g.bs.blk_read_off[b + 1] - roff);
        const uint32_t* reads = g.bs.blk_read + roff;
        const uint32_t pos0 = g.bs.blk_pos0[b];
        const uint2* ord = ARITH ? g.cell_ord + g.cell_ord_off[g.bs.blk_contig[b]] : nullptr;
        double* ev = (double*)(smem + LY.off_ev[0]);       // [state][partition] running sums of the current states (ARITH)
        double* evn = (double*)(smem + LY.off_ev[1]);

        // ---- initial beam: one empty state (global_clustering.rs:31-47) -------------------------------
        int cur = 0;
        StateRec* st = (StateRec*)(smem + LY.off_st[0]);
        StateRec* nx = (StateRec*)(smem + LY.off_st[1]);
        uint32_t nstates = 1;
        if (lane == 0) {
            st[0].q = 0; st[0].m = 0; st[0].h1 = 0; st[0].h2 = 0; st[0].score = 0.0; st[0].buf = 0; st[0].src = 0xffff; st[0].k = 0;
            s_bfree_n = 0;
            if (ARITH) for (uint32_t k = 0; k < p; ++k) ev[k] = 0.0;
            for (uint32_t i = LM - 1; i >= 1; --i) bfree[s_bfree_n++] = (uint16_t)i;   // stack: pops 1,2,3..
        }
        int32_t hi_rel = -1;            // highest position (relative to pos0) written in any live slab
        uint32_t start_rel = 0;         // hash window lower bound (= first_position of the last read)
        __syncthreads();

        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t r = reads[i];
            const uint32_t cbeg = G(cd.read_off)[r], L = G(cd.read_off)[r + 1] - cbeg;
            const uint32_t first_rel = G(cd.first)[r] - pos0;
            const int32_t  last_rel = (int32_t)(G(cd.last)[r] - pos0);
            const uint32_t limit = i < (uint32_t)EARLY_READS ? LM : B;                 // :50-53
            const uint32_t ntiles = (L + BEAM_TILE - 1) / BEAM_TILE;

            // ---- per-read hash constant Tw = sum_cells Rq[pos,allele] * w (wave-uniform) ----------------
            uint64_t tw1 = 0, tw2 = 0;
            for (uint32_t c = lane; c < L; c += 64) {
                uint32_t pr = G(cd.cell_snp)[cbeg + c] - pos0;
                uint32_t aq = G(cd.cell_aw)[cbeg + c];
                uint64_t w = (aq & 0x0fffffffu);
                uint32_t idx = pr * A + (aq >> 28);
                tw1 += g.Rq1[idx] * w;
                tw2 += g.Rq2[idx] * w;
            }
            tw1 = wave_sum_u64(tw1);
            tw2 = wave_sum_u64(tw2);

            if (lane == 0) {
                s_heap_len = 0;
                s_efree_n = 0;
                for (int e = (int)limit; e >= 0; --e) efree[s_efree_n++] = (uint16_t)e;
            }
            auto stage_tile = [&](uint32_t t) {
                __syncthreads();
                for (uint32_t c = lane; c < BEAM_TILE; c += 64) {
                    uint32_t cc = t * BEAM_TILE + c;
                    if (cc < L) {
                        uint32_t pr, aq;
                        if (ARITH) { const uint2 ca = ord[cbeg + cc]; pr = ca.x - pos0; aq = ca.y; }          // ARITH: the lanes walk the cells in the set's order
                        else { pr = G(cd.cell_snp)[cbeg + cc] - pos0; aq = G(cd.cell_aw)[cbeg + cc]; }
                        uint32_t al = aq >> 28;
                        c_pos[c] = pr;
                        c_aw[c] = (al << 28) | (aq & 0x0fffffffu);
                        c_rp1[c] = g.Rp1[pr * A + al];
                        c_rp2[c] = g.Rp2[pr * A + al];
                    }
                }
                __syncthreads();
            };
            if (ntiles == 1) stage_tile(0);

            // ---- expand every current state, in heap-array order (:71) --------------------------------------
            for (uint32_t a0 = 0; a0 < nstates; a0 += S) {
                const uint32_t a = a0 + my_sl;
                const bool act = lane_pair && a < nstates;
                uint64_t qs = 0, qd = 0, np1 = 0, np2 = 0, t1 = 0, t2 = 0;
                uint32_t m = 0;
                double df = 0.0;                                  // ARITH: the running `diff`
                const uint64_t* base = slot_states + (uint64_t)(act ? st[a].buf : 0) * state_stride + my_k * A;
                // hash of the positions that leave the window: [start_rel, first_rel) ∩ [.., hi_rel]
                {
                    int32_t tend = (int32_t)first_rel - 1 < hi_rel ? (int32_t)first_rel - 1 : hi_rel;
                    for (int32_t pr = (int32_t)start_rel; pr <= tend; ++pr) {
                        if (act) {
#pragma unroll
                            for (int al = 0; al < A; ++al) {
                                uint64_t v = base[(uint64_t)pr * PA + al];
                                if (v) {
                                    t1 += g.Rq1[pr * A + al] * (v & QMASK63) + g.Rp1[pr * A + al];
                                    t2 += g.Rq2[pr * A + al] * (v & QMASK63) + g.Rp2[pr * A + al];
                                }
                            }
                        }
                    }
                    t1 *= rk1;      // H(state) = sum_k Rk[k] * sum_{pos,allele} (Rq*Q + Rp*present)
                    t2 *= rk2;
                }
                // distance_read_haplo_epsilon_empty (utils_frags.rs:32-75) of the read vs (state a, partition k)
                for (uint32_t t = 0; t < ntiles; ++t) {
                    if (ntiles > 1) stage_tile(t);
                    const uint32_t tl = min((uint32_t)BEAM_TILE, L - t * BEAM_TILE);
                    if (act) {
                        // cells in batches of CU: the histogram pieces of a batch are requested together (one memory round trip per batch instead of one
                        // per cell), then classified one by one in the cells' order (ARITH: the running sum's order)
                        constexpr int CU = A == 2 ? 8 : 4;
                        for (uint32_t c0 = 0; c0 < tl; c0 += CU) {
                            uint32_t prs[CU], aws[CU];
                            uint64_t vs[CU][A];
#pragma unroll
                            for (int u = 0; u < CU; ++u) {
                                const uint32_t c = c0 + u < tl ? c0 + u : tl - 1;
                                prs[u] = c_pos[c]; aws[u] = c_aw[c];
                                const bool inw = (int32_t)prs[u] <= hi_rel;
                                const uint64_t* cp = base + (uint64_t)(inw ? prs[u] : 0u) * PA;
#pragma unroll
                                for (int x = 0; x < A; x += 2) { ulonglong2 vv = *(const ulonglong2*)(cp + x); vs[u][x] = vv.x; vs[u][x + 1] = vv.y; }
                            }
#pragma unroll
                            for (int u = 0; u < CU; ++u) {
                                if (c0 + u >= tl) break;
                                const uint32_t c = c0 + u, pr = prs[u], aw = aws[u];
                                const uint32_t al = aw >> 28;
                                const uint64_t w = aw & 0x0fffffffu;
                                if ((int32_t)pr > hi_rel) {               // nothing written there yet: empty position
                                    m += 1; np1 += c_rp1[c]; np2 += c_rp2[c];
                                    if (ARITH) df += g.eps;
                                    continue;
                                }
                                uint64_t mx = 0, va = 0;
#pragma unroll
                                for (int x = 0; x < A; ++x) { uint64_t qx = vs[u][x] & QMASK63; mx = qx > mx ? qx : mx; va = (x == (int)al) ? vs[u][x] : va; }
                                if (mx == 0) { m += 1; if (ARITH) df += g.eps; }       // :45-48  diff += epsilon
                                else if ((va & QMASK63) == mx) qs += w;                // :58-68  same
                                else { qd += w; if (ARITH) df += (double)w * 0x1p-24; }   // :70     diff
                                if (!(va >> 63)) { np1 += c_rp1[c]; np2 += c_rp2[c]; }
                            }
                        }
                    }
                }
                // p-value (:77-91) — stable_binom_cdf_p_rev of truncated (n, k)
                double pv = 0.0;
                if (act) {
                    const double same_f = qm_to_f64(qs, 0, g.eps), diff_f = ARITH ? df : qm_to_f64(qd, m, g.eps);
                    const uint64_t nn = (uint64_t)(same_f + diff_f), kk = (uint64_t)diff_f;
                    if (nn <= g.binom_nmax) pv = g.binom_tab[nn * (nn + 1) / 2 + kk];
                    else { pv = binom_device(nn, kk, g.eps, g.div_factor); n_fallback++; }
                    pv = 1.0 * pv;
                }
                // log_sum_exp over the p partitions of the state (:93), segment = lanes [sl*p, sl*p+p)
                const int seg0 = (int)(my_sl * p);
                double mx = 0.0, sum = 0.0;
                uint64_t ts1 = 0, ts2 = 0;
                for (uint32_t j = 0; j < p; ++j) {
                    double o = shfl_f64(pv, seg0 + (int)j);
                    mx = (j == 0) ? o : (o > mx ? o : mx);
                    ts1 += shfl_u64(t1, seg0 + (int)j);
                    ts2 += shfl_u64(t2, seg0 + (int)j);
                }
                const double ex = exp(pv - mx);              // own term once; summed in the reference's order j = 0..p-1
                for (uint32_t j = 0; j < p; ++j) sum += shfl_f64(ex, seg0 + (int)j);
                const double lse = mx + log(sum);
                bool pass = false;
                uint64_t ch1 = 0, ch2 = 0, cq = 0;
                uint32_t cm = 0;
                double cscore = 0.0;
                if (act) {
                    const double margin = (pv - lse) - g.cutoff;
                    const double am = fabs(margin);
                    min_margin = am < min_margin ? am : min_margin;
                    pass = (pv - lse) > g.cutoff;                                   // :98
                    // child: read_to_node_value (:181-208) + hash of build_truncated_hap_block (:326-376)
                    cq = st[a].q + qd;
                    cm = st[a].m + m;
                    cscore = qm_to_f64(cq, cm, g.eps);
                    if (ARITH) {                                                    // :198-202: error_vec[k].1 (+ diff for the read's partition), summed in partition order
                        double mec = 0.0;
                        for (uint32_t k = 0; k < p; ++k) { const double e = ev[a * p + k]; mec += (k == my_k) ? e + df : e; }
                        cscore = mec;
                    }
                    ch1 = (st[a].h1 - ts1) + rk1 * (tw1 + np1);
                    ch2 = (st[a].h2 - ts2) + rk2 * (tw2 + np2);
                }
                uint64_t passmask = __ballot(pass);
                // ---- children in (state, partition) order through the duplicate test and the heap -------------
                while (passmask) {
                    const int src = __ffsll((unsigned long long)passmask) - 1;
                    passmask &= passmask - 1;
                    const double s_score = shfl_f64(cscore, src);
                    const uint64_t s_h1 = shfl_u64(ch1, src), s_h2 = shfl_u64(ch2, src);
                    const uint64_t s_q = shfl_u64(cq, src);
                    const uint32_t s_m = __shfl(cm, src);
                    const double s_df = shfl_f64(df, src);
                    const uint32_t s_a = a0 + (uint32_t)src / p, s_k = (uint32_t)src % p;
                    const uint32_t hl = s_heap_len;
                    bool dup = false;                                               // :122-127
                    for (uint32_t e = lane; e < hl; e += 64) {
                        const EntryRec& E = ent[hid[e]];
                        dup |= (E.h1 == s_h1 && E.h2 == s_h2 && E.score >= s_score);
                    }
                    if (!__any(dup)) {
                        if (lane == 0) {
                            uint16_t id = efree[--s_efree_n];
                            EntryRec& E = ent[id];
                            E.score = s_score; E.h1 = s_h1; E.h2 = s_h2; E.q = s_q; E.m = s_m; E.parent = (uint16_t)s_a; E.k = (uint8_t)s_k; E.df = s_df;
                            uint32_t len = s_heap_len;
                            heap_push(hid, ent, len, id);                           // :130
                            if (len > limit) efree[s_efree_n++] = heap_pop(hid, ent, len);   // :132-134
                            s_heap_len = len;
                        }
                    }
                    __syncthreads();
                }
            }

            // ---- materialise the survivors ---------------------------------------------------------------
            const uint32_t nnext = s_heap_len;
            const int32_t new_hi = last_rel > hi_rel ? last_rel : hi_rel;
            for (uint32_t a = lane; a < nstates; a += 64) flag[a] = 0;
            __syncthreads();
            for (uint32_t e = lane; e < nnext; e += 64) flag[ent[hid[e]].parent] = 1;     // parents with >= 1 survivor
            __syncthreads();
            if (lane == 0) {
                uint32_t bn = s_bfree_n;
                for (uint32_t a = 0; a < nstates; ++a) if (!flag[a]) bfree[bn++] = st[a].buf;      // dead parents release
                for (uint32_t e = 0; e < nnext; ++e) {
                    const EntryRec& E = ent[hid[e]];
                    StateRec& N = nx[e];
                    N.q = E.q; N.m = E.m; N.h1 = E.h1; N.h2 = E.h2; N.score = E.score; N.k = E.k;
                    if (flag[E.parent] == 1) { flag[E.parent] = 2; N.buf = st[E.parent].buf; N.src = 0xffff; }   // in place
                    else {
                        if (bn == 0) { atomicAdd(&g.diag[1], 1u); N.buf = st[E.parent].buf; N.src = 0xffff; }
                        else { N.buf = bfree[--bn]; N.src = st[E.parent].buf; }
                    }
                }
                s_bfree_n = bn;
            }
            if (ARITH)                                               // the survivors' running sums: the parent's, with the read's distance added to its partition
                for (uint32_t x = lane; x < nnext * p; x += 64) { const uint32_t e = x / p, k = x - e * p; const EntryRec& E = ent[hid[e]]; const double pe = ev[(uint32_t)E.parent * p + k]; evn[x] = (k == E.k) ? pe + E.df : pe; }
            // traceback record of every survivor: (index of parent in the current array, partition)
            {
                uint32_t* hrow = slot_hist + beam_hist_off(i, LM, B);
                for (uint32_t e = lane; e < nnext; e += 64) { const EntryRec& E = ent[hid[e]]; hrow[e] = (uint32_t)E.parent | ((uint32_t)E.k << 16); }
            }
            // zero the newly reached positions (hi_rel, new_hi] of every current slab
            if (new_hi > hi_rel) {
                const uint32_t cnt = (uint32_t)(new_hi - hi_rel) * PA;
                for (uint32_t a = 0; a < nstates; ++a) {
                    uint64_t* d = slot_states + (uint64_t)st[a].buf * state_stride + (uint64_t)(hi_rel + 1) * PA;
                    for (uint32_t x = lane; x < cnt; x += 64) d[x] = 0;
                }
            }
            __syncthreads();
            // copies: further children of a parent get the parent's live window [first_rel, new_hi]
            {
                const uint32_t cnt2 = ((uint32_t)(new_hi - (int32_t)first_rel + 1) * PA) >> 1;     // ulonglong2 units (A even)
                for (uint32_t e = 0; e < nnext; ++e) {
                    const uint16_t src = nx[e].src;
                    if (src == 0xffff) continue;
                    const ulonglong2* s = (const ulonglong2*)(slot_states + (uint64_t)src * state_stride + (uint64_t)first_rel * PA);
                    ulonglong2* d = (ulonglong2*)(slot_states + (uint64_t)nx[e].buf * state_stride + (uint64_t)first_rel * PA);
                    for (uint32_t x = lane; x < cnt2; x += 64) d[x] = s[x];
                }
            }
            __syncthreads();
            // add the read to partition k of every survivor (types_structs.rs:368-373)
            for (uint32_t t = 0; t < ntiles; ++t) {
                if (ntiles > 1) stage_tile(t);
                const uint32_t tl = min((uint32_t)BEAM_TILE, L - t * BEAM_TILE);
                const uint32_t items = nnext * tl;
                for (uint32_t x = lane; x < items; x += 64) {
                    const uint32_t e = x / tl, c = x - e * tl;
                    const uint32_t aw = c_aw[c];
                    uint64_t* cp = slot_states + (uint64_t)nx[e].buf * state_stride + (uint64_t)c_pos[c] * PA + nx[e].k * A + (aw >> 28);
                    *cp = (*cp + (uint64_t)(aw & 0x0fffffffu)) | PRESENT_BIT;
                }
            }
            __syncthreads();
            // swap current/next
            cur ^= 1;
            st = (StateRec*)(smem + (cur ? LY.off_st[1] : LY.off_st[0]));
            nx = (StateRec*)(smem + (cur ? LY.off_st[0] : LY.off_st[1]));
            { double* t = ev; ev = evn; evn = t; }
            nstates = nnext;
            hi_rel = new_hi;
            start_rel = first_rel;
        }

        // ---- into_sorted_vec()[0] (:149-150) and traceback (:155-176) --------------------------------------------
        if (lane == 0) {
            for (uint32_t e = 0; e < nstates; ++e) hid[e] = (uint16_t)e;
            uint32_t ecur = heap_sorted_first(hid, st, nstates);
            uint8_t* out = g.part_out + roff;
            for (int32_t i = (int32_t)n - 1; i >= 0; --i) {
                uint32_t rec = slot_hist[beam_hist_off((uint32_t)i, LM, B) + ecur];
                out[i] = (uint8_t)(rec >> 16);
                ecur = rec & 0xffff;
            }
            atomicAdd(g.steps_done, (unsigned long long)n);
        }
        {
            const double jm = wave_min_f64(min_margin);
            if (lane == 0) g.job_margin[(uint64_t)b * g.max_ploidy + g.ploidy - 1] = jm;
        }
        __syncthreads();
    }
    n_fallback = wave_sum_u32(n_fallback);
    if (lane == 0) {
        if (n_fallback) atomicAdd(&g.diag[0], n_fallback);
    }
}

}  // namespace fl
