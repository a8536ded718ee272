// beam_wide_kernel.h — shared partition slabs for WIDE beams (ploidy*beam > 63, e.g. BASELINE config 5: -p 8 -n 40).
//
// Same state representation and step structure as beam_slab_kernel.h (a state = p slab ids; the allele-agreement
// loop runs once per live slab; new slab versions in place or after a window copy), but the bounded heap, the
// entry table and the survivor bookkeeping no longer fit one wavefront's registers: the std::BinaryHeap emulation
// of beam_kernel.h (LDS, lane 0) is used, survivors are walked in chunks of 64, and the per-slab result tables
// (up to ploidy^2*beam entries) live in the slot's HBM scratch.  Results are bit-identical to the other beam kernels.
#pragma once
#include "beam_slab_kernel.h"

namespace fl {

constexpr int WIDE_NS_MAX = 8192;

struct WideLds {
    uint32_t off_coff, off_caw, off_crp1, off_crp2;
    uint32_t off_q[2], off_h1[2], off_h2[2], off_m[2], off_sl[2];
    uint32_t off_live, off_s2l, off_ref, off_leader, off_newid;
    uint32_t off_ent, off_heap, off_efree, off_e2j;
    uint32_t off_pk, off_u, off_flag, off_copy, off_fslab, off_lead;
    uint32_t total;
};
__host__ __device__ inline WideLds wide_lds_layout(uint32_t LM, uint32_t p, bool q0) {
    WideLds L;
    const uint32_t NS = LM * p;
    uint32_t o = 0;
    auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15) & ~15u; return r; };
    L.off_coff = take(SLAB_TILE * 4); L.off_caw = take(SLAB_TILE * 4);
    L.off_crp1 = take(q0 ? SLAB_TILE * 8 : 0); L.off_crp2 = take(q0 ? SLAB_TILE * 8 : 0);
    for (int i = 0; i < 2; ++i) {
        L.off_q[i] = take(LM * 8); L.off_h1[i] = take(LM * 8); L.off_h2[i] = take(LM * 8); L.off_m[i] = take(LM * 4);
        L.off_sl[i] = take(NS * 2);
    }
    L.off_live = take(NS * 2); L.off_s2l = take(NS * 2); L.off_ref = take(NS); L.off_leader = take(NS * 4); L.off_newid = take(NS * 2);
    L.off_ent = take((LM + 1) * sizeof(EntryRec)); L.off_heap = take((LM + 1) * 2); L.off_efree = take((LM + 1) * 2); L.off_e2j = take((LM + 1) * 2);
    L.off_pk = take(LM * 4); L.off_u = take(LM * 2); L.off_flag = take(LM); L.off_copy = take(LM * 2); L.off_fslab = take(LM * 2); L.off_lead = take(LM * 2);
    L.total = o;
    return L;
}
// u32 words the kernel needs behind a slot's traceback records (per-slab result tables)
// (arith: + the states' error_vec, two parities of LM * p f64, behind the tables)
__host__ __device__ inline uint64_t wide_scratch_words(uint32_t LM, uint32_t p, bool q0, bool arith = false) { return (uint64_t)LM * p * ((q0 ? 22 : 14) + (arith ? 4 : 0)); }

__device__ inline uint16_t wide_sorted_first(uint16_t* hid, const EntryRec* ent, uint32_t len) {      // into_sorted_vec()[0]
    uint32_t end = len;
    while (end > 1) {
        --end;
        uint16_t t = hid[0]; hid[0] = hid[end]; hid[end] = t;
        const uint16_t hole = hid[0];
        const double hs = ent[hole].score;
        uint32_t pos = 0, child = 1;
        const uint32_t lim = end >= 2 ? end - 2 : 0;
        bool placed = false;
        while (child <= lim) {
            if (ent[hid[child]].score <= ent[hid[child + 1]].score) child++;
            if (hs >= ent[hid[child]].score) { hid[pos] = hole; placed = true; break; }
            hid[pos] = hid[child];
            pos = child;
            child = 2 * pos + 1;
        }
        if (!placed) {
            if (child == end - 1 && hs < ent[hid[child]].score) { hid[pos] = hid[child]; pos = child; }
            hid[pos] = hole;
        }
    }
    return hid[0];
}

// ARITH: the reference's own f64 arithmetic (floria_hip_set_option("arith", 1)) for wide beams (round 6; until then such beams took the generic kernel in this
// mode).  A read's cells are staged in the iteration order of Frag.positions (BeamArgs::cell_ord); phase A gives every live slab ONE lane, which walks the cells in that
// order and adds `diff += w | epsilon` term by term (utils_frags.rs:32-75) - a wide beam has dozens to hundreds of live slabs, so the lanes are busy anyway and the running
// sum needs no staging; a state carries error_vec (global_clustering.rs:196-202) as p f64 in the slot's HBM scratch, a child's score is their sum in partition order with
// the read's diff added to its partition first.  Slabs, hash, heap, traceback: unchanged.
template <int A, bool Q0, bool ARITH = false>
__global__ __launch_bounds__(64) void beam_wide_kernel(BeamArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_heap_len, s_efree_n;
    const uint32_t lane = threadIdx.x;
    const uint32_t p = g.ploidy, B = g.beam, LM = p * B, NS = LM * p;
    const WideLds LY = wide_lds_layout(LM, p, Q0);
    uint32_t* c_off = (uint32_t*)(smem + LY.off_coff);
    uint32_t* c_aw  = (uint32_t*)(smem + LY.off_caw);
    uint64_t* c_rp1 = (uint64_t*)(smem + LY.off_crp1);
    uint64_t* c_rp2 = (uint64_t*)(smem + LY.off_crp2);
    uint16_t* live_id = (uint16_t*)(smem + LY.off_live);
    uint16_t* s2l = (uint16_t*)(smem + LY.off_s2l);
    uint8_t*  ref = (uint8_t*)(smem + LY.off_ref);
    uint32_t* leader = (uint32_t*)(smem + LY.off_leader);
    uint16_t* newid = (uint16_t*)(smem + LY.off_newid);
    EntryRec* ent = (EntryRec*)(smem + LY.off_ent);
    uint16_t* hid = (uint16_t*)(smem + LY.off_heap);
    uint16_t* efree = (uint16_t*)(smem + LY.off_efree);
    uint16_t* e2j = (uint16_t*)(smem + LY.off_e2j);
    uint32_t* s_pk = (uint32_t*)(smem + LY.off_pk);
    uint16_t* s_u = (uint16_t*)(smem + LY.off_u);
    uint8_t*  s_flag = (uint8_t*)(smem + LY.off_flag);
    uint16_t* copy_list = (uint16_t*)(smem + LY.off_copy);
    uint16_t* freeslab = (uint16_t*)(smem + LY.off_fslab);
    uint16_t* lead_list = (uint16_t*)(smem + LY.off_lead);

    const uint32_t pos_bytes = A * 8;
    const uint32_t slab_bytes = g.span_max * pos_bytes;
    char* pool = (char*)(g.state_pool + (uint64_t)blockIdx.x * ((uint64_t)LM * g.span_max * p * A));
    uint32_t* slot_hist = g.hist_pool + (uint64_t)blockIdx.x * g.hist_stride;
    uint64_t* r_qs = (uint64_t*)(slot_hist + (g.hist_stride - wide_scratch_words(LM, p, Q0, ARITH)));
    uint64_t* r_qd = r_qs + NS; uint64_t* r_t1 = r_qd + NS; uint64_t* r_t2 = r_t1 + NS;
    uint64_t* r_np1 = r_t2 + NS; uint64_t* r_np2 = r_np1 + (Q0 ? NS : 0);
    uint32_t* r_m = (uint32_t*)(r_np2 + (Q0 ? NS : 0));
    double* const r_fd = (double*)r_qd;                                  // ARITH: the read's running diff against every live slab
    double* const ev_base = (double*)(r_qs + (uint64_t)NS * (Q0 ? 11 : 7));      // ARITH: error_vec of the current / next states, [2][LM * p]
    const uint64_t lane_lt = (1ull << lane) - 1;

    const uint32_t S = 64 / p;
    const uint32_t my_sl = lane / p, my_k = lane % p;
    const bool lane_pair = my_sl < S;
    const uint64_t rk1 = c_rk1[my_k], rk2 = c_rk2[my_k];
    const int seg0 = (int)(my_sl * p);
    double min_margin = 1e300;
    uint32_t n_fallback = 0;

    for (;;) {
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(g.queue_head, 1u);
        job = uni(__shfl(job, 0));
        if (job >= g.n_jobs) break;
        const uint32_t b = uni(g.job_block[job]);
        if (g.blk_done[b]) continue;
        if (g.stop_at && (uint32_t)__shfl((int)__hip_atomic_load(&g.stop_at[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0) < g.ploidy) continue;      // (speculative stages, see optimize_kernel.h; lane 0's reading for the whole wave)
        min_margin = 1e300;                                 // per (block, ploidy) job: the host keeps the jobs the stop rule reached
        const ContigDev cd = g.bs.contigs[g.bs.blk_contig[b]];
        const uint64_t roff = g.bs.blk_read_off[b];
        const uint32_t n = (uint32_t)(g.bs.blk_read_off[b + 1] - roff);
        const uint32_t* reads = g.bs.blk_read + roff;
        const uint32_t pos0 = g.bs.blk_pos0[b];
        const uint2* const ord = ARITH ? g.cell_ord + g.cell_ord_off[g.bs.blk_contig[b]] : nullptr;     // the contig's cells, every read's in set order

        int cur = 0;
        auto ST_ev = [&](int w) { return ev_base + (w ? NS : 0u); };
        auto ST_q = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_q[1] : LY.off_q[0])); };
        auto ST_h1 = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_h1[1] : LY.off_h1[0])); };
        auto ST_h2 = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_h2[1] : LY.off_h2[0])); };
        auto ST_m = [&](int w) { return (uint32_t*)(smem + (w ? LY.off_m[1] : LY.off_m[0])); };
        auto ST_sl = [&](int w) { return (uint16_t*)(smem + (w ? LY.off_sl[1] : LY.off_sl[0])); };
        uint32_t nstates = 1, nlive = 1;
        if (lane == 0) { ST_q(0)[0] = 0; ST_h1(0)[0] = 0; ST_h2(0)[0] = 0; ST_m(0)[0] = 0; live_id[0] = 0; s2l[0] = 0; }
        if (lane < p) ST_sl(0)[lane] = 0;
        if (ARITH && lane < p) ST_ev(0)[lane] = 0.0;                  // error_vec: vec![(0.0, 0.0); ploidy] (global_clustering.rs:37)
        int32_t hi_rel = -1;
        uint32_t start_rel = 0;
        __syncthreads();

        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t r = sload(reads + i);
            const uint32_t cbeg = sload(cd.read_off + r), L = sload(cd.read_off + r + 1) - cbeg;
            const uint32_t first_rel = sload(cd.first + r) - pos0;
            const int32_t  last_rel = (int32_t)(sload(cd.last + r) - pos0);
            const uint64_t tw1 = sload(cd.tw + 2 * (uint64_t)r), tw2 = sload(cd.tw + 2 * (uint64_t)r + 1);
            const uint32_t limit = i < (uint32_t)EARLY_READS ? LM : B;
            const uint32_t ntiles = (L + SLAB_TILE - 1) / SLAB_TILE;
            uint64_t* st_q = ST_q(cur); uint64_t* st_h1 = ST_h1(cur); uint64_t* st_h2 = ST_h2(cur);
            uint32_t* st_m = ST_m(cur); uint16_t* st_sl = ST_sl(cur);

            uint64_t rpb1 = 0, rpb2 = 0;
            uint32_t nin = 0;
            auto stage_tile = [&](uint32_t t) {
                __syncthreads();
                uint32_t cnt_in = 0;
                uint64_t b1 = 0, b2 = 0;
#pragma unroll
                for (int u = 0; u < SLAB_TILE / 64; ++u) {
                    const uint32_t c = lane + 64 * u;
                    const uint32_t cc = t * SLAB_TILE + c;
                    bool in = false;
                    if (cc < L) {
                        uint32_t snp, aq;
                        if constexpr (ARITH) { const uint64_t ca = G((const uint64_t*)ord)[cbeg + cc]; snp = (uint32_t)ca; aq = (uint32_t)(ca >> 32); }      // (set order: the cells inside the written window are not a prefix)
                        else { snp = G(cd.cell_snp)[cbeg + cc]; aq = G(cd.cell_aw)[cbeg + cc]; }
                        const uint32_t pr = snp - pos0, al = aq >> 28;
                        c_off[c] = pr * pos_bytes;
                        in = (int32_t)pr <= hi_rel;
                        c_aw[c] = (al << 28) | (aq & 0x0fffffffu) | ((ARITH && in) ? 0x80000000u : 0u);       // (ARITH: bit 31 = the position lies inside the written window)
                        if (Q0) {
                            const uint64_t r1 = g.Rp1[hash_idx(snp, al)], r2 = g.Rp2[hash_idx(snp, al)];
                            c_rp1[c] = r1; c_rp2[c] = r2;
                            if (!in) { b1 += r1; b2 += r2; }
                        }
                    }
                    cnt_in += (uint32_t)__popcll(__ballot(in));
                }
                nin = uni(cnt_in);
                if (Q0) { rpb1 = wave_sum_u64(b1); rpb2 = wave_sum_u64(b2); }
                __syncthreads();
            };
            if (ntiles == 1) stage_tile(0);

            // ---- A: read vs every live slab --------------------------------------------------------------------------
            uint32_t Gs = 1;
            while (!ARITH && Gs < 16 && nlive * (Gs * 2) <= 64) Gs *= 2;          // (ARITH: one lane per slab - the running sum is sequential in the cells)
            const int32_t tend = (int32_t)first_rel - 1 < hi_rel ? (int32_t)first_rel - 1 : hi_rel;
            const bool trunc = tend >= (int32_t)start_rel;
            const uint32_t per = 64 / Gs;
            for (uint32_t l0 = 0; l0 < nlive; l0 += per) {
                const uint32_t li = l0 + lane / Gs, sub = lane % Gs;
                const bool act = li < nlive;
                const uint32_t slab_off = act ? (uint32_t)live_id[li] * slab_bytes : 0;
                uint64_t qs = 0, qd = 0, np1 = 0, np2 = 0, t1 = 0, t2 = 0;
                uint32_t m = 0;
                double df = 0.0;                                   // ARITH: the running `diff` of utils_frags.rs:32-75 (continued across the tiles of a long read)
                for (int32_t pr = (int32_t)start_rel + (int32_t)sub; pr <= tend; pr += (int32_t)Gs) {
                    if (act) {
#pragma unroll
                        for (int al = 0; al < A; ++al) {
                            const uint64_t v = *(const uint64_t*)(pool + (slab_off + (uint32_t)pr * pos_bytes + al * 8));
                            if (v) {
                                const uint64_t qv = Q0 ? (v & QMASK63) : v;
                                const uint32_t hx = hash_idx(pos0 + (uint32_t)pr, (uint32_t)al);
                                t1 += g.Rq1[hx] * qv; t2 += g.Rq2[hx] * qv;
                                if (Q0) { t1 += g.Rp1[hx]; t2 += g.Rp2[hx]; }
                            }
                        }
                    }
                }
                uint32_t ps = 0, pd = 0;
                auto cell = [&](const ulonglong2* vv, uint32_t aw, uint32_t c, bool valid) {
                    const uint32_t al = aw >> 28;
                    const uint32_t w = aw & 0x0fffffffu;
                    uint64_t v[A];
#pragma unroll
                    for (int x = 0; x < A; x += 2) { v[x] = vv[x / 2].x; v[x + 1] = vv[x / 2].y; }
                    uint64_t mx = 0, va = 0;
#pragma unroll
                    for (int x = 0; x < A; ++x) { const uint64_t qx = Q0 ? (v[x] & QMASK63) : v[x]; mx = qx > mx ? qx : mx; va = (x == (int)al) ? v[x] : va; }
                    const bool nonempty = mx != 0;
                    const bool same = (Q0 ? (va & QMASK63) : va) == mx;
                    ps += (nonempty && same) ? w : 0u;
                    pd += (nonempty && !same) ? w : 0u;
                    m += (valid && !nonempty) ? 1u : 0u;
                    if (Q0) { const bool np = valid && !(va >> 63); np1 += np ? c_rp1[c] : 0ull; np2 += np ? c_rp2[c] : 0ull; }
                };
                for (uint32_t t = 0; t < ntiles; ++t) {
                    if (ntiles > 1) stage_tile(t);
                    const uint32_t tl = min((uint32_t)SLAB_TILE, L - t * SLAB_TILE);
                    if constexpr (ARITH) {
                        if (act)
                        for (uint32_t c0 = 0; c0 < tl; c0 += 4) {
                            uint32_t aws[4]; bool vs[4], ins[4];
                            ulonglong2 vv[4][A / 2];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const uint32_t c = c0 + u; vs[u] = c < tl; const uint32_t cx = vs[u] ? c : 0;
                                aws[u] = c_aw[cx]; ins[u] = vs[u] && (aws[u] >> 31);
                                const char* cp = pool + (slab_off + (ins[u] ? c_off[cx] : 0u));
#pragma unroll
                                for (int x = 0; x < A / 2; ++x) vv[u][x] = *(const ulonglong2*)(cp + 16 * x);
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (!vs[u]) break;
                                const uint32_t al = (aws[u] >> 28) & 3u, w = aws[u] & 0x0fffffffu;
                                uint64_t v[A], mx = 0, va = 0;
#pragma unroll
                                for (int x = 0; x < A; x += 2) { v[x] = ins[u] ? vv[u][x / 2].x : 0ull; v[x + 1] = ins[u] ? vv[u][x / 2].y : 0ull; }      // beyond the written window: not in the haplotype (:36-48)
#pragma unroll
                                for (int x = 0; x < A; ++x) { const uint64_t qx = Q0 ? (v[x] & QMASK63) : v[x]; mx = qx > mx ? qx : mx; va = (x == (int)al) ? v[x] : va; }
                                const bool nonempty = mx != 0, same = nonempty && (Q0 ? (va & QMASK63) : va) == mx;
                                if (!nonempty) df += g.eps;                                   // :45-48 diff += epsilon
                                else if (same) qs += w;                                       // :54-67 same += w
                                else df += (double)w * 0x1p-24;                               // :70    diff += w
                                if (Q0) { const bool np = !(va >> 63); np1 += np ? c_rp1[c0 + u] : 0ull; np2 += np ? c_rp2[c0 + u] : 0ull; }
                            }
                        }
                    } else
                    if (act) {
                        for (uint32_t c0 = sub; c0 < nin; c0 += 4 * Gs) {
                            uint32_t offs[4], aws[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) { const uint32_t c = c0 + u * Gs; const bool v = c < nin; const uint32_t cx = v ? c : 0; offs[u] = c_off[cx]; aws[u] = v ? c_aw[cx] : 0; }
                            ulonglong2 vv[4][A / 2];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const char* cp = pool + (slab_off + offs[u]);
#pragma unroll
                                for (int x = 0; x < A / 2; ++x) vv[u][x] = *(const ulonglong2*)(cp + 16 * x);
                            }
                            ps = 0; pd = 0;
#pragma unroll
                            for (int u = 0; u < 4; ++u) cell(vv[u], aws[u], c0 + u * Gs, c0 + u * Gs < nin);
                            qs += ps; qd += pd;
                        }
                        if (sub == 0) { m += tl - nin; if (Q0) { np1 += rpb1; np2 += rpb2; } }
                    }
                }
                if constexpr (ARITH) { if (act) r_fd[li] = df; }
                qs = seg_sum_u64(qs, Gs); qd = seg_sum_u64(qd, Gs); m = seg_sum_u32(m, Gs);
                if (trunc) { t1 = seg_sum_u64(t1, Gs); t2 = seg_sum_u64(t2, Gs); }
                if (Q0) { np1 = seg_sum_u64(np1, Gs); np2 = seg_sum_u64(np2, Gs); }
                if (act && sub == 0) {
                    r_qs[li] = qs; if (!ARITH) { r_qd[li] = qd; r_m[li] = m; }
                    if (trunc) { r_t1[li] = t1; r_t2[li] = t2; }
                    if (Q0) { r_np1[li] = np1; r_np2[li] = np2; }
                }
            }
            if (lane == 0) { s_heap_len = 0; s_efree_n = 0; for (int e = (int)limit; e >= 0; --e) efree[s_efree_n++] = (uint16_t)e; }
            __syncthreads();

            // ---- B: pairs, pruning, children through the LDS heap (beam_kernel.h) ------------------------------------------
            for (uint32_t a0 = 0; a0 < nstates; a0 += S) {
                const uint32_t a = a0 + my_sl;
                const bool act = lane_pair && a < nstates;
                uint64_t qd = 0, t1 = 0, t2 = 0, np1 = 0, np2 = 0;
                uint32_t m = 0;
                double pv = 0.0, df = 0.0, e_own = 0.0;
                if (act) {
                    const uint32_t li = s2l[st_sl[a * p + my_k]];
                    const uint64_t qs = r_qs[li];
                    if constexpr (ARITH) { df = r_fd[li]; e_own = ST_ev(cur)[a * p + my_k]; }
                    else { qd = r_qd[li]; m = r_m[li]; }
                    if (trunc) { t1 = r_t1[li] * rk1; t2 = r_t2[li] * rk2; }
                    if (Q0) { np1 = r_np1[li]; np2 = r_np2[li]; }
                    const double same_f = qm_to_f64(qs, 0, g.eps), diff_f = ARITH ? df : qm_to_f64(qd, m, g.eps);
                    const uint64_t nn = (uint64_t)(same_f + diff_f), kk = (uint64_t)diff_f;
                    if (nn <= g.binom_nmax) pv = g.binom_tab[nn * (nn + 1) / 2 + kk];
                    else { pv = binom_device(nn, kk, g.eps, g.div_factor); n_fallback++; }
                }
                double mx = 0.0, sum = 0.0;
                uint64_t ts1 = 0, ts2 = 0;
                for (uint32_t j = 0; j < p; ++j) { const double o = shfl_f64(pv, seg0 + (int)j); mx = (j == 0) ? o : (o > mx ? o : mx); }
                if (trunc) for (uint32_t j = 0; j < p; ++j) { ts1 += shfl_u64(t1, seg0 + (int)j); ts2 += shfl_u64(t2, seg0 + (int)j); }
                const double ex = exp(pv - mx);              // own term once; summed in the reference's order j = 0..p-1
                for (uint32_t j = 0; j < p; ++j) sum += shfl_f64(ex, seg0 + (int)j);
                const double lse = mx + log(sum);
                bool pass = false;
                uint64_t ch1 = 0, ch2 = 0, cq = 0;
                uint32_t cm = 0;
                double cscore = 0.0;
                if constexpr (ARITH) {
                    // read_to_node_value (global_clustering.rs:196-202): error_vec with the read's diff added to its partition, summed in partition order
                    const double ed = e_own + df;
                    double mec = 0.0;
                    for (uint32_t j = 0; j < p; ++j) { const double ej = shfl_f64(e_own, seg0 + (int)j); mec += (j == my_k) ? ed : ej; }
                    cscore = mec;
                }
                if (act) {
                    const double am = fabs((pv - lse) - g.cutoff);
                    min_margin = am < min_margin ? am : min_margin;
                    pass = (pv - lse) > g.cutoff;
                    cq = st_q[a] + qd;
                    cm = st_m[a] + m;
                    if (!ARITH) cscore = qm_to_f64(cq, cm, g.eps);
                    ch1 = (st_h1[a] - ts1) + rk1 * (tw1 + (Q0 ? np1 : 0));
                    ch2 = (st_h2[a] - ts2) + rk2 * (tw2 + (Q0 ? np2 : 0));
                }
                uint64_t passmask = __ballot(pass);
                while (passmask) {
                    const int src = __ffsll((unsigned long long)passmask) - 1;
                    passmask &= passmask - 1;
                    const double s_score = shfl_f64(cscore, src);
                    const uint64_t s_h1 = shfl_u64(ch1, src), s_h2 = shfl_u64(ch2, src), s_q = shfl_u64(cq, src);
                    const uint32_t s_m = __shfl(cm, src);
                    const double s_df = ARITH ? shfl_f64(df, src) : 0.0;
                    const uint32_t s_a = a0 + (uint32_t)src / p, s_k = (uint32_t)src % p;
                    const uint32_t hl = s_heap_len;
                    bool dup = false;
                    for (uint32_t e = lane; e < hl; e += 64) { const EntryRec& E = ent[hid[e]]; dup |= (E.h1 == s_h1 && E.h2 == s_h2 && E.score >= s_score); }
                    if (!__any(dup)) {
                        if (lane == 0) {
                            const uint16_t id = efree[--s_efree_n];
                            EntryRec& E = ent[id];
                            E.score = s_score; E.h1 = s_h1; E.h2 = s_h2; E.q = s_q; E.m = s_m; E.parent = (uint16_t)s_a; E.k = (uint8_t)s_k; E.df = s_df;
                            uint32_t len = s_heap_len;
                            heap_push(hid, ent, len, id);
                            if (len > limit) efree[s_efree_n++] = heap_pop(hid, ent, len);
                            s_heap_len = len;
                        }
                    }
                    __syncthreads();
                }
            }

            // ---- M: survivors j = heap slot j, walked in chunks of 64 ----------------------------------------------------------
            const uint32_t nnext = s_heap_len;
            const int32_t new_hi = last_rel > hi_rel ? last_rel : hi_rel;
            uint64_t* nx_q = ST_q(cur ^ 1); uint64_t* nx_h1 = ST_h1(cur ^ 1); uint64_t* nx_h2 = ST_h2(cur ^ 1);
            uint32_t* nx_m = ST_m(cur ^ 1); uint16_t* nx_sl = ST_sl(cur ^ 1);
            for (uint32_t x = lane; x < NS; x += 64) { ref[x] = 0; leader[x] = 0xffffffffu; }
            for (uint32_t j = lane; j < nnext; j += 64) {
                const uint16_t eid = hid[j];
                const EntryRec& E = ent[eid];
                nx_q[j] = E.q; nx_h1[j] = E.h1; nx_h2[j] = E.h2; nx_m[j] = E.m;
                s_pk[j] = (uint32_t)E.parent | ((uint32_t)E.k << 16);
                e2j[eid] = (uint16_t)j;
                slot_hist[beam_hist_off(i, LM, B) + j] = (uint32_t)E.parent | ((uint32_t)E.k << 16);
            }
            __syncthreads();
            for (uint32_t x = lane; x < nnext * p; x += 64) {
                const uint32_t j = x / p, k = x - j * p;
                const uint32_t pk = s_pk[j];
                const uint32_t sid = st_sl[(pk & 0xffff) * p + k];
                nx_sl[x] = (uint16_t)sid;
                if (k != (pk >> 16)) ref[sid] = 1;
                if constexpr (ARITH) { const double pe = ST_ev(cur)[(pk & 0xffff) * p + k]; ST_ev(cur ^ 1)[x] = (k == (pk >> 16)) ? pe + ent[hid[j]].df : pe; }      // the survivors' error_vec
            }
            for (uint32_t j = lane; j < nnext; j += 64) {
                const uint32_t pk = s_pk[j];
                const uint32_t u = st_sl[(pk & 0xffff) * p + (pk >> 16)];
                s_u[j] = (uint16_t)u;
                atomicMin(&leader[u], j);
            }
            __syncthreads();
            for (uint32_t j = lane; j < nnext; j += 64) {
                const uint32_t u = s_u[j];
                const bool lead = leader[u] == j;
                const bool inplace = lead && ref[u] == 0;
                s_flag[j] = (uint8_t)((lead ? 1 : 0) | ((lead && !inplace) ? 2 : 0) | (inplace ? 4 : 0));
            }
            __syncthreads();
            uint32_t ncopy = 0, nlead = 0;
            for (uint32_t j0 = 0; j0 < nnext; j0 += 64) {
                const uint32_t j = j0 + lane;
                const uint8_t f = j < nnext ? s_flag[j] : 0;
                if (f & 4) { const uint32_t u = s_u[j]; ref[u] = 2; newid[u] = (uint16_t)u; }
                const uint64_t cmask = __ballot((f & 2) != 0), lmask = __ballot((f & 1) != 0);
                if (f & 2) copy_list[ncopy + (uint32_t)__popcll(cmask & lane_lt)] = (uint16_t)j;
                if (f & 1) lead_list[nlead + (uint32_t)__popcll(lmask & lane_lt)] = (uint16_t)j;      // survivor index for now
                ncopy += (uint32_t)__popcll(cmask); nlead += (uint32_t)__popcll(lmask);
            }
            __syncthreads();
            if (ncopy) {
                uint32_t found = 0;
                for (uint32_t x0 = 0; x0 < NS && found < ncopy; x0 += 64) {
                    const uint32_t x = x0 + lane;
                    const bool fr = x < NS && ref[x] == 0;
                    const uint64_t fm = __ballot(fr);
                    const uint32_t pos = found + (uint32_t)__popcll(fm & lane_lt);
                    if (fr && pos < ncopy) freeslab[pos] = (uint16_t)x;
                    found += (uint32_t)__popcll(fm);
                }
                if (found < ncopy && lane == 0) atomicAdd(&g.diag[1], 1u);
                __syncthreads();
                for (uint32_t c = lane; c < ncopy; c += 64) { const uint32_t u = s_u[copy_list[c]]; const uint32_t f = freeslab[c]; newid[u] = (uint16_t)f; ref[f] = 2; }
                __syncthreads();
            }
            for (uint32_t j = lane; j < nnext; j += 64) nx_sl[j * p + (s_pk[j] >> 16)] = newid[s_u[j]];
            for (uint32_t c = lane; c < nlead; c += 64) lead_list[c] = newid[s_u[lead_list[c]]];             // -> target slab of the leader
            // copies of the written window for the new versions that could not go in place
            if (ncopy && hi_rel >= (int32_t)first_rel) {
                const uint32_t cnt2 = ((uint32_t)(hi_rel - (int32_t)first_rel + 1) * A) >> 1;
                for (uint32_t c = 0; c < ncopy; ++c) {
                    const uint32_t su = s_u[copy_list[c]];
                    const uint32_t du = freeslab[c];
                    const ulonglong2* s = (const ulonglong2*)(pool + (su * slab_bytes + first_rel * pos_bytes));
                    ulonglong2* d = (ulonglong2*)(pool + (du * slab_bytes + first_rel * pos_bytes));
                    uint32_t x = lane;
                    for (; x + 192 < cnt2; x += 256) {
                        const ulonglong2 v0 = s[x], v1 = s[x + 64], v2 = s[x + 128], v3 = s[x + 192];
                        d[x] = v0; d[x + 64] = v1; d[x + 128] = v2; d[x + 192] = v3;
                    }
                    for (; x < cnt2; x += 64) d[x] = s[x];
                }
            }
            __syncthreads();
            uint32_t nl = 0;
            for (uint32_t x0 = 0; x0 < NS; x0 += 64) {
                const uint32_t x = x0 + lane;
                const bool rf = x < NS && ref[x] != 0;
                const uint64_t fm = __ballot(rf);
                if (rf) { const uint32_t idx = nl + (uint32_t)__popcll(fm & lane_lt); live_id[idx] = (uint16_t)x; s2l[x] = (uint16_t)idx; }
                nl += (uint32_t)__popcll(fm);
            }
            __syncthreads();
            if (new_hi > hi_rel) {
                const uint32_t cntz = (uint32_t)(new_hi - hi_rel) * A;
                const uint32_t items = nl * cntz;
                for (uint32_t x = lane; x < items; x += 64) {
                    const uint32_t e = x / cntz, o = x - e * cntz;
                    *(uint64_t*)(pool + ((uint32_t)live_id[e] * slab_bytes + (uint32_t)(hi_rel + 1) * pos_bytes + o * 8)) = 0;
                }
            }
            __syncthreads();
            for (uint32_t t = 0; t < ntiles; ++t) {
                if (ntiles > 1) stage_tile(t);
                const uint32_t tl = min((uint32_t)SLAB_TILE, L - t * SLAB_TILE);
                const uint32_t items = nlead * tl;
                for (uint32_t x = lane; x < items; x += 64) {
                    const uint32_t e = x / tl, c = x - e * tl;
                    const uint32_t aw = c_aw[c];
                    uint64_t* cp = (uint64_t*)(pool + ((uint32_t)lead_list[e] * slab_bytes + c_off[c] + ((aw >> 28) & 3u) * 8));
                    const uint64_t nv = *cp + (uint64_t)(aw & 0x0fffffffu);
                    *cp = Q0 ? (nv | PRESENT_BIT) : nv;
                }
            }
            __syncthreads();
            cur ^= 1;
            nstates = nnext;
            nlive = nl;
            hi_rel = new_hi;
            start_rel = first_rel;
        }

        if (n > 0) {
            uint32_t ecur = 0;
            if (lane == 0) ecur = e2j[wide_sorted_first(hid, ent, nstates)];
            ecur = uni(__shfl(ecur, 0));
            if (lane == 0) {
                uint8_t* out = g.part_out + roff;
                for (int32_t i = (int32_t)n - 1; i >= 0; --i) {
                    const uint32_t rec = slot_hist[beam_hist_off((uint32_t)i, LM, B) + ecur];
                    out[i] = (uint8_t)(rec >> 16);
                    ecur = rec & 0xffff;
                }
                atomicAdd(g.steps_done, (unsigned long long)n);
            }
            { const double jm = wave_min_f64(min_margin); if (lane == 0) g.job_margin[(uint64_t)b * g.max_ploidy + g.ploidy - 1] = jm; }
        }
        __syncthreads();
    }
    n_fallback = wave_sum_u32(n_fallback);
    if (lane == 0) {
        if (n_fallback) atomicAdd(&g.diag[0], n_fallback);
    }
}

}  // namespace fl
