// beam_fast_kernel.h — the production beam-search kernel for ploidy*beam <= 63 (the CLI defaults give <= 50).
//
// Same algorithm and the same bit-exact results as beam_kernel.h (which stays as the generic path for wider
// beams); what changes is how the work is laid onto a gfx950 wavefront, driven by the round-1 SQ counters
// (profiles/r01a_sq_counters.json: 61 % of wave cycles parked in s_waitcnt, ~450 cycles exposed per global load):
//
//  * software-pipelined allele-agreement loop: the read's cells that can hit a written position are a prefix
//    (cells ascend by SNP), so the loop body has no branch; 8 independent 16-B histogram loads are issued
//    back-to-back from a wave-uniform slab base + 32-bit lane offset before the first one is consumed;
//  * the std::BinaryHeap of (score, entry) pairs and the entry table live in VGPRs, one heap slot / entry per
//    lane, and are manipulated with v_readlane/v_writelane under wave-uniform control flow — push, pop and
//    the duplicate-state test touch no memory and need no barrier;
//  * materialisation of the survivors is lane-parallel: first child of a parent found with an LDS atomicMin,
//    free slabs = complement of the slabs inherited in place (64-bit mask), rank-th free slab by popcount;
//  * pileups without q=0 observations (presence <=> weight > 0) use plain Q24 sums without the presence bit and
//    a single linear hash term per cell (template parameter Q0 = false).
#pragma once
#include "beam_kernel.h"

namespace fl {

constexpr int FAST_TILE = 256;
#ifndef FLORIA_FAST_UNROLL
#define FLORIA_FAST_UNROLL 8
#endif
#ifndef FLORIA_FAST_WAVES
#define FLORIA_FAST_WAVES 3
#endif
constexpr int FAST_UNROLL = FLORIA_FAST_UNROLL;

struct __align__(16) FastState { uint64_t q, h1, h2; uint32_t m; uint32_t bufk; };   // bufk = slab | partition << 16

struct FastLds { uint32_t off_coff, off_caw, off_crp1, off_crp2, off_st[2], off_first, off_free, off_words, total; };
__host__ __device__ inline FastLds fast_lds_layout(uint32_t LM, bool q0) {
    FastLds L;
    uint32_t o = 0;
    L.off_coff = o; o += FAST_TILE * 4;
    L.off_caw = o;  o += FAST_TILE * 4;
    L.off_crp1 = o; o += q0 ? FAST_TILE * 8 : 0;
    L.off_crp2 = o; o += q0 ? FAST_TILE * 8 : 0;
    L.off_st[0] = o; o += LM * sizeof(FastState);
    L.off_st[1] = o; o += LM * sizeof(FastState);
    L.off_first = o; o += 64 * 4;
    L.off_free = o; o += 64 * 4;
    L.off_words = o; o += 16;
    L.total = o;
    return L;
}

__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t l) { return ((uint64_t)rl32((uint32_t)(v >> 32), l) << 32) | rl32((uint32_t)v, l); }
// "writelane": x and l are wave-uniform; a compare + select per dword (clang has no writelane builtin for HIP, and the
// select form lets the compiler handle the VALU->SGPR lane-select hazards itself)
__device__ __forceinline__ void wl32(uint32_t& v, uint32_t x, uint32_t l) { v = (threadIdx.x == l) ? x : v; }
__device__ __forceinline__ void wl64(uint64_t& v, uint64_t x, uint32_t l) { v = (threadIdx.x == l) ? x : v; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// scalar (SMEM) load of read-only data at a wave-uniform address: the value lands in SGPRs, costs no VGPR and is
// tracked by lgkmcnt, so it can be requested a whole beam step before it is used
template <class T> __device__ __forceinline__ T sload(const T* p) {
    return *(const __attribute__((address_space(4))) T*)(uintptr_t)p;
}

// Scores are non-negative f64 (sums of non-negative terms), so their IEEE bit patterns order like the values:
// the heap compares u64 bit patterns.  Heap slot j lives in lane j: (hp_s, hp_id).
struct RegHeap {
    uint64_t hp_s; uint32_t hp_id;     // per-lane
    uint32_t len;                       // uniform
    __device__ __forceinline__ void sift_up(uint32_t pos, uint64_t xs, uint32_t xid) {
        while (pos > 0) {
            const uint32_t par = (pos - 1) >> 1;
            const uint64_t ps = rl64(hp_s, par);
            if (xs <= ps) break;
            const uint32_t pid = rl32(hp_id, par);
            wl64(hp_s, ps, pos); wl32(hp_id, pid, pos);
            pos = par;
        }
        wl64(hp_s, xs, pos); wl32(hp_id, xid, pos);
    }
    __device__ __forceinline__ void push(uint64_t xs, uint32_t xid) { const uint32_t pos = len++; sift_up(pos, xs, xid); }
    __device__ __forceinline__ uint32_t pop() {                    // returns the evicted (max) entry id
        --len;
        const uint64_t xs = rl64(hp_s, len);
        const uint32_t xid = rl32(hp_id, len);
        if (len == 0) return xid;
        const uint32_t root = rl32(hp_id, 0);
        const uint32_t end = len, lim = end >= 2 ? end - 2 : 0;
        uint32_t pos = 0, child = 1;
        while (child <= lim) {                                     // sift_down_to_bottom(0)
            if (rl64(hp_s, child) <= rl64(hp_s, child + 1)) child++;
            wl64(hp_s, rl64(hp_s, child), pos); wl32(hp_id, rl32(hp_id, child), pos);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { wl64(hp_s, rl64(hp_s, child), pos); wl32(hp_id, rl32(hp_id, child), pos); pos = child; }
        sift_up(pos, xs, xid);
        return root;
    }
    // into_sorted_vec()[0]: heap-sort in place, return the id at array position 0
    __device__ __forceinline__ uint32_t sorted_first() {
        uint32_t end = len;
        while (end > 1) {
            --end;
            const uint64_t s0 = rl64(hp_s, 0), se = rl64(hp_s, end);
            const uint32_t i0 = rl32(hp_id, 0), ie = rl32(hp_id, end);
            wl64(hp_s, s0, end); wl32(hp_id, i0, end);              // swap(0, end)
            const uint64_t hs = se; const uint32_t hid = ie;        // hole element = old data[end], now at 0
            uint32_t pos = 0, child = 1;
            const uint32_t lim = end >= 2 ? end - 2 : 0;
            bool placed = false;
            while (child <= lim) {                                  // sift_down_range(0, end)
                if (rl64(hp_s, child) <= rl64(hp_s, child + 1)) child++;
                const uint64_t cs = rl64(hp_s, child);
                if (hs >= cs) { placed = true; break; }
                wl64(hp_s, cs, pos); wl32(hp_id, rl32(hp_id, child), pos);
                pos = child;
                child = 2 * pos + 1;
            }
            if (!placed && child == end - 1) {
                const uint64_t cs = rl64(hp_s, child);
                if (hs < cs) { wl64(hp_s, cs, pos); wl32(hp_id, rl32(hp_id, child), pos); pos = child; }
            }
            wl64(hp_s, hs, pos); wl32(hp_id, hid, pos);
        }
        return rl32(hp_id, 0);
    }
};

template <int A, bool Q0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FLORIA_FAST_WAVES, FLORIA_FAST_WAVES))) void beam_fast_kernel(BeamArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x;
    const uint32_t p = g.ploidy, B = g.beam, LM = p * B;
    const FastLds LY = fast_lds_layout(LM, Q0);
    uint32_t* c_off = (uint32_t*)(smem + LY.off_coff);
    uint32_t* c_aw  = (uint32_t*)(smem + LY.off_caw);
    uint64_t* c_rp1 = (uint64_t*)(smem + LY.off_crp1);
    uint64_t* c_rp2 = (uint64_t*)(smem + LY.off_crp2);
    uint32_t* firstc = (uint32_t*)(smem + LY.off_first);
    uint32_t* freelist = (uint32_t*)(smem + LY.off_free);
    uint32_t* words = (uint32_t*)(smem + LY.off_words);

    const uint32_t PA = p * A;
    const uint32_t pos_bytes = PA * 8;                                 // bytes per SNP position of one state slab
    const uint32_t state_bytes = g.span_max * pos_bytes;               // host guarantees LM*state_bytes < 2^32
    char* slab = (char*)(g.state_pool + (uint64_t)blockIdx.x * LM * ((uint64_t)g.span_max * PA));
    uint32_t* slot_hist = g.hist_pool + (uint64_t)blockIdx.x * g.hist_stride;
    const uint64_t lane_lt = (1ull << lane) - 1;
    const uint64_t full_mask = LM >= 64 ? ~0ull : ((1ull << LM) - 1);

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
        min_margin = 1e300;                                 // per (block, ploidy) job: the host keeps the jobs the stop rule reached
        const ContigDev cd = g.bs.contigs[g.bs.blk_contig[b]];
        const uint64_t roff = g.bs.blk_read_off[b];
        const uint32_t n = (uint32_t)(g.bs.blk_read_off[b + 1] - roff);
        const uint32_t* reads = g.bs.blk_read + roff;
        const uint32_t pos0 = g.bs.blk_pos0[b];

        int cur = 0;
        FastState* st = (FastState*)(smem + LY.off_st[0]);
        FastState* nx = (FastState*)(smem + LY.off_st[1]);
        uint32_t nstates = 1;
        if (lane == 0) { st[0].q = 0; st[0].h1 = 0; st[0].h2 = 0; st[0].m = 0; st[0].bufk = 0; }
        int32_t hi_rel = -1;
        uint32_t start_rel = 0;
        // entry table (lane = entry id) and heap (lane = heap slot)
        uint64_t ev_s = 0, ev_h1 = 0, ev_h2 = 0, ev_q = 0;
        uint32_t ev_m = 0, ev_pk = 0;
        RegHeap H; H.hp_s = 0; H.hp_id = 0; H.len = 0;
        // metadata of the next read, fetched one step ahead
        uint32_t r_next = n > 0 ? reads[0] : 0;
        __syncthreads();

        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t r = uni(r_next);
            if (i + 1 < n) r_next = reads[i + 1];
            const uint32_t cbeg = uni(G(cd.read_off)[r]), L = uni(G(cd.read_off)[r + 1]) - cbeg;
            const uint32_t first_rel = uni(G(cd.first)[r]) - pos0;
            const int32_t  last_rel = (int32_t)(uni(G(cd.last)[r]) - pos0);
            const uint32_t limit = i < (uint32_t)EARLY_READS ? LM : B;
            const uint32_t ntiles = (L + FAST_TILE - 1) / FAST_TILE;

            uint64_t tw1 = 0, tw2 = 0;                  // per-lane partials of sum Rq*w over the read
            uint64_t rpb1 = 0, rpb2 = 0;                // Q0: sum of Rp over the cells beyond hi_rel (current tile)
            uint32_t nin = 0;                           // cells of the current tile with pos <= hi_rel (a prefix)
            auto stage_tile = [&](uint32_t t, bool with_tw) {
                __syncthreads();
                uint32_t cnt_in = 0;
                uint64_t b1 = 0, b2 = 0;
                for (uint32_t c = lane; c < FAST_TILE; c += 64) {
                    const uint32_t cc = t * FAST_TILE + c;
                    bool in = false;
                    if (cc < L) {
                        const uint32_t pr = G(cd.cell_snp)[cbeg + cc] - pos0;
                        const uint32_t aq = G(cd.cell_aw)[cbeg + cc];
                        const uint32_t al = aq >> 28;
                        const uint32_t w = (aq & 0x0fffffffu);
                        const uint32_t idx = pr * A + al;
                        c_off[c] = pr * pos_bytes;
                        c_aw[c] = (al << 28) | w;
                        in = (int32_t)pr <= hi_rel;
                        if (with_tw) { tw1 += g.Rq1[idx] * (uint64_t)w; tw2 += g.Rq2[idx] * (uint64_t)w; }
                        if (Q0) {
                            const uint64_t r1 = g.Rp1[idx], r2 = g.Rp2[idx];
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
            if (ntiles > 1) {                           // rare: reads with > 256 SNPs; hash constant in its own pass
                for (uint32_t c = lane; c < L; c += 64) {
                    const uint32_t pr = G(cd.cell_snp)[cbeg + c] - pos0;
                    const uint32_t aq = G(cd.cell_aw)[cbeg + c];
                    const uint32_t idx = pr * A + (aq >> 28);
                    const uint64_t w = (aq & 0x0fffffffu);
                    tw1 += g.Rq1[idx] * w; tw2 += g.Rq2[idx] * w;
                }
            } else stage_tile(0, true);
            tw1 = wave_sum_u64(tw1); tw2 = wave_sum_u64(tw2);

            uint64_t evalid = 0;                        // entries currently in the next heap
            H.len = 0;

            for (uint32_t a0 = 0; a0 < nstates; a0 += S) {
                const uint32_t a = a0 + my_sl;
                const bool act = lane_pair && a < nstates;
                uint64_t qs = 0, qd = 0, np1 = 0, np2 = 0, t1 = 0, t2 = 0;
                uint32_t m = 0;
                FastState sa; sa.q = 0; sa.h1 = 0; sa.h2 = 0; sa.m = 0; sa.bufk = 0;
                if (act) sa = st[a];
                const uint32_t lane_off = (sa.bufk & 0xffff) * state_bytes + my_k * (A * 8);
                // positions leaving the hash window: [start_rel, first_rel) ∩ [.., hi_rel]
                {
                    const int32_t tend = (int32_t)first_rel - 1 < hi_rel ? (int32_t)first_rel - 1 : hi_rel;
                    for (int32_t pr = (int32_t)start_rel; pr <= tend; ++pr) {
                        if (act) {
#pragma unroll
                            for (int al = 0; al < A; ++al) {
                                const uint64_t v = *(const uint64_t*)(slab + (lane_off + (uint32_t)pr * pos_bytes + al * 8));
                                if (v) {
                                    const uint64_t qv = Q0 ? (v & QMASK63) : v;
                                    t1 += g.Rq1[pr * A + al] * qv; t2 += g.Rq2[pr * A + al] * qv;
                                    if (Q0) { t1 += g.Rp1[pr * A + al]; t2 += g.Rp2[pr * A + al]; }
                                }
                            }
                        }
                    }
                    t1 *= rk1; t2 *= rk2;
                }
                // ---- distance_read_haplo_epsilon_empty (utils_frags.rs:32-75), branch-free ------------------------------
                // per cell: empty position (no non-zero count) -> m += 1; read allele (tied for) the consensus -> same += w;
                // else diff += w.  32-bit partial sums per batch of cells (w <= 2^24), folded into u64 after each batch.
                uint32_t ps = 0, pd = 0;
                auto cell = [&](const ulonglong2* vv, uint32_t aw, uint32_t c) {
                    const uint32_t al = aw >> 28;
                    const uint32_t w = aw & 0x0fffffffu;
                    bool nonempty, same;
                    uint64_t va;
                    if (A == 2) {
                        const uint64_t v0 = Q0 ? (vv[0].x & QMASK63) : vv[0].x, v1 = Q0 ? (vv[0].y & QMASK63) : vv[0].y;
                        nonempty = (v0 | v1) != 0;
                        same = al ? (v1 >= v0) : (v0 >= v1);
                        va = al ? vv[0].y : vv[0].x;
                    } else {
                        uint64_t v[A];
#pragma unroll
                        for (int x = 0; x < A; x += 2) { v[x] = vv[x / 2].x; v[x + 1] = vv[x / 2].y; }
                        uint64_t mx = 0; va = 0;
#pragma unroll
                        for (int x = 0; x < A; ++x) { const uint64_t qx = Q0 ? (v[x] & QMASK63) : v[x]; mx = qx > mx ? qx : mx; va = (x == (int)al) ? v[x] : va; }
                        nonempty = mx != 0;
                        same = (Q0 ? (va & QMASK63) : va) == mx;
                    }
                    ps += (nonempty && same) ? w : 0u;
                    pd += (nonempty && !same) ? w : 0u;
                    m += nonempty ? 0u : 1u;
                    if (Q0) { const bool np = !(va >> 63); np1 += np ? c_rp1[c] : 0ull; np2 += np ? c_rp2[c] : 0ull; }
                };
                for (uint32_t t = 0; t < ntiles; ++t) {
                    if (ntiles > 1) stage_tile(t, false);
                    const uint32_t tl = min((uint32_t)FAST_TILE, L - t * FAST_TILE);
                    if (act) {
                        uint32_t c = 0;
                        for (; c + FAST_UNROLL <= nin; c += FAST_UNROLL) {
                            uint32_t offs[FAST_UNROLL], aws[FAST_UNROLL];
#pragma unroll
                            for (int u = 0; u < FAST_UNROLL; u += 4) {
                                const uint4 o4 = *(const uint4*)(c_off + c + u), a4 = *(const uint4*)(c_aw + c + u);
                                offs[u] = o4.x; offs[u + 1] = o4.y; offs[u + 2] = o4.z; offs[u + 3] = o4.w;
                                aws[u] = a4.x; aws[u + 1] = a4.y; aws[u + 2] = a4.z; aws[u + 3] = a4.w;
                            }
                            ulonglong2 vv[FAST_UNROLL][A / 2];
#pragma unroll
                            for (int u = 0; u < FAST_UNROLL; ++u) {
                                const char* cp = slab + (lane_off + offs[u]);
#pragma unroll
                                for (int x = 0; x < A / 2; ++x) vv[u][x] = *(const ulonglong2*)(cp + 16 * x);
                            }
                            ps = 0; pd = 0;
#pragma unroll
                            for (int u = 0; u < FAST_UNROLL; ++u) cell(vv[u], aws[u], c + u);
                            qs += ps; qd += pd;
                        }
                        ps = 0; pd = 0;
                        for (; c < nin; ++c) {
                            ulonglong2 vv[A / 2];
                            const char* cp = slab + (lane_off + c_off[c]);
#pragma unroll
                            for (int x = 0; x < A / 2; ++x) vv[x] = *(const ulonglong2*)(cp + 16 * x);
                            cell(vv, c_aw[c], c);
                            if ((c & 7) == 7) { qs += ps; qd += pd; ps = 0; pd = 0; }
                        }
                        qs += ps; qd += pd;
                        m += tl - nin;                                  // cells beyond hi_rel: empty positions (:45-48)
                        if (Q0) { np1 += rpb1; np2 += rpb2; }
                    }
                }
                // ---- p-value, log-sum-exp over the state's partitions, pruning (:77-98) -----------------------------------
                double pv = 0.0;
                if (act) {
                    const double same_f = qm_to_f64(qs, 0, g.eps), diff_f = qm_to_f64(qd, m, g.eps);
                    const uint64_t nn = (uint64_t)(same_f + diff_f), kk = (uint64_t)diff_f;
                    if (nn <= g.binom_nmax) pv = g.binom_tab[nn * (nn + 1) / 2 + kk];
                    else { pv = binom_device(nn, kk, g.eps, g.div_factor); n_fallback++; }
                }
                double mx = 0.0, sum = 0.0;
                uint64_t ts1 = 0, ts2 = 0;
                for (uint32_t j = 0; j < p; ++j) {
                    const double o = shfl_f64(pv, seg0 + (int)j);
                    mx = (j == 0) ? o : (o > mx ? o : mx);
                    ts1 += shfl_u64(t1, seg0 + (int)j);
                    ts2 += shfl_u64(t2, seg0 + (int)j);
                }
                const double ex = exp(pv - mx);              // own term once; summed in the reference's order j = 0..p-1
                for (uint32_t j = 0; j < p; ++j) sum += shfl_f64(ex, seg0 + (int)j);
                const double lse = mx + log(sum);
                bool pass = false;
                uint64_t ch1 = 0, ch2 = 0, cq = 0, cs = 0;
                uint32_t cm = 0;
                if (act) {
                    const double am = fabs((pv - lse) - g.cutoff);
                    min_margin = am < min_margin ? am : min_margin;
                    pass = (pv - lse) > g.cutoff;
                    cq = sa.q + qd;
                    cm = sa.m + m;
                    cs = (uint64_t)__double_as_longlong(qm_to_f64(cq, cm, g.eps));
                    ch1 = (sa.h1 - ts1) + rk1 * (tw1 + (Q0 ? np1 : 0));
                    ch2 = (sa.h2 - ts2) + rk2 * (tw2 + (Q0 ? np2 : 0));
                }
                // ---- children through the duplicate test (:122-127) and the bounded heap (:130-134), all in registers -------
                uint64_t passmask = __ballot(pass);
                while (passmask) {
                    const uint32_t src = (uint32_t)__ffsll((unsigned long long)passmask) - 1;
                    passmask &= passmask - 1;
                    const uint64_t s_s = rl64(cs, src), s_h1 = rl64(ch1, src), s_h2 = rl64(ch2, src);
                    const bool dup = ((evalid >> lane) & 1) && ev_h1 == s_h1 && ev_h2 == s_h2 && ev_s >= s_s;
                    if (__ballot(dup)) continue;
                    const uint32_t id = (uint32_t)__ffsll((unsigned long long)~evalid) - 1;
                    evalid |= 1ull << id;
                    wl64(ev_s, s_s, id); wl64(ev_h1, s_h1, id); wl64(ev_h2, s_h2, id);
                    wl64(ev_q, rl64(cq, src), id); wl32(ev_m, rl32(cm, src), id);
                    wl32(ev_pk, (a0 + src / p) | ((src % p) << 16), id);
                    H.push(s_s, id);
                    if (H.len > limit) evalid &= ~(1ull << H.pop());
                }
            }

            // ---- materialise the survivors (lane j = heap slot j = index in the next state array) ------------------------
            const uint32_t nnext = H.len;
            const int32_t new_hi = last_rel > hi_rel ? last_rel : hi_rel;
            const bool surv = lane < nnext;
            const uint32_t eid = surv ? H.hp_id : 0;
            const uint64_t n_q = shfl_u64(ev_q, (int)eid), n_h1 = shfl_u64(ev_h1, (int)eid), n_h2 = shfl_u64(ev_h2, (int)eid);
            const uint32_t n_m = __shfl(ev_m, (int)eid), n_pk = __shfl(ev_pk, (int)eid);
            const uint32_t pj = n_pk & 0xffff, kj = n_pk >> 16;
            firstc[lane] = 0xffffffffu;
            if (lane < 2) words[lane] = 0;
            __syncthreads();
            if (surv) atomicMin(&firstc[pj], lane);
            __syncthreads();
            const bool isfirst = surv && firstc[pj] == lane;
            const uint32_t pbuf = surv ? (st[pj].bufk & 0xffff) : 0;
            if (isfirst) atomicOr(&words[pbuf >> 5], 1u << (pbuf & 31));
            __syncthreads();
            const uint64_t freem = ~(((uint64_t)words[1] << 32) | words[0]) & full_mask;
            const bool nonfirst = surv && !isfirst;
            const uint64_t nfmask = __ballot(nonfirst);
            if ((freem >> lane) & 1) freelist[__popcll(freem & lane_lt)] = lane;
            __syncthreads();
            uint32_t mybuf = pbuf;
            if (nonfirst) {
                const uint32_t rank = (uint32_t)__popcll(nfmask & lane_lt);
                if (rank >= (uint32_t)__popcll(freem)) atomicAdd(&g.diag[1], 1u);
                else mybuf = freelist[rank];
            }
            if (surv) {
                FastState N; N.q = n_q; N.h1 = n_h1; N.h2 = n_h2; N.m = n_m; N.bufk = mybuf | (kj << 16);
                nx[lane] = N;
                slot_hist[beam_hist_off(i, LM, B) + lane] = pj | (kj << 16);
            }
            // further children of a parent: copy the parent's written window [first_rel, hi_rel]
            if (hi_rel >= (int32_t)first_rel) {
                const uint32_t cnt2 = ((uint32_t)(hi_rel - (int32_t)first_rel + 1) * PA) >> 1;
                uint64_t cm2 = nfmask;
                while (cm2) {
                    const uint32_t jj = (uint32_t)__ffsll((unsigned long long)cm2) - 1;
                    cm2 &= cm2 - 1;
                    const uint32_t sb = rl32(pbuf, jj), db = rl32(mybuf, jj);
                    const ulonglong2* s = (const ulonglong2*)(slab + (sb * state_bytes + first_rel * pos_bytes));
                    ulonglong2* d = (ulonglong2*)(slab + (db * state_bytes + first_rel * pos_bytes));
                    uint32_t x = lane;
                    for (; x + 192 < cnt2; x += 256) {
                        const ulonglong2 v0 = s[x], v1 = s[x + 64], v2 = s[x + 128], v3 = s[x + 192];
                        d[x] = v0; d[x + 64] = v1; d[x + 128] = v2; d[x + 192] = v3;
                    }
                    for (; x < cnt2; x += 64) d[x] = s[x];
                }
            }
            __syncthreads();          // nx[] visible
            // zero the newly reached positions (hi_rel, new_hi] of every survivor's slab
            if (new_hi > hi_rel) {
                const uint32_t cntz = (uint32_t)(new_hi - hi_rel) * PA;
                const uint32_t items = nnext * cntz;
                for (uint32_t x = lane; x < items; x += 64) {
                    const uint32_t e = x / cntz, o = x - e * cntz;
                    *(uint64_t*)(slab + ((nx[e].bufk & 0xffff) * state_bytes + (uint32_t)(hi_rel + 1) * pos_bytes + o * 8)) = 0;
                }
            }
            __syncthreads();
            // add the read to partition k of every survivor (types_structs.rs:368-373)
            for (uint32_t t = 0; t < ntiles; ++t) {
                if (ntiles > 1) stage_tile(t, false);
                const uint32_t tl = min((uint32_t)FAST_TILE, L - t * FAST_TILE);
                const uint32_t items = nnext * tl;
                auto addr_of = [&](uint32_t x, uint32_t& w) -> uint64_t* {
                    const uint32_t e = x / tl, c = x - e * tl;
                    const uint32_t aw = c_aw[c], bk = nx[e].bufk;
                    w = aw & 0x0fffffffu;
                    return (uint64_t*)(slab + ((bk & 0xffff) * state_bytes + c_off[c] + ((bk >> 16) * A + (aw >> 28)) * 8));
                };
                uint32_t x = lane;
                for (; x + 192 < items; x += 256) {           // 4 independent read-modify-writes in flight
                    uint32_t w0, w1, w2, w3;
                    uint64_t *p0 = addr_of(x, w0), *p1 = addr_of(x + 64, w1), *p2 = addr_of(x + 128, w2), *p3 = addr_of(x + 192, w3);
                    const uint64_t v0 = *p0, v1 = *p1, v2 = *p2, v3 = *p3;
                    *p0 = Q0 ? ((v0 + w0) | PRESENT_BIT) : v0 + w0; *p1 = Q0 ? ((v1 + w1) | PRESENT_BIT) : v1 + w1;
                    *p2 = Q0 ? ((v2 + w2) | PRESENT_BIT) : v2 + w2; *p3 = Q0 ? ((v3 + w3) | PRESENT_BIT) : v3 + w3;
                }
                for (; x < items; x += 64) {
                    uint32_t w0;
                    uint64_t* p0 = addr_of(x, w0);
                    const uint64_t v0 = *p0;
                    *p0 = Q0 ? ((v0 + w0) | PRESENT_BIT) : v0 + w0;
                }
            }
            __syncthreads();
            cur ^= 1;
            st = (FastState*)(smem + (cur ? LY.off_st[1] : LY.off_st[0]));
            nx = (FastState*)(smem + (cur ? LY.off_st[0] : LY.off_st[1]));
            nstates = nnext;
            hi_rel = new_hi;
            start_rel = first_rel;
        }

        // ---- into_sorted_vec()[0] (:149-150): the heap registers still hold the last step's array; traceback (:155-176) --
        if (n > 0) {
            // lane j's heap slot carries entry ids; the state index of slot j is j
            H.hp_id = lane;
            uint32_t ecur = H.sorted_first();
            uint8_t* out = g.part_out + roff;
            for (int32_t i = (int32_t)n - 1; i >= 0; --i) {
                const uint32_t rec = slot_hist[beam_hist_off((uint32_t)i, LM, B) + ecur];
                if (lane == 0) out[i] = (uint8_t)(rec >> 16);
                ecur = uni(rec & 0xffff);
            }
            if (lane == 0) atomicAdd(g.steps_done, (unsigned long long)n);
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
