// beam_lean_kernel.h — the production beam search (biallelic pileups without q = 0 cells, ploidy 2..5, the CLI's default beam of 10): the shared-slab design
// of beam_slab_kernel.h with every DEPENDENT global round trip taken off the critical path of a step (round 4).
//
// What a step of beam_slab_kernel waits for, one after the other: (1) the code bytes of the read's cells in every live slab (global gather), (2) the binomial
// table (global gather; gone since the level-1 screen), (3) the sums the read-modify-write adds to (global loads).  Measured: a lone wave spends
// 8 k of its 14 k cycles per step in such waits, 4 k each under load.  Here:
//
//   * POSITION CODES LIVE IN LDS.  Slab ids are handed out lowest-first, so 99.9 % of the live slabs have an id < 16 (profiles/r04a_*): the slabs 0..15 keep a ROW
//     of 2-bit codes (bit a = allele a attains the position's maximal sum, 0 = nothing observed) for a circular window of W positions (W = 256 / 512 / 1024 chosen
//     by the host from the batch's largest block span; 128 B per row at W = 512).  Phase A classifies the read's cells against them: LDS reads only.  A step in
//     which some live slab has no valid row (id >= 16, or a window wider than W) classifies from the sums in HBM instead, so there are NO code bytes in HBM
//     any more: a position is one 12-byte record {lo0, lo1, hi0 | hi1 << 8} (narrow sums, < 2^40).
//   * THE READ-MODIFY-WRITE'S LOADS ARE A PREFETCH.  In a bulk step (93 % of the steps: every passing child survives) the slabs that get a new version are known
//     when the pruning test is, BEFORE the std::BinaryHeap pushes: leader election, in-place / copy decision, live list and zeroing run there, and the records of
//     (leader, cell) travel HBM -> LDS by `global_load_lds_dwordx3` (lane l's record lands at 16 * l) while the scalar heap work and the survivors' bookkeeping run.  The add phase reads them from
//     LDS, stores the changed sum and updates the LDS row (one `ds_xor_b32` on the 2-bit field).
//   * ONE vmcnt WAIT PER STEP, at the top of the add phase: it covers the prefetch, the next read's cells (LDS-DMA issued at the top of the step) and the record
//     of the read after next.
//
// Everything observable is unchanged: (same, diff, #eps) per (state, partition), p-values, pruning, child scores, the 128-bit linear state hash, the heap order —
// bit-identical to beam_slab_kernel.h / beam_kernel.h and the oracle (tests/test_gpu_parity.py runs all paths).
// Follows global_clustering.rs:10-208, types_structs.rs:326-376, utils_frags.rs:32-75, :211-258.
#pragma once
#include "beam_slab_kernel.h"

namespace fl {

constexpr int LEAN_ROWS = 16;                  // slabs with an id below this keep their codes in LDS
constexpr int LEAN_POSB = 12;                  // bytes of a position record in HBM
constexpr int LEAN_WAVES = 3;                  // waves per SIMD (LDS-limited: 10-14 KB per wave)
constexpr int LEAN_PF_STRIDE = 16;             // bytes of LDS per prefetched record: global_load_lds_dwordx3 writes lane l's 12 bytes at 16 * l (measured: scripts/probes/dma_probe.hip)
constexpr int lean_pf_cap(int tp) { return tp == 2 ? 192 : 128; }      // (leader, cell) records prefetched per step

struct LeanLds {
    uint32_t off_coff, off_caw;
    uint32_t off_q[2], off_h1[2], off_h2[2], off_m[2], off_sl[2];     // state arrays (SoA) x2
    uint32_t off_live, off_s2l, off_free, off_pk, off_lsrc, off_ldst;
    uint32_t off_rqs, off_rqd, off_rm;
    uint32_t off_ref, off_leader, off_newid;                             // alias r_qs (dead between phase B's p-values and the next phase A)
    uint32_t off_rows, off_pf;
    uint32_t total;
};
__host__ __device__ inline LeanLds lean_lds_layout(uint32_t LM, uint32_t p, uint32_t row_w, uint32_t pf_cap) {
    LeanLds L;
    const uint32_t NS = LM * p;
    uint32_t o = 0;
    auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15) & ~15u; return r; };
    L.off_coff = take(2 * SLAB_TILE * 4); L.off_caw = take(2 * SLAB_TILE * 4);
    for (int i = 0; i < 2; ++i) {
        L.off_q[i] = take((LM + 1) * 8); L.off_h1[i] = take((LM + 1) * 8); L.off_h2[i] = take((LM + 1) * 8); L.off_m[i] = take((LM + 1) * 4);
        L.off_sl[i] = take(NS * 2);
    }
    L.off_live = take(NS * 2); L.off_s2l = take(NS * 2);
    L.off_free = take(64 * 2); L.off_pk = take(64 * 4); L.off_lsrc = take(64 * 2); L.off_ldst = take(64 * 2);
    L.off_rqs = take(NS * 8); L.off_rqd = take(NS * 8); L.off_rm = take(NS * 4);
    L.off_leader = L.off_rqs; L.off_newid = L.off_rqs + NS * 4; L.off_ref = L.off_rqs + NS * 6;
    L.off_rows = take(LEAN_ROWS * (row_w / 4));
    L.off_pf = take(pf_cap * LEAN_PF_STRIDE);
    L.total = o;
    return L;
}

struct LeanRec { uint32_t lo0, lo1, hi; };      // one position of a slab: Q24 sums of allele 0 / 1 (low words), their bits 32..39 in hi (byte 0 / byte 1)
__device__ __forceinline__ uint32_t lean_code(uint64_t v0, uint64_t v1) { return (v0 | v1) ? ((v0 >= v1 ? 1u : 0u) | (v1 >= v0 ? 2u : 0u)) : 0u; }

template <int TP, int TB, bool SPEC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LEAN_WAVES, LEAN_WAVES)))
void beam_lean_kernel(BeamArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x;
    constexpr uint32_t p = TP, B = TB, LM = p * B, NS = LM * p;
    constexpr uint32_t PF = (uint32_t)lean_pf_cap(TP);
    const uint32_t W = 1u << g.row_lg, WM = W - 1u, row_bytes = W >> 2;
    const LeanLds LY = lean_lds_layout(LM, p, W, PF);
    uint32_t* const c_snp_base = (uint32_t*)(smem + LY.off_coff);
    uint32_t* const c_aw_base  = (uint32_t*)(smem + LY.off_caw);
    uint16_t* live_id = (uint16_t*)(smem + LY.off_live);
    uint16_t* s2l = (uint16_t*)(smem + LY.off_s2l);
    uint8_t*  ref = (uint8_t*)(smem + LY.off_ref);
    uint32_t* leader = (uint32_t*)(smem + LY.off_leader);
    uint16_t* newid = (uint16_t*)(smem + LY.off_newid);
    uint16_t* freelist = (uint16_t*)(smem + LY.off_free);
    uint32_t* s_pk = (uint32_t*)(smem + LY.off_pk);
    uint16_t* lsrc = (uint16_t*)(smem + LY.off_lsrc);
    uint16_t* ldst = (uint16_t*)(smem + LY.off_ldst);
    uint64_t* r_qs = (uint64_t*)(smem + LY.off_rqs);
    uint64_t* r_qd = (uint64_t*)(smem + LY.off_rqd);
    uint32_t* r_m = (uint32_t*)(smem + LY.off_rm);
    uint8_t*  rows = smem + LY.off_rows;
    uint32_t* pf = (uint32_t*)(smem + LY.off_pf);

    const uint32_t slab_bytes = g.span_max * (uint32_t)LEAN_POSB;                // host guarantees NS * slab_bytes < 2^32
    char* pool = (char*)g.state_pool + (uint64_t)blockIdx.x * g.state_stride;
    uint32_t* slot_hist = g.hist_pool + (uint64_t)blockIdx.x * g.hist_stride;
    uint64_t* r_t1 = (uint64_t*)(slot_hist + (g.hist_stride - 4ull * NS));      // window-exit hash terms (needed in 1/6 of the steps): the tail of the slot's traceback region
    uint64_t* r_t2 = r_t1 + NS;
    uint32_t* dummy = slot_hist + (g.hist_stride - 4ull * NS - SLAB_DUMMY_WORDS); // scratch for branch-free tails: 4 dwords per lane

    constexpr bool DPPSEG = TP >= 2 && TP <= 4;
    constexpr uint32_t PSC = TP == 2 ? 2 : 4;
    constexpr uint32_t psl = DPPSEG ? PSC : p;
    constexpr uint32_t S = 64 / psl;
    const float rcp_p = __builtin_amdgcn_rcpf((float)p);
    const float eps_f = (float)g.eps, rdiv_f = (float)(1.0 / g.div_factor), cutoff_f = (float)g.cutoff;
    const double margin_alone = fabs(0.0 - g.cutoff);
    const uint32_t my_sl = lane / psl, my_k = lane % psl;
    const bool lane_pair = my_sl < S && my_k < p;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint64_t rk1 = c_rk1[my_k], rk2 = c_rk2[my_k];
    const int seg0 = (int)(my_sl * psl);
    double min_margin = 1e300;
    uint32_t n_fallback = 0;
#ifdef FLORIA_PROF
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = clock64();
    const unsigned long long t_wall0 = wall_clock64(), t_core0 = t_last;
    unsigned long long c_steps = 0, c_bulk = 0, c_cached = 0, c_pfitems = 0, c_directitems = 0, c_lvl2 = 0, c_rebuild = 0;
#endif

    for (;;) {
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(g.queue_head, 1u);
        job = uni(__shfl(job, 0));
        if (job >= g.n_jobs) break;
        const uint32_t b = uni(g.job_block[job]);
        if (g.blk_done[b]) continue;
        if (SPEC && uni(__hip_atomic_load(&g.stop_at[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < p) continue;
        bool dropped = false;
        min_margin = 1e300;
        const ContigDev cd = g.bs.contigs[g.bs.blk_contig[b]];
        const uint64_t roff = g.bs.blk_read_off[b];
        const uint32_t n = (uint32_t)(g.bs.blk_read_off[b + 1] - roff);
        const uint32_t* reads = g.bs.blk_read + roff;
        const uint32_t pos0 = g.bs.blk_pos0[b];

        int cur = 0;
        auto ST_q = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_q[1] : LY.off_q[0])); };
        auto ST_h1 = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_h1[1] : LY.off_h1[0])); };
        auto ST_h2 = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_h2[1] : LY.off_h2[0])); };
        auto ST_m = [&](int w) { return (uint32_t*)(smem + (w ? LY.off_m[1] : LY.off_m[0])); };
        auto ST_sl = [&](int w) { return (uint16_t*)(smem + (w ? LY.off_sl[1] : LY.off_sl[0])); };
        uint32_t nstates = 1, nlive = 1;
        // root: every partition points at slab 0, which is logically empty (nothing written: hi_rel = -1)
        if (lane == 0) { ST_q(0)[0] = 0; ST_h1(0)[0] = 0; ST_h2(0)[0] = 0; ST_m(0)[0] = 0; live_id[0] = 0; s2l[0] = 0; }
        if (lane < p) ST_sl(0)[lane] = 0;
        int32_t hi_rel = -1;
        uint32_t start_rel = 0;
        // the LDS rows: rows_ok = every window of the job so far fitted W positions; row_valid bit r = row r holds the codes of slab r; live16 / live_hi = the live
        // slabs below / at or above LEAN_ROWS.  A step classifies from the rows when all three say so.
        bool rows_ok = true, live_hi = false;
        uint32_t row_valid = 1u, live16 = 1u;
        RegHeap H; H.hp_hi = 0; H.hp_lo = 0; H.hp_id = 0; H.len = 0;
        struct CellMeta { uint32_t cbeg, L; };
        struct StepMeta { uint32_t first, last; uint64_t tw1, tw2; };
        auto load_rec = [&](uint32_t r) -> uint32_t { return lane < 8 ? __builtin_nontemporal_load(G(cd.meta) + 8 * (uint64_t)r + lane) : 0u; };
        auto rec_cm = [&](uint32_t v) { CellMeta m; m.cbeg = rl32(v, 0); m.L = rl32(v, 1); return m; };
        auto rec_sm = [&](uint32_t v) { StepMeta m; m.first = rl32(v, 2); m.last = rl32(v, 3);
                                        m.tw1 = ((uint64_t)rl32(v, 5) << 32) | rl32(v, 4); m.tw2 = ((uint64_t)rl32(v, 7) << 32) | rl32(v, 6); return m; };
        auto dma_cells = [&](uint32_t w, const CellMeta& m) {
            if (m.L <= (uint32_t)SLAB_TILE && 4 * lane < m.L) {
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(G(cd.cell_snp) + m.cbeg + 4 * lane), (lds_void*)(c_snp_base + w * SLAB_TILE), 16, 0, FLORIA_NT_AUX);
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(G(cd.cell_aw) + m.cbeg + 4 * lane), (lds_void*)(c_aw_base + w * SLAB_TILE), 16, 0, FLORIA_NT_AUX);
            }
        };
        uint32_t rid_vec = lane < n ? reads[lane] : 0;
        uint32_t rec_n2 = 0;
        CellMeta cm_cur, cm_next;
        StepMeta sm_cur, sm_next;
        {
            const uint32_t rec0 = load_rec(rl32(rid_vec, 0));
            const uint32_t rec1 = load_rec(rl32(rid_vec, n > 1 ? 1 : 0));
            cm_cur = rec_cm(rec0); sm_cur = rec_sm(rec0);
            cm_next = rec_cm(rec1); sm_next = rec_sm(rec1);
        }
        __syncthreads();
        dma_cells(0, cm_cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (uint32_t i = 0; i < n; ++i) {
            if (SPEC && (i & 63u) == 63u && uni(__hip_atomic_load(&g.stop_at[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < p) { dropped = true; break; }
            const uint32_t cbeg = cm_cur.cbeg, L = cm_cur.L;
            const uint32_t first_rel = sm_cur.first - pos0;
            const int32_t  last_rel = (int32_t)(sm_cur.last - pos0);
            const uint64_t tw1 = sm_cur.tw1, tw2 = sm_cur.tw2;
            const uint32_t limit = i < (uint32_t)EARLY_READS ? LM : B;
            const float tol1 = 4.f * BINOM_SCREEN_C * (float)L + 1e-3f;
            const uint32_t ntiles = (L + SLAB_TILE - 1) / SLAB_TILE;
            const int32_t new_hi = last_rel > hi_rel ? last_rel : hi_rel;
            uint64_t* st_q = ST_q(cur); uint64_t* st_h1 = ST_h1(cur); uint64_t* st_h2 = ST_h2(cur);
            uint32_t* st_m = ST_m(cur); uint16_t* st_sl = ST_sl(cur);
            uint32_t* const c_snp = c_snp_base + (i & 1) * SLAB_TILE;
            uint32_t* const c_aw  = c_aw_base + (i & 1) * SLAB_TILE;
            if ((uint32_t)(new_hi - (int32_t)first_rel) >= W) rows_ok = false;        // the window [first_rel, new_hi] must map injectively into a row
#ifdef FLORIA_LEAN_NO_ROWS
            const bool cached = false;
#else
            const bool cached = rows_ok && !live_hi && (live16 & ~row_valid) == 0u;
#endif

            uint32_t nin = 0;
            auto scan_tile = [&](uint32_t tl) {       // number of the tile's cells at written positions (<= hi_rel; a prefix, cells ascend)
                uint32_t cnt_in = 0;
                uint32_t snps[SLAB_TILE / 64];
#pragma unroll
                for (int u = 0; u < SLAB_TILE / 64; ++u) { snps[u] = 0; if ((uint32_t)(64 * u) < tl) snps[u] = c_snp[lane + 64 * u]; }
#pragma unroll
                for (int u = 0; u < SLAB_TILE / 64; ++u) {
                    if ((uint32_t)(64 * u) >= tl) break;
                    const uint32_t c = lane + 64 * u;
                    const bool in = c < tl && (int32_t)(snps[u] - pos0) <= hi_rel;
                    cnt_in += (uint32_t)__popcll(__ballot(in));
                }
                nin = uni(cnt_in);
            };
            auto stage_tile = [&](uint32_t t) {       // multi-tile reads (L > SLAB_TILE) stage every tile from HBM inside the phases that walk the cells
                __syncthreads();
#pragma unroll
                for (int u = 0; u < SLAB_TILE / 64; ++u) {
                    const uint32_t c = lane + 64 * u;
                    const uint32_t cc = t * SLAB_TILE + c;
                    if (cc < L) { c_snp[c] = G(cd.cell_snp)[cbeg + cc]; c_aw[c] = G(cd.cell_aw)[cbeg + cc]; }
                }
                __syncthreads();
                scan_tile(min((uint32_t)SLAB_TILE, L - t * SLAB_TILE));
                __syncthreads();
            };
            if (ntiles == 1) scan_tile(L);
            if (i + 1 < n) dma_cells((i + 1) & 1, cm_next);
            if (i + 2 < n) {
                if (((i + 2) & 63) == 0) {
                    const uint32_t nv = (i + 2 + lane < n) ? reads[i + 2 + lane] : 0;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(rid_vec) : "v"(nv));
                }
                rec_n2 = load_rec(rl32(rid_vec, (i + 2) & 63));
            }
            BEAM_TICK(0);
#ifdef FLORIA_PROF
            c_steps++; c_cached += cached ? 1 : 0;
#endif

            // ---- A: read vs every LIVE slab -------------------------------------------------------------------------------
            const int32_t tend = (int32_t)first_rel - 1 < hi_rel ? (int32_t)first_rel - 1 : hi_rel;
            const bool trunc = tend >= (int32_t)start_rel;            // some written position leaves the hash window this step
            if (trunc) {      // (1) positions leaving the hash window: [start_rel, first_rel) ∩ [.., hi_rel] — one step in six, 1-2 positions; Gs lanes per slab
                uint32_t Gs = 1, lgGs = 0;
                while (Gs < 16 && nlive * (Gs * 2) <= 64) { Gs *= 2; ++lgGs; }
                const uint32_t per = 64u >> lgGs;
                for (uint32_t l0 = 0; l0 < nlive; l0 += per) {
                    const uint32_t li = l0 + (lane >> lgGs), sub = lane & (Gs - 1);
                    const bool act = li < nlive;
                    const uint32_t slab_off = act ? (uint32_t)live_id[li] * slab_bytes : 0;
                    uint64_t t1 = 0, t2 = 0;
                    for (int32_t pr = (int32_t)start_rel + (int32_t)sub; pr <= tend; pr += (int32_t)Gs) {
                        if (act) {
                            const LeanRec rc = *(const LeanRec*)(pool + (slab_off + (uint32_t)pr * (uint32_t)LEAN_POSB));
                            const uint64_t v0 = ((uint64_t)(rc.hi & 0xffu) << 32) | rc.lo0, v1 = ((uint64_t)((rc.hi >> 8) & 0xffu) << 32) | rc.lo1;
                            const uint32_t hx0 = hash_idx(pos0 + (uint32_t)pr, 0u), hx1 = hash_idx(pos0 + (uint32_t)pr, 1u);
                            t1 += g.Rq1[hx0] * v0 + g.Rq1[hx1] * v1; t2 += g.Rq2[hx0] * v0 + g.Rq2[hx1] * v1;
                        }
                    }
                    t1 = seg_sum_u64(t1, Gs); t2 = seg_sum_u64(t2, Gs);
                    if (act && sub == 0) { r_t1[li] = t1; r_t2[li] = t2; }
                }
            }
            // (2) distances: Gl = 64 / nlive lanes per slab, every lane walks ceil(nin / Gl) cells in batches of 2 / 4 / 6 / 8, sums in 32 bits inside a batch,
            // the lanes of a slab combine through LDS atomics
            for (uint32_t x = lane; x < nlive; x += 64) { r_qs[x] = 0; r_qd[x] = 0; r_m[x] = 0; }
            {
                const uint32_t Gl = nlive <= 64u ? div_small(64u, __builtin_amdgcn_rcpf((float)nlive)) : 1u;
                const float rcp_gl = __builtin_amdgcn_rcpf((float)Gl);
                for (uint32_t l0 = 0; l0 < nlive; l0 += 64u) {
                    const uint32_t lsl = div_small(lane, rcp_gl), sub = lane - lsl * Gl, li = l0 + lsl;
                    const bool act = li < nlive;
                    const uint32_t sid = act ? (uint32_t)live_id[li] : 0u;
                    const uint8_t* const rowp = rows + (sid & (uint32_t)(LEAN_ROWS - 1)) * row_bytes;
                    const char* const rbase = pool + sid * slab_bytes;
                    uint64_t qs = 0, qd = 0;
                    uint32_t m = 0;
                    auto batch = [&](auto NC, uint32_t u0) {
                        constexpr int N = decltype(NC)::value;
                        uint32_t prs[N], aws[N], cdb[N]; bool vs[N];
#pragma unroll
                        for (int u = 0; u < N; ++u) {
                            const uint32_t c = sub + (u0 + (uint32_t)u) * Gl; vs[u] = c < nin; const uint32_t cx = vs[u] ? c : 0u;
                            prs[u] = c_snp[cx] - pos0; const uint32_t awr = c_aw[cx]; aws[u] = vs[u] ? awr : 0u;
                        }
                        if (cached) {
#pragma unroll
                            for (int u = 0; u < N; ++u) cdb[u] = rowp[(prs[u] & WM) >> 2];
#pragma unroll
                            for (int u = 0; u < N; ++u) cdb[u] = (cdb[u] >> ((prs[u] & 3u) * 2u)) & 3u;
                        } else {
                            LeanRec rc[N];
#pragma unroll
                            for (int u = 0; u < N; ++u) rc[u] = *(const LeanRec*)(rbase + prs[u] * (uint32_t)LEAN_POSB);
#pragma unroll
                            for (int u = 0; u < N; ++u)
                                cdb[u] = lean_code(((uint64_t)(rc[u].hi & 0xffu) << 32) | rc[u].lo0, ((uint64_t)((rc[u].hi >> 8) & 0xffu) << 32) | rc[u].lo1);
                        }
                        uint32_t ps = 0, pt = 0, me = 0;
#pragma unroll
                        for (int u = 0; u < N; ++u) {
                            const uint32_t w = aws[u] & 0x0fffffffu;
                            ps += w & (uint32_t)__builtin_amdgcn_sbfe((int)cdb[u], aws[u] >> 28, 1u);      // bit `allele` of the code <=> same
                            pt += cdb[u] ? w : 0u;                                                          // observed position
                            me += (vs[u] && cdb[u] == 0u) ? 1u : 0u;
                        }
                        qs += ps; qd += pt - ps; m += me;
                    };
                    for (uint32_t t = 0; t < ntiles; ++t) {
                        if (ntiles > 1) stage_tile(t);
                        const uint32_t tl = min((uint32_t)SLAB_TILE, L - t * SLAB_TILE);
                        const uint32_t U = div_small(nin + Gl - 1u, rcp_gl);               // rounds of Gl cells
                        if (act) {
                            for (uint32_t u0 = 0; u0 < U; u0 += 8u) {
                                const uint32_t r = U - u0;
                                if (r >= 7u) batch(IC<8>{}, u0); else if (r >= 5u) batch(IC<6>{}, u0); else if (r >= 3u) batch(IC<4>{}, u0); else batch(IC<2>{}, u0);
                            }
                            if (sub == 0) m += tl - nin;                        // cells beyond hi_rel (:45-48)
                        }
                    }
                    if (act) {
                        atomicAdd((unsigned long long*)&r_qs[li], (unsigned long long)qs);
                        atomicAdd((unsigned long long*)&r_qd[li], (unsigned long long)qd);
                        atomicAdd(&r_m[li], m);
                    }
                }
            }
            __syncthreads();
            BEAM_TICK(1);

            // ---- B: per (state, partition) pair: p-value, log-sum-exp, pruning, child (:74-134) -----------------------
            uint64_t evalid = 0;
            H.len = 0;
            bool bulk = false;                       // this step took the no-duplicate / no-eviction path
            uint64_t b_h1 = 0, b_h2 = 0, b_q = 0;    // its children's state hashes and (sum of diffs, #eps) (lane = (state, partition) pair)
            uint32_t b_m = 0;
            uint64_t b_cs = 0, b_pass = 0;
            uint32_t src_map = 0;                    // lane r = child lane of entry r
            uint64_t* const E_s = ST_q(cur ^ 1); uint64_t* const E_h1 = ST_h1(cur ^ 1); uint64_t* const E_h2 = ST_h2(cur ^ 1);
            uint32_t* const E_pk = ST_m(cur ^ 1);
            for (uint32_t a0 = 0; a0 < nstates; a0 += S) {
                const uint32_t a = a0 + my_sl;
                const bool act = lane_pair && a < nstates;
                uint64_t qd = 0, t1 = 0, t2 = 0;
                uint32_t m = 0;
                uint32_t nn = 0, kk = 0;
                float pvf = 0.f;
                if (act) {
                    const uint32_t li = s2l[st_sl[a * p + my_k]];
                    const uint64_t qs = r_qs[li];
                    qd = r_qd[li]; m = r_m[li];
                    if (trunc) { t1 = r_t1[li] * rk1; t2 = r_t2[li] * rk2; }
                    const double same_f = qm_to_f64(qs, 0, g.eps), diff_f = qm_to_f64(qd, m, g.eps);
                    nn = (uint32_t)(same_f + diff_f); kk = (uint32_t)diff_f;
                    pvf = binom_screen_f32(nn, kk, g.ln_eps, g.ln_1meps, eps_f, rdiv_f);
                }
                uint64_t ts1 = 0, ts2 = 0;
                if constexpr (DPPSEG) { if (trunc) static_for<0, TP>([&](auto J) { constexpr int j = decltype(J)::value; ts1 += seg_get64<PSC, j>(t1); ts2 += seg_get64<PSC, j>(t2); }); }
                else if (trunc) for (uint32_t j = 0; j < p; ++j) { ts1 += shfl_u64(t1, seg0 + (int)j); ts2 += shfl_u64(t2, seg0 + (int)j); }
                // the pruning test (pv - lse) > ln 0.01 in two screens: see beam_slab_kernel.h, phase B
                bool pass;
                {
                    float mxf = 0.f;
                    if constexpr (DPPSEG) static_for<0, TP>([&](auto J) { constexpr int j = decltype(J)::value; const float o = seg_get_f32<PSC, j>(pvf); mxf = (j == 0) ? o : (o > mxf ? o : mxf); });
                    else for (uint32_t j = 0; j < p; ++j) { const float o = __shfl(pvf, seg0 + (int)j); mxf = (j == 0) ? o : (o > mxf ? o : mxf); }
                    const float dxf = pvf - mxf;
                    const float ef = __builtin_amdgcn_exp2f(dxf * 1.44269504088896341f);
                    float sumf = 0.f;
                    if constexpr (DPPSEG) static_for<0, TP>([&](auto J) { sumf += seg_get_f32<PSC, decltype(J)::value>(ef); });
                    else for (uint32_t j = 0; j < p; ++j) sumf += __shfl(ef, seg0 + (int)j);
                    const float dsf = (dxf - __builtin_amdgcn_logf(sumf) * 0.693147180559945309f) - cutoff_f;
                    const float asf = fabsf(dsf) - tol1;
                    const uint64_t closeb = __ballot(act && dxf > -(40.f + tol1));
                    const uint32_t segbits = (uint32_t)(closeb >> seg0) & ((1u << psl) - 1u);
                    const bool alone = act && (segbits & (segbits - 1u)) == 0u;
                    if (alone) min_margin = margin_alone < min_margin ? margin_alone : min_margin;
                    const bool far1 = alone || (asf > 0.f && (double)asf >= min_margin);
                    pass = dsf > 0.f;
                    if (__any(act && !far1)) {
#ifdef FLORIA_PROF
                        c_lvl2++;
#endif
                        const Prune2 r2 = prune_level2<DPPSEG ? TP : 0>(nn, kk, act, seg0, p, g.binom_tab, g.binom_nmax, g.eps, g.div_factor, g.cutoff, min_margin);
                        pass = r2.pass != 0; min_margin = r2.min_margin; n_fallback += r2.fallback;
                    }
                }
                pass = pass && act;
                uint64_t ch1 = 0, ch2 = 0, cq = 0, cs = 0;
                uint32_t cm = 0;
                if (act) {
                    cq = st_q[a] + qd;
                    cm = st_m[a] + m;
                    cs = (uint64_t)__double_as_longlong(qm_to_f64(cq, cm, g.eps));
                    ch1 = (st_h1[a] - ts1) + rk1 * tw1;
                    ch2 = (st_h2[a] - ts2) + rk2 * tw2;
                }
                uint64_t passmask = __ballot(pass);
                BEAM_TICK(7);
                // bulk step: one batch, no more children than the heap holds, pairwise distinct state hashes -> every child is inserted, nothing is evicted,
                // entry id = rank among the passing lanes.  Its heap pushes are DEFERRED behind the slab bookkeeping and the prefetch (below).
                if (a0 == 0 && nstates <= S && !g.no_bulk) {
                    const uint32_t npass = (uint32_t)__popcll(passmask);
                    if (npass != 0 && npass <= limit) {
                        const uint32_t slot = (uint32_t)(ch1 ^ (ch1 >> 31) ^ (ch2 >> 17)) & 255u;
                        const uint32_t tab_addr = lds_base + LY.off_pk + slot;
                        if (pass) asm volatile("ds_write_b8 %0, %1" :: "v"(tab_addr), "v"(lane) : "memory");
                        uint32_t slot_owner;
                        asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot_owner) : "v"(tab_addr) : "memory");
                        const bool coll = pass && slot_owner != lane;
                        if (!__any(coll)) {
                            bulk = true;
                            b_h1 = ch1; b_h2 = ch2; b_q = cq; b_m = cm; b_cs = cs; b_pass = passmask;
                            passmask = 0;
                        }
                    }
                }
                while (passmask) {
                    const uint32_t src = (uint32_t)__ffsll((unsigned long long)passmask) - 1;
                    passmask &= passmask - 1;
                    const uint64_t s_s = rl64(cs, src), s_h1 = rl64(ch1, src), s_h2 = rl64(ch2, src);
                    // general path (several batches, possible duplicates or evictions): the entry table lives in the NEXT parity's state arrays — lane e tests entry e
                    bool dup = false;
                    if ((evalid >> lane) & 1) dup = E_h1[lane] == s_h1 && E_h2[lane] == s_h2 && E_s[lane] >= s_s;
                    if (__ballot(dup)) continue;
                    const uint32_t id = (uint32_t)__ffsll((unsigned long long)~evalid) - 1;
                    evalid |= 1ull << id;
                    if (lane == 0) { E_s[id] = s_s; E_h1[id] = s_h1; E_h2[id] = s_h2; E_pk[id] = (a0 + rl32(my_sl, src)) | (rl32(my_k, src) << 16); }
                    H.push(s_s, id);
                    if (H.len > limit) evalid &= ~(1ull << H.pop());
                }
                BEAM_TICK(2);
            }

            // ---- slab bookkeeping for a set of surviving children (order-independent): which slabs get a new version, where it goes, who is live next, the
            // newly reached positions zeroed, the copies made — and the prefetch of the records the add phase will modify.
            // participants: lanes with part = true, each the child (parent state pj, partition kj) with table index idx (all distinct, < npart).
            uint32_t nl = nlive, nlead = 0, pf_items = 0;
            auto bookkeeping = [&](bool part, uint32_t pj, uint32_t kj, uint32_t idx, uint32_t npart) {
                for (uint32_t x = lane; x < NS; x += 64) { ref[x] = 0; leader[x] = 0xffffffffu; }
                if (part) s_pk[idx] = pj | (kj << 16);
                __syncthreads();
                for (uint32_t x = lane; x < npart * p; x += 64) {            // inherited slabs: every partition of a child but the modified one
                    const uint32_t j = div_small(x, rcp_p), k = x - j * p;
                    const uint32_t pk = s_pk[j];
                    if (k != (pk >> 16)) ref[st_sl[(pk & 0xffff) * p + k]] = 1;
                }
                const uint32_t u_old = part ? (uint32_t)st_sl[pj * p + kj] : 0u;
                if (part) atomicMin(&leader[u_old], idx);
                __syncthreads();
                const bool lead = part && leader[u_old] == idx;
                const bool inplace = lead && ref[u_old] == 0;
                const bool needcopy = lead && !inplace;
                __syncthreads();                                           // all reads of ref[] done before in-place marks
                if (inplace) { ref[u_old] = 2; newid[u_old] = (uint16_t)u_old; }
                __syncthreads();
                const uint64_t cmask = __ballot(needcopy);
                const uint32_t ncopy = (uint32_t)__popcll(cmask);
                if (ncopy) {                  // free slabs = not referenced; the first ncopy of them (ascending id) go to the copy leaders
                    uint32_t found = 0;
                    for (uint32_t x0 = 0; x0 < NS && found < ncopy; x0 += 64) {
                        const uint32_t x = x0 + lane;
                        const bool fr = x < NS && ref[x] == 0;
                        const uint64_t fm = __ballot(fr);
                        const uint32_t pos = found + mbcnt64(fm);
                        if (fr && pos < 64) freelist[pos] = (uint16_t)x;
                        found += (uint32_t)__popcll(fm);
                    }
                    if (found < ncopy && lane == 0) atomicAdd(&g.diag[1], 1u);
                    __syncthreads();
                    if (needcopy) { const uint32_t f = freelist[mbcnt64(cmask)]; newid[u_old] = (uint16_t)f; ref[f] = 2; }
                    __syncthreads();
                    // copies of the written window [first_rel, hi_rel]: the records in HBM and (whole) the LDS row
                    uint64_t cm2 = cmask;
                    while (cm2) {
                        const uint32_t jj = (uint32_t)__ffsll((unsigned long long)cm2) - 1;
                        cm2 &= cm2 - 1;
                        const uint32_t su = rl32(u_old, jj);
                        const uint32_t du = newid[su];
                        if (hi_rel >= (int32_t)first_rel) {
                            const uint32_t cw = (uint32_t)(hi_rel - (int32_t)first_rel + 1) * 3u;                      // dwords
                            const uint32_t* s = (const uint32_t*)(pool + (su * slab_bytes + first_rel * (uint32_t)LEAN_POSB));
                            uint32_t* d = (uint32_t*)(pool + (du * slab_bytes + first_rel * (uint32_t)LEAN_POSB));
                            uint32_t x = lane;
                            for (; x + 192 < cw; x += 256) {
                                const uint32_t v0 = s[x], v1 = s[x + 64], v2 = s[x + 128], v3 = s[x + 192];
                                d[x] = v0; d[x + 64] = v1; d[x + 128] = v2; d[x + 192] = v3;
                            }
                            for (; x < cw; x += 64) d[x] = s[x];
                        }
                        if (du < (uint32_t)LEAN_ROWS) {
                            const bool ok = su < (uint32_t)LEAN_ROWS && ((row_valid >> su) & 1u);
                            if (ok) { const uint32_t* rs = (const uint32_t*)(rows + su * row_bytes); uint32_t* rd = (uint32_t*)(rows + du * row_bytes);
                                      for (uint32_t y = lane; y < (row_bytes >> 2); y += 64) rd[y] = rs[y]; }
                            row_valid = ok ? (row_valid | (1u << du)) : (row_valid & ~(1u << du));
                        }
                    }
                    __syncthreads();
                }
                // leaders' (source, target) slabs, compacted
                {
                    const uint64_t lmask = __ballot(lead);
                    nlead = (uint32_t)__popcll(lmask);
                    if (lead) { const uint32_t e = mbcnt64(lmask); lsrc[e] = (uint16_t)u_old; ldst[e] = newid[u_old]; }
                }
                // next live list = every referenced slab (ascending id)
                {
                    uint32_t cnt = 0, l16 = 0; bool lhi = false;
                    for (uint32_t x0 = 0; x0 < NS; x0 += 64) {
                        const uint32_t x = x0 + lane;
                        const bool rf = x < NS && ref[x] != 0;
                        const uint64_t fm = __ballot(rf);
                        if (rf) { const uint32_t ix = cnt + mbcnt64(fm); live_id[ix] = (uint16_t)x; s2l[x] = (uint16_t)ix; }
                        cnt += (uint32_t)__popcll(fm);
                        if (x0 == 0) { l16 = (uint32_t)fm & 0xffffu; lhi = (fm >> LEAN_ROWS) != 0; } else lhi = lhi || fm != 0;
                    }
                    nl = cnt; live16 = l16; live_hi = lhi;
                }
                __syncthreads();
                // zero the newly reached positions (hi_rel, new_hi] of every next-live slab: records in HBM, 2-bit fields in the rows
                if (new_hi > hi_rel) {
                    const uint32_t cz = (uint32_t)(new_hi - hi_rel);
                    const uint32_t cw = cz * 3u, items = nl * cw;
                    const float rcp_cw = __builtin_amdgcn_rcpf((float)cw);
                    for (uint32_t x = lane; x < items; x += 64) {
                        const uint32_t e = items < (1u << 20) ? div_small(x, rcp_cw) : x / cw, o = x - e * cw;
                        *(uint32_t*)(pool + ((uint32_t)live_id[e] * slab_bytes + (uint32_t)(hi_rel + 1) * (uint32_t)LEAN_POSB + o * 4u)) = 0u;
                    }
                    if (rows_ok) {
                        // per live row: clear the fields of positions hi_rel+1 .. new_hi (mod W): dword d of the row holds positions 16d .. 16d+15
                        const uint32_t p_lo = (uint32_t)(hi_rel + 1), ndw = ((p_lo & 15u) + cz + 15u) >> 4;        // dwords touched (the run may wrap around the row)
                        const uint32_t items_r = nl * ndw;
                        const float rcp_nd = __builtin_amdgcn_rcpf((float)ndw);
                        for (uint32_t x = lane; x < items_r; x += 64) {
                            const uint32_t e = items_r < (1u << 20) ? div_small(x, rcp_nd) : x / ndw, o = x - e * ndw;
                            const uint32_t sidz = live_id[e];
                            if (sidz < (uint32_t)LEAN_ROWS) {
                                const uint32_t dpos = (p_lo & ~15u) + 16u * o;                      // first position of this dword (unwrapped)
                                const uint32_t lo_p = dpos > p_lo ? dpos : p_lo, hi_p = (dpos + 15u) < (uint32_t)new_hi ? (dpos + 15u) : (uint32_t)new_hi;
                                const uint32_t nb = (hi_p - lo_p + 1u) * 2u;                        // bits to clear
                                const uint32_t fieldmask = (nb >= 32u ? 0xffffffffu : ((1u << nb) - 1u)) << ((lo_p & 15u) * 2u);
                                atomicAnd((uint32_t*)(rows + sidz * row_bytes + (((dpos & WM) >> 4) << 2)), ~fieldmask);
                            }
                        }
                    }
                }
                // prefetch: the records of (leader e, cell c), item x = e * L + c, of the leaders' SOURCE slabs (nothing of this step has touched them), HBM -> LDS
#ifndef FLORIA_LEAN_NO_PF
                if (ntiles == 1) {
                    const uint32_t items = nlead * L;
                    pf_items = items < PF ? items : PF;
                    const float rcp_tl = __builtin_amdgcn_rcpf((float)L);
                    __syncthreads();
#pragma unroll
                    for (uint32_t k = 0; k < PF / 64; ++k) {
                        const uint32_t x = 64u * k + lane;
                        if (64u * k < pf_items) {
                            if (x < pf_items) {
                                const uint32_t e = div_small(x, rcp_tl), c = x - e * L;
                                const char* src = pool + ((uint32_t)lsrc[e] * slab_bytes + (c_snp[c] - pos0) * (uint32_t)LEAN_POSB);
                                __builtin_amdgcn_global_load_lds((gbl_cvoid*)src, (lds_void*)(pf + 256u * k), 12, 0, 0);
                            }
                        }
                    }
                }
#endif
#ifdef FLORIA_PROF
                c_rebuild++;
#endif
            };

            if (bulk) {
                const uint32_t npass = (uint32_t)__popcll(b_pass);
                const bool mypass = (b_pass >> lane) & 1;
                bookkeeping(mypass, my_sl, my_k, mbcnt64(b_pass), npass);
                BEAM_TICK(4);
                // the std::BinaryHeap pushes, in child-lane order
                uint64_t pm = b_pass;
                uint32_t r = 0;
                while (pm) {
                    const uint32_t src = (uint32_t)__ffsll((unsigned long long)pm) - 1;
                    pm &= pm - 1;
                    wlane(src_map, src, r);
                    H.push(rl64(b_cs, src), r);
                    ++r;
                }
#ifdef FLORIA_PROF
                c_bulk++;
#endif
            }
            BEAM_TICK(3);

            // ---- M: survivors (lane j = heap slot j = next state j) -----------------------------------------------------------
            const uint32_t nnext = H.len;
            const bool surv = lane < nnext;
            const uint32_t eid = surv ? H.hp_id : 0;
            uint64_t n_q = 0, n_h1 = 0, n_h2 = 0;
            uint32_t n_m = 0, n_pk = 0;
            if (bulk) {
                const int esrc = (int)__shfl(src_map, (int)eid);
                n_h1 = shfl_u64(b_h1, esrc); n_h2 = shfl_u64(b_h2, esrc); n_q = shfl_u64(b_q, esrc); n_m = __shfl(b_m, esrc);
                n_pk = __shfl(my_sl | (my_k << 16), esrc);
            } else if (surv) {
                n_h1 = E_h1[eid]; n_h2 = E_h2[eid]; n_pk = E_pk[eid];
            }
            const uint32_t pj = n_pk & 0xffff, kj = n_pk >> 16;
            if (!bulk) {
                if (surv) {                            // the child's (sum of diffs, #eps) = its parent's + the read's distance to the extended slab
                    const uint32_t li = s2l[st_sl[pj * p + kj]];
                    n_q = st_q[pj] + r_qd[li]; n_m = st_m[pj] + r_m[li];
                }
                __syncthreads();
                bookkeeping(surv, pj, kj, lane, nnext);
            }
            uint64_t* nx_q = ST_q(cur ^ 1); uint64_t* nx_h1 = ST_h1(cur ^ 1); uint64_t* nx_h2 = ST_h2(cur ^ 1);
            uint32_t* nx_m = ST_m(cur ^ 1); uint16_t* nx_sl = ST_sl(cur ^ 1);
            __syncthreads();
            s_pk[lane] = n_pk;
            __syncthreads();
            for (uint32_t x = lane; x < nnext * p; x += 64) {            // the next states' slab tables, in heap order
                const uint32_t j = div_small(x, rcp_p), k = x - j * p;
                const uint32_t pk = s_pk[j];
                const uint32_t sidn = st_sl[(pk & 0xffff) * p + k];
                nx_sl[x] = (k != (pk >> 16)) ? (uint16_t)sidn : newid[sidn];
            }
            if (surv) {
                nx_q[lane] = n_q; nx_h1[lane] = n_h1; nx_h2[lane] = n_h2; nx_m[lane] = n_m;
                __builtin_nontemporal_store(pj | (kj << 16), slot_hist + beam_hist_off(i, LM, B) + lane);
            }
            __syncthreads();
            BEAM_TICK(5);

            // ---- add the read ONCE per distinct new version (types_structs.rs:368-373) -------------------------------------------------------
            // THE vmcnt wait of the step: the prefetched records, the next read's cells (LDS-DMA from the top of the step) and the record of the read after next
            // have landed.  (The traceback store above is younger than all of them; waiting for it as well costs less than a wrong count would.)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t rec_hold;
            asm volatile("v_mov_b32 %0, %1" : "=v"(rec_hold) : "v"(rec_n2));
            {
                for (uint32_t t = 0; t < ntiles; ++t) {
                    if (ntiles > 1) stage_tile(t); else __syncthreads();
                    const uint32_t tl = min((uint32_t)SLAB_TILE, L - t * SLAB_TILE);
                    const uint32_t items = nlead * tl;
                    const float rcp_tl = __builtin_amdgcn_rcpf((float)tl);
#ifdef FLORIA_PROF
                    c_pfitems += pf_items; c_directitems += items - pf_items;
#endif
                    auto add_pass = [&](auto NC, uint32_t x0, bool from_lds) {
                        constexpr int AU = decltype(NC)::value;
                        uint32_t w[AU], al[AU], prr[AU], dsl[AU]; char* base[AU]; bool oks[AU];
                        LeanRec rc[AU];
#pragma unroll
                        for (int u = 0; u < AU; ++u) {
                            const uint32_t xx = x0 + lane + 64 * u;
                            const bool ok = xx < items;
                            const uint32_t xs = ok ? xx : 0;
                            const uint32_t e = div_small(xs, rcp_tl), c = xs - e * tl;
                            const uint32_t aw = c_aw[c], pr = c_snp[c] - pos0;
                            w[u] = aw & 0x0fffffffu; al[u] = aw >> 28; prr[u] = pr; oks[u] = ok;
                            dsl[u] = (uint32_t)ldst[e];
                            base[u] = ok ? pool + (dsl[u] * slab_bytes + pr * (uint32_t)LEAN_POSB) : (char*)(dummy + 4 * lane);
                            if (from_lds) { rc[u].lo0 = pf[4 * xs]; rc[u].lo1 = pf[4 * xs + 1]; rc[u].hi = pf[4 * xs + 2]; }
                            else rc[u] = *(const LeanRec*)(ok ? pool + ((uint32_t)lsrc[e] * slab_bytes + pr * (uint32_t)LEAN_POSB) : (const char*)(dummy + 4 * lane));
                        }
#pragma unroll
                        for (int u = 0; u < AU; ++u) {
                            uint64_t v0 = ((uint64_t)(rc[u].hi & 0xffu) << 32) | rc[u].lo0, v1 = ((uint64_t)((rc[u].hi >> 8) & 0xffu) << 32) | rc[u].lo1;
                            const uint32_t oldc = lean_code(v0, v1);
                            uint64_t nv;
                            if (al[u]) { v1 += w[u]; nv = v1; } else { v0 += w[u]; nv = v0; }
                            ((uint32_t*)base[u])[al[u]] = (uint32_t)nv;
                            if ((uint32_t)nv < w[u]) ((uint8_t*)base[u])[8 + al[u]] = (uint8_t)(nv >> 32);          // the low word wrapped: one time in ~256 adds
                            const uint32_t newc = lean_code(v0, v1);
                            if (oks[u] && rows_ok && dsl[u] < (uint32_t)LEAN_ROWS && newc != oldc)
                                atomicXor((uint32_t*)(rows + dsl[u] * row_bytes + (((prr[u] & WM) >> 4) << 2)), (oldc ^ newc) << ((prr[u] & 15u) * 2u));
                        }
                    };
                    const uint32_t n_lds = (t == 0) ? pf_items : 0u;            // (multi-tile reads are not prefetched)
                    uint32_t x0 = 0;
                    for (; x0 < n_lds; x0 += 256u) {
                        const uint32_t r = n_lds - x0;
                        if (r > 192u) add_pass(IC<4>{}, x0, true); else if (r > 128u) add_pass(IC<3>{}, x0, true); else if (r > 64u) add_pass(IC<2>{}, x0, true); else add_pass(IC<1>{}, x0, true);
                    }
                    for (x0 = n_lds; x0 < items; x0 += 256u) {
                        const uint32_t r = items - x0;
                        if (r > 192u) add_pass(IC<4>{}, x0, false); else if (r > 128u) add_pass(IC<3>{}, x0, false); else if (r > 64u) add_pass(IC<2>{}, x0, false); else add_pass(IC<1>{}, x0, false);
                    }
                }
            }
            cm_cur = cm_next; sm_cur = sm_next;
            if (i + 2 < n) { cm_next = rec_cm(rec_hold); sm_next = rec_sm(rec_hold); }
            __syncthreads();
            BEAM_TICK(6);
            cur ^= 1;
            nstates = nnext;
            nlive = nl;
            hi_rel = new_hi;
            start_rel = first_rel;
        }

        if (SPEC && dropped) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (n > 0 && !(SPEC && dropped)) {
            H.hp_id = lane;
            uint32_t ecur = H.sorted_first();
            uint8_t* out = g.part_out + roff;
            for (int32_t i = (int32_t)n - 1; i >= 0; i -= 8) {                 // 8 traceback rows per memory round trip
                uint32_t row[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) row[u] = (i - u >= 0) ? slot_hist[beam_hist_off((uint32_t)(i - u), LM, B) + lane] : 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i - u >= 0) {
                        const uint32_t rec = rl32(row[u], ecur);
                        if (lane == 0) out[i - u] = (uint8_t)(rec >> 16);
                        ecur = rec & 0xffff;
                    }
                }
            }
            if (lane == 0) atomicAdd(g.steps_done, (unsigned long long)n);
            { const double jm = wave_min_f64(min_margin); if (lane == 0) g.job_margin[(uint64_t)b * g.max_ploidy + p - 1] = jm; }
        }
        __syncthreads();
    }
#ifdef FLORIA_PROF
    if (lane == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&g.prof[16 + i], t_acc[i]);
                     atomicAdd(&g.prof[24], wall_clock64() - t_wall0); atomicAdd(&g.prof[25], clock64() - t_core0); atomicAdd(&g.prof[26], 1ull);
                     atomicAdd(&g.prof[26 + g.ploidy], wall_clock64() - t_wall0);
                     atomicAdd(&g.prof[54], c_steps); atomicAdd(&g.prof[55], c_bulk); atomicAdd(&g.prof[56], c_cached); atomicAdd(&g.prof[57], c_pfitems);
                     atomicAdd(&g.prof[58], c_directitems); atomicAdd(&g.prof[51], c_lvl2); atomicAdd(&g.prof[59], c_rebuild); }
#endif
    n_fallback = wave_sum_u32(n_fallback);
    if (lane == 0 && n_fallback) atomicAdd(&g.diag[0], n_fallback);
}

}  // namespace fl
