// optimize_kernel.h — iterative MEC optimisation + MEC statistics of one (SNP block, ploidy) job per
// 256-thread workgroup, then the ploidy stop rule and the final gather.
//
// Follows local_clustering.rs:71-130 (optimize_clustering), :292-358 (opt_iterate), :218-260
// (get_mec_stats_epsilon), :187-215 (get_mec_stats_epsilon_no_phred), utils_frags.rs:160-184
// (hap_block_from_partition) and graph_processing.rs:156-162,196-251 (mec_vector, stop rule).
//
// Layout: one histogram slab per resident job, [pos][partition][allele] u64 cells holding
// (read count << 44 | Q24 weighted sum): the phred histogram (hap_block_from_partition(.., true)) and the
// unit-count histogram (.., false) of the reference are the two fields of the same cell, so
// get_mec_stats_epsilon_no_phred needs no second pass over the reads.  opt_iterate's candidate moves are
// sorted with a bitonic network on the key (gain desc, partition asc, read asc, target asc), which is the
// reference's stable sort_by over its enumeration order with set iteration canonicalised to ascending
// counter_id (DESIGN.md "Iteration order").
#pragma once
#include "common.h"
#include "arith_kernel.h"

namespace fl {

// workgroup size is a template parameter (128 / 512 / 1024): one job's LDS histogram is shared by that many threads;
// the host picks it from the batch's mean reads per block
constexpr int OPT_SORT_LDS = 512;
constexpr int OPT_META_MAX = 4096;        // reads per block whose (cell offset, length, partition) are staged in LDS        // candidates sorted in LDS up to this many
constexpr int NUM_ITER_OPTIMIZE = 20;     // constants.rs:3
constexpr int OPT_U = 8;                  // cells per software-pipelined batch of the distance loop
constexpr int OPT_BUCKETS = 16;           // visiting order of the build / distance passes: reads bucketed by ceil(#cells / 64) (the distance loop's trips), 16+ trips together

struct OptArgs {
    BlockSet bs;
    const uint32_t* job_block;
    uint32_t  n_jobs;
    uint32_t  ploidy, max_ploidy;
    uint32_t  span_max, n_max;
    uint32_t* queue_head;
    const uint8_t* blk_done;
    double    eps;
    const uint8_t* part_in;      // beam output  [total reads]
    uint8_t*  part_out;          // optimised    [total reads]  (this ploidy's plane)
    uint64_t* hist_pool;         // [slots][span_max*ploidy*A]
    double*   dist_pool;         // [slots][n_max*ploidy]
    uint64_t* cand_gain_pool;    // [slots][cand_cap]   (f64 bits of the gain)
    uint32_t* cand_key_pool;     // [slots][cand_cap]
    uint32_t* moves_pool;        // [slots][n_max]      applied moves (read_local << 8 | from << 4 | to)
    uint64_t  cand_cap;          // pow2 >= n_max*(ploidy-1)
    double*   mec;               // [n_blocks*max_ploidy]   mec_vector[p-1]
    double*   num_alleles;       // [n_blocks*max_ploidy]
    uint32_t* iters;             // [n_blocks*max_ploidy]   optimisation rounds run (diagnostic)
    unsigned long long* prof;    // [16] cycle counters per phase (only with -DFLORIA_PROF)
    // the ploidy stop rule (graph_processing.rs:196-251), applied by the job's workgroup right after its MEC statistics when the
    // stage holds this ploidy alone (a tiny separate launch between two persistent grids waits for wave slots like a big one)
    uint32_t  fuse_select;
    int32_t   stopping_heuristic;
    double    mec_threshold;     // threshold for THIS ploidy, computed on the host with libm pow (:204-220)
    uint32_t  release_tried;     // != 0: a beam launch in flight waits for tried[b] / blk_done[b] (tail_overlap): they are published with agent-scope stores
    uint8_t*  blk_done_w;
    uint32_t* best_ploidy;
    uint32_t* tried;
    // speculative stages (several ploidies of a block in flight at once): the stop rule is published per block as soon as the two MEC
    // values it compares exist — stop_at[b] = the smallest ploidy at which the reference's loop breaks (as far as known), ready[b] =
    // bit p set once mec[b][p-1] is final.  Jobs of a block with ploidy > stop_at[b] are dropped wherever they are (beam: at dequeue
    // and every 64 reads; here: at dequeue): their results could not be looked at by select_kernel.  Null outside speculative stages.
    uint32_t* stop_at;
    uint32_t* ready;
    double    thresholds[FLORIA_MAX_PLOIDY + 2];      // mec_threshold of every ploidy (host libm pow)
    // reference-arithmetic mode (optimize_kernel<.., ARITH = true>, arith_kernel.h)
    const uint2*    cell_ord;    // the reads' cells {SNP, allele << 28 | weight} in the iteration order of Frag.positions
    const uint64_t* cell_ord_off;
    uint64_t* fk_pool;           // [slots][ploidy*span_max]  first-insertion key of every (partition, position)
    uint64_t* sk_pool;           // [slots][sort_cap]         sort keys
    uint32_t* sp_pool;           // [slots][sort_cap]         sorted (partition, position) entries
    uint8_t*  fx_pool;           // [slots][ploidy][2][fx_ctrl + fx_slot] the emulated position maps: control bytes, keys
    uint32_t* ol_pool;           // [slots][2][ploidy][span_max] every partition's positions (block-relative) in the bucket order of its map, two parities
    uint64_t  sort_cap, fx_ctrl, fx_slot;
    uint32_t  fx_lds_off;        // != 0: the maps' control bytes sit in the workgroup's LDS at this offset instead ([ploidy][2][fx_ctrl], then [ploidy][fx_tags] words)
    uint32_t  fx_replay;         // (tests) != 0: every map is replayed, the home-bucket rule is off
    uint32_t  fx_tags;           // words of a map's conflict-detection table (a power of two, FX_TAGS_MIN..FX_TAGS_MAX)
    uint32_t  pm_lds_off;        // != 0: the visiting order of the build and distance passes (u16 read indices, longest reads first) in LDS at this offset ([n_max])
    uint32_t  fk_lds_off;        // != 0: the first-insertion keys as 32-bit words (read << 12 | cell rank: reads < 2^20, cells per read < 2^12) in LDS at this offset ([ploidy*span_max])
};
// fired(q): the reference's loop breaks at ploidy q (graph_processing.rs:196-251); needs mec[q-1] (q > 1) and mec[q], num_alleles[q]
__device__ inline bool stop_rule_fires(const OptArgs& g, uint32_t b, uint32_t q) {
    const double mec_q = __hip_atomic_load(&g.mec[(uint64_t)b * g.max_ploidy + q - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double na_q = __hip_atomic_load(&g.num_alleles[(uint64_t)b * g.max_ploidy + q - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double expected = na_q * g.eps;
    if (q > 1) {
        const double mec_prev = __hip_atomic_load(&g.mec[(uint64_t)b * g.max_ploidy + q - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((mec_q / mec_prev) < g.thresholds[q]) { /* do nothing */ }
        else if (g.stopping_heuristic) return true;
        return mec_q < expected;
    }
    return mec_q < expected;
}
#ifdef FLORIA_PROF
#define OPT_TICK(ph) do { __syncthreads(); if (tid == 0) { const unsigned long long _t = clock64(); atomicAdd(&g.prof[ph], _t - t_last); t_last = _t; } } while (0)
#else
#define OPT_TICK(ph) do {} while (0)
#endif

// all-reduce (sum) over each row of 16 lanes with DPP (no LDS crossbar): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
template <int CTRL> __device__ __forceinline__ uint64_t dpp_u64(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t row16_sum_u64(uint64_t v) {
    v += dpp_u64<0xB1>(v); v += dpp_u64<0x4E>(v); v += dpp_u64<0x141>(v); v += dpp_u64<0x140>(v);
    return v;
}

// whole-wave sums by DPP inside the rows of 16 lanes and four v_readlane across them.  (The __shfl_xor butterflies of common.h need six lane-address registers that
// hipcc computes before the job loop and keeps - spilled to scratch in the register-capped instances - until the loop is left: round 5 found the same in the beam kernel.)
__device__ __forceinline__ uint64_t opt_wave_sum_u64(uint64_t v) {
    v = row16_sum_u64(v);
    auto rl = [](uint64_t x, int l) { return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l); };
    return rl(v, 0) + rl(v, 16) + rl(v, 32) + rl(v, 48);
}
__device__ __forceinline__ uint32_t opt_wave_sum_u32(uint32_t v) { return (uint32_t)opt_wave_sum_u64((uint64_t)v); }

__device__ __forceinline__ bool cand_before(uint64_t ga, uint32_t ka, uint64_t gb, uint32_t kb) {
    return ga > gb || (ga == gb && ka < kb);
}

// bitonic sort of n (pow2) candidates, ascending in cand_before order
template <class G, class K>
__device__ inline void bitonic_sort(G gain, K key, uint32_t n, int tid, int nthreads) {
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < n; i += nthreads) {
                uint32_t ixj = i ^ j;
                if (ixj > i) {
                    uint64_t ga = gain[i], gb = gain[ixj];
                    uint32_t ka = key[i], kb = key[ixj];
                    bool up = (i & k) == 0;
                    bool sw = up ? cand_before(gb, kb, ga, ka) : cand_before(ga, ka, gb, kb);
                    if (sw) { gain[i] = gb; gain[ixj] = ga; key[i] = kb; key[ixj] = ka; }
                }
            }
            __syncthreads();
        }
    }
}

// HL = the job's histogram slab lives in LDS (span_max*ploidy*A*8 bytes fit): distance loads, the build/move atomics and the
// MEC reductions then never leave the CU; only the reads' cells stream from HBM/L2.
// TP: ploidy as a compile-time constant (0 = read it from the arguments): the per-partition loops of the distance pass are exact
// instead of MAX_PLOIDY predicated iterations.
// ARITH: the reference's own f64 arithmetic (floria_hip_set_option("arith", 1)): a read's distance is the running sum over its cells in the order
// of its position set (utils_frags.rs:33-72), a partition's `errors` the running sum over its positions in the bucket order of its position
// map (local_clustering.rs:226-256), which is emulated per partition (arith_kernel.h).
// (the ploidy 1-3 instances of 512 threads are held to the 80 VGPRs that let three workgroups share a CU: six waves per SIMD)
#ifndef FLORIA_ARITH_OPT_WAVES
#define FLORIA_ARITH_OPT_WAVES 6
#endif
constexpr int opt_min_waves(int tp, int threads, bool arith, int ow = 0) { return arith ? (threads == 512 ? (ow ? ow : FLORIA_ARITH_OPT_WAVES) : 1) : ((threads == 512 && tp >= 1 && tp <= 3) ? 6 : 1); }      // (ARITH, 512 threads: 80 VGPRs = three workgroups per CU; the 149 hipcc takes when left alone leave one.  Measured on config 4, arith = 1: 4 / 5 / 6 waves 177 / 175 / 171 ms)
// OW (ARITH): waves per SIMD to compile for, 0 = the default above (where LDS allows two workgroups per CU anyway — ploidy >= 3 on config 4 — four waves' worth of registers)
template <int A, bool HL, int OPT_THREADS, int TP = 0, bool ARITH = false, int OW = 0>
__global__ __launch_bounds__(OPT_THREADS) __attribute__((amdgpu_waves_per_eu(opt_min_waves(TP, OPT_THREADS, ARITH, OW))))
void optimize_kernel(OptArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];   // moved bitset [n_max/8 rounded] | histogram (HL)
    __shared__ uint64_t s_gain[OPT_SORT_LDS];
    __shared__ uint32_t s_key[OPT_SORT_LDS];
    __shared__ uint64_t s_errq[MAX_PLOIDY], s_goodq[MAX_PLOIDY];
    __shared__ uint32_t s_errm[MAX_PLOIDY];
    __shared__ uint32_t s_size[MAX_PLOIDY];
    __shared__ uint32_t s_ncand, s_nmoves, s_job, s_skip, s_dq;
    __shared__ uint32_t s_chg_lo, s_chg_hi;          // positions whose code byte changed in the last batch of moves (HL)
    __shared__ uint32_t s_bkt[OPT_BUCKETS];          // visiting order: reads per bucket of ceil(#cells / 64), then the buckets' write cursors
    __shared__ double s_score;
    __shared__ double s_errf[MAX_PLOIDY];            // ARITH: running `errors` of every partition
    __shared__ uint32_t s_cntk[MAX_PLOIDY + 1];      // ARITH: positions in every partition's map
    __shared__ uint32_t s_cnt2[2][MAX_PLOIDY];       // ... of the two position lists
    __shared__ uint32_t s_nk[MAX_PLOIDY], s_kmin[MAX_PLOIDY], s_kmax[MAX_PLOIDY], s_dirC[MAX_PLOIDY];      // ARITH: positions partition k covers, the first and the last (relative); != 0: buckets of its map when every key sits in its home bucket
    __shared__ uint32_t s_ldir[2][MAX_PLOIDY];       // ... list [parity][k] came from the home-bucket rule (no first-insertion keys stand behind it)
    __shared__ uint32_t s_bpar[MAX_PLOIDY], s_tpar[MAX_PLOIDY], s_keep[MAX_PLOIDY];      // ARITH: which of its two lists holds partition k's ACCEPTED order / the order of the partition being scored; keys unchanged
    uint32_t* s_moved = (uint32_t*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t p = TP ? (uint32_t)TP : g.ploidy, PA = p * A;
    constexpr int KMAX = TP ? TP : MAX_PLOIDY;
    const uint32_t moved_bytes = (((g.n_max + 31) / 32) * 4 + 15) & ~15u;
    const uint32_t meta_n = g.n_max <= (uint32_t)OPT_META_MAX ? g.n_max : 0;            // 0 = read the metadata from HBM every time
    const uint32_t meta_bytes = (meta_n * 12 + 15) & ~15u;
    uint32_t* m_cb = (uint32_t*)(smem + moved_bytes);
    uint32_t* m_lk = m_cb + meta_n;                                                   // cell count | partition << 24
    uint32_t* m_fl = m_lk + meta_n;                                                   // first | last << 16 position index of the read (span <= 65535)
    // The four 16-lane groups of a wavefront walk their reads' cells in lockstep, so a pass costs the LONGEST of the four reads (config 4: 2.66 trips of the distance
    // loop instead of the mean 1.94).  With an order table the passes visit the reads longest first, reads of equal trip count side by side (counting sort while
    // the metadata is staged): neither the histogram (integer atomics) nor the distances (one slot per read) depend on the visiting order.
    uint16_t* m_pm = (!ARITH && g.pm_lds_off && meta_n) ? (uint16_t*)(smem + g.pm_lds_off) : nullptr;
    uint64_t* hist = HL ? (uint64_t*)(smem + moved_bytes + meta_bytes) : g.hist_pool + (uint64_t)blockIdx.x * g.span_max * PA;
    // HL: one CODE byte per (position, partition) next to the histogram — bit a = allele a attains the position's maximal phred sum, 0 =
    // nothing observed — refreshed after the build and after every batch of moves; the distance pass (70 % of the kernel) then reads
    // p bytes per cell instead of p 16-byte histogram pieces and needs no 64-bit compares
    uint8_t* codes = (uint8_t*)(smem + moved_bytes + meta_bytes + (((size_t)g.span_max * PA * 8 + 15) & ~(size_t)15));
    double* dist = g.dist_pool + (uint64_t)blockIdx.x * g.n_max * p;
    uint64_t* cgain = g.cand_gain_pool + (uint64_t)blockIdx.x * g.cand_cap;
    uint32_t* ckey = g.cand_key_pool + (uint64_t)blockIdx.x * g.cand_cap;
    uint32_t* moves = g.moves_pool + (uint64_t)blockIdx.x * g.n_max;

    for (;;) {
        __syncthreads();
        if (tid == 0) {
            const uint32_t j0 = atomicAdd(g.queue_head, 1u);
            s_job = j0;
            // stop_at[] changes while this launch runs (other workgroups publish their blocks' stop rule): ONE thread reads it and the
            // workgroup acts on that value — threads reading it separately could disagree and part ways around the barriers below
            s_skip = (j0 < g.n_jobs && g.stop_at && __hip_atomic_load(&g.stop_at[g.job_block[j0]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p) ? 1u : 0u;
        }
        __syncthreads();
        const uint32_t job = s_job;
        if (job >= g.n_jobs) break;
        const uint32_t b = g.job_block[job];
        if (g.blk_done[b]) continue;                    // (written only by the workgroup that owns block b in a launch: the same for every thread)
        if (s_skip) continue;
        const ContigDev cd = g.bs.contigs[g.bs.blk_contig[b]];
        const uint64_t roff = g.bs.blk_read_off[b];
        const uint32_t n = (uint32_t)(g.bs.blk_read_off[b + 1] - roff);
        const uint32_t* reads = g.bs.blk_read + roff;
        const uint32_t pos0 = g.bs.blk_pos0[b], span = g.bs.blk_span[b];
        const uint8_t* pin = g.part_in + roff;
        uint8_t* part = g.part_out + roff;
        const uint32_t ncell = span * PA;
        const uint2* ord = ARITH ? g.cell_ord + g.cell_ord_off[g.bs.blk_contig[b]] : nullptr;
#ifdef FLORIA_PROF
        unsigned long long t_last = clock64();
#endif

        // ---- hap_block_from_partition (utils_frags.rs:177-184): phred sums and unit counts in one cell -----
        for (uint32_t x = tid; x < ncell; x += OPT_THREADS) hist[x] = 0;
        if (tid < MAX_PLOIDY) s_size[tid] = 0;
        if (tid < OPT_BUCKETS) s_bkt[tid] = 0;
        __syncthreads();
        // stage (first cell, #cells, partition) of every read once: the chain reads[i] -> read_off[r] -> cells is then one hop
        const bool meta = meta_n != 0;
        for (uint32_t i = tid; i < n; i += OPT_THREADS) {
            const uint32_t r = reads[i], k = pin[i];
            part[i] = (uint8_t)k;
            atomicAdd(&s_size[k], 1u);
            if (meta) { const uint4 mr = *(const uint4*)(cd.meta + 8 * (uint64_t)r);     // {cell offset, #cells, first, last}: one 16-B load
                        m_cb[i] = mr.x; m_lk[i] = mr.y | (k << 24); m_fl[i] = (mr.z - pos0) | ((mr.w - pos0) << 16);
                        if (m_pm) atomicAdd(&s_bkt[min((mr.y + 63u) >> 6, (uint32_t)OPT_BUCKETS - 1u)], 1u); }
        }
        __syncthreads();
        if (m_pm) {
            if (tid == 0) { uint32_t acc = 0; for (int bk = OPT_BUCKETS - 1; bk >= 0; --bk) { const uint32_t c = s_bkt[bk]; s_bkt[bk] = acc; acc += c; } }      // longest first
            __syncthreads();
            for (uint32_t i = tid; i < n; i += OPT_THREADS) m_pm[atomicAdd(&s_bkt[min(((m_lk[i] & 0xffffffu) + 63u) >> 6, (uint32_t)OPT_BUCKETS - 1u)], 1u)] = (uint16_t)i;
            __syncthreads();
        }
        auto read_meta = [&](uint32_t i, uint32_t& cb, uint32_t& len, uint32_t& k) {
            if (meta) { cb = m_cb[i]; const uint32_t lk = m_lk[i]; len = lk & 0xffffffu; k = lk >> 24; }
            else { const uint32_t r = reads[i]; cb = G(cd.read_off)[r]; len = G(cd.read_off)[r + 1] - cb; k = part[i]; }
        };
        // `track`: record the interval of positions whose code changed — a read's distances depend on the codes at its positions only,
        // so after a batch of moves only the reads that reach into that interval are re-evaluated (moves rarely flip a consensus)
        auto refresh_codes = [&](bool track) {
            if (!HL) return;
            if (track) { if (tid == 0) { s_chg_lo = 0xffffffffu; s_chg_hi = 0; } __syncthreads(); }
            for (uint32_t x = tid; x < span * p; x += OPT_THREADS) {            // x = position * p + partition
                const uint64_t* cp = hist + (uint64_t)x * A;
                uint64_t q[A], mx = 0;
#pragma unroll
                for (int a = 0; a < A; ++a) { q[a] = cp[a] & QMASK44; mx = q[a] > mx ? q[a] : mx; }
                uint32_t c = 0;
#pragma unroll
                for (int a = 0; a < A; ++a) c |= (mx != 0 && q[a] == mx) ? (1u << a) : 0u;
                if (!track) codes[x] = (uint8_t)c;
                else if (codes[x] != (uint8_t)c) { codes[x] = (uint8_t)c; const uint32_t pr = x / p; atomicMin(&s_chg_lo, pr); atomicMax(&s_chg_hi, pr); }
            }
            __syncthreads();
        };
        // 16 lanes per read (4 reads per wavefront, 16 per workgroup pass): 64-B coalesced cell segments, up to 8 cells
        // per lane loaded before the first atomic
        const uint32_t grp = tid >> 4, sub = tid & 15;
        const uint32_t n16 = (n + 15) & ~15u;
        for (uint32_t iv = grp; iv < n16; iv += OPT_THREADS / 16) {
            const uint32_t i = (m_pm && iv < n) ? (uint32_t)m_pm[iv] : iv;
            uint32_t cb = 0, len = 0, k = 0;
            if (i < n) read_meta(i, cb, len, k);
            for (uint32_t c0 = sub; c0 < len; c0 += 16 * 8) {
                uint32_t sn[8], aqs[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const uint32_t c = c0 + 16 * u; const bool v = c < len; sn[u] = v ? G(cd.cell_snp)[cb + c] : 0; aqs[u] = v ? G(cd.cell_aw)[cb + c] : 0xffffffffu; }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (aqs[u] != 0xffffffffu)
                        atomicAdd((unsigned long long*)&hist[(uint64_t)(sn[u] - pos0) * PA + k * A + (aqs[u] >> 28)], (unsigned long long)((1ull << CNT_SHIFT) | (aqs[u] & 0x0fffffffu)));
            }
        }
        __syncthreads();

        // get_mec_stats_epsilon (local_clustering.rs:218-260) -> per-partition (errors Q24, #eps) and
        // _no_phred (:187-215) -> (good count, bad count, #eps); returns score = -(sum_k errors_k) via s_score
        auto mec_stats = [&](bool phred) {
            if (tid < MAX_PLOIDY) { s_errq[tid] = 0; s_goodq[tid] = 0; s_errm[tid] = 0; }
            __syncthreads();
            for (uint32_t k = 0; k < p; ++k) {
                uint64_t eq = 0, gq = 0;
                uint32_t em = 0;
                for (uint32_t pr = tid; pr < span; pr += OPT_THREADS) {
                    const uint64_t* cp = hist + (uint64_t)pr * PA + k * A;
                    uint64_t mx = 0, tot = 0, cn = 0;
#pragma unroll
                    for (int al = 0; al < A; ++al) {
                        const uint64_t v = cp[al];
                        const uint64_t q = phred ? (v & QMASK44) : (v >> CNT_SHIFT);
                        mx = q > mx ? q : mx; tot += q; cn += v >> CNT_SHIFT;
                    }
                    if (cn) {                                   // position key exists in the partition's map
                        gq += mx; eq += tot - mx;
                        if (mx <= (phred ? ONE_Q24 : 1ull)) em += 1;           // cons_bases <= 1. -> errors += epsilon
                    }
                }
                eq = opt_wave_sum_u64(eq); gq = opt_wave_sum_u64(gq); em = opt_wave_sum_u32(em);
                if (lane == 0) { atomicAdd((unsigned long long*)&s_errq[k], (unsigned long long)eq); atomicAdd((unsigned long long*)&s_goodq[k], (unsigned long long)gq); atomicAdd(&s_errm[k], em); }
            }
            __syncthreads();
            if (tid == 0 && phred) {
                double s = 0.0;                                 // binom_vec.iter().map(|x| x.1).sum() * -1.
                for (uint32_t k = 0; k < p; ++k) s += qm_to_f64(s_errq[k], s_errm[k], g.eps);
                s_score = s * -1.0;
            }
            __syncthreads();
        };

        // ARITH, opt_iterate's distances (local_clustering.rs:292-326): one thread per (read, partition) — the running sum cannot be split over lanes.  They depend on
        // the histogram only, so the pass for round r + 1 runs beside the statistics of round r (a rejected round r makes it useless, and is the last): on the
        // wavefronts that have no position map to list at once, on the others when they are done.  inc: only the reads that reach into the interval of positions
        // whose code byte changed in the last batch of moves.
        // The pairs are handed out 64 at a time from a counter in LDS (s_dq, zeroed behind a barrier before the pass): the wavefronts that replay a position map join
        // when they are done, so everybody ends together.
        auto dist_arith = [&](bool inc) __attribute__((always_inline)) {
            const uint32_t chg_lo = inc ? s_chg_lo : 0u, chg_hi = inc ? s_chg_hi : 0xffffffffu;
            // (measured: a thread per read folding up to four partitions at once - the cells loaded once for all of them - is slower, 274 against 250 ms per call: half as many
            // threads have work)
            for (;;) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_dq, 64u);
                base = (uint32_t)__shfl((int)base, 0);
                if (base >= n * p) break;
                const uint32_t pair = base + (uint32_t)lane;
                bool act = pair < n * p;                                 // (no `continue` out of the body: every lane is back at the counter together)
                const uint32_t i = act ? pair / p : 0u, k = pair - i * p;
                uint32_t cb = 0, len = 0, kk = 0;
                read_meta(i, cb, len, kk);
                if (inc) { const uint32_t fl = m_fl[i]; if ((fl >> 16) < chg_lo || (fl & 0xffffu) > chg_hi) act = false; }      // no code changed at any position of this read: its distances stand
                if (!act) len = 0;
                double df = 0.0;
                constexpr int DU = HL ? 8 : 4;                           // cells per batch: order entries, cells and histogram rows / code bytes requested together (measured with the prefetch below: 4 / 6 / 8 / 12 / 16 cells 99.6 / 97.3 / 97.5 / 123 / 136 ms per call - beyond 8 the 80-VGPR instance spills)
                // (the order entries come from HBM or the Infinity Cache - a block's reads do not stay in L2 between two passes -, a microsecond away: the NEXT batch's are
                // requested before this batch's are used, so a read of 92 cells waits for six round trips instead of twelve)
                uint2 nx[DU];
                if (len) {
#pragma unroll
                    for (int u = 0; u < DU; ++u) nx[u] = ord[cb + ((uint32_t)u < len ? (uint32_t)u : len - 1)];
                }
                for (uint32_t c0 = 0; c0 < len; c0 += DU) {
                    uint32_t aqs[DU], sn[DU]; uint64_t row[DU][A];
#pragma unroll
                    for (int u = 0; u < DU; ++u) { sn[u] = nx[u].x; aqs[u] = nx[u].y; }
                    if (c0 + DU < len) {
#pragma unroll
                        for (int u = 0; u < DU; ++u) nx[u] = ord[cb + (c0 + DU + u < len ? c0 + DU + u : len - 1)];
                    }
                    if constexpr (HL) {                                  // the code byte says it all: 0 = nothing observed, bit a = allele a attains the maximal sum
                        uint32_t cds[DU];
#pragma unroll
                        for (int u = 0; u < DU; ++u) cds[u] = codes[(sn[u] - pos0) * p + k];
                        // (branch-free: x + 0.0 == x for the non-negative running sum, so a `same` cell or a slot past the read's end adds an exact zero instead of
                        // branching around the add - the lanes of a wavefront walk different reads, every path was executed under masks anyway)
#pragma unroll
                        for (int u = 0; u < DU; ++u) {
                            const bool valid = c0 + u < len;
                            const bool same = ((cds[u] >> (aqs[u] >> 28)) & 1u) != 0u;
                            double tv = cds[u] == 0u ? g.eps : (double)(aqs[u] & 0x0fffffffu) * 0x1p-24;      // :45-48 diff += epsilon | :70 diff += w
                            tv = (cds[u] != 0u && same) ? 0.0 : tv;                                           // :54-67 same: nothing is added to diff
                            df += valid ? tv : 0.0;
                        }
                        continue;
                    }
#pragma unroll
                    for (int u = 0; u < DU; ++u) {
                        const uint64_t* rp = hist + (uint64_t)(sn[u] - pos0) * PA + k * A;
#pragma unroll
                        for (int x = 0; x < A; ++x) row[u][x] = rp[x];
                    }
#pragma unroll
                    for (int u = 0; u < DU; ++u) {
                        if (c0 + u >= len) break;
                        const uint32_t al = aqs[u] >> 28;
                        uint64_t mx = 0, va = 0;
#pragma unroll
                        for (int x = 0; x < A; ++x) { const uint64_t q = row[u][x] & QMASK44; mx = q > mx ? q : mx; va = (x == (int)al) ? q : va; }
                        if (mx == 0) df += g.eps;
                        else if (va != mx) df += (double)(aqs[u] & 0x0fffffffu) * 0x1p-24;
                    }
                }
                if (act) dist[pair] = df;
            }
        };
        // ARITH: the same statistics with `errors` as the reference's running sum.  The position map of partition k is filled by its reads in ascending
        // order, every read's cells in set order, through `hap_map.entry(*pos).or_insert(..)` (utils_frags.rs:165): std looks the key up first and reserves
        // room only for a key that is not there, so only the FIRST insertion of a position moves anything.  (1) first-insertion key of every (partition,
        // position) by atomicMin — behind a plain read: the reads are visited in ascending order, so most keys lose against what is already there and
        // never issue the atomic; (2) sort; (3) one WAVEFRONT per partition replays the insertions into the emulated table, many per round
        // (arith_kernel.h: FxWave::insert_batch); (4) and walks its buckets, adding the terms of :244-253 in that order — and writing the positions down
        // in that order (`olist`, two parities): the statistics of the same partition in the other weighting (the final unit-count pass, :187-215) or
        // after a rejected round walk the list again instead of replaying the map.
        // par = which list is written (reuse = false) or walked again (reuse = true).
        // (always_inline: called from three places, hipcc would otherwise make it a real function whose captures — every local it touches — live in scratch memory:
        // 1 KB of scratch per lane, 149 VGPRs and one workgroup per CU, measured)
        // first = the job's first call (nothing to compare with); reuse = walk the accepted lists again, nothing else.
        auto mec_stats_arith = [&](bool phred, bool first, bool reuse, bool with_dist, bool dist_inc) __attribute__((always_inline)) {
            auto olist_of = [&](uint32_t par, uint32_t k) { return g.ol_pool + (((uint64_t)blockIdx.x * 2 + par) * p + k) * g.span_max; };          // [slot][parity][partition][span_max]
            uint32_t* const kprev = (uint32_t*)(g.fk_pool + (uint64_t)blockIdx.x * g.span_max * p);      // [parity][partition][span_max] the first-insertion keys behind every list (the u64 key pool, unused while the keys are 32-bit words in LDS)
            auto kprev_of = [&](uint32_t par, uint32_t k) { return kprev + ((uint64_t)par * p + k) * g.span_max; };
            const uint64_t one = phred ? ONE_Q24 : 1ull;
            const double scale = phred ? 0x1p-24 : 1.0;
            // the terms of 64 listed positions (lane = entry; `have` = this lane holds one), added in lane order onto ef; returns the lanes' consensus counts
            auto fold64 = [&](uint32_t k, uint32_t posrel, bool have, double& ef) __attribute__((always_inline)) -> uint64_t {
                const uint64_t* cp = hist + (uint64_t)(have ? posrel : 0u) * PA + k * A;
                uint64_t q[A];
#pragma unroll
                for (int al = 0; al < A; ++al) { const uint64_t v = cp[al]; q[al] = phred ? (v & QMASK44) : (v >> CNT_SHIFT); }
#pragma unroll
                for (int x = 1; x < A; ++x)                                  // allele_counts.sort_by(count) (:244): ascending, absent alleles are zeros (x + 0.0 == x)
#pragma unroll
                    for (int y = A - 1; y >= x; --y) if (q[y] < q[y - 1]) { const uint64_t tq = q[y]; q[y] = q[y - 1]; q[y - 1] = tq; }
                double term[A];                                              // [A - 1]: errors += epsilon if cons_bases <= 1 (:251-253), else + 0.0
#pragma unroll
                for (int x = 0; x + 1 < A; ++x) term[x] = (double)q[x] * scale;
                term[A - 1] = q[A - 1] <= one ? g.eps : 0.0;
                uint64_t nz[A];                                              // (x + 0.0 == x for the non-negative sums here: the zero terms - most of them - stay off the chain)
#pragma unroll
                for (int x = 0; x < A; ++x) nz[x] = __ballot(have && term[x] != 0.0);
                uint64_t hm = 0;
#pragma unroll
                for (int x = 0; x < A; ++x) hm |= nz[x];
                while (hm) {                                                 // (wave-uniform: the sum is the one sequential thing here)
                    const uint32_t l = (uint32_t)__builtin_ctzll(hm);
                    hm &= hm - 1;
#pragma unroll
                    for (int x = 0; x < A; ++x) {
                        if (!((nz[x] >> l) & 1ull)) continue;
                        const uint64_t tb = (uint64_t)__double_as_longlong(term[x]);
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tb, (int)l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tb >> 32), (int)l);
                        ef += __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));          // :248-250 all but the last, then :251-253
                    }
                }
                return have ? q[A - 1] : 0ull;
            };
            if (reuse) {
                for (uint32_t k = wid; k < p; k += OPT_THREADS / 64) {
                    const uint32_t par = s_bpar[k], D = s_cnt2[par][k];
                    const uint32_t* const olist = olist_of(par, k);
                    double ef = 0.0;
                    uint64_t good = 0;
                    for (uint32_t d0 = 0; d0 < D; d0 += 64) {
                        const bool have = d0 + lane < D;
                        const uint32_t posrel = have ? olist[d0 + lane] : 0u;
                        good += fold64(k, posrel, have, ef);
                    }
                    good = opt_wave_sum_u64(good);
                    if (lane == 0) { s_errf[k] = ef; s_goodq[k] = good; }
                }
                __syncthreads();
                if (tid == 0 && phred) { double sc = 0.0; for (uint32_t k = 0; k < p; ++k) sc += s_errf[k]; s_score = sc * -1.0; }
                __syncthreads();
                return;
            }
            uint64_t* fk = g.fk_pool + (uint64_t)blockIdx.x * g.span_max * p;
            const uint32_t M = span * p;
            uint32_t M2 = 1;
            while (M2 < M) M2 <<= 1;
            uint64_t* sk = M2 <= (uint32_t)OPT_SORT_LDS ? s_gain : g.sk_pool + (uint64_t)blockIdx.x * g.sort_cap;        // (the candidate sort's LDS arrays are free here)
            uint32_t* sp = M2 <= (uint32_t)OPT_SORT_LDS ? s_key : g.sp_pool + (uint64_t)blockIdx.x * g.sort_cap;
            uint32_t* fk32 = (uint32_t*)(smem + g.fk_lds_off);
            const bool k32 = g.fk_lds_off != 0;                                 // (LDS atomics: the HBM ones were a third of this kernel's time)
            if (k32) for (uint32_t x = tid; x < M; x += OPT_THREADS) fk32[x] = ~0u;
            else for (uint32_t x = tid; x < M; x += OPT_THREADS) fk[x] = ~0ull;
            if (tid < MAX_PLOIDY) { s_errf[tid] = 0.0; s_goodq[tid] = 0; s_cntk[tid] = 0; s_nk[tid] = 0; s_kmin[tid] = ~0u; s_kmax[tid] = 0; }
            if (tid == 0) s_dq = 0;
            if (g.fx_lds_off) for (uint32_t x = tid; x < p * g.fx_tags; x += OPT_THREADS) ((uint32_t*)(smem + g.fx_lds_off + (uint64_t)p * 2 * g.fx_ctrl))[x] = 0xffffffffu;
            __syncthreads();
            // (0) THE HOME-BUCKET RULE.  A key's first probe is bucket (key * K) mod C (FxHash's low bits, arith_kernel.h), K odd: keys that differ mod C have different
            // home buckets, an insertion takes another bucket only when its home is full, and a resize re-inserts by the same rule — so if the positions of partition
            // k span fewer than C of them, C the buckets of a map grown to their number, every key of the final map sits in its home bucket WHATEVER the order of
            // insertion was, and the iteration order is a function of the set: bucket b holds the position congruent to b * K^-1 mod C, if the partition covers it.
            // No first-insertion keys, no sort, no replay for such a partition (every one of BASELINE's configs: a block spans ~500 positions, a map of >= 449 keys
            // has 1024 buckets); the others go the long way below.
            for (uint32_t k = 0; k < p; ++k)
                for (uint32_t pr0 = (uint32_t)tid & ~63u; pr0 < span; pr0 += OPT_THREADS) {
                    const uint32_t pr = pr0 + lane;
                    bool cov = false;
                    if (pr < span) {
#pragma unroll
                        for (int al = 0; al < A; ++al) cov |= hist[(uint64_t)pr * PA + k * A + al] != 0;
                    }
                    const uint64_t cm = __ballot(cov);
                    if (cm && lane == 0) { atomicAdd(&s_nk[k], (uint32_t)__popcll(cm)); atomicMin(&s_kmin[k], pr0 + (uint32_t)__builtin_ctzll(cm)); atomicMax(&s_kmax[k], pr0 + 63u - (uint32_t)__builtin_clzll(cm)); }
                }
            __syncthreads();
            if ((uint32_t)tid < p) {
                const uint32_t nk = s_nk[tid];
                uint32_t C = 4;
                while (fx_cap_of(C) < nk) C <<= 1;
                s_dirC[tid] = (!g.fx_replay && (nk == 0 || s_kmax[tid] - s_kmin[tid] < C)) ? C : 0u;
            }
            __syncthreads();
            OPT_TICK(14);    // (ARITH) key table cleared, home-bucket rule decided
            bool any_replay = false;
            for (uint32_t k = 0; k < p; ++k) any_replay |= s_dirC[k] == 0;
            if (any_replay)
            for (uint32_t i = grp; i < n16; i += OPT_THREADS / 16) {
                uint32_t cb = 0, len = 0, k = 0;
                if (i < n) { read_meta(i, cb, len, k); if (!meta) k = part[i]; if (s_dirC[k]) len = 0; }
                // A read can only win a position that no earlier read of its partition holds.  The reads are sorted by their first position, so for most of
                // them every position of their span is already taken when their turn comes: 16 lanes look at the span's keys in LDS first and skip the read's
                // cells (the global loads this pass waits for) unless some key could still lose.  (Keys only ever decrease: a stale look errs towards loading.)
                if (k32 && meta && span <= 65535u) {
                    bool open = false;
                    if (i < n && len) {
                        const uint32_t fl = m_fl[i], thr = i << 12, lastp = fl >> 16;
                        for (uint32_t pr0 = (fl & 0xffffu) + sub; pr0 <= lastp; pr0 += 128) {          // eight keys per lane and round trip
                            uint32_t v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { const uint32_t pr = pr0 + 16u * u; v[u] = pr <= lastp ? fk32[k * span + pr] : 0u; }
#pragma unroll
                            for (int u = 0; u < 8; ++u) open |= v[u] >= thr;
                        }
                    }
                    if (!((__ballot(open) >> (lane & 48)) & 0xffffull)) len = 0;
                }
                for (uint32_t c0 = sub; c0 < len; c0 += 16 * 8) {              // (as the build pass: eight cells per lane requested before the first atomic)
                    uint32_t sn[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const uint32_t c = c0 + 16 * u; sn[u] = c < len ? ord[cb + c].x : 0u; }          // (SNP indices are 1-based: 0 = no cell)
                    uint32_t curv[8];                                          // (a stale value can only be larger than the current one: then the atomic is issued needlessly, never skipped wrongly)
                    if (k32) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) curv[u] = sn[u] ? fk32[k * span + (sn[u] - pos0)] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (sn[u]) {
                            if (k32) { const uint32_t key = (i << 12) | (c0 + 16 * u); if (curv[u] > key) atomicMin(&fk32[k * span + (sn[u] - pos0)], key); }
                            else { const unsigned long long key = ((unsigned long long)i << 24) | (c0 + 16 * u); unsigned long long* const a = (unsigned long long*)&fk[k * span + (sn[u] - pos0)]; if (*(volatile unsigned long long*)a > key) atomicMin(a, key); }
                        }
                }
            }
            __syncthreads();
            OPT_TICK(15);    // (ARITH) atomicMin pass
            // (2) the order of the first insertions.  Keys in LDS and at most 256 positions (every BASELINE config): partition k's wavefront ranks its own row by
            // counting — key e's rank = the number of smaller keys, every key read by all lanes at once — and writes the positions back IN PLACE in that order
            // (a wave's LDS operations execute in order: every read of the row precedes the first write).  No workgroup-wide sort, no barrier.
            const bool wave_sort = k32 && span <= 1024u;
            // (1b) A batch of moves rarely changes WHO inserts a position first (only where a moved read was, or becomes, the lowest-numbered read of its partition at a
            // position: the frontier and the gaps), and the map's layout is a function of the first-insertion keys alone: a partition whose keys equal those behind its
            // accepted list keeps that list - no sort, no replay, only the walk with the new counts.  Compared exactly, key by key, against the copy in HBM scratch.
            if (tid < p) { s_keep[tid] = (wave_sort && !first && !s_dirC[tid] && !s_ldir[s_bpar[tid]][tid]) ? 1u : 0u; }
            __syncthreads();
            if (wave_sort && !first) {
                for (uint32_t k = 0; k < p; ++k) {
                    if (!s_keep[k]) continue;
                    const uint32_t* const prev = kprev_of(s_bpar[k], k);
                    bool differs = false;
                    for (uint32_t pr = tid; pr < span; pr += OPT_THREADS) differs |= fk32[k * span + pr] != prev[pr];
                    if (differs) s_keep[k] = 0u;
                }
                __syncthreads();
            }
            if (tid < p) s_tpar[tid] = first ? 0u : (s_keep[tid] ? s_bpar[tid] : s_bpar[tid] ^ 1u);
            __syncthreads();
            if (wave_sort)
                for (uint32_t k = 0; k < p; ++k) {
                    if (s_keep[k] || s_dirC[k]) continue;
                    uint32_t* const dst = kprev_of(s_tpar[k], k);
                    for (uint32_t pr = tid; pr < span; pr += OPT_THREADS) dst[pr] = fk32[k * span + pr];
                }
            __syncthreads();          // (the rows are sorted in place below)
            if (!wave_sort) {
            for (uint32_t x = tid; x < M2; x += OPT_THREADS) {
                uint64_t f = ~0ull;
                if (x < M) { if (k32) { const uint32_t f32 = fk32[x]; f = f32 == ~0u ? ~0ull : ((uint64_t)(f32 >> 12) << 24) | (f32 & 0xfffu); } else f = fk[x]; }
                const uint32_t k = x < M ? x / span : 0;
                sk[x] = f == ~0ull ? 0ull : ~(((uint64_t)k << 56) | f);          // bitonic_sort puts the largest "gain" first: the smallest (partition, first insertion)
                sp[x] = x < M ? x : 0xffffffffu;
                if (f != ~0ull) atomicAdd(&s_cntk[k], 1u);
            }
            __syncthreads();
            OPT_TICK(10);    // (ARITH) first-insertion keys
            bitonic_sort(sk, sp, M2, tid, OPT_THREADS);
            }
            OPT_TICK(11);    // (ARITH) sort
            for (uint32_t k = wid; k < p; k += OPT_THREADS / 64) {            // a wavefront per partition: the table is driven by all 64 lanes (arith_kernel.h: FxWave)
                const uint32_t par = s_tpar[k];
                uint32_t* const olist = olist_of(par, k);
                if (s_keep[k]) {                                             // the accepted order stands: walk its list with the new counts
                    const uint32_t Dk = s_cnt2[par][k];
                    double ef = 0.0;
                    uint64_t good = 0;
                    for (uint32_t d0 = 0; d0 < Dk; d0 += 64) {
                        const bool have = d0 + lane < Dk;
                        const uint32_t posrel = have ? olist[d0 + lane] : 0u;
                        good += fold64(k, posrel, have, ef);
                    }
                    good = opt_wave_sum_u64(good);
                    if (lane == 0) { s_errf[k] = ef; s_goodq[k] = good; }
                    continue;
                }
                if (s_dirC[k]) {                                             // every key in its home bucket: the buckets in order, straight from the histogram
                    const uint32_t C = s_dirC[k], nk = s_nk[k], kmin = pos0 + s_kmin[k], kmax = pos0 + s_kmax[k];
                    double ef = 0.0;
                    uint64_t good = 0;
                    uint32_t written = 0;
                    for (uint32_t b0 = 0; b0 < (nk ? C : 0u); b0 += 64) {
                        const uint32_t b = b0 + lane;
                        const uint32_t key = kmin + ((b * FX_KINV32 - kmin) & (C - 1u));          // the one position of [kmin, kmin + C) whose home is bucket b
                        bool full = b < C && key <= kmax;
                        if (full) {
                            bool cov = false;
#pragma unroll
                            for (int al = 0; al < A; ++al) cov |= hist[(uint64_t)(key - pos0) * PA + k * A + al] != 0;
                            full = cov;
                        }
                        const uint32_t posrel = full ? key - pos0 : 0u;
                        const uint64_t fm = __ballot(full);
                        if (full) olist[written + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = posrel;
                        written += (uint32_t)__popcll(fm);
                        good += fold64(k, posrel, full, ef);
                    }
                    good = opt_wave_sum_u64(good);
                    if (lane == 0) { s_errf[k] = ef; s_goodq[k] = good; s_cnt2[par][k] = nk; s_ldir[par][k] = 1u; }
#ifdef FLORIA_PROF
                    if (tid == 0) { const unsigned long long _t = clock64(); atomicAdd(&g.prof[13], _t - t_last); }      // (ARITH) partition 0's walk by the home-bucket rule
#endif
                    continue;
                }
                uint32_t start = 0, D = 0;
                uint32_t* const row = fk32 + k * span;
                if (wave_sort) {
                    uint32_t mykey[16], rk[16];                                // lane l owns the keys l, l + 64, ...: four per sweep over the row
#pragma unroll
                    for (int j = 0; j < 16; ++j) { const uint32_t e = lane + 64u * j; mykey[j] = e < span ? row[e] : ~0u; rk[j] = 0; }
#pragma unroll
                    for (int j0 = 0; j0 < 16; j0 += 4) {
                        if (64u * j0 < span)
                            for (uint32_t x = 0; x < span; x += 4) {          // four keys per round trip (the sweep waits for LDS, not for the compares)
                                uint32_t o[4];
#pragma unroll
                                for (int u = 0; u < 4; ++u) o[u] = x + u < span ? row[x + u] : ~0u;
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int j = j0; j < j0 + 4; ++j) rk[j] += o[u] < mykey[j] ? 1u : 0u;
                            }
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) D += (uint32_t)__popcll(__ballot(mykey[j] != ~0u));
#pragma unroll
                    for (int j = 0; j < 16; ++j) if (mykey[j] != ~0u) row[rk[j]] = lane + 64u * j;          // (keys are distinct: (read, cell rank) of one first insertion each)
                } else {
                    for (uint32_t q = 0; q < k; ++q) start += s_cntk[q];
                    D = s_cntk[k];
                }
                const uint64_t fxb = g.fx_ctrl + g.fx_slot;
                uint8_t* gmem = g.fx_pool + ((uint64_t)blockIdx.x * p + k) * 2 * fxb;
                uint8_t* c0 = g.fx_lds_off ? smem + g.fx_lds_off + (uint64_t)k * 2 * g.fx_ctrl : gmem;
                uint8_t* spare_c = g.fx_lds_off ? c0 + g.fx_ctrl : gmem + fxb;
                uint32_t* spare_s = (uint32_t*)(gmem + fxb + g.fx_ctrl);
                uint32_t* const tag = (uint32_t*)(smem + g.fx_lds_off + (uint64_t)p * 2 * g.fx_ctrl) + k * g.fx_tags;        // (LDS tables only)
                FxWave t;
                t.hbm = g.fx_lds_off == 0; t.tag_mask = g.fx_tags - 1u;
                if (D) { if (g.fx_lds_off) t.bind_lds(c0, (uint32_t*)(gmem + g.fx_ctrl), fx_buckets_for(1), lane); else t.bind(c0, (uint32_t*)(gmem + g.fx_ctrl), fx_buckets_for(1), lane); }
                for (uint32_t d0 = 0; d0 < D; d0 += 64) {                    // 64 positions of the sorted first-insertion list at a time
                    const uint32_t mine = d0 + lane < D ? (wave_sort ? row[d0 + lane] : sp[start + d0 + lane] - k * span) + pos0 : 0u;
                    const uint32_t cnt = D - d0 < 64u ? D - d0 : 64u;
                    if (g.fx_lds_off) t.template insert_batch<true>(mine, lane < cnt, lane, cnt, tag, spare_c, spare_s, lane);
                    else for (uint32_t l = 0; l < cnt; ++l) t.insert_new((uint32_t)__shfl((int)mine, (int)l), spare_c, spare_s, lane);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");          // the keys (HBM scratch) were stored lane by lane; the walk reads them bucket by bucket
#ifdef FLORIA_PROF
                if (tid == 0) { const unsigned long long _t = clock64(); atomicAdd(&g.prof[13], _t - t_last); }      // (ARITH) partition 0's replay
#endif
                // the buckets in order, 64 at a time: every lane prepares its bucket's terms, then they are added one by one
                double ef = 0.0;
                uint64_t good = 0;
                uint32_t written = 0;
                for (uint32_t i0 = 0; i0 < (D ? t.buckets : 0u); i0 += 64) {
                    const bool in = i0 + lane < t.buckets;
                    const bool full = in && !(t.ctrl[in ? i0 + lane : 0] & 0x80);
                    const uint32_t key = t.slot[in ? i0 + lane : 0];
                    const uint32_t posrel = full ? key - pos0 : 0u;
                    const uint64_t fm = __ballot(full);
                    if (full) olist[written + __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u))] = posrel;
                    written += (uint32_t)__popcll(fm);
                    good += fold64(k, posrel, full, ef);
                }
                good = opt_wave_sum_u64(good);
                if (lane == 0) { s_errf[k] = ef; s_goodq[k] = good; s_cnt2[par][k] = D; s_ldir[par][k] = 0u; }
            }
            // the next round's distances on the wavefronts without a map to replay (dist_arith); with as many partitions as wavefronts, by everybody afterwards
            if (with_dist) dist_arith(dist_inc);          // (the wavefronts without a map start here at once, the others when their replay and walk are done)
            __syncthreads();
            OPT_TICK(12);    // (ARITH) replay + walk, slowest partition
            if (tid == 0 && phred) {
                double sc = 0.0;
                for (uint32_t k = 0; k < p; ++k) sc += s_errf[k];
                s_score = sc * -1.0;
            }
            __syncthreads();
        };
        auto stats = [&](bool phred, bool first, bool reuse, bool with_dist = false, bool dist_inc = false) __attribute__((always_inline)) { if constexpr (ARITH) mec_stats_arith(phred, first, reuse, with_dist, dist_inc); else mec_stats(phred); };
        // ARITH: the lists of the partition just scored become the accepted ones
        auto accept_lists = [&]() __attribute__((always_inline)) { if constexpr (ARITH) { __syncthreads(); if ((uint32_t)tid < p) s_bpar[tid] = s_tpar[tid]; __syncthreads(); } };

        refresh_codes(false);
        OPT_TICK(0);     // build
        bool not_empty = n > 0;                                 // :76-85 (a job always has reads)
        uint32_t iters_done = 0;
        if (not_empty) {
            stats(true, true, false, p > 1, false);
            accept_lists();
            OPT_TICK(1);     // first stats
            double prev_score = s_score;
            for (int it = 0; it < NUM_ITER_OPTIMIZE; ++it) {   // :105-127
                iters_done = it + 1;
                if (p == 1) break;      // one partition: opt_iterate has no target (j != i), new_part == best_part, not accepted
                // ---- opt_iterate (:292-358): distance of every read to every partition ------------------------
                // 16 lanes per read: every lane classifies its cells against all p partitions (p x 16 B of one histogram row),
                // packed partial (diff Q24 << 16 | #eps) per partition, DPP row all-reduce, lane k of the row stores
#ifdef FLORIA_OPT_FULL_DIST
                const bool incremental = false;
#else
                const bool incremental = HL && meta && it > 0 && span <= 65535u;
#endif
                const uint32_t chg_lo = incremental ? s_chg_lo : 0u, chg_hi = incremental ? s_chg_hi : 0xffffffffu;
                if constexpr (ARITH) { /* the distances of this round were computed beside the last statistics call (dist_arith) */                } else
                for (uint32_t iv = grp; iv < n16; iv += OPT_THREADS / 16) {
                    const uint32_t i = (m_pm && iv < n) ? (uint32_t)m_pm[iv] : iv;
                    uint32_t cb = 0, len = 0, kk = 0;
                    if (i < n) read_meta(i, cb, len, kk);
                    if (incremental && i < n) {             // no code changed at any position of this read: its distances stand
                        const uint32_t fl = m_fl[i];
                        if ((fl >> 16) < chg_lo || (fl & 0xffffu) > chg_hi) len = 0;
                    }
                    uint64_t acc[KMAX];
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) acc[k] = 0;
                    for (uint32_t c0 = sub; c0 < len; c0 += 16 * 4) {
                        uint32_t sn[4], aqs[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const uint32_t c = c0 + 16 * u; const bool v = c < len; sn[u] = v ? G(cd.cell_snp)[cb + c] : 0; aqs[u] = v ? G(cd.cell_aw)[cb + c] : 0xffffffffu; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (aqs[u] != 0xffffffffu) {
                                const uint32_t al = aqs[u] >> 28;
                                const uint64_t w = (aqs[u] & 0x0fffffffu);
                                if (HL) {
                                    const uint8_t* crow = codes + (sn[u] - pos0) * p;
#pragma unroll
                                    for (int k = 0; k < KMAX; ++k) {
                                        if ((uint32_t)k < p) {
                                            const uint32_t c = crow[k];
                                            acc[k] += c == 0 ? 1ull : (((c >> al) & 1u) ? 0ull : (w << 16));
                                        }
                                    }
                                } else {
                                    const uint64_t* row = hist + (uint64_t)(sn[u] - pos0) * PA;
#pragma unroll
                                    for (int k = 0; k < KMAX; ++k) {
                                        if ((uint32_t)k < p) {
                                            uint64_t mx = 0, va = 0;
#pragma unroll
                                            for (int x = 0; x < A; ++x) { const uint64_t q = row[k * A + x] & QMASK44; mx = q > mx ? q : mx; va = (x == (int)al) ? q : va; }
                                            acc[k] += mx == 0 ? 1ull : ((va != mx) ? (w << 16) : 0ull);
                                        }
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        if ((uint32_t)k < p) {
                            const uint64_t t = row16_sum_u64(acc[k]);
                            if (sub == (uint32_t)k && i < n && len != 0) dist[i * p + k] = qm_to_f64(t >> 16, t & 0xffff, g.eps);
                        }
                    }
                }
                OPT_TICK(2);     // dist
                if (tid == 0) { s_ncand = 0; s_nmoves = 0; }
                for (uint32_t x = tid; x < (n + 31) / 32; x += OPT_THREADS) s_moved[x] = 0;
                __syncthreads();
                for (uint32_t pair = tid; pair < n * p; pair += OPT_THREADS) {
                    const uint32_t i = pair / p, j = pair - i * p;
                    const uint32_t pi = part[i];
                    if (j == pi || s_size[pi] <= 1) continue;                   // :300-302, :309-311
                    const double diff_score = dist[i * p + pi] - dist[pair];    // :320
                    if (diff_score > 0.0) {
                        const uint32_t idx = atomicAdd(&s_ncand, 1u);
                        cgain[idx] = (uint64_t)__double_as_longlong(diff_score);
                        ckey[idx] = (pi << 28) | (i << 4) | j;
                    }
                }
                __syncthreads();
                const uint32_t M = s_ncand;
                OPT_TICK(3);     // candidates
                if (M == 0) break;          // new_part == best_part -> new_score == prev_score -> not accepted (:114-126)
                uint32_t M2 = 1;
                while (M2 < M) M2 <<= 1;
                const bool in_lds = M2 <= OPT_SORT_LDS;
                if (in_lds) {
                    for (uint32_t x = tid; x < M2; x += OPT_THREADS) { s_gain[x] = x < M ? cgain[x] : 0; s_key[x] = x < M ? ckey[x] : 0xffffffffu; }
                    __syncthreads();
                    bitonic_sort(s_gain, s_key, M2, tid, OPT_THREADS);          // best_moves.sort_by(desc) :330
                } else {
                    for (uint32_t x = M + tid; x < M2; x += OPT_THREADS) { cgain[x] = 0; ckey[x] = 0xffffffffu; }
                    __syncthreads();
                    bitonic_sort(cgain, ckey, M2, tid, OPT_THREADS);
                }
                OPT_TICK(4);     // sort
                // ---- serial application (:336-356) --------------------------------------------------------------
                if (tid == 0) {
                    uint32_t number_of_moves = M / 10;
                    if (number_of_moves == 0 && M > 0) number_of_moves = M / 3 + 1;
                    uint32_t nm = 0;
                    for (uint32_t mv = 0; mv < M; ++mv) {
                        const uint32_t key = in_lds ? s_key[mv] : ckey[mv];
                        const uint32_t i = key >> 28, rl = (key >> 4) & 0xffffffu, j = key & 15;
                        if (s_moved[rl >> 5] & (1u << (rl & 31))) continue;
                        if (s_size[i] == 1) continue;
                        s_size[j] += 1; s_size[i] -= 1;
                        s_moved[rl >> 5] |= 1u << (rl & 31);
                        moves[nm++] = (rl << 8) | (i << 4) | j;
                        if (mv > number_of_moves) break;
                    }
                    s_nmoves = nm;
                }
                __syncthreads();
                const uint32_t nm = s_nmoves;
                OPT_TICK(5);     // serial apply
                // apply to histogram + partition (direction +1), or undo (direction -1)
                auto apply_moves = [&](bool undo) {
                    for (uint32_t x = wid; x < nm; x += OPT_THREADS / 64) {
                        const uint32_t mvv = moves[x], rl = mvv >> 8;
                        uint32_t from = (mvv >> 4) & 15, to = mvv & 15;
                        if (undo) { uint32_t t = from; from = to; to = t; }
                        const uint32_t r = reads[rl];
                        if (ARITH && lane == 0 && meta) m_lk[rl] = (m_lk[rl] & 0xffffffu) | (to << 24);      // (the staged partition follows the moves: read_meta's k stays current)
                        const uint32_t cb = G(cd.read_off)[r], ce = G(cd.read_off)[r + 1];
                        for (uint32_t c = cb + lane; c < ce; c += 64) {
                            const uint32_t aq = G(cd.cell_aw)[c];
                            const unsigned long long d = (1ull << CNT_SHIFT) | (aq & 0x0fffffffu);
                            uint64_t* cp = hist + (uint64_t)(G(cd.cell_snp)[c] - pos0) * PA + (aq >> 28);
                            atomicAdd((unsigned long long*)(cp + from * A), 0ull - d);
                            atomicAdd((unsigned long long*)(cp + to * A), d);
                        }
                        if (lane == 0) part[rl] = (uint8_t)to;
                    }
                    __syncthreads();
                    refresh_codes(true);
                };
                apply_moves(false);
                OPT_TICK(6);     // moves
                stats(true, false, false, true, HL && meta && span <= 65535u);
                OPT_TICK(7);     // round stats
                const double new_score = s_score;
                if (new_score > prev_score) { prev_score = new_score; accept_lists(); }
                else {                                   // rejected: keep best_part / prev_hap_block
                    apply_moves(true);
                    if (tid == 0) for (uint32_t x = 0; x < nm; ++x) { const uint32_t mvv = moves[x]; s_size[(mvv >> 4) & 15] += 1; s_size[mvv & 15] -= 1; }
                    __syncthreads();
                    break;
                }
            }
        }
        // ---- get_mec_stats_epsilon_no_phred of the optimised partition (graph_processing.rs:156-162) -----------
        OPT_TICK(8);
        stats(false, false, true);          // (ARITH: the accepted partition's position maps were replayed by the statistics call that accepted it)
        OPT_TICK(62);    // final stats (slot 62: slots 9-15 also hold the beam kernels' step counters, which made this phase look like 2.7 Gcycles in earlier profiles)
        if (tid == 0) {
            double mecv = 0.0, na = 0.0;
            for (uint32_t k = 0; k < p; ++k) {
                const double good = (double)s_goodq[k];
                const double bad = ARITH ? s_errf[k] : (double)s_errq[k] + (double)s_errm[k] * g.eps;
                mecv += bad; na += good; na += bad;
            }
            // What OTHER workgroups of launches in flight read (the stop rule of a speculative stage, the waiting jobs of a tail launch) is stored with agent-scope atomics —
            // written through — and ordered by waiting for the stores: an agent-scope FENCE writes the whole L2 of the XCD back (and an acquire invalidates it), under the
            // beam kernels that live on its contents.  Everything else is read by later launches only (kernel boundary).
            const bool publish = g.stop_at != nullptr || g.release_tried;
            if (publish) {
                __hip_atomic_store(&g.mec[(uint64_t)b * g.max_ploidy + p - 1], mecv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g.num_alleles[(uint64_t)b * g.max_ploidy + p - 1], na, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                g.mec[(uint64_t)b * g.max_ploidy + p - 1] = mecv;
                g.num_alleles[(uint64_t)b * g.max_ploidy + p - 1] = na;
            }
            g.iters[(uint64_t)b * g.max_ploidy + p - 1] = iters_done;
            if (g.fuse_select) {                                  // == select_kernel below; only this workgroup touches block b in this launch
                const double expected = na * g.eps;                                                    // :196
                uint32_t best = p;
                bool stop = false;
                if (p > 1) {
                    const double mec_prev = g.mec[(uint64_t)b * g.max_ploidy + p - 2];
                    if ((mecv / mec_prev) < g.mec_threshold) { /* do nothing */ }
                    else if (g.stopping_heuristic) { best = p - 1; stop = true; }                     // :233-238
                    if (!stop && mecv < expected) stop = true;                                         // :240-243
                } else if (mecv < expected) stop = true;                                               // :247-250
                if (g.release_tried) {
                    // a beam launch of the next ploidy runs beside this launch (tail_overlap): it starts block b's job when it sees tried[b] = p, and reads blk_done[b] then
                    if (stop || p == g.max_ploidy) { __hip_atomic_store(&g.blk_done_w[b], (uint8_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); g.best_ploidy[b] = best; }
                    // blk_done / best_ploidy must be visible before tried: a workgroup-scope release emits no vmcnt wait on gfx950, so wait for the
                    // written-through stores explicitly — no L2 write-back, no invalidate
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&g.tried[b], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    if (stop || p == g.max_ploidy) { g.blk_done_w[b] = 1; g.best_ploidy[b] = best; }
                    g.tried[b] = p;
                }
            }
            if (g.stop_at) {
                // mec / num_alleles of (b, p) must have reached L2 before the ready bit is set: the sc1 stores above and the atomic below go to different addresses
                // (different channels) and a workgroup-scope release does not wait for them on gfx950 — an explicit wait for the store acknowledgements
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint32_t have = atomicOr(&g.ready[b], 1u << p) | (1u << p);
                // (the other ploidy's mec / num_alleles: agent-scope loads in stop_rule_fires, issued after the atomic has returned)
                for (uint32_t q = p; q <= p + 1 && q <= g.max_ploidy; ++q)
                    if (((have >> q) & 1u) && (q == 1 || ((have >> (q - 1)) & 1u)) && stop_rule_fires(g, b, q)) atomicMin(&g.stop_at[b], q);
            }
        }
    }
}

// ---- ploidy stop rule (graph_processing.rs:196-251), one thread per block, after ploidy p finished ----------
struct SelectArgs {
    const uint32_t* job_block;       // the non-empty blocks this launch decides (empty blocks keep best_ploidy = tried = 0)
    uint32_t n_jobs, ploidy, max_ploidy;
    int32_t  stopping_heuristic;
    double   eps, mec_threshold;     // threshold for THIS ploidy, computed on the host with libm pow (:204-220)
    double* mec;
    double* num_alleles;
    uint32_t* iters;
    uint8_t*  blk_done;
    uint32_t* best_ploidy;
    uint32_t* tried;
    uint32_t clear_from, stage_last; // last select of a speculative stage: forget what the stage computed beyond `tried`
};
__global__ void select_kernel(SelectArgs g) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= g.n_jobs) return;
    const uint32_t b = g.job_block[j];
    const uint32_t p = g.ploidy;
    if (!g.blk_done[b]) {
        const double mec_p = g.mec[(uint64_t)b * g.max_ploidy + p - 1];
        const double expected = g.num_alleles[(uint64_t)b * g.max_ploidy + p - 1] * g.eps;         // :196
        uint32_t best = p;
        bool stop = false;
        if (p > 1) {
            const double mec_prev = g.mec[(uint64_t)b * g.max_ploidy + p - 2];
            if ((mec_p / mec_prev) < g.mec_threshold) { /* do nothing */ }
            else if (g.stopping_heuristic) { best = p - 1; stop = true; }                         // :233-238
            if (!stop && mec_p < expected) stop = true;                                           // :240-243
        } else if (mec_p < expected) stop = true;                                                 // :247-250
        g.tried[b] = p;
        if (stop || p == g.max_ploidy) { g.blk_done[b] = 1; g.best_ploidy[b] = best; }
    }
    // a stage that ran several ploidies at once computed ploidies the reference never reaches for this block: the result
    // reports mec_vector entries of 0 for them, exactly as the one-ploidy-per-stage path leaves them
    if (g.clear_from && g.blk_done[b])
        for (uint32_t q = g.tried[b] + 1; q <= g.stage_last; ++q) {
            g.mec[(uint64_t)b * g.max_ploidy + q - 1] = 0.0; g.num_alleles[(uint64_t)b * g.max_ploidy + q - 1] = 0.0; g.iters[(uint64_t)b * g.max_ploidy + q - 1] = 0;
        }
}

// final partition of every read of every block at the chosen ploidy
__global__ void gather_kernel(uint32_t n_blocks, const uint64_t* blk_read_off, const uint32_t* best_ploidy,
                              const uint8_t* part_planes, uint64_t plane_stride, uint8_t* out) {
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) return;
    const uint32_t bp = best_ploidy[b];
    if (bp == 0) return;
    const uint8_t* src = part_planes + (uint64_t)(bp - 1) * plane_stride;
    for (uint64_t i = blk_read_off[b] + threadIdx.x; i < blk_read_off[b + 1]; i += blockDim.x) out[i] = src[i];
}

}  // namespace fl
