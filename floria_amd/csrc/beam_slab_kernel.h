// beam_slab_kernel.h — beam search over SHARED partition slabs (production path for ploidy*beam <= 63).
//
// Measured on the reference algorithm (profiles/r01_unique_partitions.txt): the beam is full of label-permuted
// mirrors of the same clustering — at ploidy 4 a step has ~41 (state, partition) pairs but only ~4 distinct
// partition histograms (ploidy 3: 23 vs 4; ploidy 2: 13 vs 6.5).  The reference recomputes the read<->haplotype
// distance for every pair (global_clustering.rs:74-91); here a partition histogram is a reference-shared SLAB:
//
//   * a state is p slab ids; a child inherits p-1 ids and gets ONE new slab version "old slab + read";
//     children that extend the same old slab share the new version (mirrors stay shared by construction, starting
//     from the single empty slab all p partitions of the root point to);
//   * distance_read_haplo_epsilon_empty (utils_frags.rs:32-75) runs once per LIVE slab, with 64/nlive lanes per
//     slab striding over the read's cells and a segmented butterfly to combine them;
//   * a new version is written in place when no survivor still inherits the old one, else to a free slab after
//     a copy of the live SNP window; adds happen once per distinct new version.
//
// Everything observable is unchanged: per-pair (same, diff), p-values, pruning, child scores, the 128-bit
// linear state hash for the duplicate test, the std::BinaryHeap order — bit-identical to beam_kernel.h.
// Slab layout: [pos][allele] u64 (16 B per SNP for biallelic data): consecutive cells of a read are consecutive
// 16-B pieces, so four cells share a 64-B line.  Next to the sums every slab keeps one CODE byte per position — bit a set <=> allele a
// attains the position's maximal sum, 0 <=> nothing observed — which is all the distance needs (`same` <=> bit of the read's allele,
// empty <=> 0): phase A reads 1 byte per (slab, cell) instead of 16, the sums are only touched by the read-modify-write that refreshes the
// code, by copies and by the window-exit hash terms.  (Pileups with q = 0 cells keep classifying from the sums: their presence bit is part of it.)
// NARROW sums (biallelic data without q = 0 cells, i.e. every BASELINE config): a sum is < n_reads * 2^24 < 2^40 (the host sends blocks of >= 65 536 reads
// down the wide / generic kernels), so it is kept as a u32 low word in the [pos][allele] plane (8 B per SNP: eight positions per 64-B line instead of four)
// and a u8 high byte in a second plane (2 B per SNP) that the read-modify-write reads but only writes on a carry.  The lines an add dirties halve:
// measured on BASELINE config 4, write-back 129.6 -> 73.1 M KiB and fetch 58.0 -> 43.3 M KiB per two S1 calls (HBM traffic 35 x -> 23 x the algorithmic
// bytes) at an unchanged step time (the inverse experiment, padding a position to 32 / 64 B, costs +13 % / +52 %: the step sits just left of the knee
// of L2 capacity).
#pragma once
#include "wave_util.h"

namespace fl {

constexpr int SLAB_TILE = 256;      // cells of a read staged per pass (x2 arrays x2 buffers x4 B of LDS)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_cvoid;

// segmented all-reduce (sum) over aligned groups of Gs = 2,4,8,16 lanes with DPP row operations
// log(sum_j exp(dx_j)) over the p lanes of a state, own term evaluated once, summed in the reference's order j = 0..p-1.  Runs in the few
// steps that the f32 screen cannot decide; kept out of line so that its f64 exp/log temporaries do not count against the kernel's
// steady-state register budget.
__device__ __attribute__((noinline)) double log_sum_exp_terms(double dx, int seg0, uint32_t p) {
    const double ex = exp(dx);
    double sum = 0.0;
    for (uint32_t j = 0; j < p; ++j) sum += shfl_f64(ex, seg0 + (int)j);
    return log(sum);
}

template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ uint64_t dpp_mov64(uint64_t v) { return ((uint64_t)dpp_mov<CTRL>((uint32_t)(v >> 32)) << 32) | dpp_mov<CTRL>((uint32_t)v); }
// Whole-wave reductions at the end of a job / of the kernel by DPP inside the rows of 16 lanes and four v_readlane across them.  (The __shfl_xor butterflies of common.h
// need six lane-address registers that hipcc computes before the job loop and keeps - spilled to scratch - until the loop is left.)
__device__ __forceinline__ double wave_min_f64_dpp(double v) {
    auto mn = [](double a, uint64_t bb) { const double b = __longlong_as_double((long long)bb); return b < a ? b : a; };
    v = mn(v, dpp_mov64<0xB1>((uint64_t)__double_as_longlong(v)));
    v = mn(v, dpp_mov64<0x4E>((uint64_t)__double_as_longlong(v)));
    v = mn(v, dpp_mov64<0x141>((uint64_t)__double_as_longlong(v)));
    v = mn(v, dpp_mov64<0x140>((uint64_t)__double_as_longlong(v)));
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    double r = __longlong_as_double((long long)rl64(b, 0));
    r = mn(r, rl64(b, 16)); r = mn(r, rl64(b, 32)); r = mn(r, rl64(b, 48));
    return r;
}
__device__ __forceinline__ uint32_t wave_sum_u32_dpp(uint32_t v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return rl32(v, 0) + rl32(v, 16) + rl32(v, 32) + rl32(v, 48);
}
// 1e300 built where it is used: as a plain constant it is a loop-invariant register pair that hipcc spills for the length of the job loop
__device__ __forceinline__ double huge_margin() {
    uint32_t lo, hi;
    asm volatile("v_mov_b32 %0, 0x8800759c" : "=v"(lo));
    asm volatile("v_mov_b32 %0, 0x7e37e43c" : "=v"(hi));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ uint64_t seg_sum_u64(uint64_t v, uint32_t Gs) {
    if (Gs >= 2) v += dpp_mov64<0xB1>(v);      // quad_perm [1,0,3,2]
    if (Gs >= 4) v += dpp_mov64<0x4E>(v);      // quad_perm [2,3,0,1]
    if (Gs >= 8) v += dpp_mov64<0x141>(v);     // row_half_mirror
    if (Gs >= 16) v += dpp_mov64<0x140>(v);    // row_mirror
    return v;
}
__device__ __forceinline__ uint32_t seg_sum_u32(uint32_t v, uint32_t Gs) {
    if (Gs >= 2) v += dpp_mov<0xB1>(v);
    if (Gs >= 4) v += dpp_mov<0x4E>(v);
    if (Gs >= 8) v += dpp_mov<0x141>(v);
    if (Gs >= 16) v += dpp_mov<0x140>(v);
    return v;
}
// x / d for 0 <= x < 2^20 without an integer division (a u32 division costs ~30 VALU instructions): the exact quotient of x + 0.5 is
// at least 0.5/d away from every integer, while the float error (1-ulp reciprocal, one product) is < 4e-7 * x/d < 0.42/d
__device__ __forceinline__ uint32_t div_small(uint32_t x, float rcp_d) { return (uint32_t)(((float)x + 0.5f) * rcp_d); }
// cache policy of the read-once / write-once streams of a job (the next read's cells, its record, the traceback rows): nt, so that they do not push the slabs'
// lines out of L2 (measured: 91.8 against 92.8 ms per resident step, twice)
constexpr int FLORIA_NT_AUX = 2;
constexpr int SLAB_NS_MAX = 512;
constexpr int SLAB_PAD_IDX = 64;     // dummy position indices behind a slot's slabs (narrow-sum layout)
constexpr int SLAB_LOW_P_MAX = 3, SLAB_WAVES_LOW_P = 4;
constexpr int slab_waves(int tp) { return (tp >= 2 && tp <= SLAB_LOW_P_MAX) ? SLAB_WAVES_LOW_P : SLAB_WAVES; }
// ARITH instances (the reference's running f64 sums, below): the distance terms of a step wait in LDS for the sequential folds — SLAB_TERM_CAP f64 per wave.
// Measured on config 4 at eps 0.04 (scripts/arith_ab.sh, ms per call): 768 terms / two waves per SIMD 172.5-174.6, 512 / three 163.3-163.4, 384 / three 168.1-168.2,
// 768 / three 167.2-167.4 -> 512 terms (five slabs of a 92-cell read per pass) and the register budget of three waves
#ifndef FLORIA_TERM_CAP
#define FLORIA_TERM_CAP 512
#endif
constexpr int SLAB_TERM_CAP = FLORIA_TERM_CAP;
#ifndef FLORIA_ARITH_BEAM_WAVES
#define FLORIA_ARITH_BEAM_WAVES 3
#endif
constexpr int SLAB_WAVES_ARITH = FLORIA_ARITH_BEAM_WAVES;
template <int N> struct IC { static constexpr int value = N; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) { if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); } }
// value of lane (segment base + J) for aligned segments of PS = 2 or 4 lanes: one DPP quad_perm move, no LDS crossbar round trip
template <int PS, int J> __device__ __forceinline__ uint32_t seg_get(uint32_t v) {
    if constexpr (PS == 2) return dpp_mov<J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6)>(v);
    else return dpp_mov<J | (J << 2) | (J << 4) | (J << 6)>(v);
}
template <int PS, int J> __device__ __forceinline__ uint64_t seg_get64(uint64_t v) { return ((uint64_t)seg_get<PS, J>((uint32_t)(v >> 32)) << 32) | seg_get<PS, J>((uint32_t)v); }
template <int PS, int J> __device__ __forceinline__ double seg_get_f64(double v) { return __longlong_as_double((long long)seg_get64<PS, J>((uint64_t)__double_as_longlong(v))); }
template <int PS, int J> __device__ __forceinline__ float seg_get_f32(float v) { return __uint_as_float(seg_get<PS, J>(__float_as_uint(v))); }
constexpr int SLAB_DUMMY_WORDS = 64 * FLORIA_MAX_ALLELES * 2 + 16;      // u32 words of per-slot scratch behind the traceback rows (host reserves them)
constexpr double PRUNE_SCREEN = 1e-3;     // >> the f32 screen's error bound (2e-5)
constexpr int SLAB_U = 6;      // 16-B slab loads in flight per lane (q = 0 pileups: classification from the sums)

// stable_binom_cdf_p_rev (utils_frags.rs:211-248) in f32 with the hardware reciprocal and log2: the SCREEN of the pruning test (the decisions it cannot make take
// the exact host-libm table).  a = k/n and 1 - a = (n - k)/n are formed from the integers, so neither loses bits next to 0 or 1; the two clamps are the reference's.
// |result - exact| <= BINOM_SCREEN_C * n for n <= 1024 and 1e-3 <= eps <= 0.2: measured 4.7e-6 * n at eps = 2^-5, 9.2e-6 * n at eps = 1e-3 with every rcp / log2
// result moved one ulp against the sign of the error (numpy emulation), and on the device over the whole table by floria_hip_selftest (tests/test_gpu_parity.py).
constexpr float BINOM_SCREEN_C = 2e-5f;
__device__ __forceinline__ float binom_screen_f32(uint32_t nn, uint32_t kk, float ln_eps, float ln_1meps, float eps_f, float rdiv_f) {
    const float n = (float)nn, k = (float)kk;
    const float rn = __builtin_amdgcn_rcpf(n);
    float a = k * rn, b = (n - k) * rn;
    if (kk == nn) { a = 0.9999999f; b = 1.0000000000287557e-07f; }      // (1.0 - 0.9999999 in f64)
    if (kk == 0) { a = 1e-7f; b = 0.9999999f; }
    const float la = __builtin_amdgcn_logf(a) * 0.693147180559945309f - ln_eps;
    const float lb = __builtin_amdgcn_logf(b) * 0.693147180559945309f - ln_1meps;
    float rel = a * la + b * lb;
    rel = a < eps_f ? -rel : rel;
    return nn ? -(n * rdiv_f) * rel : 0.f;
}

// floria_hip_selftest: max over the table of |f32 screen - host-libm table entry| / n (one thread per n, all k): the error bound the level-1 screen assumes
__global__ void binom_screen_selftest_kernel(const double* tab, uint32_t nmax, float ln_eps, float ln_1meps, float eps_f, float rdiv_f, double* out_max) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x + 1;
    double worst = 0.0;
    if (n <= nmax)
        for (uint32_t k = 0; k <= n; ++k) {
            const double e = fabs((double)binom_screen_f32(n, k, ln_eps, ln_1meps, eps_f, rdiv_f) - tab[n * (n + 1) / 2 + k]) / (double)n;
            worst = e > worst ? e : worst;
        }
    worst = wave_min_f64(-worst);                  // (max through the wave-min helper)
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)out_max, (unsigned long long)__double_as_longlong(-worst));     // non-negative doubles order like their bit patterns
}

// Level 2 of the pruning test (see phase B): exact p-values from the host-libm table, the f32 exp2 / log2 screen on their exact differences and, where that
// is not decisive, the f64 exp / log of the reference formula.  Out of line: reached in a few percent of the steps, and its f64 temporaries would otherwise count
// against the steady-state register budget.  The whole wave calls it (wave-uniform branch); TPS = compile-time ploidy of the DPP-segmented instances, else 0.
struct Prune2 { double min_margin; uint32_t pass, fallback, exact; };
template <int TPS>
__device__ __attribute__((noinline)) Prune2 prune_level2(uint32_t nn, uint32_t kk, bool act, int seg0, uint32_t p, const double* tab, uint32_t nmax, double eps, double div_factor,
                                                         double cutoff, double min_margin) {
    constexpr uint32_t PSC = TPS == 2 ? 2 : 4;
    Prune2 r; r.fallback = 0; r.exact = 0;
    double pv = 0.0;
    if (act) {
        if (nn <= nmax) pv = tab[nn * (nn + 1) / 2 + kk];
        else { pv = binom_device(nn, kk, eps, div_factor); r.fallback = 1; }
    }
    double mx = 0.0;
    if constexpr (TPS != 0) static_for<0, TPS>([&](auto J) { constexpr int j = decltype(J)::value; const double o = seg_get_f64<PSC, j>(pv); mx = (j == 0) ? o : (o > mx ? o : mx); });
    else for (uint32_t j = 0; j < p; ++j) { const double o = shfl_f64(pv, seg0 + (int)j); mx = (j == 0) ? o : (o > mx ? o : mx); }
    const double dx = pv - mx;
    const float ef2 = __builtin_amdgcn_exp2f((float)dx * 1.44269504088896341f);
    float sumf2 = 0.f;
    if constexpr (TPS != 0) static_for<0, TPS>([&](auto J) { sumf2 += seg_get_f32<PSC, decltype(J)::value>(ef2); });
    else for (uint32_t j = 0; j < p; ++j) sumf2 += __shfl(ef2, seg0 + (int)j);
    const double dscr = (dx - (double)(__builtin_amdgcn_logf(sumf2) * 0.693147180559945309f)) - cutoff;
    const double ascr = fabs(dscr);
    const bool far_enough = ascr >= PRUNE_SCREEN && ascr - PRUNE_SCREEN >= min_margin;     // false for NaN
    bool pass = dscr > 0.0;
    if (__any(act && !far_enough)) {
        r.exact = 1;
        const double lse = mx + log_sum_exp_terms(dx, seg0, p);
        const double am = fabs((pv - lse) - cutoff);
        if (act) min_margin = am < min_margin ? am : min_margin;
        pass = (pv - lse) > cutoff;
    }
    r.pass = pass ? 1u : 0u; r.min_margin = min_margin;
    return r;
}

// A kernel-argument field read from the KERNARG SEGMENT at its point of use (one scalar load) instead of from the by-value copy that hipcc loads into SGPRs at
// kernel entry and then keeps alive - or spills into VGPR lanes - across the whole step loop.  For the fields a job touches once (queue, block tables, outputs)
// or a step touches rarely (binomial table, diagnostics): the step loop of beam_slab_kernel had 70-100 SGPR spill moves per step before (round 4).
template <class T> __device__ __forceinline__ T kernarg_at(size_t off) {
    return *(const __attribute__((address_space(4))) T*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
}
#define GCOLD(f) kernarg_at<decltype(BeamArgs::f)>(offsetof(BeamArgs, f))
#define GCOLD_BS(f) kernarg_at<decltype(BlockSet::f)>(offsetof(BeamArgs, bs) + offsetof(BlockSet, f))

struct SlabLds {
    uint32_t off_coff, off_caw, off_crp1, off_crp2;
    uint32_t off_q[2], off_h1[2], off_h2[2], off_m[2], off_sl[2];     // state arrays (SoA) x2
    uint32_t off_live, off_ref, off_leader, off_newid, off_free, off_pk;
    uint32_t off_rqs, off_rqd, off_rm, off_rt1, off_rt2, off_rnp1, off_rnp2;
    uint32_t off_ev[2], off_term;     // ARITH: error_vec of every state (p f64, x2 parities), the step's distance terms
    uint32_t total;
};
__host__ __device__ inline SlabLds slab_lds_layout(uint32_t LM, uint32_t p, bool q0, bool arith = false) {
    SlabLds L;
    const uint32_t NS = LM * p;
    uint32_t o = 0;
    auto take = [&](uint32_t bytes) { uint32_t r = o; o += (bytes + 15) & ~15u; return r; };
    // cells of a read (first tile: raw SNP index and attribute word), double-buffered: read i+1 arrives by LDS-DMA during step i
    L.off_coff = take(2 * SLAB_TILE * 4); L.off_caw = take(2 * SLAB_TILE * 4);
    L.off_crp1 = take(q0 ? SLAB_TILE * 8 : 0); L.off_crp2 = take(q0 ? SLAB_TILE * 8 : 0);
    for (int i = 0; i < 2; ++i) {
        L.off_q[i] = take((LM + 1) * 8); L.off_h1[i] = take((LM + 1) * 8); L.off_h2[i] = take((LM + 1) * 8); L.off_m[i] = take((LM + 1) * 4);   // +1: see the entry table of phase B
        L.off_sl[i] = take(NS * 2);
    }
    L.off_live = take(NS * 2);
    L.off_free = take(64 * 2); L.off_pk = take(64 * 4);
    L.off_rqs = take(NS * 8); L.off_rqd = take(NS * 8); L.off_rm = take(NS * 4);
    L.off_rt1 = 0; L.off_rt2 = 0;                    // the window-exit hash terms live in HBM scratch (needed in ~1/4 of the steps)
    L.off_rnp1 = take(q0 ? NS * 8 : 0); L.off_rnp2 = take(q0 ? NS * 8 : 0);
    // the materialisation tables are only live between phase B and the next phase A: they alias the r_qs array
    // (leader 4 B + newid 2 B + ref 1 B = 7 B per slab <= 8 B)
    L.off_leader = L.off_rqs; L.off_newid = L.off_rqs + NS * 4; L.off_ref = L.off_rqs + NS * 6;
    L.off_ev[0] = take(arith ? LM * p * 8 : 0); L.off_ev[1] = take(arith ? LM * p * 8 : 0);
    L.off_term = take(arith ? SLAB_TERM_CAP * 8 : 0);
    L.total = o;
    return L;
}

#ifdef FLORIA_PROF
#define BEAM_TICK(ph) do { const unsigned long long _t = clock64(); t_acc[ph] += _t - t_last; t_last = _t; } while (0)
#else
#ifdef FLORIA_MARK
#define BEAM_TICK(ph) asm volatile("; ====PHASE_END " #ph)
#else
#define BEAM_TICK(ph) do {} while (0)
#endif
#endif

// TP / TB: ploidy and beam width as compile-time constants (0 = read them from the arguments): LDS offsets become immediates, the
// per-partition loops unroll and divisions by p turn into multiplications; the host picks TP = p, TB = 10 for the CLI's default beam.
// SPEC: the launch belongs to a speculative stage (stop_at is set): only that instance carries the checks that drop a job whose ploidy turned out not to be needed
// ARITH: the reference's own f64 arithmetic (floria_hip_set_option("arith", 1)) on the shared slabs.  What changes against the canonical (Q24, #eps) form:
//   * a read's cells are staged in the iteration order of Frag.positions (BeamArgs::cell_ord, arith_kernel.h: {SNP, attribute word} pairs, so the LDS image is
//     interleaved and "beyond the written window" is a test per cell, not a prefix);
//   * phase A classifies lane-parallel as always, but instead of summing it leaves one f64 TERM per (live slab, cell) in LDS — 0.0 same, eps empty / beyond the window,
//     w * 2^-24 different — and then ONE lane per live slab folds its row in cell order: `diff += ..` of utils_frags.rs:32-75 term by term (x + 0.0 == x, so the
//     `same` cells cost nothing but do not perturb the sum); `same` stays an exact integer sum;
//   * a state carries error_vec (global_clustering.rs:196-202) as p f64 in LDS, a child's score is their sum in partition order with the read's diff added to its
//     partition first; the survivors' vectors are the parents' with that one element replaced.
// Slabs, hash, heap, pruning screen, traceback: unchanged.  Biallelic or not; pileups with q = 0 cells classify from the sums (their presence bit is part of it) and leave
// the same terms behind.
template <int A, bool Q0, int TP = 0, int TB = 0, bool SPEC = false, bool ARITH = false>
// waves per SIMD: four for the ploidy 2 and 3 instances (126-128 VGPRs, < 10 KB of LDS per wave), SLAB_WAVES = 3 where LDS limits (ploidy >= 4, runtime-parameter
// instances).  Measured and left alone: five waves spill 17-29 VGPRs and starve the co-running optimise kernels.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ARITH ? 1 : slab_waves(TP), ARITH ? SLAB_WAVES_ARITH : slab_waves(TP))))
void beam_slab_kernel(BeamArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x;
    const uint32_t p = TP ? (uint32_t)TP : g.ploidy, B = TB ? (uint32_t)TB : g.beam, LM = p * B, NS = LM * p;
    const SlabLds LY = slab_lds_layout(LM, p, Q0, ARITH);
    // cells of the current / next read: two arrays (SNP index, attribute word) x two buffers; ARITH: ONE interleaved {SNP, attribute} image per buffer,
    // buffer 0 where the SNP arrays would be, buffer 1 where the attribute arrays would be (same bytes), indexed with stride CST = 2
    constexpr uint32_t CST = ARITH ? 2u : 1u;
    uint32_t* const c_snp_base = (uint32_t*)(smem + LY.off_coff);
    uint32_t* const c_aw_base  = ARITH ? c_snp_base + 1 : (uint32_t*)(smem + LY.off_caw);
    constexpr uint32_t CBUF = ARITH ? 2u * SLAB_TILE : (uint32_t)SLAB_TILE;      // words between the two buffers
    uint64_t* const c_rp1 = (uint64_t*)(smem + LY.off_crp1);       // q=0 pileups only: presence-hash words of the current tile
    uint64_t* const c_rp2 = (uint64_t*)(smem + LY.off_crp2);
    uint16_t* live_id = (uint16_t*)(smem + LY.off_live);
    uint8_t*  ref = (uint8_t*)(smem + LY.off_ref);
    uint32_t* leader = (uint32_t*)(smem + LY.off_leader);
    uint16_t* newid = (uint16_t*)(smem + LY.off_newid);
    uint16_t* freelist = (uint16_t*)(smem + LY.off_free);
    uint32_t* s_pk = (uint32_t*)(smem + LY.off_pk);
    uint64_t* r_qs = (uint64_t*)(smem + LY.off_rqs);
    uint64_t* r_qd = (uint64_t*)(smem + LY.off_rqd);
    uint32_t* r_m = (uint32_t*)(smem + LY.off_rm);
    uint64_t* r_np1 = (uint64_t*)(smem + LY.off_rnp1);
    uint64_t* r_np2 = (uint64_t*)(smem + LY.off_rnp2);
    double* const r_fd = (double*)r_qd;                             // ARITH: running `diff` of the read against every live slab (the (Q24, #eps) tables are not used)
    double* const terms = (double*)(smem + LY.off_term);

    constexpr bool CODES = !Q0;
    constexpr bool NARROW = CODES && A == 2;
    const uint32_t pos_bytes = NARROW ? 8 : A * 8;
    const uint32_t span_pad = (g.span_max + 15u) & ~15u;
    // NARROW: the three planes of a slot — low words [idx][2] u32, high bytes [idx][2] u8, code bytes [idx] — share ONE position index idx = slab * span_pad +
    // position, followed by SLAB_PAD_IDX dummy indices (one per lane) for the branch-free tails of the add phase: an add computes idx once and addresses
    // all three with it, every access as pool (scalar) + 32-bit offset.  Otherwise: [NS slabs][span_max][A] u64, then [NS][span_pad] code bytes.
    const uint32_t slab_bytes = (NARROW ? span_pad : g.span_max) * pos_bytes;                // host guarantees the slot's bytes < 2^32
    char* pool = (char*)g.state_pool + (uint64_t)blockIdx.x * g.state_stride;
    const uint32_t n_idx = NS * span_pad + (uint32_t)SLAB_PAD_IDX;
    const uint32_t hi_base = NARROW ? n_idx * 8u : NS * slab_bytes;
    const uint32_t code_base = NARROW ? hi_base + n_idx * 2u : hi_base;
    uint8_t* const hi_plane = (uint8_t*)(pool + hi_base);
    const uint32_t hi_slab_bytes = NARROW ? span_pad * 2u : 0u;
    uint8_t* const codes = (uint8_t*)(pool + code_base);
    uint32_t* slot_hist = g.hist_pool + (uint64_t)blockIdx.x * g.hist_stride;
    uint64_t* r_t1 = (uint64_t*)(slot_hist + (g.hist_stride - 4ull * NS));      // tail of the slot's traceback region (host reserves it)
    uint64_t* r_t2 = r_t1 + NS;
    uint64_t* dummy = (uint64_t*)(slot_hist + (g.hist_stride - 4ull * NS - SLAB_DUMMY_WORDS));      // scratch for branch-free tails: 64 lanes x A sums, then 64 code bytes
    uint8_t* const dummy_code = (uint8_t*)(dummy + 64 * A);

    // phase B: lane = (state, partition).  The instances with a compile-time ploidy of 2..4 give a state an aligned group of PS = 2 / 4 / 4 lanes
    // (ploidy 3: every fourth lane idles), so that the sums and maxima over a state's partitions are DPP quad permutes instead of LDS shuffles
    constexpr bool DPPSEG = TP >= 2 && TP <= 4;
    constexpr uint32_t PSC = TP == 2 ? 2 : 4;
    const uint32_t psl = DPPSEG ? PSC : p;
    const uint32_t S = 64 / psl;
    const float rcp_p = __builtin_amdgcn_rcpf((float)p);
    const float eps_f = (float)g.eps, rdiv_f = (float)(1.0 / GCOLD(div_factor)), cutoff_f = (float)g.cutoff;
    const double margin_alone = fabs(0.0 - g.cutoff);      // |(p_k - lse) - ln 0.01| with p_k == lse
    const bool screen_ok = g.eps >= 1e-3 && g.eps <= 0.2;
    const uint32_t screen_nmax = GCOLD(binom_nmax);
    const uint32_t my_sl = lane / psl, my_k = lane % psl;
    const bool lane_pair = my_sl < S && my_k < p;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const uint64_t rk1 = c_rk1[my_k], rk2 = c_rk2[my_k];
    const int seg0 = (int)(my_sl * psl);
    double min_margin = 0.0;
    uint32_t n_fallback = 0;
#ifdef FLORIA_PROF
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = clock64();
    const unsigned long long t_wall0 = wall_clock64(), t_core0 = t_last;
    uint32_t c_pass = 0, c_push = 0, c_pop = 0;
    unsigned long long c_nlive = 0, c_nin = 0, c_nstates = 0, c_L = 0, c_copy_pos = 0, c_ncopy = 0, c_add_items = 0, c_zero_items = 0, c_nlead = 0, c_trunc = 0;
    unsigned long long c_nl = 0, c_id8 = 0, c_id16 = 0, c_id32 = 0, c_w128 = 0, c_w256 = 0, c_w512 = 0, c_wsum = 0, c_it64 = 0, c_it128 = 0, c_it256 = 0, c_lvl2 = 0, c_general = 0, c_exact = 0, c_boring = 0, c_heapkeep = 0, c_code = 0;
#endif

    bool gave_up = false;
    for (;;) {
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(GCOLD(queue_head), 1u);
        job = uni(__shfl(job, 0));
        if (job >= GCOLD(n_jobs)) break;
        const uint32_t b = uni(GCOLD(job_block)[job]);
        if (GCOLD(blk_done)[b]) continue;
        if (SPEC && GCOLD(wait_tried)) {
            // per-block dataflow (last ploidy stage): the optimise launch of the ploidy below runs beside this one and decides block b — done, or tried[b] = that
            // ploidy.  One lane polls (agent-scope loads, a sleep between two), bounded by the wall clock: a job that ran without being needed is ignored.
            const uint32_t want = GCOLD(wait_tried);
            const unsigned long long t0 = wall_clock64(), tmax = gave_up ? 0ull : (unsigned long long)GCOLD(wait_ticks);
            uint32_t st = 0;            // 1 = decided: go on, 2 = decided: done, 3 = waited in vain (from here on this wave does not wait)
            for (;;) {
                if (lane == 0) {
                    const uint32_t tr = __hip_atomic_load(&GCOLD(tried)[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (blk_done is stored, and acknowledged, before tried: optimize_kernel.h)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // ... and loaded after it: the tried load has returned before the blk_done load is issued
                    const uint32_t dn = __hip_atomic_load(&GCOLD(blk_done)[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    st = dn ? 2u : (tr >= want ? 1u : 0u);
                    if (st == 0 && wall_clock64() - t0 >= tmax) st = 3;
                }
                st = uni(__shfl(st, 0));
                if (st) break;
                __builtin_amdgcn_s_sleep(64);
            }
            if (st == 2) continue;
            if (st == 3) gave_up = true;
        }
        if (SPEC && GCOLD(stop_at) && uni(__hip_atomic_load(&GCOLD(stop_at)[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < p) continue;   // nobody will look at this ploidy of the block
        bool dropped = false;
        min_margin = huge_margin();                                 // per (block, ploidy) job: the host keeps the jobs the stop rule reached
        const ContigDev cd = GCOLD_BS(contigs)[GCOLD_BS(blk_contig)[b]];
        const uint64_t roff = GCOLD_BS(blk_read_off)[b];
        const uint32_t n = (uint32_t)(GCOLD_BS(blk_read_off)[b + 1] - roff);
        const uint32_t* reads = GCOLD_BS(blk_read) + roff;
        const uint32_t pos0 = GCOLD_BS(blk_pos0)[b];
        const uint2* const ord = ARITH ? GCOLD(cell_ord) + GCOLD(cell_ord_off)[GCOLD_BS(blk_contig)[b]] : nullptr;     // the contig's cells, every read's in set order

        int cur = 0;
        // (select between the two static carve-outs; indexing LY.off_*[cur] dynamically would put LY in scratch)
        auto ST_q = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_q[1] : LY.off_q[0])); };
        auto ST_h1 = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_h1[1] : LY.off_h1[0])); };
        auto ST_h2 = [&](int w) { return (uint64_t*)(smem + (w ? LY.off_h2[1] : LY.off_h2[0])); };
        auto ST_m = [&](int w) { return (uint32_t*)(smem + (w ? LY.off_m[1] : LY.off_m[0])); };
        auto ST_sl = [&](int w) { return (uint16_t*)(smem + (w ? LY.off_sl[1] : LY.off_sl[0])); };
        auto ST_ev = [&](int w) { return (double*)(smem + (w ? LY.off_ev[1] : LY.off_ev[0])); };
        uint32_t nstates = 1, nlive = 1;
        // root: every partition points at slab 0, which is logically empty (nothing written: hi_rel = -1)
        if (lane == 0) { ST_q(0)[0] = 0; ST_h1(0)[0] = 0; ST_h2(0)[0] = 0; ST_m(0)[0] = 0; live_id[0] = 0; }
        if (lane < p) ST_sl(0)[lane] = 0;
        if (ARITH && lane < p) ST_ev(0)[lane] = 0.0;              // error_vec: vec![(0.0, 0.0); ploidy] (global_clustering.rs:37)
        int32_t hi_rel = -1;
        uint32_t start_rel = 0;
        RegHeap H; H.hp_hi = 0; H.hp_lo = 0; H.hp_id = 0; H.len = 0;
        // software pipeline over reads (every request is issued in the shadow of phase A's slab loads):
        //   step i top:   LDS buffer i&1 holds read i's cells (raw snp / attribute words, written by LDS-DMA during step i-1),
        //                 SGPRs hold cell metadata (offset, length) of reads i, i+1 and step metadata of reads i, i+1
        //   step i, in A: LDS-DMA of read i+1's cells into buffer (i+1)&1 (two global_load_lds_dwordx4: no VGPRs, no ds_write),
        //                 scalar loads of the metadata of read i+2 and of the id of read i+3
        // Metadata travels through the VECTOR memory pipe (scalar loads share lgkmcnt with LDS and would stall every LDS wait of
        // the step): lanes 0..7 fetch the packed 32-B record of a read one step before it is needed, v_readlane turns it into SGPRs;
        // read ids come 64 at a time (lane j = reads[base + j]).
        struct CellMeta { uint32_t cbeg, L; };
        struct StepMeta { uint32_t first, last; uint64_t tw1, tw2; };
        auto load_rec = [&](uint32_t r) -> uint32_t { return lane < 8 ? (FLORIA_NT_AUX ? __builtin_nontemporal_load(G(cd.meta) + 8 * (uint64_t)r + lane) : G(cd.meta)[8 * (uint64_t)r + lane]) : 0u; };
        auto rec_cm = [&](uint32_t v) { CellMeta m; m.cbeg = rl32(v, 0); m.L = rl32(v, 1); return m; };
        auto rec_sm = [&](uint32_t v) { StepMeta m; m.first = rl32(v, 2); m.last = rl32(v, 3);
                                        m.tw1 = ((uint64_t)rl32(v, 5) << 32) | rl32(v, 4); m.tw2 = ((uint64_t)rl32(v, 7) << 32) | rl32(v, 6); return m; };
        // lane l moves cells 4l..4l+3 (16 B) of each array; the LDS image is lane-linear = cell order.  The last lane may read up to
        // 3 cells past the read (the next read's cells or the arrays' 16-B tail padding, see floria_hip_contig_upload); never used.
        auto dma_cells = [&](uint32_t w, const CellMeta& m) {
            if constexpr (ARITH) {                 // lane l moves the pairs of cells 2l, 2l+1 and 128+2l, 128+2l+1 (16 B each); the last lane may read one pair past the read (padding)
                if (m.L <= (uint32_t)SLAB_TILE) {
                    if (2 * lane < m.L) __builtin_amdgcn_global_load_lds((gbl_cvoid*)(G(ord) + m.cbeg + 2 * lane), (lds_void*)(c_snp_base + w * CBUF), 16, 0, FLORIA_NT_AUX);
                    if (128 + 2 * lane < m.L) __builtin_amdgcn_global_load_lds((gbl_cvoid*)(G(ord) + m.cbeg + 128 + 2 * lane), (lds_void*)(c_snp_base + w * CBUF + SLAB_TILE), 16, 0, FLORIA_NT_AUX);
                }
            } else
            if (m.L <= (uint32_t)SLAB_TILE && 4 * lane < m.L) {
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(G(cd.cell_snp) + m.cbeg + 4 * lane), (lds_void*)(c_snp_base + w * SLAB_TILE), 16, 0, FLORIA_NT_AUX);
                __builtin_amdgcn_global_load_lds((gbl_cvoid*)(G(cd.cell_aw) + m.cbeg + 4 * lane), (lds_void*)(c_aw_base + w * SLAB_TILE), 16, 0, FLORIA_NT_AUX);
            }
        };
        uint64_t rpb1 = 0, rpb2 = 0;
        uint32_t rid_vec = lane < n ? reads[lane] : 0;                  // ids of reads [0, 64)
        uint32_t rec_n2 = 0;                                             // record of read i+2 (in flight during step i)
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
            if (SPEC && (i & 63u) == 63u && GCOLD(stop_at) && uni(__hip_atomic_load(&GCOLD(stop_at)[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < p) { dropped = true; break; }
            const uint32_t cbeg = cm_cur.cbeg, L = cm_cur.L;
            const uint32_t first_rel = sm_cur.first - pos0;
            const int32_t  last_rel = (int32_t)(sm_cur.last - pos0);
            const uint64_t tw1 = sm_cur.tw1, tw2 = sm_cur.tw2;
            const uint32_t limit = i < (uint32_t)EARLY_READS ? LM : B;
            // level-1 screen of the pruning test (phase B): n <= L in every lane.  Its error bound is validated (floria_hip_selftest) for 1e-3 <= eps <= 0.2 and n within the
            // host-built table: outside of that every decision takes level 2 (an infinite tolerance: no lane is ever "far enough", no winner "alone")
            const float tol1 = screen_ok && L <= screen_nmax ? 4.f * BINOM_SCREEN_C * (float)L + 1e-3f : 3e38f;
            const uint32_t ntiles = (L + SLAB_TILE - 1) / SLAB_TILE;
            const int32_t new_hi = last_rel > hi_rel ? last_rel : hi_rel;
            uint64_t* st_q = ST_q(cur); uint64_t* st_h1 = ST_h1(cur); uint64_t* st_h2 = ST_h2(cur);
            uint32_t* st_m = ST_m(cur); uint16_t* st_sl = ST_sl(cur);
            uint32_t* const c_snp = c_snp_base + (i & 1) * CBUF;        // cell c: c_snp[c * CST], c_aw[c * CST]
            uint32_t* const c_aw  = c_aw_base + (i & 1) * CBUF;

            uint32_t nin = 0;
            // number of the tile's cells at written positions (<= hi_rel; a prefix, cells ascend) and, for q=0 pileups,
            // the presence-hash words of the cells (LDS) and their sum over the cells beyond hi_rel
            auto scan_tile = [&](uint32_t tl) {
                if constexpr (ARITH) {              // (set order: the cells inside the written window are not a prefix; phase A tests every cell)
                    if constexpr (Q0) {             // q = 0 pileups: the presence-hash words of the tile's cells (LDS), as below
#pragma unroll
                        for (int u = 0; u < SLAB_TILE / 64; ++u) {
                            const uint32_t c = lane + 64 * u;
                            if (c < tl) { const uint32_t hx = hash_idx(c_snp[c * CST], c_aw[c * CST] >> 28); c_rp1[c] = g.Rp1[hx]; c_rp2[c] = g.Rp2[hx]; }
                        }
                    }
                    return;
                }
                uint32_t cnt_in = 0;
                uint64_t b1 = 0, b2 = 0;
                uint32_t snps[SLAB_TILE / 64];
#pragma unroll
                for (int u = 0; u < SLAB_TILE / 64; ++u) { snps[u] = 0; if ((uint32_t)(64 * u) < tl) snps[u] = c_snp[lane + 64 * u]; }       // (stale words beyond tl are masked below)
#pragma unroll
                for (int u = 0; u < SLAB_TILE / 64; ++u) {
                    if ((uint32_t)(64 * u) >= tl) break;
                    const uint32_t c = lane + 64 * u;
                    const bool in = c < tl && (int32_t)(snps[u] - pos0) <= hi_rel;
                    if (Q0) {
                        if (c < tl) {
                            const uint32_t hx = hash_idx(snps[u], c_aw[c] >> 28);
                            const uint64_t r1 = g.Rp1[hx], r2 = g.Rp2[hx];
                            c_rp1[c] = r1; c_rp2[c] = r2;
                            if (!in) { b1 += r1; b2 += r2; }
                        }
                    }
                    cnt_in += (uint32_t)__popcll(__ballot(in));
                }
                nin = uni(cnt_in);
                if (Q0) { rpb1 = wave_sum_u64(b1); rpb2 = wave_sum_u64(b2); }
            };
            // multi-tile reads (L > SLAB_TILE) stage every tile from HBM inside the phases that walk the cells
            auto stage_tile = [&](uint32_t t) {
                __syncthreads();
#pragma unroll
                for (int u = 0; u < SLAB_TILE / 64; ++u) {
                    const uint32_t c = lane + 64 * u;
                    const uint32_t cc = t * SLAB_TILE + c;
                    if constexpr (ARITH) { if (cc < L) { const uint64_t ca = G((const uint64_t*)ord)[cbeg + cc]; c_snp[2 * c] = (uint32_t)ca; c_aw[2 * c] = (uint32_t)(ca >> 32); } }
                    else
                    if (cc < L) { c_snp[c] = G(cd.cell_snp)[cbeg + cc]; c_aw[c] = G(cd.cell_aw)[cbeg + cc]; }
                }
                __syncthreads();
                scan_tile(min((uint32_t)SLAB_TILE, L - t * SLAB_TILE));
                __syncthreads();
            };
            if (ntiles == 1) { scan_tile(L); if (Q0) __syncthreads(); }
            // issued ahead of phase A's slab loads (same in-order pipe, so no wait of its own): cells of read i+1 by LDS-DMA,
            // record of read i+2, and every 64 steps the next 64 read ids
            if (i + 1 < n) dma_cells((i + 1) & 1, cm_next);
            if (i + 2 < n) {
                if (((i + 2) & 63) == 0) {               // (the wait stays inside this branch: at the join hipcc would wait vmcnt(0) every step)
                    const uint32_t nv = (i + 2 + lane < n) ? reads[i + 2 + lane] : 0;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(rid_vec) : "v"(nv));
                }
                rec_n2 = load_rec(rl32(rid_vec, (i + 2) & 63));
            }
            BEAM_TICK(0);

            // ---- A: read vs every LIVE slab; Gs lanes per slab stride over the cells -------------------------------
#ifdef FLORIA_PROF
            c_nlive += nlive; c_nin += nin; c_nstates += nstates; c_L += L; c_code += (unsigned long long)nlive * (ARITH ? L : nin);
#endif
            const int32_t tend = (int32_t)first_rel - 1 < hi_rel ? (int32_t)first_rel - 1 : hi_rel;
            const bool trunc = tend >= (int32_t)start_rel;            // some written position leaves the hash window this step
            if constexpr (!CODES && ARITH) {
                // q = 0 pileups in the reference's arithmetic: classification from the sums (their presence bit is part of it), otherwise as the code-byte form below:
                // one f64 term per (live slab, cell) in LDS, one fold lane per slab; the exact side sums (same, presence hash of the read's new allele keys) by LDS atomics
                {   // (1) positions leaving the hash window
                    uint32_t Gs = 1, lgGs = 0;
                    while (Gs < 16 && nlive * (Gs * 2) <= 64) { Gs *= 2; ++lgGs; }
                    const uint32_t per = 64u >> lgGs;
                    if (trunc)
                    for (uint32_t l0 = 0; l0 < nlive; l0 += per) {
                        const uint32_t li = l0 + (lane >> lgGs), sub = lane & (Gs - 1);
                        const bool act = li < nlive;
                        const uint32_t slab_off = act ? (uint32_t)live_id[li] * slab_bytes : 0;
                        uint64_t t1 = 0, t2 = 0;
                        for (int32_t pr = (int32_t)start_rel + (int32_t)sub; pr <= tend; pr += (int32_t)Gs) {
                            if (act) {
                                uint64_t v[A];
                                const char* vp = pool + (slab_off + (uint32_t)pr * pos_bytes);
#pragma unroll
                                for (int h = 0; h < A / 2; ++h) { const ulonglong2 w2 = *(const ulonglong2*)(vp + 16 * h); v[2 * h] = w2.x; v[2 * h + 1] = w2.y; }
#pragma unroll
                                for (int al = 0; al < A; ++al) {
                                    const uint32_t hx = hash_idx(pos0 + (uint32_t)pr, (uint32_t)al);
                                    const uint64_t qv = Q0 ? (v[al] & QMASK63) : v[al];
                                    t1 += g.Rq1[hx] * qv; t2 += g.Rq2[hx] * qv;
                                    if (Q0) { t1 += v[al] ? g.Rp1[hx] : 0ull; t2 += v[al] ? g.Rp2[hx] : 0ull; }
                                }
                            }
                        }
                        t1 = seg_sum_u64(t1, Gs); t2 = seg_sum_u64(t2, Gs);
                        if (act && sub == 0) { const uint32_t sidr = live_id[li]; r_t1[sidr] = t1; r_t2[sidr] = t2; }
                    }
                }
                for (uint32_t x = lane; x < nlive; x += 64) { const uint32_t sidr = live_id[x]; r_qs[sidr] = 0; if (Q0) { r_np1[sidr] = 0; r_np2[sidr] = 0; } }
                for (uint32_t t = 0; t < ntiles; ++t) {
                    if (ntiles > 1) stage_tile(t); else if (Q0) __syncthreads();          // (the presence-hash words of scan_tile)
                    const uint32_t tl = min((uint32_t)SLAB_TILE, L - t * SLAB_TILE);
                    const uint32_t RS = (tl + 7u) & ~7u;
                    const uint32_t cap_s = div_small((uint32_t)SLAB_TERM_CAP, __builtin_amdgcn_rcpf((float)RS));
                    const uint32_t spp = min(nlive, min(64u, cap_s));
                    for (uint32_t l0 = 0; l0 < nlive; l0 += spp) {
                        const uint32_t ns = min(spp, nlive - l0);
                        const uint32_t Gl = div_small(64u, __builtin_amdgcn_rcpf((float)ns));
                        const float rcp_gl = __builtin_amdgcn_rcpf((float)Gl);
                        const uint32_t lsl = div_small(lane, rcp_gl), sub = lane - lsl * Gl;
                        const bool act = lsl < ns;
                        const uint32_t sidr = act ? (uint32_t)live_id[l0 + lsl] : 0u;
                        const uint32_t slab_off = sidr * slab_bytes;
                        double* const trow = terms + lsl * RS;
                        uint64_t qs = 0, np1 = 0, np2 = 0;
                        if (act) {
                            constexpr int N = 4;                                 // 16-B slab pieces in flight per lane
                            const uint32_t U = div_small(RS + Gl - 1u, rcp_gl);
                            for (uint32_t u0 = 0; u0 < U; u0 += N) {
                                uint32_t aws[N], cc[N]; bool vs[N], ins[N]; ulonglong2 vv[N][A / 2];
#pragma unroll
                                for (int u = 0; u < N; ++u) {
                                    cc[u] = sub + (u0 + (uint32_t)u) * Gl; vs[u] = cc[u] < tl; const uint32_t cx = vs[u] ? cc[u] : 0u;
                                    const uint32_t o = c_snp[cx * CST] - pos0; aws[u] = c_aw[cx * CST];
                                    ins[u] = vs[u] && (int32_t)o <= hi_rel;
                                    const char* cp = pool + (slab_off + (ins[u] ? o : 0u) * pos_bytes);
#pragma unroll
                                    for (int x = 0; x < A / 2; ++x) vv[u][x] = *(const ulonglong2*)(cp + 16 * x);
                                }
#pragma unroll
                                for (int u = 0; u < N; ++u) {
                                    const uint32_t al = aws[u] >> 28, w = aws[u] & 0x0fffffffu;
                                    uint64_t v[A], mx = 0, va = 0;
#pragma unroll
                                    for (int x = 0; x < A; x += 2) { v[x] = ins[u] ? vv[u][x / 2].x : 0ull; v[x + 1] = ins[u] ? vv[u][x / 2].y : 0ull; }
#pragma unroll
                                    for (int x = 0; x < A; ++x) { const uint64_t qx = Q0 ? (v[x] & QMASK63) : v[x]; mx = qx > mx ? qx : mx; va = (x == (int)al) ? v[x] : va; }
                                    const bool nonempty = mx != 0, same = nonempty && (Q0 ? (va & QMASK63) : va) == mx;
                                    qs += same ? w : 0u;
                                    if (Q0) { const bool np = vs[u] && !(va >> 63); np1 += np ? c_rp1[vs[u] ? cc[u] : 0u] : 0ull; np2 += np ? c_rp2[vs[u] ? cc[u] : 0u] : 0ull; }
                                    double tv = nonempty ? (double)w * 0x1p-24 : g.eps;
                                    tv = same ? 0.0 : tv;
                                    tv = vs[u] ? tv : 0.0;
                                    if (cc[u] < RS) trow[cc[u]] = tv;
                                }
                            }
                            atomicAdd((unsigned long long*)&r_qs[sidr], (unsigned long long)qs);
                            if (Q0) { atomicAdd((unsigned long long*)&r_np1[sidr], (unsigned long long)np1); atomicAdd((unsigned long long*)&r_np2[sidr], (unsigned long long)np2); }
                        }
                        __syncthreads();
                        if (lane < ns) {
                            const uint32_t sidf = live_id[l0 + lane];
                            double d = t == 0 ? 0.0 : r_fd[sidf];
                            const double2* row = (const double2*)(terms + lane * RS);
                            for (uint32_t c8 = 0; c8 < RS; c8 += 8u) {
                                const double2 t0 = row[0], t1 = row[1], t2 = row[2], t3 = row[3];
                                row += 4;
                                d += t0.x; d += t0.y; d += t1.x; d += t1.y; d += t2.x; d += t2.y; d += t3.x; d += t3.y;
                            }
                            r_fd[sidf] = d;
                        }
                        __syncthreads();
                    }
                }
            } else if constexpr (!CODES) {
                uint32_t Gs = 1, lgGs = 0;
                while (Gs < 16 && nlive * (Gs * 2) <= 64) { Gs *= 2; ++lgGs; }
                const uint32_t per = 64u >> lgGs;                   // slabs per pass
                for (uint32_t l0 = 0; l0 < nlive; l0 += per) {
                    const uint32_t li = l0 + (lane >> lgGs), sub = lane & (Gs - 1);
                    const bool act = li < nlive;
                    const uint32_t slab_off = act ? (uint32_t)live_id[li] * slab_bytes : 0;
                    uint64_t qs = 0, qd = 0, np1 = 0, np2 = 0, t1 = 0, t2 = 0;
                    uint32_t m = 0;
                    {   // positions leaving the hash window: [start_rel, first_rel) ∩ [.., hi_rel] (in ~45 % of the steps, 1-2 positions).  The
                        // multipliers depend on the position only, so they are requested together with the sums: one memory round trip, no
                        // branch on the loaded value (a zero sum contributes zero).
                        for (int32_t pr = (int32_t)start_rel + (int32_t)sub; pr <= tend; pr += (int32_t)Gs) {
                            if (act) {
                                uint64_t v[A], r1[A], r2[A], p1[A], p2[A];
                                const char* vp = pool + (slab_off + (uint32_t)pr * pos_bytes);
    #pragma unroll
                                for (int h = 0; h < A / 2; ++h) { const ulonglong2 w2 = *(const ulonglong2*)(vp + 16 * h); v[2 * h] = w2.x; v[2 * h + 1] = w2.y; }
    #pragma unroll
                                for (int al = 0; al < A; ++al) {
                                    const uint32_t hx = hash_idx(pos0 + (uint32_t)pr, (uint32_t)al);
                                    r1[al] = g.Rq1[hx]; r2[al] = g.Rq2[hx];
                                    if (Q0) { p1[al] = g.Rp1[hx]; p2[al] = g.Rp2[hx]; }
                                }
    #pragma unroll
                                for (int al = 0; al < A; ++al) {
                                    const uint64_t qv = Q0 ? (v[al] & QMASK63) : v[al];
                                    t1 += r1[al] * qv; t2 += r2[al] * qv;
                                    if (Q0) { t1 += v[al] ? p1[al] : 0ull; t2 += v[al] ? p2[al] : 0ull; }
                                }
                            }
                        }
                    }
                    uint32_t ps = 0, pd = 0;
                    auto cell = [&](const ulonglong2* vv, uint32_t aw, uint32_t c, bool valid) {
                        const uint32_t al = aw >> 28;
                        const uint32_t w = aw & 0x0fffffffu;
                        bool nonempty, same;
                        uint64_t va;
                        if (A == 2) {
                            const uint64_t v0 = Q0 ? (vv[0].x & QMASK63) : vv[0].x, v1 = Q0 ? (vv[0].y & QMASK63) : vv[0].y;
                            nonempty = (v0 | v1) != 0;
                            // same <=> the read's allele holds the larger-or-equal sum <=> equal sums, or (allele == 1) != (v1 < v0); integer form
                            // (sums are < 2^63, so the sign of the difference is the comparison) - hipcc turns the select form into 0/1 VGPR traffic
                            const uint64_t d = v1 - v0;
                            same = d == 0 || ((al ^ (uint32_t)(d >> 63)) & 1u) != 0;
                            va = Q0 ? (al ? vv[0].y : vv[0].x) : 0;
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
                        m += (valid && !nonempty) ? 1u : 0u;
                        if (Q0) { const bool np = valid && !(va >> 63); np1 += np ? c_rp1[c] : 0ull; np2 += np ? c_rp2[c] : 0ull; }
                    };
                    for (uint32_t t = 0; t < ntiles; ++t) {
                        if (ntiles > 1) stage_tile(t);
                        const uint32_t tl = min((uint32_t)SLAB_TILE, L - t * SLAB_TILE);
                        // SLAB_U independent 16-B loads in flight per lane; the loop is wave-uniform (invalid slots and idle lanes read
                        // cell 0 of a slab with weight 0: no branches), and the next read is staged behind the first batch of loads
                        if (act && !CODES) {
                            for (uint32_t c0 = sub; c0 < nin; c0 += SLAB_U * Gs) {
                                uint32_t offs[SLAB_U], aws[SLAB_U];
    #pragma unroll
                                for (int u = 0; u < SLAB_U; ++u) {
                                    const uint32_t c = c0 + u * Gs; const bool v = c < nin; const uint32_t cx = v ? c : 0;
                                    offs[u] = (c_snp[cx] - pos0) * pos_bytes; aws[u] = v ? c_aw[cx] : 0;
                                }
                                ulonglong2 vv[SLAB_U][A / 2];
    #pragma unroll
                                for (int u = 0; u < SLAB_U; ++u) {
                                    const char* cp = pool + (slab_off + offs[u]);
    #pragma unroll
                                    for (int x = 0; x < A / 2; ++x) vv[u][x] = *(const ulonglong2*)(cp + 16 * x);
                                }
                                ps = 0; pd = 0;
    #pragma unroll
                                for (int u = 0; u < SLAB_U; ++u) cell(vv[u], aws[u], c0 + u * Gs, c0 + u * Gs < nin);
                                qs += ps; qd += pd;
                            }
                        }
                        if (act && sub == 0) { m += tl - nin; if (Q0) { np1 += rpb1; np2 += rpb2; } }   // cells beyond hi_rel (:45-48)
                    }
                    // segmented all-reduce over the Gs lanes of a slab (DPP, no LDS crossbar)
                    qs = seg_sum_u64(qs, Gs); qd = seg_sum_u64(qd, Gs); m = seg_sum_u32(m, Gs);
                    if (trunc) { t1 = seg_sum_u64(t1, Gs); t2 = seg_sum_u64(t2, Gs); }
                    if (Q0) { np1 = seg_sum_u64(np1, Gs); np2 = seg_sum_u64(np2, Gs); }
                    if (act && sub == 0) {
                        const uint32_t sidr = live_id[li];
                        r_qs[sidr] = qs; r_qd[sidr] = qd; r_m[sidr] = m;
                        if (trunc) { r_t1[sidr] = t1; r_t2[sidr] = t2; }
                        if (Q0) { r_np1[sidr] = np1; r_np2[sidr] = np2; }
                    }
                }
            } else {
                // (1) positions leaving the hash window: [start_rel, first_rel) ∩ [.., hi_rel] — one step in six, 1-2 positions; Gs lanes per slab
                if (trunc) {
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
                                uint64_t v[A], r1[A], r2[A];
                                const char* vp = pool + (slab_off + (uint32_t)pr * pos_bytes);
                                if constexpr (NARROW) {
                                    const uint2 lo = *(const uint2*)vp;
                                    const uint32_t hb = *(const uint16_t*)(hi_plane + ((uint32_t)live_id[li] * hi_slab_bytes + (uint32_t)pr * 2u));
                                    v[0] = ((uint64_t)(hb & 0xffu) << 32) | lo.x; v[1] = ((uint64_t)(hb >> 8) << 32) | lo.y;
                                } else {
#pragma unroll
                                for (int h = 0; h < A / 2; ++h) { const ulonglong2 w2 = *(const ulonglong2*)(vp + 16 * h); v[2 * h] = w2.x; v[2 * h + 1] = w2.y; }
                                }
#pragma unroll
                                for (int al = 0; al < A; ++al) { const uint32_t hx = hash_idx(pos0 + (uint32_t)pr, (uint32_t)al); r1[al] = g.Rq1[hx]; r2[al] = g.Rq2[hx]; }
#pragma unroll
                                for (int al = 0; al < A; ++al) { t1 += r1[al] * v[al]; t2 += r2[al] * v[al]; }
                            }
                        }
                        t1 = seg_sum_u64(t1, Gs); t2 = seg_sum_u64(t2, Gs);
                        if (act && sub == 0) { const uint32_t sidr = live_id[li]; r_t1[sidr] = t1; r_t2[sidr] = t2; }
                    }
                }
                // (2, ARITH) the reference's running `diff` (utils_frags.rs:32-75): the lanes classify as below — Gl lanes per slab, batches of independent byte loads —
                // but leave one f64 term per (slab, cell) in LDS, rows of RS = tl rounded up to 8 (padding: 0.0), for as many live slabs as SLAB_TERM_CAP holds at a
                // time; then one lane per slab adds its row in cell order onto the slab's running sum (continued across the tiles of a long read).  `same` is an exact
                // integer sum as ever.
                if constexpr (ARITH) {
                    for (uint32_t x = lane; x < nlive; x += 64) r_qs[live_id[x]] = 0;
                    for (uint32_t t = 0; t < ntiles; ++t) {
                        if (ntiles > 1) stage_tile(t);
                        const uint32_t tl = min((uint32_t)SLAB_TILE, L - t * SLAB_TILE);
                        const uint32_t RS = (tl + 7u) & ~7u;
                        const uint32_t cap_s = div_small((uint32_t)SLAB_TERM_CAP, __builtin_amdgcn_rcpf((float)RS));          // rows that fit (>= 3: RS <= SLAB_TILE)
                        const uint32_t spp = min(nlive, min(64u, cap_s));
                        for (uint32_t l0 = 0; l0 < nlive; l0 += spp) {
                            const uint32_t ns = min(spp, nlive - l0);
                            const uint32_t Gl = div_small(64u, __builtin_amdgcn_rcpf((float)ns));
                            const float rcp_gl = __builtin_amdgcn_rcpf((float)Gl);
                            const uint32_t lsl = div_small(lane, rcp_gl), sub = lane - lsl * Gl;
                            const bool act = lsl < ns;
                            const uint32_t sidr = act ? (uint32_t)live_id[l0 + lsl] : 0u;
                            const uint8_t* const cbase = codes + sidr * span_pad;
                            double* const trow = terms + lsl * RS;
                            uint64_t qs = 0;
                            auto batch = [&](auto NC, uint32_t u0) {
                                constexpr int N = decltype(NC)::value;
                                uint32_t offs[N], aws[N], cdb[N], cc[N]; bool vs[N], ins[N];
#pragma unroll
                                for (int u = 0; u < N; ++u) {
                                    cc[u] = sub + (u0 + (uint32_t)u) * Gl; vs[u] = cc[u] < tl; const uint32_t cx = vs[u] ? cc[u] : 0u;
                                    const uint32_t o = c_snp[cx * CST] - pos0; aws[u] = c_aw[cx * CST];
                                    ins[u] = vs[u] && (int32_t)o <= hi_rel;                     // a position beyond the written window is not in the haplotype: empty (:36-48)
                                    offs[u] = ins[u] ? o : 0u;
                                }
#pragma unroll
                                for (int u = 0; u < N; ++u) cdb[u] = cbase[offs[u]];
                                uint32_t ps = 0;
#pragma unroll
                                for (int u = 0; u < N; ++u) {
                                    const uint32_t w = aws[u] & 0x0fffffffu;
                                    const uint32_t code = ins[u] ? cdb[u] : 0u;
                                    const uint32_t sm = (uint32_t)__builtin_amdgcn_sbfe((int)code, aws[u] >> 28, 1u);        // all ones <=> same
                                    ps += w & sm;
                                    double tv = code ? (double)w * 0x1p-24 : g.eps;             // :70 diff += w  |  :45-48 diff += epsilon
                                    tv = sm ? 0.0 : tv;                                         // :54-67 same: nothing is added to diff
                                    tv = vs[u] ? tv : 0.0;                                      // row padding
                                    if (cc[u] < RS) trow[cc[u]] = tv;
                                }
                                qs += ps;
                            };
                            if (act) {
                                const uint32_t U = div_small(RS + Gl - 1u, rcp_gl);             // rounds of Gl cells
#if defined(FLORIA_ARITH_BATCH) && FLORIA_ARITH_BATCH == 4
                                for (uint32_t u0 = 0; u0 < U; u0 += 4u) { if (U - u0 >= 3u) batch(IC<4>{}, u0); else batch(IC<2>{}, u0); }
#else
                                for (uint32_t u0 = 0; u0 < U; u0 += 8u) {
                                    const uint32_t r = U - u0;
                                    if (r >= 7u) batch(IC<8>{}, u0); else if (r >= 5u) batch(IC<6>{}, u0); else if (r >= 3u) batch(IC<4>{}, u0); else batch(IC<2>{}, u0);
                                }
#endif
                                if (L < (uint32_t)SLAB_TILE) atomicAdd((uint32_t*)&r_qs[sidr], (uint32_t)qs);         // (< 256 cells of weight <= 2^24: the sum stays below 2^32)
                                else atomicAdd((unsigned long long*)&r_qs[sidr], (unsigned long long)qs);
                            }
                            __syncthreads();
                            if (lane < ns) {
                                const uint32_t sidf = live_id[l0 + lane];
                                double d = t == 0 ? 0.0 : r_fd[sidf];
                                const double2* row = (const double2*)(terms + lane * RS);
                                for (uint32_t c8 = 0; c8 < RS; c8 += 8u) {
                                    const double2 t0 = row[0], t1 = row[1], t2 = row[2], t3 = row[3];
                                    row += 4;
                                    d += t0.x; d += t0.y; d += t1.x; d += t1.y; d += t2.x; d += t2.y; d += t3.x; d += t3.y;
                                }
                                r_fd[sidf] = d;
                            }
                            __syncthreads();
                        }
                    }
                } else {
                // (2) distances from the code bytes.  Gl = 64 / nlive lanes per slab (any integer, not a power of two: 6 live slabs get 10 lanes each, not 8),
                // every lane walks ceil(nin / Gl) cells in batches of 2 / 4 / 6 / 8 independent byte loads chosen by the exact count, sums in 32 bits inside a
                // batch, and the lanes of a slab combine through LDS atomics (3 instructions instead of a 4-stage DPP butterfly on three values)
                for (uint32_t x = lane; x < nlive; x += 64) { const uint32_t sidr = live_id[x]; r_qs[sidr] = 0; r_qd[sidr] = 0; r_m[sidr] = 0; }
#ifdef FLORIA_MW_A_SHIFT      // (experiment, profiles/r06_multiwave_ab.txt: phase A with 1 / 2^k of its lanes per slab - if the phase were bound by lanes, more waves per job would pay)
                const uint32_t Gl = nlive <= 64u ? max(1u, div_small(64u, __builtin_amdgcn_rcpf((float)nlive)) >> FLORIA_MW_A_SHIFT) : 1u;
#else
                const uint32_t Gl = nlive <= 64u ? div_small(64u, __builtin_amdgcn_rcpf((float)nlive)) : 1u;
#endif
                const float rcp_gl = __builtin_amdgcn_rcpf((float)Gl);
                for (uint32_t l0 = 0; l0 < nlive; l0 += 64u) {             // (one pass unless more than 64 slabs are live: then Gl == 1)
                    const uint32_t lsl = div_small(lane, rcp_gl), sub = lane - lsl * Gl, li = l0 + lsl;
                    const bool act = li < nlive;
                    const uint32_t sidr = act ? (uint32_t)live_id[li] : 0u;
                    const uint8_t* const cbase = codes + sidr * span_pad;
                    uint64_t qs = 0, qd = 0;
                    uint32_t m = 0;
                    auto batch = [&](auto NC, uint32_t u0) {
                        constexpr int N = decltype(NC)::value;
                        uint32_t offs[N], aws[N], cdb[N]; bool vs[N];
#pragma unroll
                        for (int u = 0; u < N; ++u) {
                            const uint32_t c = sub + (u0 + (uint32_t)u) * Gl; vs[u] = c < nin; const uint32_t cx = vs[u] ? c : 0u;
                            offs[u] = c_snp[cx] - pos0; const uint32_t awr = c_aw[cx]; aws[u] = vs[u] ? awr : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < N; ++u) cdb[u] = cbase[offs[u]];
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
                        // (the sums of a read of fewer than 256 cells stay below 2^32 — the weight of a cell is <= 2^24, with equality from q = 73 on — so the 32-bit
                        // atomic on the low word is the whole add; a read of exactly 256 cells of weight 1.0 would wrap: ADVICE r5)
                        if (L < (uint32_t)SLAB_TILE) { atomicAdd((uint32_t*)&r_qs[sidr], (uint32_t)qs); atomicAdd((uint32_t*)&r_qd[sidr], (uint32_t)qd); }
                        else { atomicAdd((unsigned long long*)&r_qs[sidr], (unsigned long long)qs); atomicAdd((unsigned long long*)&r_qd[sidr], (unsigned long long)qd); }
                        atomicAdd(&r_m[sidr], m);
                    }
                }
                }      // (!ARITH)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the LDS-DMA of the next read has landed (hipcc does not track it)
            // Pin the record's wait HERE, where nothing younger is in flight: consumed at the end of the step, hipcc would wait
            // vmcnt(0) there and stall on the acknowledgements of the step's slab stores.
            uint32_t rec_hold;
            asm volatile("v_mov_b32 %0, %1" : "=v"(rec_hold) : "v"(rec_n2));
            __syncthreads();
            BEAM_TICK(1);

            // ---- B: per (state, partition) pair: p-value, log-sum-exp, pruning, child (:74-134) -----------------------
            uint64_t evalid = 0;
            H.len = 0;
            bool bulk = false;                       // this step took the no-duplicate / no-eviction path
            bool fastm = false;                      // ... and the slab structure carries over unchanged (see phase M)
            bool heap_kept = false;                  // ... and so does the heap: state a's child sits in slot a
            uint32_t nlead_f = 0;
            uint64_t b_h1 = 0, b_h2 = 0;             // its children's state hashes (lane = (state, partition) pair)
            uint32_t src_map = 0;                    // lane r = child lane of entry r
            uint64_t* const E_s = ST_q(cur ^ 1); uint64_t* const E_h1 = ST_h1(cur ^ 1); uint64_t* const E_h2 = ST_h2(cur ^ 1);
            uint32_t* const E_pk = ST_m(cur ^ 1);
            for (uint32_t a0 = 0; a0 < nstates; a0 += S) {
                const uint32_t a = a0 + my_sl;
                const bool act = lane_pair && a < nstates;
                uint64_t qd = 0, t1 = 0, t2 = 0, np1 = 0, np2 = 0;
                uint32_t m = 0;
                uint32_t nn = 0, kk = 0;
                float pvf = 0.f;
                double df = 0.0, e_own = 0.0;            // ARITH: the read's running diff against this lane's slab; error_vec[my_k].1 of state a
                if (act) {
                    const uint32_t li = st_sl[a * p + my_k];          // (the per-slab tables of phase A are indexed by slab id)
                    const uint64_t qs = r_qs[li];
                    if constexpr (ARITH) { df = r_fd[li]; e_own = ST_ev(cur)[a * p + my_k]; }
                    else { qd = r_qd[li]; m = r_m[li]; }
                    if (trunc) { t1 = r_t1[li] * rk1; t2 = r_t2[li] * rk2; }
                    if (Q0) { np1 = r_np1[li]; np2 = r_np2[li]; }
                    const double same_f = qm_to_f64(qs, 0, g.eps), diff_f = ARITH ? df : qm_to_f64(qd, m, g.eps);
                    // `as usize` of the two sums: both are < 2^32 (a read has < 2^32 cells of weight <= 1), so the 1-instruction u32 conversion is exact
                    nn = (uint32_t)(same_f + diff_f); kk = (uint32_t)diff_f;
                    pvf = binom_screen_f32(nn, kk, g.ln_eps, g.ln_1meps, eps_f, rdiv_f);
                }
                uint64_t ts1 = 0, ts2 = 0;
                if constexpr (DPPSEG) { if (trunc) static_for<0, TP>([&](auto J) { constexpr int j = decltype(J)::value; ts1 += seg_get64<PSC, j>(t1); ts2 += seg_get64<PSC, j>(t2); }); }
                else if (trunc) for (uint32_t j = 0; j < p; ++j) { ts1 += shfl_u64(t1, seg0 + (int)j); ts2 += shfl_u64(t2, seg0 + (int)j); }
                // Pruning test (pv - lse) > ln 0.01 and its margin, decided in two screens.
                // Level 1, every step, no memory: pv from an f32 evaluation of stable_binom_cdf_p_rev (hardware rcp / log2; |error| <= BINOM_SCREEN_C * n, n <= L,
                // checked against the host table by floria_hip_selftest), log-sum-exp in f32.  |d_f32 - d_exact| <= tol1 (log-sum-exp is 1-Lipschitz in the
                // max norm of its arguments): a decision further than tol1 from the threshold AND from the job's running minimum margin has the same outcome and
                // cannot lower the minimum.  Level 2, only when some lane is closer: the exact p-values from the host-libm table (one gather), then the round-3
                // screen on them (f32 exp2 / log2 of exact differences) and, where that is not decisive either, the f64 exp / log of the reference formula.
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
                    // The commonest decision of all — ONE partition explains the read and every other is more than 40 log units behind — is known exactly
                    // without any of this: the other terms of the log-sum-exp are below e^-40, their sum is below 2^-53 for p <= 16, so the f64 sum of the
                    // reference formula is exactly 1.0, lse == max, the winner's margin is |0 - ln 0.01| to the bit and the losers' margins exceed 35.  (Left
                    // to the screen, that winner would tie with the job's running minimum margin - the same constant - and force the exact path every step.)
                    const uint64_t closeb = __ballot(act && dxf > -(40.f + tol1));
                    const uint32_t segbits = (uint32_t)(closeb >> seg0) & ((1u << psl) - 1u);
                    const bool alone = act && (segbits & (segbits - 1u)) == 0u;           // (an active lane's segment has at least its maximum in the set)
                    if (alone) min_margin = margin_alone < min_margin ? margin_alone : min_margin;
                    const bool far1 = alone || (asf > 0.f && (double)asf >= min_margin);     // false for NaN
                    pass = dsf > 0.f;
#ifdef FLORIA_NO_BINOM_SCREEN
                    if (true) {
#else
                    if (__any(act && !far1)) {
#endif
#ifdef FLORIA_PROF
                        c_lvl2++;
#endif
                        const Prune2 r2 = prune_level2<DPPSEG ? TP : 0>(nn, kk, act, seg0, p, GCOLD(binom_tab), GCOLD(binom_nmax), g.eps, GCOLD(div_factor), g.cutoff, min_margin);
                        pass = r2.pass != 0; min_margin = r2.min_margin; n_fallback += r2.fallback;
#ifdef FLORIA_PROF
                        c_exact += r2.exact;
#endif
                    }
                }
                pass = pass && act;
                uint64_t ch1 = 0, ch2 = 0, cq = 0, cs = 0;
                uint32_t cm = 0;
                if constexpr (ARITH) {
                    // read_to_node_value (global_clustering.rs:196-202): error_vec with the read's diff added to its partition, summed in partition order
                    const double ed = e_own + df;
                    double mec = 0.0;
                    if constexpr (DPPSEG) static_for<0, TP>([&](auto J) { constexpr int j = decltype(J)::value; const double ej = seg_get_f64<PSC, j>(e_own); mec += ((uint32_t)j == my_k) ? ed : ej; });
                    else for (uint32_t j = 0; j < p; ++j) { const double ej = shfl_f64(e_own, seg0 + (int)j); mec += (j == my_k) ? ed : ej; }
                    cs = (uint64_t)__double_as_longlong(mec);
                }
                if (act) {
                    if constexpr (!ARITH) {
                    cq = st_q[a] + qd;
                    cm = st_m[a] + m;
                    cs = (uint64_t)__double_as_longlong(qm_to_f64(cq, cm, g.eps));
                    }
                    ch1 = (st_h1[a] - ts1) + rk1 * (tw1 + (Q0 ? np1 : 0));
                    ch2 = (st_h2[a] - ts2) + rk2 * (tw2 + (Q0 ? np2 : 0));
                }
                uint64_t passmask = __ballot(pass);
                BEAM_TICK(7);
#ifdef FLORIA_PROF
                c_pass += (uint32_t)__popcll(passmask);
#endif
                // Common case (measured: 4.2 children pass per step, 0.06 % are duplicates, 1 % of the pushes overflow the heap): one batch,
                // no more children than the heap holds, and pairwise distinct state hashes (tested with a 256-slot LDS table: a slot
                // collision only sends the step down the general path).  Then every child is inserted, nothing is evicted, entry id =
                // rank among the passing lanes: the entry table is skipped (the survivors gather straight from the child lanes in M)
                // and only the std::BinaryHeap pushes remain.
                if (a0 == 0 && nstates <= S && !g.no_bulk) {
                    const uint32_t npass = (uint32_t)__popcll(passmask);
                    if (npass != 0 && npass <= limit) {
                        // the table is the 256 B of s_pk, free until phase M.  Explicit DS instructions: other LANES write the slot too, so the compiler must
                        // not forward this lane's store to its load (a volatile C++ access would do that as well, but hipcc turns it into FLAT
                        // operations that also wait for every outstanding global request).  LDS requests of a wave are served in order.
                        // (no clearing needed: every passing lane overwrites its own slot, so it reads back its own id or another PASSING lane's)
                        const uint32_t slot = (uint32_t)(ch1 ^ (ch1 >> 31) ^ (ch2 >> 17)) & 255u;
                        const uint32_t tab_addr = lds_base + LY.off_pk + slot;
                        if (pass) asm volatile("ds_write_b8 %0, %1" :: "v"(tab_addr), "v"(lane) : "memory");
                        uint32_t slot_owner;
                        asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot_owner) : "v"(tab_addr) : "memory");
                        const bool coll = pass && slot_owner != lane;
                        if (!__any(coll)) {
                            bulk = true;
                            b_h1 = ch1; b_h2 = ch2;
#ifndef FLORIA_NO_FASTM
                            // STRUCTURE-PRESERVING step (6 steps in 10): every state has exactly one passing child and no child inherits a slab that another
                            // child extends.  Then every new version goes in place, every slab keeps its id, the live list and the states' slab tables carry
                            // over (a child's table is its parent's), and phase M needs no reference counts, no leader election by atomics, no free list and
                            // no live-list rebuild.  Tested here, on the child lanes: a byte per slab id (the `ref` array, dead until phase M) — every
                            // (state, partition) lane clears its slab's byte, the passing lanes write 1 + lane into theirs, everybody reads back: a passing
                            // lane that finds its own number leads its slab; a non-passing lane's slab is inherited by its state's child, so a number there
                            // means "extended by someone AND inherited".  Explicit DS instructions, as for the hash-slot table above.
                            {
                                const uint32_t segp = (uint32_t)(passmask >> seg0) & ((1u << psl) - 1u);
                                const bool one = segp != 0u && (segp & (segp - 1u)) == 0u;
                                if (npass == nstates && !__any(act && !one)) {
                                    const uint32_t sid_f = act ? (uint32_t)st_sl[a * p + my_k] : 0u;
                                    const uint32_t t_addr = lds_base + LY.off_ref + sid_f;
                                    const uint32_t zero = 0u, mine = 1u + lane;
                                    if (act) asm volatile("ds_write_b8 %0, %1" :: "v"(t_addr), "v"(zero) : "memory");
                                    if (pass) asm volatile("ds_write_b8 %0, %1" :: "v"(t_addr), "v"(mine) : "memory");
                                    uint32_t t_val;
                                    asm volatile("ds_read_u8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t_val) : "v"(t_addr) : "memory");
                                    if (!__any(act && !pass && t_val != 0u)) {
                                        fastm = true;
                                        const bool lead_f = pass && t_val == mine;
                                        const uint64_t lm_f = __ballot(lead_f);
                                        nlead_f = (uint32_t)__popcll(lm_f);
                                        if (lead_f) freelist[mbcnt64(lm_f)] = (uint16_t)sid_f;
                                    }
                                }
                            }
#endif
#ifndef FLORIA_NO_HEAPKEEP
                            // ... and when, on top of that, the children's scores still satisfy the heap property slot by slot (child of state a = entry a = pushed a-th,
                            // and s_a <= s_parent(a) for every a makes std's sift_up stop at once: global_clustering.rs:130, Appendix A), the new heap IS the children in
                            // their parents' slots: no scalar pushes, one lane-parallel comparison.  (Mirrored states tie before and after: `<=` holds.)
                            if (fastm) {
                                const bool ina = lane < npass;
                                const uint32_t segp_a = (uint32_t)(passmask >> ((lane * psl) & 63u)) & ((1u << psl) - 1u);       // the passing partition of state `lane`
                                const uint32_t esrc_a = ina ? lane * psl + (uint32_t)__ffs((int)segp_a) - 1u : 0u;
                                const uint64_t cs_a = shfl_u64(cs, (int)esrc_a);
                                const uint64_t cs_par = shfl_u64(cs_a, (int)(lane ? (lane - 1u) >> 1 : 0u));
                                if (!__any(ina && lane != 0u && cs_a > cs_par)) {
                                    H.hp_hi = (uint32_t)(cs_a >> 32); H.hp_lo = (uint32_t)cs_a; H.hp_id = lane; H.len = npass;
                                    src_map = esrc_a;
                                    passmask = 0;
                                    heap_kept = true;
#ifdef FLORIA_PROF
                                    c_heapkeep++;
#endif
                                }
                            }
#endif
                            uint32_t r = 0;
                            while (passmask) {
                                const uint32_t src = (uint32_t)__ffsll((unsigned long long)passmask) - 1;
                                passmask &= passmask - 1;
                                wlane(src_map, src, r);
                                H.push(rl64(cs, src), r);
                                ++r;
                            }
#ifdef FLORIA_PROF
                            c_push += npass;
#endif
                        }
                    }
                }
                while (passmask) {
                    const uint32_t src = (uint32_t)__ffsll((unsigned long long)passmask) - 1;
                    passmask &= passmask - 1;
                    const uint64_t s_s = rl64(cs, src), s_h1 = rl64(ch1, src), s_h2 = rl64(ch2, src);
                    // general path (several batches, possible duplicates or evictions: ~3 % of the steps): the entry table (score, hash,
                    // parent | partition) lives in the NEXT parity's state arrays, which are dead until phase M — lane e tests entry e
                    bool dup = false;
                    if ((evalid >> lane) & 1) dup = E_h1[lane] == s_h1 && E_h2[lane] == s_h2 && E_s[lane] >= s_s;
                    if (__ballot(dup)) continue;
                    const uint32_t id = (uint32_t)__ffsll((unsigned long long)~evalid) - 1;
                    evalid |= 1ull << id;
                    if (lane == 0) { E_s[id] = s_s; E_h1[id] = s_h1; E_h2[id] = s_h2; E_pk[id] = (a0 + rl32(my_sl, src)) | (rl32(my_k, src) << 16); }
                    H.push(s_s, id);
#ifdef FLORIA_PROF
                    c_push++; if (H.len > limit) c_pop++;
#endif
                    if (H.len > limit) evalid &= ~(1ull << H.pop());
                }
                BEAM_TICK(2);
            }

            // ---- M: survivors (lane j = heap slot j = next state j) and their slabs -----------------------------------
            const uint32_t nnext = H.len;
            const bool surv = lane < nnext;
            const uint32_t eid = surv ? H.hp_id : 0;
            uint64_t n_q = 0, n_h1 = 0, n_h2 = 0;
            uint32_t n_m = 0, n_pk = 0;
            if (heap_kept) {                    // child of state `lane` in slot `lane`: its lane and partition are known without a shuffle
                n_h1 = shfl_u64(b_h1, (int)src_map); n_h2 = shfl_u64(b_h2, (int)src_map);
                n_pk = lane | ((src_map - lane * psl) << 16);
            } else if (bulk) {
                const int esrc = (int)__shfl(src_map, (int)eid);
                n_h1 = shfl_u64(b_h1, esrc); n_h2 = shfl_u64(b_h2, esrc);
                n_pk = __shfl(my_sl | (my_k << 16), esrc);
            } else if (surv) {
                n_h1 = E_h1[eid]; n_h2 = E_h2[eid]; n_pk = E_pk[eid];
            }
            const uint32_t pj = n_pk & 0xffff, kj = n_pk >> 16;
            if (!ARITH && surv) {                  // the child's (sum of diffs, #eps) = its parent's + the read's distance to the extended slab
                const uint32_t li = st_sl[pj * p + kj];
                n_q = st_q[pj] + r_qd[li]; n_m = st_m[pj] + r_m[li];
            }
            double* const st_ev = ST_ev(cur); double* const nx_ev = ST_ev(cur ^ 1);      // ARITH: error_vec of the current / next states
            uint64_t* nx_q = ST_q(cur ^ 1); uint64_t* nx_h1 = ST_h1(cur ^ 1); uint64_t* nx_h2 = ST_h2(cur ^ 1);
            uint32_t* nx_m = ST_m(cur ^ 1); uint16_t* nx_sl = ST_sl(cur ^ 1);
            uint32_t u_old = 0, ncopy = 0;
            uint64_t cmask = 0;
            bool lead = false;
            if (!heap_kept) s_pk[lane] = n_pk;
            if (fastm) {
                // structure-preserving step (phase B): the next states' slab tables are their parents', everything else about the slabs stands
                __syncthreads();
                if (heap_kept) { }          // child a in slot a: the states are updated IN PLACE below (their slab tables stand, the parity of the state arrays does not flip)
                else
                for (uint32_t x = lane; x < nnext * p; x += 64) {
                    const uint32_t j = div_small(x, rcp_p), k = x - j * p;
                    const uint32_t pk = s_pk[j];
                    const uint32_t sid = st_sl[(pk & 0xffff) * p + k];
                    nx_sl[x] = (uint16_t)sid;
                    if constexpr (ARITH) { const double pe = st_ev[(pk & 0xffff) * p + k]; nx_ev[x] = (k == (pk >> 16)) ? pe + r_fd[sid] : pe; }
                }
            } else {
            for (uint32_t x = lane; x < NS; x += 64) { ref[x] = 0; leader[x] = 0xffffffffu; }
            __syncthreads();
            // inherited pointers (every partition but the modified one) and the modified slab's leader
            for (uint32_t x = lane; x < nnext * p; x += 64) {
                const uint32_t j = div_small(x, rcp_p), k = x - j * p;
                const uint32_t pk = s_pk[j];
                const uint32_t sid = st_sl[(pk & 0xffff) * p + k];
                nx_sl[x] = (uint16_t)sid;
                if (k != (pk >> 16)) ref[sid] = 1;
                if constexpr (ARITH) { const double pe = st_ev[(pk & 0xffff) * p + k]; nx_ev[x] = (k == (pk >> 16)) ? pe + r_fd[sid] : pe; }
            }
            u_old = surv ? st_sl[pj * p + kj] : 0;
            if (surv) atomicMin(&leader[u_old], lane);
            __syncthreads();
            lead = surv && leader[u_old] == lane;
            const bool inplace = lead && ref[u_old] == 0;
            const bool needcopy = lead && !inplace;
            __syncthreads();                                           // all reads of ref[] done before in-place marks
            if (inplace) { ref[u_old] = 2; newid[u_old] = (uint16_t)u_old; }
            __syncthreads();
            // free slabs = not referenced; the first ncopy of them (ascending id) go to the copy leaders
            cmask = __ballot(needcopy);
            ncopy = (uint32_t)__popcll(cmask);
            if (ncopy) {
                uint32_t found = 0;
                for (uint32_t x0 = 0; x0 < NS && found < ncopy; x0 += 64) {
                    const uint32_t x = x0 + lane;
                    const bool fr = x < NS && ref[x] == 0;
                    const uint64_t fm = __ballot(fr);
                    const uint32_t pos = found + mbcnt64(fm);
                    if (fr && pos < 64) freelist[pos] = (uint16_t)x;
                    found += (uint32_t)__popcll(fm);
                }
                if (found < ncopy && lane == 0) atomicAdd(&GCOLD(diag)[1], 1u);
                __syncthreads();
                if (needcopy) { const uint32_t f = freelist[mbcnt64(cmask)]; newid[u_old] = (uint16_t)f; ref[f] = 2; }
                __syncthreads();
            }
            }
            // survivor records
            if (surv) {
                if constexpr (ARITH) {
                    if (heap_kept) { st_h1[lane] = n_h1; st_h2[lane] = n_h2; const uint32_t ix = lane * p + kj; st_ev[ix] = st_ev[ix] + r_fd[st_sl[ix]]; }     // (pj == lane)
                    else { nx_h1[lane] = n_h1; nx_h2[lane] = n_h2; }
                } else
                if (heap_kept) { st_q[lane] = n_q; st_h1[lane] = n_h1; st_h2[lane] = n_h2; st_m[lane] = n_m; }
                else { nx_q[lane] = n_q; nx_h1[lane] = n_h1; nx_h2[lane] = n_h2; nx_m[lane] = n_m; }
                if (!fastm) nx_sl[lane * p + kj] = newid[u_old];
                if (FLORIA_NT_AUX) __builtin_nontemporal_store(pj | (kj << 16), slot_hist + beam_hist_off(i, LM, B) + lane);
                else slot_hist[beam_hist_off(i, LM, B) + lane] = pj | (kj << 16);
            }
            BEAM_TICK(3);
            // copies of the written window [first_rel, hi_rel] for the new versions that could not go in place
#ifdef FLORIA_PROF
            c_ncopy += ncopy; if (ncopy && hi_rel >= (int32_t)first_rel) c_copy_pos += (unsigned long long)ncopy * (uint32_t)(hi_rel - (int32_t)first_rel + 1);
            c_trunc += trunc ? 1 : 0;
#endif
            if (ncopy && hi_rel >= (int32_t)first_rel) {
                const uint32_t cnt2 = ((uint32_t)(hi_rel - (int32_t)first_rel + 1) * A) >> 1;
                uint64_t cm2 = cmask;
                while (cm2) {
                    const uint32_t jj = (uint32_t)__ffsll((unsigned long long)cm2) - 1;
                    cm2 &= cm2 - 1;
                    const uint32_t su = rl32(u_old, jj);
                    const uint32_t du = newid[su];
                    if constexpr (NARROW) {
                        const uint32_t cb = (uint32_t)(hi_rel - (int32_t)first_rel + 1);
                        const uint2* s = (const uint2*)(pool + (su * slab_bytes + first_rel * pos_bytes));
                        uint2* d = (uint2*)(pool + (du * slab_bytes + first_rel * pos_bytes));
                        const uint16_t* sh = (const uint16_t*)(hi_plane + (su * hi_slab_bytes + first_rel * 2u));
                        uint16_t* dh = (uint16_t*)(hi_plane + (du * hi_slab_bytes + first_rel * 2u));
                        uint32_t x = lane;
                        for (; x + 192 < cb; x += 256) {
                            const uint2 v0 = s[x], v1 = s[x + 64], v2 = s[x + 128], v3 = s[x + 192];
                            const uint16_t h0 = sh[x], h1 = sh[x + 64], h2 = sh[x + 128], h3 = sh[x + 192];
                            d[x] = v0; d[x + 64] = v1; d[x + 128] = v2; d[x + 192] = v3;
                            dh[x] = h0; dh[x + 64] = h1; dh[x + 128] = h2; dh[x + 192] = h3;
                        }
                        for (; x < cb; x += 64) { d[x] = s[x]; dh[x] = sh[x]; }
                    } else {
                    const ulonglong2* s = (const ulonglong2*)(pool + (su * slab_bytes + first_rel * pos_bytes));
                    ulonglong2* d = (ulonglong2*)(pool + (du * slab_bytes + first_rel * pos_bytes));
                    uint32_t x = lane;
                    for (; x + 192 < cnt2; x += 256) {
                        const ulonglong2 v0 = s[x], v1 = s[x + 64], v2 = s[x + 128], v3 = s[x + 192];
                        d[x] = v0; d[x + 64] = v1; d[x + 128] = v2; d[x + 192] = v3;
                    }
                    for (; x < cnt2; x += 64) d[x] = s[x];
                    }
                    if (CODES) {
                        const uint8_t* sc = codes + (su * span_pad + first_rel);
                        uint8_t* dc = codes + (du * span_pad + first_rel);
                        const uint32_t cb = (uint32_t)(hi_rel - (int32_t)first_rel + 1);
                        for (uint32_t y = lane; y < cb; y += 64) dc[y] = sc[y];
                    }
                }
            }
            __syncthreads();
            // next live list = every referenced slab (ascending id); zero their newly reached positions (hi_rel, new_hi]
            uint32_t nl = fastm ? nlive : 0u;
            if (!fastm)
            for (uint32_t x0 = 0; x0 < NS; x0 += 64) {
                const uint32_t x = x0 + lane;
                const bool rf = x < NS && ref[x] != 0;
                const uint64_t fm = __ballot(rf);
                if (rf) live_id[nl + mbcnt64(fm)] = (uint16_t)x;
                nl += (uint32_t)__popcll(fm);
#ifdef FLORIA_PROF
                c_id8 += (unsigned)__popcll(__ballot(rf && x >= 8)); c_id16 += (unsigned)__popcll(__ballot(rf && x >= 16)); c_id32 += (unsigned)__popcll(__ballot(rf && x >= 32));
#endif
            }
#ifdef FLORIA_PROF
            { c_nl += nl; const uint32_t Wd = (uint32_t)(new_hi - (int32_t)first_rel + 1); c_wsum += Wd; c_w128 += Wd > 128; c_w256 += Wd > 256; c_w512 += Wd > 512; if (!bulk) c_general++; if (fastm) c_boring++; }
#endif
            __syncthreads();
            if (new_hi > hi_rel) {
                const uint32_t cntz = (uint32_t)(new_hi - hi_rel) * (NARROW ? 1u : (uint32_t)A);      // 8-B words per slab
                const uint32_t items = nl * cntz;
#ifdef FLORIA_PROF
                c_zero_items += items / A;
#endif
                const float rcp_cntz = __builtin_amdgcn_rcpf((float)cntz);
                for (uint32_t x = lane; x < items; x += 64) {
                    const uint32_t e = items < (1u << 20) ? div_small(x, rcp_cntz) : x / cntz, o = x - e * cntz;
                    *(uint64_t*)(pool + ((uint32_t)live_id[e] * slab_bytes + (uint32_t)(hi_rel + 1) * pos_bytes + o * 8)) = 0;
                }
                if (CODES) {
                    const uint32_t cz = (uint32_t)(new_hi - hi_rel), items_c = nl * cz;
                    const float rcp_cz = __builtin_amdgcn_rcpf((float)cz);
                    for (uint32_t x = lane; x < items_c; x += 64) {
                        const uint32_t e = items_c < (1u << 20) ? div_small(x, rcp_cz) : x / cz, o = x - e * cz;
                        codes[(uint32_t)live_id[e] * span_pad + (uint32_t)(hi_rel + 1) + o] = 0;
                        if constexpr (NARROW) *(uint16_t*)(hi_plane + ((uint32_t)live_id[e] * hi_slab_bytes + ((uint32_t)(hi_rel + 1) + o) * 2u)) = 0;
                    }
                }
            }
            __syncthreads();
            BEAM_TICK(4);
            // add the read ONCE per distinct new version (types_structs.rs:368-373)
            {
                const uint64_t lmask = __ballot(lead);
                const uint32_t nlead = fastm ? nlead_f : (uint32_t)__popcll(lmask);
#ifdef FLORIA_PROF
                c_nlead += nlead; c_add_items += (unsigned long long)nlead * L;
                { const uint32_t it = nlead * L; c_it64 += it > 64; c_it128 += it > 128; c_it256 += it > 256; }
#endif
                // leaders' target slabs, compacted into freelist[] (reused as scratch)
                if (lead) freelist[mbcnt64(lmask)] = newid[u_old];            // (a structure-preserving step filled the list in phase B)
                for (uint32_t t = 0; t < ntiles; ++t) {
                    if (ntiles > 1) stage_tile(t); else __syncthreads();
                    const uint32_t tl = min((uint32_t)SLAB_TILE, L - t * SLAB_TILE);
                    const uint32_t items = nlead * tl;
                    const float rcp_tl = __builtin_amdgcn_rcpf((float)tl);
                    auto addr_of = [&](uint32_t x, uint32_t& w) -> uint64_t* {
                        const uint32_t e = div_small(x, rcp_tl), c = x - e * tl;
                        const uint32_t aw = c_aw[c * CST];
                        w = aw & 0x0fffffffu;
                        return (uint64_t*)(pool + ((uint32_t)freelist[e] * slab_bytes + (c_snp[c * CST] - pos0) * pos_bytes + (aw >> 28) * 8));
                    };
                    // read-modify-writes in flight per lane, branch-free: the tail slots go to the lane's dummy words in the slot's
                    // scratch.  (A load left unconsumed on some path makes hipcc wait vmcnt(0) at the top of the next step, i.e.
                    // for the acknowledgement of these stores, before phase A can issue its loads.)
                    if constexpr (NARROW) {
                        // lane <-> cell (chunks of 64 cells), leaders in an outer scalar loop: no index division, the cell words are read once per chunk and
                        // leader, a leader's slab offset is a scalar, and every address is pool + a 32-bit offset built from ONE position index (layout above).
                        // NCH chunks x NE leaders are in flight together (<= 4 read-modify-writes per lane); tail lanes go to their dummy index.
                        const uint32_t flv = lane < nlead ? (uint32_t)freelist[lane] : 0u;          // leaders' slabs, lane e = leader e (nlead <= 63)
                        const uint32_t nch = (tl + 63u) >> 6;
                        auto add_round = [&](auto NCHC, auto NEC, uint32_t e0) {
                            constexpr int NCH = decltype(NCHC)::value, NE = decltype(NEC)::value, AU = NCH * NE;
                            uint32_t w[NCH], al[NCH], pr[NCH]; bool okc[NCH];
#pragma unroll
                            for (int c = 0; c < NCH; ++c) {
                                const uint32_t cc = lane + 64u * c;
                                okc[c] = cc < tl;
                                const uint32_t cx = okc[c] ? cc : 0u;
                                const uint32_t aw = c_aw[cx * CST];
                                pr[c] = c_snp[cx * CST] - pos0; w[c] = aw & 0x0fffffffu; al[c] = aw >> 28;
                            }
                            uint32_t idx[AU]; uint2 lo[AU]; uint32_t hv[AU];
#pragma unroll
                            for (int e = 0; e < NE; ++e) {
                                const uint32_t sb = rl32(flv, e0 + (uint32_t)e) * span_pad;                 // scalar
#pragma unroll
                                for (int c = 0; c < NCH; ++c) idx[e * NCH + c] = okc[c] ? sb + pr[c] : NS * span_pad + lane;
                            }
#pragma unroll
                            for (int u = 0; u < AU; ++u) { lo[u] = *(const uint2*)(pool + idx[u] * 8u); hv[u] = *(const uint16_t*)(pool + (hi_base + idx[u] * 2u)); }
#pragma unroll
                            for (int u = 0; u < AU; ++u) {
                                const int c = u % NCH;
                                uint64_t v0 = ((uint64_t)(hv[u] & 0xffu) << 32) | lo[u].x, v1 = ((uint64_t)(hv[u] >> 8) << 32) | lo[u].y;
                                uint64_t nv;
                                if (al[c]) { v1 += w[c]; nv = v1; } else { v0 += w[c]; nv = v0; }
                                *(uint32_t*)(pool + (idx[u] * 8u + al[c] * 4u)) = (uint32_t)nv;
                                if ((uint32_t)nv < w[c]) *(uint8_t*)(pool + (hi_base + idx[u] * 2u + al[c])) = (uint8_t)(nv >> 32);      // the low word wrapped: one time in ~256 adds
                                const uint32_t code = (v0 | v1) ? ((v0 >= v1 ? 1u : 0u) | (v1 >= v0 ? 2u : 0u)) : 0u;
                                *(uint8_t*)(pool + (code_base + idx[u])) = (uint8_t)code;
                            }
                        };
                        uint32_t e0 = 0;
                        if (nch <= 2) {                      // two leaders at a time
                            for (; e0 + 2 <= nlead; e0 += 2) { if (nch == 1) add_round(IC<1>{}, IC<2>{}, e0); else add_round(IC<2>{}, IC<2>{}, e0); }
                        }
                        for (; e0 < nlead; ++e0) {
                            if (nch == 1) add_round(IC<1>{}, IC<1>{}, e0); else if (nch == 2) add_round(IC<2>{}, IC<1>{}, e0);
                            else if (nch == 3) add_round(IC<3>{}, IC<1>{}, e0); else add_round(IC<4>{}, IC<1>{}, e0);
                        }
                    } else if (CODES) {
                        // the position's A sums come in together (one 16-B piece for biallelic data): add the read's weight, store the changed
                        // sum, and refresh the position's code byte from the new sums
                        // up to 4 read-modify-writes in flight per lane and pass, exactly as many as the pass has (a step has ~105 of them over 64 lanes: 2)
                        auto add_pass = [&](auto NC, uint32_t x0) {
                            constexpr int AU = decltype(NC)::value;
                            uint32_t w[AU], al[AU]; uint64_t* base[AU]; uint8_t* cptr[AU]; uint8_t* hptr[AU];
#pragma unroll
                            for (int u = 0; u < AU; ++u) {
                                const uint32_t xx = x0 + lane + 64 * u;
                                const bool ok = xx < items;
                                const uint32_t xs = ok ? xx : 0;
                                const uint32_t e = div_small(xs, rcp_tl), c = xs - e * tl;
                                const uint32_t aw = c_aw[c * CST], pr = c_snp[c * CST] - pos0, sl = (uint32_t)freelist[e];
                                w[u] = aw & 0x0fffffffu; al[u] = aw >> 28;
                                base[u] = ok ? (uint64_t*)(pool + (sl * slab_bytes + pr * pos_bytes)) : dummy + lane * A;
                                cptr[u] = ok ? codes + (sl * span_pad + pr) : dummy_code + lane;
                                if constexpr (NARROW) hptr[u] = ok ? hi_plane + (sl * hi_slab_bytes + pr * 2u) : dummy_code + 64 + 2 * lane;
                            }
                            ulonglong2 vv[AU][A / 2];
                            uint32_t hv[AU];
#pragma unroll
                            for (int u = 0; u < AU; ++u) {
                                if constexpr (NARROW) { const uint2 lo = *(const uint2*)base[u]; vv[u][0].x = lo.x; vv[u][0].y = lo.y; hv[u] = *(const uint16_t*)hptr[u]; }
                                else {
#pragma unroll
                                for (int x = 0; x < A / 2; ++x) vv[u][x] = ((const ulonglong2*)base[u])[x];
                                }
                            }
#pragma unroll
                            for (int u = 0; u < AU; ++u) {
                                uint64_t v[A];
#pragma unroll
                                for (int x = 0; x < A; x += 2) { v[x] = vv[u][x / 2].x; v[x + 1] = vv[u][x / 2].y; }
                                if constexpr (NARROW) { v[0] |= (uint64_t)(hv[u] & 0xffu) << 32; v[1] |= (uint64_t)(hv[u] >> 8) << 32; }
                                uint64_t nv = 0;
#pragma unroll
                                for (int x = 0; x < A; ++x) { if (x == (int)al[u]) { v[x] += w[u]; nv = v[x]; } }
                                if constexpr (NARROW) {
                                    ((uint32_t*)base[u])[al[u]] = (uint32_t)nv;
                                    if ((uint32_t)nv < w[u]) hptr[u][al[u]] = (uint8_t)(nv >> 32);          // the low word wrapped: one time in ~256 adds
                                } else
                                base[u][al[u]] = nv;
                                uint32_t code;
                                if (A == 2) code = (v[0] | v[1]) ? ((v[0] >= v[1] ? 1u : 0u) | (v[1] >= v[0] ? 2u : 0u)) : 0u;
                                else {
                                    uint64_t mx = 0;
#pragma unroll
                                    for (int x = 0; x < A; ++x) mx = v[x] > mx ? v[x] : mx;
                                    code = 0;
#pragma unroll
                                    for (int x = 0; x < A; ++x) code |= (mx != 0 && v[x] == mx) ? (1u << x) : 0u;
                                }
                                *cptr[u] = (uint8_t)code;
                            }
                        };
                        for (uint32_t x0 = 0; x0 < items; x0 += 256u) {
                            const uint32_t r = items - x0;
                            if (r > 192u) add_pass(IC<4>{}, x0); else if (r > 128u) add_pass(IC<3>{}, x0); else if (r > 64u) add_pass(IC<2>{}, x0); else add_pass(IC<1>{}, x0);
                        }
                    } else
                    for (uint32_t x0 = 0; x0 < items; x0 += 256) {
                        uint32_t w[4]; uint64_t* ptr[4]; uint64_t v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const uint32_t xx = x0 + lane + 64 * u;
                            uint64_t* pa = addr_of(xx < items ? xx : 0, w[u]);
                            ptr[u] = xx < items ? pa : dummy + lane;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = *ptr[u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) *ptr[u] = Q0 ? ((v[u] + w[u]) | PRESENT_BIT) : v[u] + w[u];
                    }
                }
            }
            cm_cur = cm_next; sm_cur = sm_next;
            if (i + 2 < n) { cm_next = rec_cm(rec_hold); sm_next = rec_sm(rec_hold); }
            __syncthreads();
            BEAM_TICK(5);
            if (!heap_kept) cur ^= 1;
            nstates = nnext;
            nlive = nl;
            hi_rel = new_hi;
            start_rel = first_rel;
        }

        if (SPEC && dropped) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }     // (an LDS-DMA of the next read may still be in flight)
        if (n > 0 && !(SPEC && dropped)) {
            H.hp_id = lane;
            uint32_t ecur = H.sorted_first();
            uint8_t* out = GCOLD(part_out) + roff;
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
            if (lane == 0) atomicAdd(GCOLD(steps_done), (unsigned long long)n);
            { const double jm = wave_min_f64_dpp(min_margin); if (lane == 0) GCOLD(job_margin)[(uint64_t)b * GCOLD(max_ploidy) + p - 1] = jm; }
        }
        __syncthreads();
        BEAM_TICK(6);
    }
#ifdef FLORIA_PROF
    if (lane == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&g.prof[16 + i], t_acc[i]);
                     atomicAdd(&g.prof[24], wall_clock64() - t_wall0); atomicAdd(&g.prof[25], clock64() - t_core0); atomicAdd(&g.prof[26], 1ull);
                     atomicAdd(&g.prof[26 + g.ploidy], wall_clock64() - t_wall0);
                     atomicAdd(&g.prof[10], (unsigned long long)c_pass); atomicAdd(&g.prof[11], (unsigned long long)c_push); atomicAdd(&g.prof[12], (unsigned long long)c_pop);
                     atomicAdd(&g.prof[32], c_copy_pos); atomicAdd(&g.prof[33], c_ncopy); atomicAdd(&g.prof[34], c_add_items); atomicAdd(&g.prof[35], c_zero_items); atomicAdd(&g.prof[36], c_nlead); atomicAdd(&g.prof[37], c_trunc);
                     atomicAdd(&g.prof[38], c_nl); atomicAdd(&g.prof[39], c_id8); atomicAdd(&g.prof[40], c_id16); atomicAdd(&g.prof[41], c_id32); atomicAdd(&g.prof[42], c_w128); atomicAdd(&g.prof[43], c_w256);
                     atomicAdd(&g.prof[44], c_w512); atomicAdd(&g.prof[45], c_wsum); atomicAdd(&g.prof[48], c_it64); atomicAdd(&g.prof[49], c_it128); atomicAdd(&g.prof[50], c_it256); atomicAdd(&g.prof[51], c_lvl2); atomicAdd(&g.prof[52], c_general); atomicAdd(&g.prof[53], c_exact); atomicAdd(&g.prof[60], c_boring); atomicAdd(&g.prof[61], c_heapkeep); atomicAdd(&g.prof[54], c_code);      // [54]: code bytes gathered by the distance phase
                     atomicAdd(&g.prof[13], c_nlive); atomicAdd(&g.prof[14], c_nin); atomicAdd(&g.prof[15], c_nstates); atomicAdd(&g.prof[9], c_L); }     // [28..31]: wave wall ticks of the ploidy 2..5 launches
#endif
    n_fallback = wave_sum_u32_dpp(n_fallback);
    if (lane == 0) {
        if (n_fallback) atomicAdd(&GCOLD(diag)[0], n_fallback);
    }
}

}  // namespace fl
