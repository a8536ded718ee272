// floria_hip.hip — host side of libfloria_hip.so (C ABI in include/floria_hip.h) + kernel launches.
//
// Host responsibilities (the reference does these on its rayon workers / main thread):
//   get_range_with_lengths            utils_frags.rs:405-463      -> floria_hip_block_ranges
//   find_reads_in_interval            local_clustering.rs:12-59   -> build_block_lists (binary search + filter)
//   per-ploidy loop of get_local_hap_blocks graph_processing.rs:132-252 -> one launch triple per ploidy:
//       beam_kernel -> optimize_kernel -> select_kernel (the stop rule sets blk_done so later
//       ploidies skip finished blocks: no speculative work, results identical)
//   separate_broken_haplogroups / sort_parts  part_block_manip.rs:27-98,276-288 -> host bookkeeping after
//       reassign_kernel
// There is NO CPU compute fallback: every entry point needs a working HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <utility>
#include <mutex>
#include <vector>

#include "../../include/floria_hip.h"
#include "beam_kernel.h"
#include "beam_fast_kernel.h"
#include "beam_slab_kernel.h"
#include "beam_wide_kernel.h"
#include "optimize_kernel.h"
#include "reassign_kernel.h"
#include "hapq_kernel.h"
#include "blocks_kernel.h"
#include "graph_kernel.h"
#include "stats_kernel.h"

static_assert(FLORIA_MAX_PLOIDY == fl::MAX_PLOIDY, "ploidy limits out of sync");

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(_e == hipErrorOutOfMemory ? FLORIA_E_NOMEM : FLORIA_E_DEVICE,                  \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                            \
    } while (0)

constexpr double DIV_FACTOR  = 0.25;   // constants.rs:5
constexpr double PROB_CUTOFF = 0.01;   // constants.rs:6
constexpr uint32_t BINOM_NMAX_CAP = 1024;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(FLORIA_E_NOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

}  // namespace

// Large result arrays (read ids, partitions: tens of MB per S1 call) come from a small process-wide cache: a fresh malloc of
// that size is an mmap whose pages fault one by one under the device->host copy.  Header = capacity, 64 B before the payload.
namespace {
struct BigCache {
    std::mutex m;
    std::vector<std::pair<void*, size_t>> v;      // (base, capacity)
    void* get(size_t n) {
        {
            std::lock_guard<std::mutex> l(m);
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i].second >= n && v[i].second <= 2 * n + 4096) { void* b = v[i].first; v.erase(v.begin() + i); return (char*)b + 64; }
        }
        void* b = malloc(n + 64);
        if (!b) return nullptr;
        *(size_t*)b = n;
        return (char*)b + 64;
    }
    void put(void* payload) {
        if (!payload) return;
        void* b = (char*)payload - 64;
        const size_t cap = *(size_t*)b;
        std::lock_guard<std::mutex> l(m);
        if (cap >= (1u << 20) && v.size() < 8) v.push_back({b, cap}); else free(b);
    }
};
BigCache g_big;
}  // namespace

struct floria_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t user_slots = 0, user_groups = 0;
    int n_cu = 256;
    // job groups: the (block, ploidy) launch triples of disjoint block groups run on their own streams, so one group's
    // launch tail (persistent waves draining their last jobs) is filled by the other groups' kernels
    static constexpr uint32_t MAX_GROUPS = 8;
    hipStream_t gstream[MAX_GROUPS] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_GROUPS] = {};
    hipStream_t copy_stream = nullptr;        // read-id lists go back to the host while the launch loop runs
    hipEvent_t ev_rids = nullptr;
    // cached tables
    double binom_eps = -1.0;
    uint32_t binom_nmax = 0;
    DevBuf d_binom;
    DevBuf d_hash;            // Rq1 | Rp1 | Rq2 | Rp2, each hash_len u64
    uint32_t hash_len = 0;
    std::vector<uint64_t> h_rq1, h_rq2;   // host copies of the Rq tables (per-read hash constants are computed at upload)
    uint32_t w24[256];
    uint64_t Rk1[FLORIA_MAX_PLOIDY], Rk2[FLORIA_MAX_PLOIDY];
    // scratch pools
    DevBuf state_pool, hist_pool, opt_hist, opt_dist, opt_gain, opt_key, opt_moves, misc, misc0;
    floria_timing timing{};
    // device-resident copy of the last S1 batch (floria_hip_hap_graph)
    uint64_t batch_token = 0, token_counter = 0;
    fl::BlockSet last_bs{};
    const uint8_t* last_part = nullptr;
    const uint32_t* last_best = nullptr;
    uint32_t last_nall = 2;
    std::vector<uint32_t> last_bc, last_start, last_end;
    DevBuf graph_buf, graph_hist, graph_sort;
};

struct floria_hip_contig {
    floria_hip_ctx* ctx = nullptr;
    uint32_t n_reads = 0;
    uint64_t n_cells = 0;
    uint32_t max_len = 0;       // max cells per read
    uint32_t n_alleles = 2;     // 2 or 4 (kernel template)
    bool has_q0 = false;        // some cell has qual 0 (weight 0): presence != (weight sum > 0)
    std::vector<uint32_t> h_first, h_last, h_read_off;
    DevBuf d_read_off, d_first, d_last, d_snp, d_aq, d_tw, d_meta;
    fl::ContigDev dev{};
};

namespace {

int validate_pileup(const floria_pileup* p, uint32_t* max_len, uint32_t* max_allele) {
    if (!p) return fail(FLORIA_E_INVALID, "null pileup");
    if (p->n_reads && (!p->read_off || !p->snp || !p->allele || !p->qual || !p->first || !p->last))
        return fail(FLORIA_E_INVALID, "null pileup field");
    uint32_t ml = 0, ma = 0;
    for (uint32_t r = 0; r < p->n_reads; ++r) {
        const uint32_t b = p->read_off[r], e = p->read_off[r + 1];
        if (e <= b) return fail(FLORIA_E_INVALID, "read " + std::to_string(r) + " has no cells");
        if (p->snp[b] != p->first[r] || p->snp[e - 1] != p->last[r]) return fail(FLORIA_E_INVALID, "first/last of read " + std::to_string(r) + " do not match its cells");
        if (p->snp[b] == 0) return fail(FLORIA_E_INVALID, "SNP positions are 1-based");
        for (uint32_t c = b; c < e; ++c) {
            if (c > b && p->snp[c] <= p->snp[c - 1]) return fail(FLORIA_E_INVALID, "cells of read " + std::to_string(r) + " not strictly ascending");
            if (p->allele[c] >= FLORIA_MAX_ALLELES) return fail(FLORIA_E_UNSUPPORTED, "allele index > 3 (read " + std::to_string(r) + ")");
            ma = std::max<uint32_t>(ma, p->allele[c]);
        }
        ml = std::max(ml, e - b);
        if (r > 0) {   // Frag::cmp (types_structs.rs:87-93)
            const bool ok = p->first[r - 1] < p->first[r] || (p->first[r - 1] == p->first[r] && p->last[r - 1] >= p->last[r]);
            if (!ok) return fail(FLORIA_E_INVALID, "reads not sorted by Frag::cmp at read " + std::to_string(r));
        }
    }
    *max_len = ml; *max_allele = ma;
    return 0;
}

int ensure_binom(floria_hip_ctx* ctx, double eps, uint32_t nmax) {
    nmax = std::min(nmax, BINOM_NMAX_CAP);
    if (ctx->binom_eps == eps && ctx->binom_nmax >= nmax && ctx->d_binom.p) return 0;
    nmax = std::max(nmax, ctx->binom_eps == eps ? ctx->binom_nmax : 0u);
    std::vector<double> tab((size_t)(nmax + 1) * (nmax + 2) / 2);
    for (uint64_t n = 0; n <= nmax; ++n)
        for (uint64_t k = 0; k <= n; ++k) {
            // stable_binom_cdf_p_rev (utils_frags.rs:211-248), host libm
            double v = 0.0;
            if (n != 0) {
                double n64 = (double)n, k64 = (double)k;
                double a = k64 / n64;
                if (a == 1.0) a = 0.9999999;
                if (a == 0.0) a = 0.0000001;
                double rel_ent = a * std::log(a / eps) + (1.0 - a) * std::log((1.0 - a) / (1.0 - eps));
                if (a < eps) rel_ent = -rel_ent;
                v = -1.0 * n64 / DIV_FACTOR * rel_ent;
            }
            tab[n * (n + 1) / 2 + k] = v;
        }
    int rc = ctx->d_binom.ensure(tab.size() * sizeof(double));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_binom.p, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->binom_eps = eps; ctx->binom_nmax = nmax;
    return 0;
}

int ensure_hash(floria_hip_ctx* ctx, uint32_t len) {
    if (ctx->hash_len >= len && ctx->d_hash.p) return 0;
    len = std::max<uint32_t>(len, 4 * fl::HASH_M);
    std::vector<uint64_t> t((size_t)len * 4);
    uint64_t s = 0x1577f10a1aull;
    for (uint32_t i = 0; i < len; ++i) {
        t[i] = splitmix64(s) | 1ull;                 // Rq1 (odd)
        t[(size_t)len + i] = splitmix64(s);          // Rp1
        t[(size_t)2 * len + i] = splitmix64(s) | 1ull;
        t[(size_t)3 * len + i] = splitmix64(s);
    }
    int rc = ctx->d_hash.ensure(t.size() * 8);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->d_hash.p, t.data(), t.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->hash_len = len;
    ctx->h_rq1.assign(t.begin(), t.begin() + len);
    ctx->h_rq2.assign(t.begin() + 2 * (size_t)len, t.begin() + 3 * (size_t)len);
    return 0;
}

struct EventTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<int> kind;
    hipStream_t s;
    explicit EventTimer(hipStream_t st) : s(st) {}
    ~EventTimer() { for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); } }
    std::vector<hipStream_t> on;
    int begin(int k, hipStream_t st = nullptr) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        ev.push_back({a, b}); kind.push_back(k); on.push_back(st ? st : s);
        (void)hipEventRecord(a, on.back());
        return (int)ev.size() - 1;
    }
    void end(int i) { if (i >= 0) (void)hipEventRecord(ev[i].second, on[i]); }
    double elapsed(int i) { float ms = 0; if (i >= 0 && hipEventElapsedTime(&ms, ev[i].first, ev[i].second) == hipSuccess) return ms; return 0; }
    double sum(int k) {
        double t = 0;
        for (size_t i = 0; i < ev.size(); ++i) if (kind[i] == k) { float ms = 0; if (hipEventElapsedTime(&ms, ev[i].first, ev[i].second) == hipSuccess) t += ms; }
        return t;
    }
    double span() {
        if (ev.empty()) return 0;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ev.front().first, ev.back().second);
        return ms;
    }
};
enum { K_BEAM = 0, K_OPT = 1, K_SEL = 2, K_H2D = 3, K_D2H = 4, K_REASSIGN = 5, K_PHASE = 6 };

void sync_all(floria_hip_ctx* ctx) {
    for (uint32_t g = 0; g < floria_hip_ctx::MAX_GROUPS; ++g) if (ctx->gstream[g]) (void)hipStreamSynchronize(ctx->gstream[g]);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
}

// One launch triple (beam -> optimise -> stop rule) per ploidy and JOB GROUP.  `jobs` holds the non-empty blocks, longest
// first inside each group; group g owns jobs [group_off[g], group_off[g+1]) and runs on its own stream with its own scratch.
template <int A>
int run_phase(floria_hip_ctx* ctx, bool any_q0, const fl::BlockSet& bs, const std::vector<uint32_t>& jobs, const std::vector<uint32_t>& group_off,
              const uint32_t* d_jobs, uint64_t tot_reads, uint32_t n_max, uint32_t span_max, const floria_params* prm, uint8_t* d_planes,
              uint8_t* d_beam_part, double* d_mec, double* d_na, uint32_t* d_iters, uint8_t* d_done, uint32_t* d_best,
              uint32_t* d_tried, uint32_t* d_queue, unsigned long long* d_margin, uint32_t* d_diag,
              unsigned long long* d_steps, EventTimer& T, bool& p1_shortcut) {
    const uint32_t P = prm->max_ploidy, B = prm->beam;
    p1_shortcut = false;
    const uint32_t n_jobs = (uint32_t)jobs.size();
    const uint32_t G = (uint32_t)group_off.size() - 1;
    if (n_jobs == 0) return 0;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const double cutoff = std::log(PROB_CUTOFF);      // graph_processing.rs:146
    // fork: every group stream starts after what the main stream has queued so far (uploads, memsets, block_reads_kernel)
    hipStream_t gs[floria_hip_ctx::MAX_GROUPS];
    gs[0] = ctx->stream;
    if (G > 1) {
        if (!ctx->ev_fork) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        for (uint32_t g = 1; g < G; ++g) {                 // group 0 stays on the main stream (HIP maps streams onto few hardware queues)
            if (!ctx->gstream[g]) HIPCHK(hipStreamCreateWithFlags(&ctx->gstream[g], hipStreamNonBlocking));
            if (!ctx->ev_join[g]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_join[g], hipEventDisableTiming));
            gs[g] = ctx->gstream[g];
        }
    }
    const int t_phase = T.begin(K_PHASE);
    bool forked = false;
    auto fork = [&]() -> int {
        if (G > 1 && !forked) {
            HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
            for (uint32_t g = 1; g < G; ++g) HIPCHK(hipStreamWaitEvent(gs[g], ctx->ev_fork, 0));
            forked = true;
        }
        return 0;
    };
    for (uint32_t p = 1; p <= P; ++p) {
        const uint32_t LM = p * B;
        if (LM > 65000) return fail(FLORIA_E_UNSUPPORTED, "ploidy*beam too large");
        // ---- beam search ---------------------------------------------------------------------------------------
        // ploidy 1 has nothing to search: one state, one partition, every child passes (p_k - lse == 0 > ln 0.01), the
        // partition is "all reads in haplotype 0" (global_clustering.rs:74-134 with ploidy == 1) -> a memset.
        const bool shortcut = p == 1 && !getenv("FLORIA_HIP_NO_P1_SHORTCUT");
        if (shortcut) {
            HIPCHK(hipMemsetAsync(d_beam_part, 0, tot_reads, ctx->stream));
            p1_shortcut = true;
        }
        if (int rc = fork()) return rc;
        // sizes shared by the groups of this ploidy
        const uint64_t state_bytes = (uint64_t)LM * span_max * p * A * 8;
        // traceback records + the slab kernels' per-slab tables that live in HBM (window-exit hash terms; everything for wide beams)
        const uint64_t hist_stride = (((uint64_t)fl::beam_hist_off(n_max, LM, B) + LM + 64 + 1) & ~1ull) + 128 + std::max<uint64_t>(4ull * LM * p, fl::wide_scratch_words(LM, p, any_q0));   // + 64 dummy u64 words (beam_slab_kernel's branch-free tails)
        const fl::SlabLds SL = fl::slab_lds_layout(LM, p, any_q0);
        const fl::WideLds WL = fl::wide_lds_layout(LM, p, any_q0);
        const fl::BeamLds LY = fl::beam_lds_layout(LM);
        const uint32_t by_lds = std::max<uint32_t>(1, (uint32_t)((158 * 1024) / (SL.total + 256)));
        const uint32_t waves_per_simd = (A == 2 && !any_q0 && B == 10 && p >= 2 && p <= FLORIA_SLAB_LOW_P_MAX && !getenv("FLORIA_HIP_NO_SPECIALIZED")) ? FLORIA_SLAB_WAVES_LOW_P : FLORIA_FAST_WAVES;
        uint32_t beam_slots = ctx->user_slots ? ctx->user_slots : (uint32_t)ctx->n_cu * std::min<uint32_t>(4 * waves_per_simd, by_lds);
        beam_slots = std::min(beam_slots, n_jobs);
        const uint64_t budget = (uint64_t)((double)(free_b + ctx->state_pool.cap + ctx->hist_pool.cap) * 0.6);
        while (beam_slots > 1 && (state_bytes + hist_stride * 4) * beam_slots * G > budget) beam_slots /= 2;
        const char* force = getenv("FLORIA_HIP_BEAM");       // dev/test knob: generic | fast | slab | wide
        const bool fits32 = (uint64_t)LM * span_max * p * A * 8 < 0xf0000000ull;
        const bool small = LM <= 63 && fits32;
        bool wide = !small && fits32 && LM * p <= (uint32_t)fl::WIDE_NS_MAX && LM < 65000 && WL.total <= 150 * 1024;
        if (force && !strcmp(force, "wide") && fits32 && LM * p <= (uint32_t)fl::WIDE_NS_MAX && WL.total <= 150 * 1024) wide = true;
        if (force && strcmp(force, "wide")) wide = false;
        bool slab = small && LM * p <= (uint32_t)fl::SLAB_NS_MAX && SL.total <= 60 * 1024;
        bool fast = small;
        if (force && !strcmp(force, "generic")) { slab = false; fast = false; }
        if (force && !strcmp(force, "fast")) slab = false;
        if (!shortcut) {
            int rc = ctx->state_pool.ensure(state_bytes * beam_slots * G); if (rc) return rc;
            rc = ctx->hist_pool.ensure(hist_stride * 4 * beam_slots * G); if (rc) return rc;
            if (LY.total > 160 * 1024 - 64) return fail(FLORIA_E_UNSUPPORTED, "ploidy*beam needs more LDS than a CU has");
            if (!wide && !slab && !fast && LY.total > 48 * 1024)
                HIPCHK(hipFuncSetAttribute((const void*)fl::beam_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LY.total));
            if (wide && WL.total > 48 * 1024) {
                if (any_q0) HIPCHK(hipFuncSetAttribute((const void*)fl::beam_wide_kernel<A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WL.total));
                else HIPCHK(hipFuncSetAttribute((const void*)fl::beam_wide_kernel<A, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WL.total));
            }
        }
        // optimise: sizes
        uint64_t cand_cap = 1;
        while (cand_cap < (uint64_t)n_max * std::max(1u, p - 1)) cand_cap <<= 1;
        const uint32_t mean_n = (uint32_t)(tot_reads / std::max<uint32_t>(1, n_jobs));
        uint32_t threads = mean_n >= 384 ? 1024 : (mean_n >= 96 ? 512 : 128);
        if (const char* te = getenv("FLORIA_HIP_OPT_THREADS")) { const int tv = atoi(te); if (tv == 1024 || tv == 512 || tv == 128) threads = (uint32_t)tv; }   // dev knob
        const size_t moved_bytes = ((((size_t)n_max + 31) / 32) * 4 + 15) & ~(size_t)15;
        const size_t hist_bytes = (size_t)span_max * p * A * 8;
        const size_t meta_bytes = n_max <= (uint32_t)fl::OPT_META_MAX ? (((size_t)n_max * 12 + 15) & ~(size_t)15) : 0;
        const bool hl = hist_bytes + (size_t)span_max * p + 32 + moved_bytes + meta_bytes <= 60 * 1024 && !getenv("FLORIA_HIP_OPT_GLOBAL");
        const size_t code_bytes = hl ? ((size_t)span_max * p + 15) & ~(size_t)15 : 0;       // one byte per (position, partition), see optimize_kernel.h
        const size_t lds = moved_bytes + meta_bytes + (hl ? ((hist_bytes + 15) & ~(size_t)15) + code_bytes : 0) + 16;
        // where the ploidy-specialised instances apply (75-92 VGPRs), three 512-thread workgroups per CU beat one of 1024 threads
        if (A == 2 && hl && p <= 5 && threads == 1024 && !getenv("FLORIA_HIP_OPT_THREADS") && !getenv("FLORIA_HIP_NO_SPECIALIZED")) threads = 512;
        uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)((156 * 1024) / (lds + 8 * 1024)), 2048 / threads));
        per_cu = std::min<uint32_t>(per_cu, 8);
        const uint32_t opt_slots = std::min<uint32_t>((uint32_t)ctx->n_cu * per_cu, n_jobs);
        {
            int rc = ctx->opt_hist.ensure((uint64_t)opt_slots * G * span_max * p * A * 8); if (rc) return rc;
            rc = ctx->opt_dist.ensure((uint64_t)opt_slots * G * n_max * p * 8); if (rc) return rc;
            rc = ctx->opt_gain.ensure((uint64_t)opt_slots * G * cand_cap * 8); if (rc) return rc;
            rc = ctx->opt_key.ensure((uint64_t)opt_slots * G * cand_cap * 4); if (rc) return rc;
            rc = ctx->opt_moves.ensure((uint64_t)opt_slots * G * n_max * 4); if (rc) return rc;
        }
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t nj = group_off[g + 1] - group_off[g];
            if (nj == 0) continue;
            hipStream_t st = gs[g];
            const uint32_t* gjobs = d_jobs + group_off[g];
            uint32_t* gqueue = d_queue + 2 * g;              // [0] beam, [1] optimise
            if (!shortcut) {
                const uint32_t slots = std::min(beam_slots, nj);
                fl::BeamArgs a{};
                a.bs = bs; a.job_block = gjobs; a.n_jobs = nj; a.ploidy = p; a.beam = B; a.span_max = span_max; a.n_max = n_max;
                a.queue_head = gqueue; a.blk_done = d_done;
                a.state_pool = ctx->state_pool.as<uint64_t>() + (state_bytes / 8) * beam_slots * g;
                a.hist_pool = ctx->hist_pool.as<uint32_t>() + hist_stride * beam_slots * g; a.hist_stride = hist_stride;
                a.binom_tab = ctx->d_binom.as<double>(); a.binom_nmax = ctx->binom_nmax;
                a.eps = prm->epsilon; a.div_factor = DIV_FACTOR; a.cutoff = cutoff;
                const uint64_t* H = ctx->d_hash.as<uint64_t>();
                a.Rq1 = H; a.Rp1 = H + ctx->hash_len; a.Rq2 = H + 2ull * ctx->hash_len; a.Rp2 = H + 3ull * ctx->hash_len;
                a.part_out = d_beam_part; a.min_margin_bits = d_margin; a.diag = d_diag; a.steps_done = d_steps;
                a.prof = (unsigned long long*)(d_diag + 4);
                HIPCHK(hipMemsetAsync(gqueue, 0, 4, st));
                int t = T.begin(K_BEAM, st);
                if (wide) {
                    const uint32_t wslots = std::min<uint32_t>(slots, (uint32_t)ctx->n_cu * std::max<uint32_t>(1, (uint32_t)((158 * 1024) / (WL.total + 512))));
                    if (any_q0) hipLaunchKernelGGL((fl::beam_wide_kernel<A, true>), dim3(wslots), dim3(64), WL.total, st, a);
                    else hipLaunchKernelGGL((fl::beam_wide_kernel<A, false>), dim3(wslots), dim3(64), WL.total, st, a);
                } else if (slab) {
                    const bool spec = A == 2 && !any_q0 && B == 10 && p >= 2 && p <= 5 && !getenv("FLORIA_HIP_NO_SPECIALIZED");
                    if (any_q0) hipLaunchKernelGGL((fl::beam_slab_kernel<A, true>), dim3(slots), dim3(64), SL.total, st, a);
                    else if (spec && p == 2) hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 2, 10>), dim3(slots), dim3(64), SL.total, st, a);
                    else if (spec && p == 3) hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 3, 10>), dim3(slots), dim3(64), SL.total, st, a);
                    else if (spec && p == 4) hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 4, 10>), dim3(slots), dim3(64), SL.total, st, a);
                    else if (spec && p == 5) hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 5, 10>), dim3(slots), dim3(64), SL.total, st, a);
                    else hipLaunchKernelGGL((fl::beam_slab_kernel<A, false>), dim3(slots), dim3(64), SL.total, st, a);
                } else if (fast) {
                    const fl::FastLds FL = fl::fast_lds_layout(LM, any_q0);
                    if (any_q0) hipLaunchKernelGGL((fl::beam_fast_kernel<A, true>), dim3(slots), dim3(64), FL.total, st, a);
                    else hipLaunchKernelGGL((fl::beam_fast_kernel<A, false>), dim3(slots), dim3(64), FL.total, st, a);
                } else
                    hipLaunchKernelGGL(fl::beam_kernel<A>, dim3(slots), dim3(64), LY.total, st, a);
                T.end(t);
                HIPCHK(hipGetLastError());
                ctx->timing.beam_launches++;
            }
            // ---- optimise + MEC stats ------------------------------------------------------------------------------
            {
                const uint32_t slots = std::min(opt_slots, nj);
                const uint64_t so = (uint64_t)opt_slots * g;
                fl::OptArgs a{};
                a.bs = bs; a.job_block = gjobs; a.n_jobs = nj; a.ploidy = p; a.max_ploidy = P; a.span_max = span_max; a.n_max = n_max;
                a.queue_head = gqueue + 1; a.blk_done = d_done; a.eps = prm->epsilon;
                a.part_in = d_beam_part; a.part_out = d_planes + (uint64_t)(p - 1) * tot_reads;
                a.hist_pool = ctx->opt_hist.as<uint64_t>() + so * span_max * p * A; a.dist_pool = ctx->opt_dist.as<double>() + so * n_max * p;
                a.cand_gain_pool = ctx->opt_gain.as<uint64_t>() + so * cand_cap; a.cand_key_pool = ctx->opt_key.as<uint32_t>() + so * cand_cap;
                a.moves_pool = ctx->opt_moves.as<uint32_t>() + so * n_max; a.cand_cap = cand_cap;
                a.mec = d_mec; a.num_alleles = d_na; a.iters = d_iters;
                a.prof = (unsigned long long*)(d_diag + 4);
                HIPCHK(hipMemsetAsync(gqueue + 1, 0, 4, st));
                int t = T.begin(K_OPT, st);
                auto launch = [&](auto kern) -> hipError_t {
                    if (lds > 48 * 1024) { hipError_t e2 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e2 != hipSuccess) return e2; }
                    hipLaunchKernelGGL(kern, dim3(slots), dim3(threads), lds, st, a);
                    return hipGetLastError();
                };
                hipError_t le;
                const bool ospec = A == 2 && hl && threads >= 512 && p <= 5 && !getenv("FLORIA_HIP_NO_SPECIALIZED");
                if (ospec && threads == 1024)
                    le = p == 1 ? launch(fl::optimize_kernel<2, true, 1024, 1>) : p == 2 ? launch(fl::optimize_kernel<2, true, 1024, 2>) : p == 3 ? launch(fl::optimize_kernel<2, true, 1024, 3>)
                       : p == 4 ? launch(fl::optimize_kernel<2, true, 1024, 4>) : launch(fl::optimize_kernel<2, true, 1024, 5>);
                else if (ospec)
                    le = p == 1 ? launch(fl::optimize_kernel<2, true, 512, 1>) : p == 2 ? launch(fl::optimize_kernel<2, true, 512, 2>) : p == 3 ? launch(fl::optimize_kernel<2, true, 512, 3>)
                       : p == 4 ? launch(fl::optimize_kernel<2, true, 512, 4>) : launch(fl::optimize_kernel<2, true, 512, 5>);
                else if (hl) le = threads == 1024 ? launch(fl::optimize_kernel<A, true, 1024>) : threads == 512 ? launch(fl::optimize_kernel<A, true, 512>) : launch(fl::optimize_kernel<A, true, 128>);
                else    le = threads == 1024 ? launch(fl::optimize_kernel<A, false, 1024>) : threads == 512 ? launch(fl::optimize_kernel<A, false, 512>) : launch(fl::optimize_kernel<A, false, 128>);
                T.end(t);
                if (le != hipSuccess) return fail(FLORIA_E_DEVICE, std::string("optimize_kernel launch: ") + hipGetErrorString(le));
                ctx->timing.optimize_launches++;
            }
            // ---- stop rule ---------------------------------------------------------------------------------------------
            {
                fl::SelectArgs s{};
                s.job_block = gjobs; s.n_jobs = nj; s.ploidy = p; s.max_ploidy = P; s.stopping_heuristic = prm->stopping_heuristic; s.eps = prm->epsilon;
                const double eps = prm->epsilon, pl = (double)p;    // graph_processing.rs:204-220
                if (prm->ploidy_sensitivity == 1)      s.mec_threshold = 1.0 / (1.0 - eps) / (1.0 + 1.0 / (std::pow(pl, 0.50) + 1.00));
                else if (prm->ploidy_sensitivity == 2) s.mec_threshold = 1.0 / (1.0 - eps) / (1.0 + 1.0 / (std::pow(pl, 1.00) + 1. / 3.));
                else                                   s.mec_threshold = 1.0 / (1.0 - eps) / (1.0 + 1.0 / (std::pow(pl, 1.00) + 1.00));
                s.mec = d_mec; s.num_alleles = d_na; s.blk_done = d_done; s.best_ploidy = d_best; s.tried = d_tried;
                int t = T.begin(K_SEL, st);
                hipLaunchKernelGGL(fl::select_kernel, dim3((nj + 255) / 256), dim3(256), 0, st, s);
                T.end(t);
                HIPCHK(hipGetLastError());
            }
        }
    }
    // join: the main stream continues after every group
    if (G > 1 && forked) {
        for (uint32_t g = 1; g < G; ++g) { HIPCHK(hipEventRecord(ctx->ev_join[g], gs[g])); HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join[g], 0)); }
    }
    T.end(t_phase);
    ctx->timing.streams = G;
    return 0;
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* floria_hip_last_error(void) { return g_err.c_str(); }
const char* floria_hip_version(void) { return "floria_hip 0.1.0 (gfx950)"; }

int floria_hip_create(int device, floria_hip_ctx** out) {
    if (!out) return fail(FLORIA_E_INVALID, "null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return fail(FLORIA_E_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(FLORIA_E_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    floria_hip_ctx* c = new floria_hip_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(FLORIA_E_DEVICE, "hipStreamCreate failed"); }
    // phred_scale (utils_frags.rs:702-711): w = 1f32 - 10f32.powf(q as f32 / -10.), exact multiple of 2^-24
    uint32_t w24[256];
    for (int q = 0; q < 256; ++q) {
        float x = (float)q / -10.0f;
        float prob = 1.0f - powf(10.0f, x);
        double s = (double)prob * 16777216.0;
        w24[q] = (uint32_t)s;
        if ((double)w24[q] != s) { delete c; return fail(FLORIA_E_DEVICE, "quality weight is not a multiple of 2^-24"); }
    }
    memcpy(c->w24, w24, sizeof(w24));
    uint64_t s = 0xf10a1a2024ull;
    for (int k = 0; k < FLORIA_MAX_PLOIDY; ++k) { c->Rk1[k] = splitmix64(s) | 1ull; c->Rk2[k] = splitmix64(s) | 1ull; }
    e = hipMemcpyToSymbol(HIP_SYMBOL(fl::c_rk1), c->Rk1, sizeof(c->Rk1));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(fl::c_rk2), c->Rk2, sizeof(c->Rk2));
    if (e != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return fail(FLORIA_E_DEVICE, std::string("hash multiplier upload: ") + hipGetErrorString(e)); }
    if (int rc = ensure_hash(c, 4 * fl::HASH_M)) { (void)hipStreamDestroy(c->stream); delete c; return rc; }
    *out = c;
    return 0;
}

void floria_hip_destroy(floria_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&c->d_binom, &c->d_hash, &c->state_pool, &c->hist_pool, &c->opt_hist, &c->opt_dist, &c->opt_gain, &c->opt_key, &c->opt_moves, &c->misc, &c->misc0, &c->graph_buf, &c->graph_hist, &c->graph_sort}) b->release();
    for (uint32_t g = 0; g < floria_hip_ctx::MAX_GROUPS; ++g) {
        if (c->gstream[g]) (void)hipStreamDestroy(c->gstream[g]);
        if (c->ev_join[g]) (void)hipEventDestroy(c->ev_join[g]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_rids) (void)hipEventDestroy(c->ev_rids);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int floria_hip_set_slots(floria_hip_ctx* ctx, uint32_t beam_slots) {
    if (!ctx) return fail(FLORIA_E_INVALID, "null ctx");
    ctx->user_slots = beam_slots;
    return 0;
}

int floria_hip_last_timing(const floria_hip_ctx* ctx, floria_timing* out) {
    if (!ctx || !out) return fail(FLORIA_E_INVALID, "null argument");
    *out = ctx->timing;
    return 0;
}

// ---- get_range_with_lengths (utils_frags.rs:405-463) ------------------------------------------------------
int floria_hip_block_ranges(const uint64_t* g, uint32_t n, uint64_t block_length, uint64_t overlap_len,
                            double minimal_density, floria_ranges** out) {
    if (!g || !out) return fail(FLORIA_E_INVALID, "null argument");
    if (n == 0) return fail(FLORIA_E_INVALID, "empty snp_to_genome_pos");
    std::vector<uint32_t> S, E;
    uint64_t cum_pos = 0, last_pos = g[0];
    uint32_t left_endpoint = 0, new_left_end = 0;
    bool hit_new_left = false;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t pos = g[i];
        if (i == n - 1) { S.push_back(left_endpoint); E.push_back(i); break; }                         // :418-421
        if (pos < last_pos) return fail(FLORIA_E_INVALID, "VCF malformed. Positions are not increasing");   // :422-425
        cum_pos += pos - last_pos;
        last_pos = pos;
        if (cum_pos > block_length - overlap_len && !hit_new_left) { new_left_end = i; hit_new_left = true; }
        if (cum_pos > block_length) {
            cum_pos = 0;
            const double snp_density = (double)(i - left_endpoint) / (double)block_length;
            if (snp_density > minimal_density) { S.push_back(left_endpoint); E.push_back(i - 1); }
            left_endpoint = (g[new_left_end] + block_length < g[new_left_end + 1]) ? new_left_end : new_left_end + 1;   // :447-453
            last_pos = g[left_endpoint];
            hit_new_left = false;
        }
    }
    floria_ranges* r = (floria_ranges*)calloc(1, sizeof(floria_ranges));
    if (!r) return fail(FLORIA_E_NOMEM, "calloc");
    r->n = (uint32_t)S.size();
    r->start = (uint32_t*)malloc(sizeof(uint32_t) * (S.size() + 1));
    r->end = (uint32_t*)malloc(sizeof(uint32_t) * (S.size() + 1));
    for (size_t i = 0; i < S.size(); ++i) { r->start[i] = S[i] + 1; r->end[i] = E[i] + 1; }   // 1-indexed :461
    *out = r;
    return 0;
}
void floria_hip_ranges_free(floria_ranges* r) { if (r) { free(r->start); free(r->end); free(r); } }

// ---- contig upload ------------------------------------------------------------------------------------------
int floria_hip_contig_upload(floria_hip_ctx* ctx, const floria_pileup* p, floria_hip_contig** out) {
    if (!ctx || !out) return fail(FLORIA_E_INVALID, "null argument");
    *out = nullptr;
    uint32_t ml = 0, ma = 0;
    int rc = validate_pileup(p, &ml, &ma);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    floria_hip_contig* c = new floria_hip_contig();
    c->ctx = ctx; c->n_reads = p->n_reads; c->max_len = ml; c->n_alleles = ma >= 2 ? 4 : 2;
    const uint64_t nc = p->n_reads ? p->read_off[p->n_reads] : 0;
    c->n_cells = nc;
    c->h_first.assign(p->first, p->first + p->n_reads);
    c->h_last.assign(p->last, p->last + p->n_reads);
    c->h_read_off.assign(p->read_off, p->read_off + p->n_reads + (p->n_reads ? 1 : 0));
    std::vector<uint32_t> aq(nc);     // allele << 28 | Q24 weight of the cell (weights are < 2^24)
    std::vector<uint64_t> tw((size_t)p->n_reads * 2, 0);
    std::vector<uint32_t> meta((size_t)p->n_reads * 8, 0);
    for (uint32_t r = 0; r < p->n_reads; ++r) {
        uint64_t t1 = 0, t2 = 0;
        for (uint64_t i = p->read_off[r]; i < p->read_off[r + 1]; ++i) {
            aq[i] = ((uint32_t)p->allele[i] << 28) | ctx->w24[p->qual[i]];
            if (p->qual[i] == 0) c->has_q0 = true;
            const uint32_t idx = fl::hash_idx(p->snp[i], p->allele[i]);
            const uint64_t w = ctx->w24[p->qual[i]];
            t1 += ctx->h_rq1[idx] * w; t2 += ctx->h_rq2[idx] * w;
        }
        tw[2 * (size_t)r] = t1; tw[2 * (size_t)r + 1] = t2;
        uint32_t* mr = &meta[8 * (size_t)r];
        mr[0] = p->read_off[r]; mr[1] = p->read_off[r + 1] - p->read_off[r]; mr[2] = p->first[r]; mr[3] = p->last[r];
        mr[4] = (uint32_t)t1; mr[5] = (uint32_t)(t1 >> 32); mr[6] = (uint32_t)t2; mr[7] = (uint32_t)(t2 >> 32);
    }
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        int r2 = b.ensure(bytes + 16);            // 16-B tail padding: the beam kernel's LDS-DMA moves cells in 16-B pieces
        if (r2) return r2;
        if (bytes) { hipError_t e = hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream); if (e != hipSuccess) return fail(FLORIA_E_DEVICE, hipGetErrorString(e)); }
        return 0;
    };
    rc = up(c->d_read_off, p->read_off, (size_t)(p->n_reads + 1) * 4 * (p->n_reads ? 1 : 0));
    if (!rc) rc = up(c->d_first, p->first, (size_t)p->n_reads * 4);
    if (!rc) rc = up(c->d_last, p->last, (size_t)p->n_reads * 4);
    if (!rc) rc = up(c->d_snp, p->snp, nc * 4);
    if (!rc) rc = up(c->d_aq, aq.data(), nc * 4);
    if (!rc) rc = up(c->d_tw, tw.data(), tw.size() * 8);
    if (!rc) rc = up(c->d_meta, meta.data(), meta.size() * 4);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(FLORIA_E_DEVICE, "upload sync failed");
    if (rc) { floria_hip_contig_free(c); return rc; }
    c->dev.read_off = c->d_read_off.as<uint32_t>(); c->dev.first = c->d_first.as<uint32_t>(); c->dev.last = c->d_last.as<uint32_t>();
    c->dev.cell_snp = c->d_snp.as<uint32_t>(); c->dev.cell_aw = c->d_aq.as<uint32_t>(); c->dev.tw = c->d_tw.as<uint64_t>(); c->dev.meta = c->d_meta.as<uint32_t>(); c->dev.n_reads = p->n_reads;
    *out = c;
    return 0;
}
void floria_hip_contig_free(floria_hip_contig* c) {
    if (!c) return;
    for (DevBuf* b : {&c->d_read_off, &c->d_first, &c->d_last, &c->d_snp, &c->d_aq, &c->d_tw, &c->d_meta}) b->release();
    delete c;
}

// ---- S1 --------------------------------------------------------------------------------------------------------
int floria_hip_phase_blocks_batch(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs, uint32_t n_contigs,
                                  const uint32_t* blk_contig, const uint32_t* blk_start, const uint32_t* blk_end,
                                  uint32_t n_blocks, const floria_params* prm, floria_block_result** out) {
    if (!ctx || !out || !prm || (n_blocks && (!blk_start || !blk_end || !contigs))) return fail(FLORIA_E_INVALID, "null argument");
    *out = nullptr;
    if (prm->max_ploidy < 1 || prm->max_ploidy > FLORIA_MAX_PLOIDY) return fail(FLORIA_E_INVALID, "max_ploidy must be in 1..16");
    if (prm->beam < 1) return fail(FLORIA_E_INVALID, "beam (max_number_solns) must be >= 1");
    if (!(prm->epsilon > 0.0 && prm->epsilon < 1.0)) return fail(FLORIA_E_INVALID, "epsilon must be in (0,1)");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->timing = floria_timing{};
    ctx->batch_token = 0;
    const uint32_t P = prm->max_ploidy;

    // ---- block read lists: find_reads_in_interval on the device (blocks_kernel.h) -------------------------------
    std::vector<uint32_t> bc(n_blocks, 0);
    uint32_t n_max = 1, span_max = 1, len_max = 1, nall = 2;
    bool any_q0 = false;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t ci = blk_contig ? blk_contig[b] : 0;
        if (ci >= n_contigs || !contigs[ci]) return fail(FLORIA_E_INVALID, "blk_contig out of range");
        if (contigs[ci]->ctx != ctx) return fail(FLORIA_E_INVALID, "contig belongs to another context");
        bc[b] = ci;
    }
    for (uint32_t i = 0; i < n_contigs; ++i) if (contigs[i]) {
        len_max = std::max(len_max, contigs[i]->max_len);
        nall = std::max(nall, contigs[i]->n_alleles);
        any_q0 = any_q0 || contigs[i]->has_q0;
    }
    std::vector<fl::ContigDev> cdev(n_contigs);
    for (uint32_t i = 0; i < n_contigs; ++i) if (contigs[i]) cdev[i] = contigs[i]->dev;
    struct Seg { size_t off, bytes; };
    size_t cursor = 0;
    auto seg = [&](size_t bytes) { Seg s{cursor, bytes}; cursor += (bytes + 255) & ~(size_t)255; return s; };
    const Seg s_cdev = seg(sizeof(fl::ContigDev) * std::max(1u, n_contigs)), s_bc = seg(4ull * n_blocks + 4), s_bs = seg(4ull * n_blocks + 4),
              s_be = seg(4ull * n_blocks + 4), s_p0 = seg(4ull * n_blocks + 4), s_sp = seg(4ull * n_blocks + 4), s_cnt = seg(4ull * n_blocks + 4),
              s_bytes = seg(8ull * n_blocks + 8), s_roff = seg(8ull * (n_blocks + 1));
    int rc = ctx->misc0.ensure(cursor + 256); if (rc) return rc;
    char* M0 = ctx->misc0.as<char>();
    EventTimer T(ctx->stream);
    int th = T.begin(K_H2D);
    HIPCHK(hipMemcpyAsync(M0 + s_cdev.off, cdev.data(), sizeof(fl::ContigDev) * n_contigs, hipMemcpyHostToDevice, ctx->stream));
    if (n_blocks) {
        HIPCHK(hipMemcpyAsync(M0 + s_bc.off, bc.data(), 4ull * n_blocks, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(M0 + s_bs.off, blk_start, 4ull * n_blocks, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(M0 + s_be.off, blk_end, 4ull * n_blocks, hipMemcpyHostToDevice, ctx->stream));
    }
    T.end(th);
    fl::ScanArgs sa{};
    sa.contigs = (const fl::ContigDev*)(M0 + s_cdev.off);
    sa.blk_contig = (const uint32_t*)(M0 + s_bc.off); sa.blk_start = (const uint32_t*)(M0 + s_bs.off); sa.blk_end = (const uint32_t*)(M0 + s_be.off);
    sa.n_blocks = n_blocks; sa.max_ploidy = P;
    sa.cnt = (uint32_t*)(M0 + s_cnt.off); sa.pos0 = (uint32_t*)(M0 + s_p0.off); sa.span = (uint32_t*)(M0 + s_sp.off); sa.bytes = (uint64_t*)(M0 + s_bytes.off);
    std::vector<uint32_t> cnt(n_blocks, 0), span(n_blocks, 0);
    std::vector<uint64_t> blk_bytes(n_blocks, 0), roff(n_blocks + 1, 0);
    if (n_blocks) {
        int tk = T.begin(K_SEL);
        hipLaunchKernelGGL(fl::block_reads_kernel<false>, dim3(n_blocks), dim3(64), 0, ctx->stream, sa);
        T.end(tk);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(cnt.data(), sa.cnt, 4ull * n_blocks, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(span.data(), sa.span, 4ull * n_blocks, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipMemcpyAsync(blk_bytes.data(), sa.bytes, 8ull * n_blocks, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    uint64_t algo_bytes = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        roff[b + 1] = roff[b] + cnt[b];
        n_max = std::max(n_max, cnt[b]); span_max = std::max(span_max, span[b]);
        algo_bytes += blk_bytes[b];
    }
    if (n_max >= (1u << 20)) return fail(FLORIA_E_UNSUPPORTED, "more than 2^20 reads in one block");
    const uint64_t tot = roff[n_blocks];
    std::vector<uint32_t> jobs;
    for (uint32_t b = 0; b < n_blocks; ++b) if (cnt[b]) jobs.push_back(b);
    std::stable_sort(jobs.begin(), jobs.end(), [&](uint32_t a, uint32_t b2) { return cnt[a] > cnt[b2]; });
    // job groups (longest-first inside each group, dealt round-robin so every group sees the same size mix)
    uint32_t G = ctx->user_groups ? ctx->user_groups : 2;
    if (const char* ge = getenv("FLORIA_HIP_GROUPS")) G = (uint32_t)std::max(1, atoi(ge));
    G = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(G, floria_hip_ctx::MAX_GROUPS), (uint32_t)(jobs.size() / 1024)));
    std::vector<uint32_t> group_off(G + 1, 0);
    if (G > 1) {
        std::vector<uint32_t> dealt; dealt.reserve(jobs.size());
        for (uint32_t g = 0; g < G; ++g) { for (size_t j = g; j < jobs.size(); j += G) dealt.push_back(jobs[j]); group_off[g + 1] = (uint32_t)dealt.size(); }
        jobs.swap(dealt);
    } else group_off[1] = (uint32_t)jobs.size();

    rc = ensure_binom(ctx, prm->epsilon, len_max); if (rc) return rc;
    if (span_max > fl::HASH_M) return fail(FLORIA_E_UNSUPPORTED, "a block's reads span more than 65536 SNPs");
    rc = ensure_hash(ctx, span_max * nall); if (rc) return rc;

    // ---- device staging of the per-call arrays ----------------------------------------------------------------------
    cursor = 0;
    const Seg s_rids = seg(4ull * tot + 4), s_jobs = seg(4ull * jobs.size() + 4), s_planes = seg((uint64_t)P * tot + 16), s_bpart = seg(tot + 16),
              s_out = seg(tot + 16), s_mec = seg(8ull * n_blocks * P + 8), s_na = seg(8ull * n_blocks * P + 8), s_it = seg(4ull * n_blocks * P + 4),
              s_done = seg(n_blocks + 4), s_best = seg(4ull * n_blocks + 4), s_tried = seg(4ull * n_blocks + 4), s_q = seg(8 * floria_hip_ctx::MAX_GROUPS + 16), s_margin = seg(16),
              s_diag = seg(16 + 8 * 32), s_steps = seg(16);
    rc = ctx->misc.ensure(cursor + 256); if (rc) return rc;
    char* M = ctx->misc.as<char>();
    th = T.begin(K_H2D);
    HIPCHK(hipMemcpyAsync(M0 + s_roff.off, roff.data(), 8ull * (n_blocks + 1), hipMemcpyHostToDevice, ctx->stream));
    if (!jobs.empty()) HIPCHK(hipMemcpyAsync(M + s_jobs.off, jobs.data(), 4ull * jobs.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_mec.off, 0, s_mec.bytes, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_na.off, 0, s_na.bytes, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_it.off, 0, s_it.bytes, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_done.off, 0, s_done.bytes, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_best.off, 0, s_best.bytes, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_tried.off, 0, s_tried.bytes, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_out.off, 0, s_out.bytes, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_diag.off, 0, 16 + 8 * 32, ctx->stream));
    HIPCHK(hipMemsetAsync(M + s_steps.off, 0, 16, ctx->stream));
    const double inf = std::numeric_limits<double>::infinity();
    HIPCHK(hipMemcpyAsync(M + s_margin.off, &inf, 8, hipMemcpyHostToDevice, ctx->stream));
    T.end(th);
    if (n_blocks) {
        sa.roff = (const uint64_t*)(M0 + s_roff.off); sa.rids = (uint32_t*)(M + s_rids.off);
        int tk = T.begin(K_SEL);
        hipLaunchKernelGGL(fl::block_reads_kernel<true>, dim3(n_blocks), dim3(64), 0, ctx->stream, sa);
        T.end(tk);
        HIPCHK(hipGetLastError());
        if (!ctx->ev_rids) HIPCHK(hipEventCreateWithFlags(&ctx->ev_rids, hipEventDisableTiming));
        if (!ctx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        HIPCHK(hipEventRecord(ctx->ev_rids, ctx->stream));
    }

    fl::BlockSet bs{};
    bs.contigs = (const fl::ContigDev*)(M0 + s_cdev.off);
    bs.blk_contig = (const uint32_t*)(M0 + s_bc.off); bs.blk_start = (const uint32_t*)(M0 + s_bs.off); bs.blk_end = (const uint32_t*)(M0 + s_be.off);
    bs.blk_pos0 = (const uint32_t*)(M0 + s_p0.off); bs.blk_span = (const uint32_t*)(M0 + s_sp.off);
    bs.blk_read_off = (const uint64_t*)(M0 + s_roff.off); bs.blk_read = (const uint32_t*)(M + s_rids.off); bs.n_blocks = n_blocks;

    // ---- result buffers (allocated first: the read-id lists are copied back while the launch loop runs) -----------------
    floria_block_result* R = (floria_block_result*)calloc(1, sizeof(floria_block_result));
    if (!R) return fail(FLORIA_E_NOMEM, "calloc");
    R->n_blocks = n_blocks; R->max_ploidy = P;
    R->best_ploidy = (uint32_t*)calloc(n_blocks + 1, 4);
    R->ploidies_tried = (uint32_t*)calloc(n_blocks + 1, 4);
    R->read_off = (uint64_t*)calloc(n_blocks + 1, 8);
    R->read_id = (uint32_t*)g_big.get(4 * (tot + 1));
    R->part = (uint8_t*)g_big.get(tot + 1);
    R->mec = (double*)calloc((size_t)n_blocks * P + 1, 8);
    struct ResultGuard { floria_hip_ctx* c; floria_block_result* r; ~ResultGuard() { if (r) { sync_all(c); floria_hip_block_result_free(r); } } } guard{ctx, R};   // every early return below
    if (!R->best_ploidy || !R->ploidies_tried || !R->read_off || !R->read_id || !R->part || !R->mec) { return fail(FLORIA_E_NOMEM, "malloc"); }
    bool p1_shortcut = false;
    auto run = nall == 2 ? run_phase<2> : run_phase<4>;
    rc = run(ctx, any_q0, bs, jobs, group_off, (const uint32_t*)(M + s_jobs.off), tot, n_max, span_max, prm, (uint8_t*)(M + s_planes.off),
             (uint8_t*)(M + s_bpart.off), (double*)(M + s_mec.off), (double*)(M + s_na.off), (uint32_t*)(M + s_it.off),
             (uint8_t*)(M + s_done.off), (uint32_t*)(M + s_best.off), (uint32_t*)(M + s_tried.off), (uint32_t*)(M + s_q.off),
             (unsigned long long*)(M + s_margin.off), (uint32_t*)(M + s_diag.off), (unsigned long long*)(M + s_steps.off), T, p1_shortcut);
    if (rc) { sync_all(ctx); return rc; }
    int t_rids = -1;
    if (n_blocks && tot) {              // everything is queued: the copy (pageable destination, the host may block here) overlaps the kernels
        HIPCHK(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_rids, 0));
        t_rids = T.begin(K_D2H, ctx->copy_stream);
        hipError_t ce = hipMemcpyAsync(R->read_id, M + s_rids.off, 4ull * tot, hipMemcpyDeviceToHost, ctx->copy_stream);
        T.end(t_rids);
        if (ce != hipSuccess) { sync_all(ctx); return fail(FLORIA_E_DEVICE, hipGetErrorString(ce)); }
    }
    {
        int t = T.begin(K_SEL);
        if (n_blocks) hipLaunchKernelGGL(fl::gather_kernel, dim3(n_blocks), dim3(64), 0, ctx->stream, n_blocks, bs.blk_read_off,
                                         (const uint32_t*)(M + s_best.off), (const uint8_t*)(M + s_planes.off), tot, (uint8_t*)(M + s_out.off));
        T.end(t);
        HIPCHK(hipGetLastError());
    }

    int td = T.begin(K_D2H);
    hipError_t e = hipSuccess;
    uint32_t diag[4] = {0, 0, 0, 0};
    unsigned long long steps = 0;
    double margin = inf;
    if (n_blocks) {
        e = hipMemcpyAsync(R->best_ploidy, M + s_best.off, 4ull * n_blocks, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(R->ploidies_tried, M + s_tried.off, 4ull * n_blocks, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(R->mec, M + s_mec.off, 8ull * n_blocks * P, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && tot) e = hipMemcpyAsync(R->part, M + s_out.off, tot, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(diag, M + s_diag.off, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&steps, M + s_steps.off, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&margin, M + s_margin.off, 8, hipMemcpyDeviceToHost, ctx->stream);
    T.end(td);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && ctx->copy_stream) e = hipStreamSynchronize(ctx->copy_stream);
    if (e != hipSuccess) { sync_all(ctx); return fail(FLORIA_E_DEVICE, std::string("phase_blocks: ") + hipGetErrorString(e)); }
    if (diag[1]) { return fail(FLORIA_E_DEVICE, "internal: beam slab free-list underflow"); }
#ifdef FLORIA_PROF
    { unsigned long long prof[32]; (void)hipMemcpy(prof, M + s_diag.off + 16, sizeof(prof), hipMemcpyDeviceToHost); fprintf(stderr, "[prof]"); for (int i = 0; i < 32; ++i) fprintf(stderr, " %d:%.1fM", i, prof[i] / 1e6); fprintf(stderr, "\n"); }
#endif
    memcpy(R->read_off, roff.data(), 8ull * (n_blocks + 1));
    if (p1_shortcut && !jobs.empty()) margin = std::min(margin, std::fabs(0.0 - std::log(PROB_CUTOFF)));   // the ploidy-1 decisions: p_k - lse == 0
    R->min_prune_margin = margin;
    ctx->timing.beam_ms = T.sum(K_BEAM); ctx->timing.optimize_ms = T.sum(K_OPT); ctx->timing.select_ms = T.sum(K_SEL);
    ctx->timing.h2d_ms = T.sum(K_H2D); ctx->timing.d2h_ms = T.sum(K_D2H); ctx->timing.total_ms = T.span(); ctx->timing.phase_ms = T.sum(K_PHASE);
    ctx->timing.algorithmic_bytes = algo_bytes; ctx->timing.beam_steps = steps;
    for (uint32_t b = 0; b < n_blocks; ++b) {          // a beam launch for ploidy p phases the blocks with tried >= p (ploidy 1 needs no launch)
        const uint32_t launches_b = R->ploidies_tried[b] - ((p1_shortcut && R->ploidies_tried[b]) ? 1 : 0);
        ctx->timing.beam_launch_bytes += blk_bytes[b] * launches_b; ctx->timing.jobs += R->ploidies_tried[b];
    }
    ctx->batch_token = R->batch_token = ++ctx->token_counter;
    ctx->last_bs = bs; ctx->last_part = (const uint8_t*)(M + s_out.off); ctx->last_best = (const uint32_t*)(M + s_best.off); ctx->last_nall = nall;
    ctx->last_bc = bc; ctx->last_start.assign(blk_start, blk_start + n_blocks); ctx->last_end.assign(blk_end, blk_end + n_blocks);
    guard.r = nullptr;
    *out = R;
    return 0;
}

int floria_hip_phase_blocks_resident(floria_hip_ctx* ctx, const floria_hip_contig* contig, const uint32_t* blk_start,
                                     const uint32_t* blk_end, uint32_t n_blocks, const floria_params* params,
                                     floria_block_result** out) {
    if (!contig) return fail(FLORIA_E_INVALID, "null contig");
    const floria_hip_contig* arr[1] = {contig};
    return floria_hip_phase_blocks_batch(ctx, arr, 1, nullptr, blk_start, blk_end, n_blocks, params, out);
}

int floria_hip_phase_blocks(floria_hip_ctx* ctx, const floria_pileup* pileup, const uint32_t* blk_start, const uint32_t* blk_end,
                            uint32_t n_blocks, const floria_params* params, floria_block_result** out) {
    floria_hip_contig* c = nullptr;
    int rc = floria_hip_contig_upload(ctx, pileup, &c);
    if (rc) return rc;
    rc = floria_hip_phase_blocks_resident(ctx, c, blk_start, blk_end, n_blocks, params, out);
    floria_hip_contig_free(c);
    return rc;
}

void floria_hip_block_result_free(floria_block_result* r) {
    if (!r) return;
    free(r->best_ploidy); free(r->ploidies_tried); free(r->read_off); g_big.put(r->read_id); g_big.put(r->part); free(r->mec); free(r);
}

// ---- hap-graph nodes + edges on the resident batch ------------------------------------------------
int floria_hip_hap_graph(floria_hip_ctx* ctx, const floria_block_result* res, floria_hap_graph** out) {
    if (!ctx || !res || !out) return fail(FLORIA_E_INVALID, "null argument");
    *out = nullptr;
    if (!ctx->batch_token || res->batch_token != ctx->batch_token) return fail(FLORIA_E_INVALID, "the batch is no longer resident: call floria_hip_hap_graph directly after phase_blocks");
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t nb = res->n_blocks;
    floria_hap_graph* G = (floria_hap_graph*)calloc(1, sizeof(floria_hap_graph));
    if (!G) return fail(FLORIA_E_NOMEM, "calloc");
    G->n_blocks = nb;
    G->node_off = (uint64_t*)calloc(nb + 1, 8); G->edge_off = (uint64_t*)calloc(nb + 1, 8); G->pred = (int32_t*)calloc(nb + 1, 4);
    uint32_t range_max = 1, pmax = 1;
    {
        std::vector<int32_t> last_of_contig;                    // process_chunks: columns = non-empty blocks in block order (graph_processing.rs:306-323)
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t ci = ctx->last_bc[b];
            if (ci >= last_of_contig.size()) last_of_contig.resize(ci + 1, -1);
            const uint32_t p2 = res->best_ploidy[b];
            G->pred[b] = -1;
            G->node_off[b + 1] = G->node_off[b] + p2;
            G->edge_off[b + 1] = G->edge_off[b];
            if (p2) {
                G->pred[b] = last_of_contig[ci];
                if (G->pred[b] >= 0) G->edge_off[b + 1] += (uint64_t)res->best_ploidy[G->pred[b]] * p2;
                last_of_contig[ci] = (int32_t)b;
                range_max = std::max(range_max, ctx->last_end[b] - ctx->last_start[b] + 1);
                pmax = std::max(pmax, p2);
            }
        }
    }
    const uint64_t n_nodes = G->node_off[nb], n_edges = G->edge_off[nb];
    G->node_cov = (double*)calloc(n_nodes + 1, 8); G->edge_w = (uint32_t*)calloc(n_edges + 1, 4);
    if (!G->node_off || !G->edge_off || !G->pred || !G->node_cov || !G->edge_w) { floria_hip_hap_graph_free(G); return fail(FLORIA_E_NOMEM, "calloc"); }
    if (nb == 0) { *out = G; return 0; }
    const uint32_t A = ctx->last_nall;
    struct Seg { size_t off, bytes; };
    size_t cursor = 0;
    auto seg = [&](size_t bytes) { Seg sg{cursor, bytes}; cursor += (bytes + 255) & ~(size_t)255; return sg; };
    const Seg s_pred = seg(4ull * nb), s_noff = seg(8ull * (nb + 1)), s_eoff = seg(8ull * (nb + 1)), s_cov = seg(8ull * n_nodes + 8), s_ew = seg(4ull * n_edges + 4);
    int rc = ctx->graph_buf.ensure(cursor + 256);
    const size_t hist_bytes = (size_t)range_max * pmax * A * 8;
    const bool in_lds = hist_bytes <= 40 * 1024;
    uint64_t sort_cap = 1;
    while (sort_cap < (uint64_t)range_max * A) sort_cap <<= 1;
    if (!rc && !in_lds) rc = ctx->graph_hist.ensure((uint64_t)nb * hist_bytes);
    if (!rc && (uint64_t)range_max * A > (uint64_t)fl::GRAPH_SORT_CAP) rc = ctx->graph_sort.ensure((uint64_t)nb * sort_cap * 8);
    if (rc) { floria_hip_hap_graph_free(G); return rc; }
    char* B = ctx->graph_buf.as<char>();
    hipError_t e = hipMemcpyAsync(B + s_pred.off, G->pred, 4ull * nb, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(B + s_noff.off, G->node_off, 8ull * (nb + 1), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(B + s_eoff.off, G->edge_off, 8ull * (nb + 1), hipMemcpyHostToDevice, ctx->stream);
    fl::GraphArgs a{};
    a.bs = ctx->last_bs; a.best_ploidy = ctx->last_best; a.part = ctx->last_part;
    a.pred = (const int32_t*)(B + s_pred.off); a.node_off = (const uint64_t*)(B + s_noff.off); a.edge_off = (const uint64_t*)(B + s_eoff.off);
    a.node_cov = (double*)(B + s_cov.off); a.edge_w = (uint32_t*)(B + s_ew.off);
    a.hist_pool = ctx->graph_hist.as<uint64_t>(); a.sort_pool = ctx->graph_sort.as<uint64_t>(); a.sort_cap = sort_cap;
    a.range_max = range_max; a.hist_in_lds = in_lds ? 1 : 0; a.hist_stride = hist_bytes / 8;
    EventTimer T(ctx->stream);
    int tk = T.begin(K_SEL);
    if (e == hipSuccess) {
        if (A == 2) hipLaunchKernelGGL(fl::graph_kernel<2>, dim3(nb), dim3(fl::GRAPH_THREADS), in_lds ? hist_bytes : 0, ctx->stream, a);
        else hipLaunchKernelGGL(fl::graph_kernel<4>, dim3(nb), dim3(fl::GRAPH_THREADS), in_lds ? hist_bytes : 0, ctx->stream, a);
        e = hipGetLastError();
    }
    T.end(tk);
    if (e == hipSuccess && n_nodes) e = hipMemcpyAsync(G->node_cov, B + s_cov.off, 8ull * n_nodes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && n_edges) e = hipMemcpyAsync(G->edge_w, B + s_ew.off, 4ull * n_edges, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { floria_hip_hap_graph_free(G); return fail(FLORIA_E_DEVICE, std::string("hap_graph: ") + hipGetErrorString(e)); }
    ctx->timing.select_ms = T.sum(K_SEL);
    *out = G;
    return 0;
}
void floria_hip_hap_graph_free(floria_hap_graph* g) {
    if (!g) return;
    free(g->node_off); free(g->node_cov); free(g->pred); free(g->edge_off); free(g->edge_w); free(g);
}

// ---- haploset coverage / error statistics (first half) ---------------------------------------------
int floria_hip_haploset_stats(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs, uint32_t n_contigs,
                              const uint32_t* grp_contig, const uint64_t* grp_off, const uint32_t* grp_read,
                              const uint32_t* grp_range, uint32_t n_groups, double* out4) {
    if (!ctx || (n_contigs && !contigs) || (n_groups && (!grp_off || !grp_range || !out4))) return fail(FLORIA_E_INVALID, "null argument");
    if (n_groups == 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->batch_token = 0;
    uint32_t A = 2;
    std::vector<fl::ContigDev> cdev(n_contigs);
    for (uint32_t i = 0; i < n_contigs; ++i) {
        if (!contigs[i] || contigs[i]->ctx != ctx) return fail(FLORIA_E_INVALID, "bad contig handle");
        A = std::max(A, contigs[i]->n_alleles); cdev[i] = contigs[i]->dev;
    }
    std::vector<uint32_t> gc(n_groups, 0);
    std::vector<uint64_t> hoff(n_groups + 1, 0);
    for (uint32_t g = 0; g < n_groups; ++g) {
        gc[g] = grp_contig ? grp_contig[g] : 0;
        if (gc[g] >= n_contigs) return fail(FLORIA_E_INVALID, "grp_contig out of range");
        for (uint64_t i = grp_off[g]; i < grp_off[g + 1]; ++i) if (grp_read[i] >= contigs[gc[g]]->n_reads) return fail(FLORIA_E_INVALID, "group read id out of range");
        const uint32_t lo = grp_range[2 * g], hi = grp_range[2 * g + 1];
        hoff[g + 1] = hoff[g] + (hi >= lo ? (uint64_t)(hi - lo + 1) * A : 0);
    }
    const uint64_t n_reads_tot = grp_off[n_groups];
    struct Seg { size_t off, bytes; };
    size_t cursor = 0;
    auto seg = [&](size_t bytes) { Seg sg{cursor, bytes}; cursor += (bytes + 255) & ~(size_t)255; return sg; };
    const Seg s_cd = seg(sizeof(fl::ContigDev) * n_contigs), s_gc = seg(4ull * n_groups), s_go = seg(8ull * (n_groups + 1)), s_gr = seg(4ull * n_reads_tot + 4),
              s_rg = seg(8ull * n_groups), s_ho = seg(8ull * (n_groups + 1)), s_h = seg(4ull * hoff[n_groups] + 4), s_out = seg(32ull * n_groups);
    int rc = ctx->misc.ensure(cursor + 256); if (rc) return rc;
    char* M = ctx->misc.as<char>();
    auto h2d = [&](Seg sg, const void* src, size_t bytes) -> hipError_t { return bytes ? hipMemcpyAsync(M + sg.off, src, bytes, hipMemcpyHostToDevice, ctx->stream) : hipSuccess; };
    HIPCHK(h2d(s_cd, cdev.data(), sizeof(fl::ContigDev) * n_contigs)); HIPCHK(h2d(s_gc, gc.data(), 4ull * n_groups));
    HIPCHK(h2d(s_go, grp_off, 8ull * (n_groups + 1))); HIPCHK(h2d(s_gr, grp_read, 4ull * n_reads_tot));
    HIPCHK(h2d(s_rg, grp_range, 8ull * n_groups)); HIPCHK(h2d(s_ho, hoff.data(), 8ull * (n_groups + 1)));
    HIPCHK(hipMemsetAsync(M + s_h.off, 0, s_h.bytes, ctx->stream));
    fl::StatsArgs a{};
    a.contigs = (const fl::ContigDev*)(M + s_cd.off); a.grp_contig = (const uint32_t*)(M + s_gc.off); a.grp_off = (const uint64_t*)(M + s_go.off);
    a.grp_read = (const uint32_t*)(M + s_gr.off); a.grp_range = (const uint32_t*)(M + s_rg.off); a.hist_off = (const uint64_t*)(M + s_ho.off);
    a.hist = (uint32_t*)(M + s_h.off); a.out = (double*)(M + s_out.off); a.n_groups = n_groups;
    if (A == 2) hipLaunchKernelGGL(fl::stats_kernel<2>, dim3(n_groups), dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(fl::stats_kernel<4>, dim3(n_groups), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out4, M + s_out.off, 32ull * n_groups, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- get_hapq (part_block_manip.rs:517-616) for the haplosets of many contigs (the reference calls it once per contig) ----------
int floria_hip_hapq_batch(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs, uint32_t n_contigs, const uint32_t* grp_contig,
                          const uint64_t* grp_off, const uint32_t* grp_read, const uint32_t* grp_range, uint32_t n_groups,
                          const uint64_t* const* snp_to_genome_pos, const uint32_t* n_snps, uint64_t block_length,
                          uint8_t* hapq, double* rel_err, double* avg_err) {
    if (!ctx || (n_contigs && (!contigs || !avg_err)) || (n_groups && (!grp_off || !grp_range || !hapq || !rel_err))) return fail(FLORIA_E_INVALID, "null argument");
    if (block_length == 0) return fail(FLORIA_E_INVALID, "block_length must be positive");
    for (uint32_t c = 0; c < n_contigs; ++c) {
        if (!contigs[c] || contigs[c]->ctx != ctx) return fail(FLORIA_E_INVALID, "bad contig handle");
        avg_err[c] = std::numeric_limits<double>::quiet_NaN();                   // 0. / 0. for a contig without haplosets (:540)
    }
    if (n_groups == 0) return 0;
    // (1) get_errors_cov_from_frags per haploset (:529-539)
    std::vector<double> st(4ull * n_groups);
    int rc = floria_hip_haploset_stats(ctx, contigs, n_contigs, grp_contig, grp_off, grp_read, grp_range, n_groups, st.data());
    if (rc) return rc;
    std::vector<uint32_t> gc(n_groups, 0);
    std::vector<double> weight(n_contigs, 0.), error(n_contigs, 0.);
    for (uint32_t g = 0; g < n_groups; ++g) {                                    // (accumulated in group order inside each contig, as the reference does)
        gc[g] = grp_contig ? grp_contig[g] : 0;
        weight[gc[g]] += st[4ull * g + 3]; error[gc[g]] += st[4ull * g + 2];
    }
    for (uint32_t c = 0; c < n_contigs; ++c) avg_err[c] = error[c] / weight[c];
    // spans of the haplosets' reads (the consensus haplotype has a key wherever a read has a cell) and base ranges (:584-600)
    uint32_t A = 2;
    for (uint32_t c = 0; c < n_contigs; ++c) A = std::max(A, contigs[c]->n_alleles);
    std::vector<uint32_t> lo(n_groups, 0), len(n_groups, 0);
    std::vector<uint64_t> coff(n_groups + 1, 0), base_range(n_groups, 0);
    for (uint32_t g = 0; g < n_groups; ++g) {
        const floria_hip_contig* c = contigs[gc[g]];
        uint32_t r0 = 0xffffffffu, r1 = 0;
        for (uint64_t i = grp_off[g]; i < grp_off[g + 1]; ++i) {
            const uint32_t r = grp_read[i];                                 // (validated by floria_hip_haploset_stats)
            r0 = std::min(r0, c->h_first[r]); r1 = std::max(r1, c->h_last[r]);
        }
        if (!(r0 > r1)) {
            lo[g] = r0; len[g] = r1 - r0 + 1;
            const uint32_t x1 = grp_range[2 * g], x2 = grp_range[2 * g + 1];
            const uint64_t* pos = snp_to_genome_pos ? snp_to_genome_pos[gc[g]] : nullptr;
            const uint32_t ns = n_snps ? n_snps[gc[g]] : 0;
            if (!pos || x1 == 0 || x2 == 0 || x1 > ns || x2 > ns) return fail(FLORIA_E_INVALID, "haploset range outside snp_to_genome_pos");
            base_range[g] = pos[x2 - 1] - pos[x1 - 1];
        }
        coff[g + 1] = coff[g] + len[g];
    }
    // (2) find_overlapping_blocks (:453-513) inside each contig: rust-lapper's half-open overlap, overlap_percent (:13-24) > 0.05
    std::vector<std::vector<uint32_t>> by_contig(n_contigs);
    for (uint32_t g = 0; g < n_groups; ++g) by_contig[gc[g]].push_back(g);
    std::vector<uint32_t> pi, pj;
    std::vector<double> pol;
    std::vector<uint64_t> pair_off(n_groups + 1, 0);
    {
        std::vector<std::vector<uint32_t>> pj_of(n_groups);
        std::vector<std::vector<double>> ol_of(n_groups);
        for (uint32_t c = 0; c < n_contigs; ++c)
            for (uint32_t i : by_contig[c]) {
                const uint32_t x1 = grp_range[2 * i], x2 = grp_range[2 * i + 1];
                for (uint32_t j : by_contig[c]) {
                    if (j == i) continue;
                    const uint32_t y1 = grp_range[2 * j], y2 = grp_range[2 * j + 1];
                    if (!(y1 < x2 && y2 > x1)) continue;
                    const uint32_t a = x2 - y1 + 1, b = y2 - x1 + 1;
                    double ol = (double)std::min(a, b) / (double)(x2 - x1 + 1);
                    if (ol > 1.) ol = 1.;
                    if (!(ol > 0.05)) continue;
                    pj_of[i].push_back(j); ol_of[i].push_back(ol);
                }
            }
        for (uint32_t i = 0; i < n_groups; ++i) {
            for (size_t x = 0; x < pj_of[i].size(); ++x) { pi.push_back(i); pj.push_back(pj_of[i][x]); pol.push_back(ol_of[i][x]); }
            pair_off[i + 1] = pi.size();
        }
    }
    const uint32_t n_pairs = (uint32_t)pi.size();
    std::vector<uint32_t> sd(2ull * n_pairs + 2, 0);
    if (n_pairs) {
        HIPCHK(hipSetDevice(ctx->device));
        ctx->batch_token = 0;
        std::vector<fl::ContigDev> cdev(n_contigs);
        for (uint32_t c = 0; c < n_contigs; ++c) cdev[c] = contigs[c]->dev;
        const uint64_t n_reads_tot = grp_off[n_groups];
        struct Seg { size_t off, bytes; };
        size_t cursor = 0;
        auto seg = [&](size_t bytes) { Seg sg{cursor, bytes}; cursor += (bytes + 255) & ~(size_t)255; return sg; };
        const Seg s_cd = seg(sizeof(fl::ContigDev) * n_contigs), s_gc = seg(4ull * n_groups), s_go = seg(8ull * (n_groups + 1)), s_gr = seg(4ull * n_reads_tot + 4),
                  s_lo = seg(4ull * n_groups), s_len = seg(4ull * n_groups), s_co = seg(8ull * (n_groups + 1)), s_h = seg(8ull * coff[n_groups] * A + 8),
                  s_c = seg(coff[n_groups] + 8), s_pi = seg(4ull * n_pairs), s_pj = seg(4ull * n_pairs), s_sd = seg(8ull * n_pairs);
        rc = ctx->misc.ensure(cursor + 256); if (rc) return rc;
        char* M = ctx->misc.as<char>();
        auto h2d = [&](Seg sg, const void* src, size_t bytes) -> hipError_t { return bytes ? hipMemcpyAsync(M + sg.off, src, bytes, hipMemcpyHostToDevice, ctx->stream) : hipSuccess; };
        HIPCHK(h2d(s_cd, cdev.data(), sizeof(fl::ContigDev) * n_contigs)); HIPCHK(h2d(s_gc, gc.data(), 4ull * n_groups));
        HIPCHK(h2d(s_go, grp_off, 8ull * (n_groups + 1))); HIPCHK(h2d(s_gr, grp_read, 4ull * n_reads_tot));
        HIPCHK(h2d(s_lo, lo.data(), 4ull * n_groups)); HIPCHK(h2d(s_len, len.data(), 4ull * n_groups)); HIPCHK(h2d(s_co, coff.data(), 8ull * (n_groups + 1)));
        HIPCHK(h2d(s_pi, pi.data(), 4ull * n_pairs)); HIPCHK(h2d(s_pj, pj.data(), 4ull * n_pairs));
        HIPCHK(hipMemsetAsync(M + s_h.off, 0, s_h.bytes, ctx->stream));
        fl::ConsensusArgs ca{};
        ca.contigs = (const fl::ContigDev*)(M + s_cd.off); ca.grp_contig = (const uint32_t*)(M + s_gc.off);
        ca.grp_off = (const uint64_t*)(M + s_go.off); ca.grp_read = (const uint32_t*)(M + s_gr.off);
        ca.span_lo = (const uint32_t*)(M + s_lo.off); ca.span_len = (const uint32_t*)(M + s_len.off); ca.cons_off = (const uint64_t*)(M + s_co.off);
        ca.hist = (unsigned long long*)(M + s_h.off); ca.cons = (uint8_t*)(M + s_c.off); ca.n_groups = n_groups;
        if (A == 2) hipLaunchKernelGGL(fl::consensus_kernel<2>, dim3(n_groups), dim3(256), 0, ctx->stream, ca);
        else hipLaunchKernelGGL(fl::consensus_kernel<4>, dim3(n_groups), dim3(256), 0, ctx->stream, ca);
        HIPCHK(hipGetLastError());
        fl::PairArgs pa{};
        pa.pair_i = (const uint32_t*)(M + s_pi.off); pa.pair_j = (const uint32_t*)(M + s_pj.off); pa.span_lo = ca.span_lo; pa.span_len = ca.span_len;
        pa.cons_off = ca.cons_off; pa.cons = ca.cons; pa.same_diff = (uint32_t*)(M + s_sd.off); pa.n_pairs = n_pairs;
        hipLaunchKernelGGL(fl::pair_kernel, dim3(n_pairs), dim3(64), 0, ctx->stream, pa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(sd.data(), M + s_sd.off, 8ull * n_pairs, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    // (3) the scalar tail (:543-615), same operation order as the reference
    for (uint32_t i = 0; i < n_groups; ++i) {
        double max_penalty = 0.;
        for (uint64_t x = pair_off[i]; x < pair_off[i + 1]; ++x) {
            const double same = (double)sd[2 * x], diff = (double)sd[2 * x + 1];
            const double dist = (same + diff) == 0. ? 1. : diff / (same + diff);
            if (pol[x] * (1. - dist) > max_penalty) max_penalty = pol[x] * (1. - dist);
        }
        const uint64_t n_i = grp_off[i + 1] - grp_off[i];
        const double t1 = 40. * (1. - max_penalty);                                   // constants::HAPQ_CONSTANT
        const double t2 = std::min(1., (double)n_i / 3.);
        const double t3 = std::max(0.0, std::log(((double)base_range[i] / (double)block_length) + 1.));
        const double prod = t1 * t2 * t3;
        uint64_t hq = prod > 0. ? (prod >= 18446744073709551615. ? ~0ull : (uint64_t)prod) : 0;      // `as usize`: saturating, NaN -> 0
        if (n_i == 1) hq = 0;
        hapq[i] = (uint8_t)std::min<uint64_t>(hq, 60);
        rel_err[i] = st[4ull * i + 1] / avg_err[gc[i]];
    }
    return 0;
}

int floria_hip_hapq(floria_hip_ctx* ctx, const floria_hip_contig* contig, const uint64_t* grp_off, const uint32_t* grp_read,
                    const uint32_t* grp_range, uint32_t n_groups, const uint64_t* snp_to_genome_pos, uint32_t n_snps,
                    uint64_t block_length, uint8_t* hapq, double* rel_err, double* avg_err) {
    if (!contig || !avg_err) return fail(FLORIA_E_INVALID, "null argument");
    const floria_hip_contig* one[1] = {contig};
    const uint64_t* pos[1] = {snp_to_genome_pos};
    return floria_hip_hapq_batch(ctx, one, 1, nullptr, grp_off, grp_read, grp_range, n_groups, pos, &n_snps, block_length, hapq, rel_err, avg_err);
}

// ---- S2 --------------------------------------------------------------------------------------------------------
// process_reads_for_final_parts for MANY contigs in one launch (one wavefront per contig): the reference calls it once
// per contig from its serial contig loop (floria.rs:229,359-366); the chain is sequential inside a contig and independent
// across contigs.
int floria_hip_reassign_batch(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs, uint32_t n_contigs,
                              const uint32_t* grp_contig, const uint64_t* grp_off, const uint32_t* grp_read,
                              const uint32_t* grp_range, uint32_t n_groups, const uint32_t* read_order, const uint64_t* order_off,
                              double epsilon, floria_groups*** out) {
    if (!ctx || !out || (n_contigs && !contigs) || (n_groups && (!grp_off || !grp_range))) return fail(FLORIA_E_INVALID, "null argument");
    if ((read_order == nullptr) != (order_off == nullptr)) return fail(FLORIA_E_INVALID, "read_order and order_off go together");
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    ctx->timing = floria_timing{};
    ctx->batch_token = 0;                      // misc is about to be overwritten
    uint32_t A = 2;
    for (uint32_t i = 0; i < n_contigs; ++i) {
        if (!contigs[i] || contigs[i]->ctx != ctx) return fail(FLORIA_E_INVALID, "bad contig handle");
        A = std::max(A, contigs[i]->n_alleles);
    }
    // groups of each contig, in input order (contig-local group id = rank among the contig's groups)
    std::vector<std::vector<uint32_t>> cg(n_contigs);
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t ci = grp_contig ? grp_contig[g] : 0;
        if (ci >= n_contigs) return fail(FLORIA_E_INVALID, "grp_contig out of range");
        cg[ci].push_back(g);
    }
    std::vector<uint64_t> r2g_off_base(n_contigs), r2g_base(n_contigs), grp_base(n_contigs), assign_base(n_contigs);
    std::vector<uint64_t> r2g_off_all, hist_off_all;
    std::vector<uint32_t> r2g_all, gpos0_all;
    std::vector<fl::ContigDev> cdev(n_contigs);
    uint64_t hist_cells = 0, n_assign = 0;
    for (uint32_t ci = 0; ci < n_contigs; ++ci) {
        const floria_hip_contig* c = contigs[ci];
        cdev[ci] = c->dev;
        const uint32_t N = c->n_reads;
        // read -> groups (part_block_manip.rs:185-193) by counting sort: groups are visited in ascending local id, so every read's candidate
        // list comes out ascending; groups are sets (a repeated id inside one group counts once)
        std::vector<uint32_t> p0(cg[ci].size(), UINT32_MAX), p1(cg[ci].size(), 0);
        std::vector<uint64_t> off(N + 1, 0);
        std::vector<uint32_t> last(N, UINT32_MAX);
        for (uint32_t lg = 0; lg < cg[ci].size(); ++lg) {
            const uint32_t g = cg[ci][lg];
            for (uint64_t i = grp_off[g]; i < grp_off[g + 1]; ++i) {
                const uint32_t r = grp_read[i];
                if (r >= N) return fail(FLORIA_E_INVALID, "group read id out of range");
                if (last[r] == lg) continue;
                last[r] = lg;
                off[r + 1]++;
                p0[lg] = std::min(p0[lg], c->h_first[r]); p1[lg] = std::max(p1[lg], c->h_last[r]);
            }
        }
        for (uint32_t r = 0; r < N; ++r) off[r + 1] += off[r];
        r2g_off_base[ci] = r2g_off_all.size(); r2g_base[ci] = r2g_all.size(); grp_base[ci] = gpos0_all.size(); assign_base[ci] = n_assign;
        {
            const size_t base = r2g_all.size();
            r2g_all.resize(base + off[N]);
            std::vector<uint64_t> fill(off.begin(), off.end() - 1);
            std::fill(last.begin(), last.end(), UINT32_MAX);
            for (uint32_t lg = 0; lg < cg[ci].size(); ++lg) {
                const uint32_t g = cg[ci][lg];
                for (uint64_t i = grp_off[g]; i < grp_off[g + 1]; ++i) {
                    const uint32_t r = grp_read[i];
                    if (last[r] == lg) continue;
                    last[r] = lg;
                    r2g_all[base + fill[r]++] = lg;
                }
            }
        }
        if (read_order) {                                   // every read that sits in a group must be visited exactly once
            std::vector<uint8_t> seen(N, 0);
            for (uint64_t i = order_off[ci]; i < order_off[ci + 1]; ++i) {
                const uint32_t r = read_order[i];
                if (r >= N || seen[r]) return fail(FLORIA_E_INVALID, "read_order: id out of range or repeated");
                seen[r] = 1;
            }
            for (uint32_t r = 0; r < N; ++r) if (off[r + 1] > off[r] && !seen[r]) return fail(FLORIA_E_INVALID, "read_order misses a read that sits in a group");
        }
        r2g_off_all.insert(r2g_off_all.end(), off.begin(), off.end());
        for (uint32_t lg = 0; lg < cg[ci].size(); ++lg) {
            hist_off_all.push_back(hist_cells);
            gpos0_all.push_back(p0[lg] == UINT32_MAX ? 0 : p0[lg]);
            hist_cells += p0[lg] == UINT32_MAX ? 0 : (uint64_t)(p1[lg] - p0[lg] + 1) * A;
        }
        n_assign += N;
    }
    std::vector<int32_t> assign(n_assign, -1);
    if (n_assign && n_groups) {
        struct Seg { size_t off, bytes; };
        size_t cursor = 0;
        auto seg = [&](size_t bytes) { Seg s{cursor, bytes}; cursor += (bytes + 255) & ~(size_t)255; return s; };
        const Seg s_cd = seg(sizeof(fl::ContigDev) * n_contigs), s_rob = seg(8ull * n_contigs), s_rb = seg(8ull * n_contigs), s_gb = seg(8ull * n_contigs),
                  s_ab = seg(8ull * n_contigs), s_ro = seg(8ull * r2g_off_all.size() + 8), s_r2g = seg(4ull * r2g_all.size() + 4),
                  s_ho = seg(8ull * hist_off_all.size() + 8), s_p0 = seg(4ull * gpos0_all.size() + 4), s_hist = seg(8ull * hist_cells + 8),
                  s_as = seg(4ull * n_assign + 4), s_q = seg(16), s_ord = seg(read_order ? 4ull * order_off[n_contigs] + 4 : 4),
                  s_oo = seg(8ull * (n_contigs + 1));
        int rc = ctx->misc.ensure(cursor + 256); if (rc) return rc;
        char* M = ctx->misc.as<char>();
        EventTimer T(ctx->stream);
        int th = T.begin(K_H2D);
        auto h2d = [&](Seg sg, const void* src, size_t bytes) -> hipError_t { return bytes ? hipMemcpyAsync(M + sg.off, src, bytes, hipMemcpyHostToDevice, ctx->stream) : hipSuccess; };
        HIPCHK(h2d(s_cd, cdev.data(), sizeof(fl::ContigDev) * n_contigs));
        HIPCHK(h2d(s_rob, r2g_off_base.data(), 8ull * n_contigs)); HIPCHK(h2d(s_rb, r2g_base.data(), 8ull * n_contigs));
        HIPCHK(h2d(s_gb, grp_base.data(), 8ull * n_contigs)); HIPCHK(h2d(s_ab, assign_base.data(), 8ull * n_contigs));
        HIPCHK(h2d(s_ro, r2g_off_all.data(), 8ull * r2g_off_all.size())); HIPCHK(h2d(s_r2g, r2g_all.data(), 4ull * r2g_all.size()));
        HIPCHK(h2d(s_ho, hist_off_all.data(), 8ull * hist_off_all.size())); HIPCHK(h2d(s_p0, gpos0_all.data(), 4ull * gpos0_all.size()));
        HIPCHK(hipMemsetAsync(M + s_hist.off, 0, s_hist.bytes, ctx->stream));
        HIPCHK(hipMemsetAsync(M + s_q.off, 0, 16, ctx->stream));
        HIPCHK(hipMemsetAsync(M + s_as.off, 0xff, s_as.bytes, ctx->stream));            // -1 = not assigned
        if (read_order) { HIPCHK(h2d(s_ord, read_order, 4ull * order_off[n_contigs])); HIPCHK(h2d(s_oo, order_off, 8ull * (n_contigs + 1))); }
        T.end(th);
        fl::ReassignArgs a{};
        a.contigs = (const fl::ContigDev*)(M + s_cd.off); a.n_contigs = n_contigs;
        a.r2g_off_base = (const uint64_t*)(M + s_rob.off); a.r2g_off = (const uint64_t*)(M + s_ro.off);
        a.r2g_base = (const uint64_t*)(M + s_rb.off); a.r2g = (const uint32_t*)(M + s_r2g.off);
        a.grp_base = (const uint64_t*)(M + s_gb.off); a.grp_hist_off = (const uint64_t*)(M + s_ho.off); a.grp_pos0 = (const uint32_t*)(M + s_p0.off);
        a.hist = (uint64_t*)(M + s_hist.off); a.assign = (int32_t*)(M + s_as.off); a.assign_base = (const uint64_t*)(M + s_ab.off);
        a.eps = epsilon; a.queue_head = (uint32_t*)(M + s_q.off);
        a.order = read_order ? (const uint32_t*)(M + s_ord.off) : nullptr; a.order_off = (const uint64_t*)(M + s_oo.off);
        const uint32_t grid = std::min<uint32_t>(n_contigs, (uint32_t)ctx->n_cu * 16);
        int tk = T.begin(K_REASSIGN);
        if (A == 2) hipLaunchKernelGGL(fl::reassign_kernel<2>, dim3(grid), dim3(64), 0, ctx->stream, a);
        else hipLaunchKernelGGL(fl::reassign_kernel<4>, dim3(grid), dim3(64), 0, ctx->stream, a);
        T.end(tk);
        HIPCHK(hipGetLastError());
        int td = T.begin(K_D2H);
        HIPCHK(hipMemcpyAsync(assign.data(), M + s_as.off, 4ull * n_assign, hipMemcpyDeviceToHost, ctx->stream));
        T.end(td);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        ctx->timing.reassign_ms = T.sum(K_REASSIGN); ctx->timing.h2d_ms = T.sum(K_H2D); ctx->timing.d2h_ms = T.sum(K_D2H); ctx->timing.total_ms = T.span();
    }
    // ---- host bookkeeping per contig: rebuild groups, separate_broken_haplogroups (:27-98), sort_parts (:276-288) ----
    floria_groups** arr = (floria_groups**)calloc(std::max(1u, n_contigs), sizeof(floria_groups*));
    if (!arr) return fail(FLORIA_E_NOMEM, "calloc");
    for (uint32_t ci = 0; ci < n_contigs; ++ci) {
        const floria_hip_contig* c = contigs[ci];
        const uint32_t N = c->n_reads, ng = (uint32_t)cg[ci].size();
        std::vector<std::vector<uint32_t>> parts(ng);
        std::vector<std::pair<uint32_t, uint32_t>> ranges(ng);
        for (uint32_t lg = 0; lg < ng; ++lg) ranges[lg] = {grp_range[2 * cg[ci][lg]], grp_range[2 * cg[ci][lg] + 1]};
        const int32_t* as = assign.data() + assign_base[ci];
        for (uint32_t r = 0; r < N; ++r) if (as[r] >= 0) parts[as[r]].push_back(r);       // ascending id
        {
            const auto& F = c->h_first; const auto& L = c->h_last;
            std::vector<std::pair<size_t, std::vector<uint32_t>>> all_breaks;
            const size_t n0 = ranges.size();
            for (size_t i = 0; i < n0; ++i) {
                uint32_t latest = 0;
                std::vector<uint32_t> breaks;
                for (uint32_t r : parts[i]) {                                // sorted by first_position (ids ascend)
                    if (latest != 0 && F[r] > latest && latest >= ranges[i].first && latest < ranges[i].second) breaks.push_back(latest);
                    if (L[r] > latest) latest = L[r];
                }
                if (!breaks.empty()) all_breaks.push_back({i, std::move(breaks)});
            }
            std::vector<std::vector<uint32_t>> new_parts;
            std::vector<std::pair<uint32_t, uint32_t>> new_ranges;
            for (auto& bi : all_breaks) {
                size_t spot = 0;
                uint32_t break_start = ranges[bi.first].first, end_spot = bi.second[0];
                std::vector<uint32_t> np;
                for (uint32_t r : parts[bi.first]) {
                    if (L[r] <= end_spot) np.push_back(r);
                    else {                                                   // :69-84: this read is not re-inserted
                        new_parts.push_back(std::move(np)); np.clear();
                        new_ranges.push_back({break_start, end_spot});
                        break_start = end_spot + 1;
                        ++spot;
                        end_spot = spot != bi.second.size() ? bi.second[spot] : UINT32_MAX;
                    }
                }
                new_parts.push_back(std::move(np));
                new_ranges.push_back({break_start, ranges[bi.first].second});
            }
            for (auto& bi : all_breaks) parts[bi.first].clear();
            for (size_t i = 0; i < new_parts.size(); ++i) { parts.push_back(std::move(new_parts[i])); ranges.push_back(new_ranges[i]); }
        }
        std::vector<size_t> idx(parts.size());
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return ranges[x] < ranges[y]; });
        floria_groups* G = (floria_groups*)calloc(1, sizeof(floria_groups));
        if (!G) { floria_hip_groups_array_free(arr, n_contigs); return fail(FLORIA_E_NOMEM, "calloc"); }
        arr[ci] = G;
        G->n_groups = (uint32_t)parts.size();
        G->grp_off = (uint64_t*)calloc(parts.size() + 1, 8);
        G->range = (uint32_t*)calloc(2 * parts.size() + 2, 4);
        uint64_t tot = 0;
        for (size_t k = 0; k < idx.size(); ++k) { G->grp_off[k] = tot; tot += parts[idx[k]].size(); G->range[2 * k] = ranges[idx[k]].first; G->range[2 * k + 1] = ranges[idx[k]].second; }
        G->grp_off[idx.size()] = tot;
        G->grp_read = (uint32_t*)malloc(4 * (tot + 1));
        for (size_t k = 0; k < idx.size(); ++k) std::copy(parts[idx[k]].begin(), parts[idx[k]].end(), G->grp_read + G->grp_off[k]);
    }
    *out = arr;
    return 0;
}

int floria_hip_reassign_ordered(floria_hip_ctx* ctx, const floria_hip_contig* c, const uint64_t* grp_off, const uint32_t* grp_read,
                                const uint32_t* grp_range, uint32_t n_groups, const uint32_t* read_order, uint32_t n_order,
                                double epsilon, floria_groups** out) {
    if (!c || !out) return fail(FLORIA_E_INVALID, "null argument");
    *out = nullptr;
    const floria_hip_contig* arr[1] = {c};
    const uint64_t oo[2] = {0, n_order};
    floria_groups** res = nullptr;
    int rc = floria_hip_reassign_batch(ctx, arr, 1, nullptr, grp_off, grp_read, grp_range, n_groups, read_order, read_order ? oo : nullptr, epsilon, &res);
    if (rc) return rc;
    *out = res[0];
    free(res);
    return 0;
}
int floria_hip_reassign(floria_hip_ctx* ctx, const floria_hip_contig* c, const uint64_t* grp_off, const uint32_t* grp_read,
                        const uint32_t* grp_range, uint32_t n_groups, double epsilon, floria_groups** out) {
    return floria_hip_reassign_ordered(ctx, c, grp_off, grp_read, grp_range, n_groups, nullptr, 0, epsilon, out);
}
void floria_hip_groups_array_free(floria_groups** arr, uint32_t n) {
    if (!arr) return;
    for (uint32_t i = 0; i < n; ++i) floria_hip_groups_free(arr[i]);
    free(arr);
}
void floria_hip_groups_free(floria_groups* g) { if (g) { free(g->grp_off); free(g->grp_read); free(g->range); free(g); } }

}  // extern "C"
