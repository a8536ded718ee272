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
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/floria_hip.h"
#include "beam_kernel.h"
#include "wave_util.h"
#include "beam_slab_kernel.h"
#include "beam_wide_kernel.h"
#include "optimize_kernel.h"
#include "reassign_kernel.h"
#include "hapq_kernel.h"
#include "blocks_kernel.h"
#include "graph_kernel.h"
#include "stats_kernel.h"
#include "realign_kernel.h"
#include "upload_kernel.h"

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

// Pinned, device-mapped host memory for the small latency-critical transfers of a call (block counts down, offsets and job lists up): kernels read and
// write it directly.  A hipMemcpyAsync of a few KB shares the SDMA queues with the pileup's bulk upload and was seen waiting behind all of it
// (block counts on the host after 14.7 ms instead of 4.3 ms in two calls out of three).
struct HostBox {
    char* h = nullptr;       // host address
    char* d = nullptr;       // the same bytes as the device sees them
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        release();
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc((void**)&h, want, hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&d, h, 0);
        if (e != hipSuccess) { if (h) (void)hipHostFree(h); h = nullptr; d = nullptr; return fail(FLORIA_E_NOMEM, std::string("hipHostMalloc(mapped, ") + std::to_string(want) + "): " + hipGetErrorString(e)); }
        cap = want;
        return 0;
    }
    void release() { if (h) (void)hipHostFree(h); h = nullptr; d = nullptr; cap = 0; }
};

// Pageable sources go through a ring of pinned staging buffers filled by a few host threads (a single pageable hipMemcpy is
// staged by the runtime on one thread at a few GB/s); pinned sources are handed to the DMA engine as they are.
struct StagePool {
    static constexpr size_t SEG = 16u << 20;      // large pieces: every copy in a stream pays ~0.2 ms of completion-signal latency
    static constexpr uint32_t NBUF = 8;
    char* buf[NBUF] = {};
    hipEvent_t ev[NBUF] = {};
    bool used[NBUF] = {};
    int init() {
        for (uint32_t i = 0; i < NBUF; ++i) if (!buf[i]) {
            if (hipHostMalloc((void**)&buf[i], SEG, hipHostMallocDefault) != hipSuccess) return fail(FLORIA_E_NOMEM, "hipHostMalloc(staging) failed");
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return fail(FLORIA_E_DEVICE, "hipEventCreate failed");
        }
        return 0;
    }
    void release() { for (uint32_t i = 0; i < NBUF; ++i) { if (buf[i]) (void)hipHostFree(buf[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); buf[i] = nullptr; ev[i] = nullptr; used[i] = false; } }
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

// Tuning / test knobs of a context (floria_hip_set_option); the defaults are what bench.py measures.  Read from the environment
// ONCE, at floria_hip_create (FLORIA_HIP_GROUPS, FLORIA_HIP_BEAM=generic|slab|wide, FLORIA_HIP_NO_SPECIALIZED, FLORIA_HIP_NO_P1_SHORTCUT,
// FLORIA_HIP_OPT_THREADS, FLORIA_HIP_OPT_GLOBAL, FLORIA_HIP_SPECULATE); none of them changes results.
struct Knobs {
    uint32_t groups = 0;          // job groups (0 = auto: 2 when the batch has >= 48 x CUs non-empty blocks)
    uint32_t beam_path = 0;       // 0 auto | 1 generic | 2 slab | 3 wide
    uint32_t no_specialized = 0;  // runtime ploidy / beam width instead of the template instances
    uint32_t no_p1_shortcut = 0;  // run the beam kernel for ploidy 1 too
    uint32_t opt_threads = 0;     // 0 auto | 128 | 512 | 1024
    uint32_t opt_global = 0;      // optimise histogram in HBM
    int32_t  speculate = -1;      // ploidy stages: -1 auto | 0 one ploidy per stage | 1 all ploidies at once | 2 {1,2,3} then {4..P}
    uint32_t upload_chunks = 0;   // floria_hip_phase_pileups_batch: chunks the cell arrays travel in (0 = auto by size, <= 8)
    uint32_t trace = 0;           // print host-side timestamps of the S1 call to stderr
    uint32_t spec_gate_div = 2;   // speculative stages: grid of the gated (ploidy >= 4) beam launches = slots / this
    uint32_t reassign_path = 0;   // S2 kernel: 0 auto | 1 workgroup-parallel | 2 one-wavefront chain
    uint32_t no_bulk = 0;         // (tests) beam_slab_kernel: every step through the general insert path (entry table, duplicate test, evictions)
    uint32_t fx_tags = 0;         // (A/B) reference arithmetic: cap on the words of a position map's claim table (0 = as many as cost no workgroup per CU)
    uint32_t arith_ow6 = 0;       // (A/B) ... the optimise kernel compiled for six waves per SIMD at every ploidy
    uint32_t arith_replay = 0;    // (tests) reference arithmetic: replay every position map insertion by insertion (the home-bucket rule of optimize_kernel.h off)
    uint32_t arith_hbm = 0;       // (tests) reference arithmetic: position-map tables and first-insertion keys in HBM scratch even where they fit into LDS
    uint32_t opt_block_order = 0; // (A/B, tests) optimise: the build / distance passes visit the reads in block order instead of longest first
    uint32_t s2_assign_only = 0;  // S2 returns the haplogroups as re-inserted (input order): separate_broken_haplogroups and sort_parts are left to a host that iterates its own sets
    uint32_t arith = 0;           // 1 = the reference's own running f64 sums in its own orders (arith_kernel.h; slower kernels), 0 = the canonical (Q24, #eps) form
    int32_t  tail_overlap = 0;    // one ploidy per stage: the LAST ploidy's beam launch runs beside the optimise launch of the ploidy below, every job waiting for its block's
                                  // stop rule (run_phase): 0 off (default: measured level) | 1 on
    uint32_t tail_waves = 2;      // ... with this many waves per CU
};

struct Arena;

struct floria_hip_ctx {
    int device = 0;
    uint32_t hw_queues = 0;                   // streams that really run side by side on this device (probe_hw_queues at create)
    hipStream_t stream = nullptr;
    uint32_t user_slots = 0;
    Knobs knobs;
    int n_cu = 256;
    // launch lanes (job group x position inside a ploidy stage): every lane has its own stream and scratch slice, see run_phase
    static constexpr uint32_t MAX_GROUPS = 8;
    static constexpr uint32_t MAX_LANES = 32;
    hipStream_t gstream[MAX_LANES] = {};
    hipStream_t gstream_low[MAX_LANES] = {};      // speculative stages: the lanes of ploidy >= 4 (dispatched after the ploidies every block needs)
    hipEvent_t ev_fork[MAX_LANES] = {}, ev_join[MAX_LANES] = {};
    hipEvent_t ev_gate[MAX_GROUPS] = {};       // speculative stages: the beam search of ploidy 2 of group g has finished (ploidies >= 4 start behind it)
    hipStream_t tstream[MAX_GROUPS] = {};      // the last ploidy's beam launch of group g when it runs beside the optimise launch below it (Knobs::tail_overlap)
    hipEvent_t ev_tail[MAX_GROUPS] = {}, ev_tail_join[MAX_GROUPS] = {};
    hipStream_t copy_stream = nullptr;        // read-id lists go back to the host while the launch loop runs
    hipEvent_t ev_rids = nullptr;
    hipEvent_t ev_chunk[MAX_GROUPS + 1] = {};  // floria_hip_phase_pileups_batch: chunk g of the cell arrays has landed and is flattened
    hipEvent_t ev_copied[MAX_GROUPS + 1] = {}; //   ... has landed (the flatten launches run on their own stream, so the DMA queue never waits for a kernel)
    hipStream_t flat_stream = nullptr;
    HostBox box;                              // s1_core's small transfers
    // cached tables
    double binom_eps = -1.0;
    uint32_t binom_nmax = 0;
    DevBuf d_binom;
    DevBuf d_hash;            // Rq1 | Rp1 | Rq2 | Rp2, each hash_len u64
    uint32_t hash_len = 0;
    uint32_t w24[256];
    DevBuf d_w24;             // the same table on the device (flatten_kernel)
    uint64_t Rk1[FLORIA_MAX_PLOIDY], Rk2[FLORIA_MAX_PLOIDY];
    // scratch pools
    DevBuf state_pool, hist_pool, opt_hist, opt_dist, opt_gain, opt_key, opt_moves, misc, misc0;
    uint32_t cur_len_max = 0;                 // longest read (cells) of the S1 call in flight
    uint64_t upload_epoch = 0, ord_epoch = ~0ull, ord_sig = 0;      // the cell orders in arith_ord belong to this set of contigs (signature) as uploaded then (epoch): S2 after S1, or the next step over a resident batch, reuses them
    const uint2* cur_ord = nullptr; const uint64_t* cur_ord_off = nullptr;      // ... of the S1 call in flight
    DevBuf arith_ord, arith_scr, arith_tab, arith_pool;      // reference-arithmetic mode: cell orders of the call's contigs, the order kernel's tables, prefix arrays, optimise scratch
    floria_timing timing{};
    // device-resident copy of the last S1 batch (floria_hip_hap_graph)
    uint64_t batch_token = 0, token_counter = 0;
    fl::BlockSet last_bs{};
    const uint8_t* last_part = nullptr;
    const uint32_t* last_best = nullptr;
    uint32_t last_nall = 2;
    std::vector<uint32_t> last_bc, last_start, last_end;
    DevBuf graph_buf, graph_hist, graph_sort;
    // uploads
    DevBuf up_tmp;                            // raw allele / qual bytes + the flatten kernel's tables (transient per upload)
    std::vector<Arena*> arena_cache;          // released batch arenas, reused by the next upload
    StagePool stage;                          // pinned staging ring for pageable sources
    uint32_t stage_threads = 8;
};

// A batch of contigs uploaded together shares ONE device allocation (an arena): raw CSR regions filled by DMA, flattened
// regions written by flatten_kernel.  The arena is released when its last contig handle is freed (and then kept in a small
// per-context cache, so a host that uploads batch after batch never reallocates).
struct Arena {
    floria_hip_ctx* ctx = nullptr;
    DevBuf buf;
    uint32_t n_contigs = 0, refs = 0;
    uint64_t R = 0, C = 0;                                   // reads / cells of the batch
    size_t off_ro = 0, off_first = 0, off_last = 0;          // region offsets (bytes)
    std::vector<uint64_t> read_prefix;                       // [n_contigs+1]
    // host copies of first/last (haplogroup bookkeeping of S2 / get_hapq), fetched from the device on first use
    bool host_meta = false;
    std::vector<uint32_t> h_first, h_last;
};

struct floria_hip_contig {
    floria_hip_ctx* ctx = nullptr;
    Arena* arena = nullptr;
    uint32_t idx = 0;           // position inside the arena's batch
    uint32_t n_reads = 0;
    uint64_t n_cells = 0;
    uint32_t max_len = 0;       // max cells per read
    uint32_t n_alleles = 2;     // 2 or 4 (kernel template)
    bool has_q0 = false;        // some cell has qual 0 (weight 0): presence != (weight sum > 0)
    const uint32_t *h_first = nullptr, *h_last = nullptr;   // valid after host_meta(c)
    fl::ContigDev dev{};
};

namespace {

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
    // wall time during which at least one launch of kind k was in flight (the launches of the job groups overlap on separate streams: their durations add up to
    // more than the wall clock, this does not)
    double union_ms(int k) {
        std::vector<std::pair<float, float>> iv;
        for (size_t i = 0; i < ev.size(); ++i) if (kind[i] == k) {
            float a = 0, b = 0;
            if (hipEventElapsedTime(&a, ev.front().first, ev[i].first) == hipSuccess && hipEventElapsedTime(&b, ev.front().first, ev[i].second) == hipSuccess) iv.push_back({a, b});
        }
        std::sort(iv.begin(), iv.end());
        double t = 0; float lo = 0, hi = -1;
        for (auto& x : iv) { if (hi < lo || x.first > hi) { if (hi >= lo) t += hi - lo; lo = x.first; hi = x.second; } else if (x.second > hi) hi = x.second; }
        if (hi >= lo) t += hi - lo;
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
    for (uint32_t g = 0; g < floria_hip_ctx::MAX_LANES; ++g) if (ctx->gstream[g]) (void)hipStreamSynchronize(ctx->gstream[g]);
    for (uint32_t g = 0; g < floria_hip_ctx::MAX_LANES; ++g) if (ctx->gstream_low[g]) (void)hipStreamSynchronize(ctx->gstream_low[g]);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->flat_stream) (void)hipStreamSynchronize(ctx->flat_stream);
    (void)hipStreamSynchronize(ctx->stream);
}

// ---- S1 launch plan ---------------------------------------------------------------------------------------------------------
// The per-ploidy loop of get_local_hap_blocks (graph_processing.rs:132-252) runs as STAGES: a stage is a set of consecutive
// ploidies whose (beam -> optimise) chains run concurrently, each on its own stream with its own scratch slice ("lane"), followed
// by the stop rule (select_kernel) for the stage's ploidies in ascending order.  The default is one ploidy per stage (no
// speculative work: the launches of ploidy p+1 skip the blocks whose stop rule fired at p).  A ploidy run never reads an earlier
// ploidy's result (only the stop rule does, :198-251), so running several ploidies of a block at once gives identical results;
// it trades extra (block, ploidy) jobs for a shorter dependent chain and is used when the batch cannot fill the chip.
// JOB GROUPS: the non-empty blocks are dealt into G groups whose stage sequences run independently on separate streams, so one
// group's launch tail (persistent waves draining their last jobs) is filled by the other group's kernels.
// A lane (group g, position j inside the stage) owns the SAME byte slice of every scratch pool for the whole call — sized for
// the largest ploidy — so kernels of different groups/ploidies that are live at the same time can never alias.
// threshold of the MEC-ratio stop rule for ploidy p (graph_processing.rs:204-220), libm pow as in the reference
double mec_threshold(const floria_params* prm, uint32_t p) {
    const double eps = prm->epsilon, pl = (double)p;
    if (prm->ploidy_sensitivity == 1) return 1.0 / (1.0 - eps) / (1.0 + 1.0 / (std::pow(pl, 0.50) + 1.00));
    if (prm->ploidy_sensitivity == 2) return 1.0 / (1.0 - eps) / (1.0 + 1.0 / (std::pow(pl, 1.00) + 1. / 3.));
    return 1.0 / (1.0 - eps) / (1.0 + 1.0 / (std::pow(pl, 1.00) + 1.00));
}

struct PloidyPlan {
    uint32_t p = 0, LM = 0;
    bool shortcut = false, wide = false, slab = false, beam_spec = false;
    uint64_t state_bytes = 0, hist_stride = 0;
    fl::SlabLds SL{}; fl::WideLds WL{}; fl::BeamLds LY{};
    uint32_t beam_slots = 0;
    uint64_t cand_cap = 1;
    uint32_t threads = 128, opt_slots = 0;
    size_t opt_lds = 0;
    bool hl = false, opt_spec = false;
    uint32_t pm_lds_off = 0;      // optimise: where the visiting order of the reads (u16, longest first) sits in the workgroup's LDS (0 = none: the reads in block order)
    uint32_t fk_lds_off = 0;      // ... and the first-insertion keys as 32-bit words
    uint32_t fx_tags = 0;         // ... and the words of a map's claim table
    uint32_t fx_lds_off = 0;      // reference-arithmetic mode: where the emulated position maps sit in the workgroup's LDS (0 = in HBM scratch)
};

template <int A>
int run_phase(floria_hip_ctx* ctx, bool any_q0, const fl::BlockSet& bs, const std::vector<uint32_t>& jobs, const std::vector<uint32_t>& group_off,
              const uint32_t* d_jobs, uint64_t tot_reads, uint32_t n_max, uint32_t span_max, const floria_params* prm, uint8_t* d_planes,
              uint8_t* d_beam_part, const std::vector<std::vector<uint32_t>>& stages, hipEvent_t* chunk_ev, double* d_mec, double* d_na, uint32_t* d_iters, uint8_t* d_done, uint32_t* d_best,
              uint32_t* d_tried, uint32_t* d_queue, double* d_margin, uint32_t* d_diag,
              unsigned long long* d_steps, EventTimer& T, bool& p1_shortcut, uint32_t* d_stop, uint32_t* d_ready) {
    const uint32_t P = prm->max_ploidy, B = prm->beam;
    const Knobs& K = ctx->knobs;
    p1_shortcut = false;
    const uint32_t n_jobs = (uint32_t)jobs.size();
    const uint32_t G = (uint32_t)group_off.size() - 1;
    if (n_jobs == 0) return 0;
    uint32_t W = 1;                                        // widest stage
    for (auto& st : stages) W = std::max<uint32_t>(W, (uint32_t)st.size());
    const uint32_t n_lanes = G * W;
    if (n_lanes > floria_hip_ctx::MAX_LANES) return fail(FLORIA_E_INVALID, "internal: too many launch lanes");
    uint32_t nj_max = 0;
    for (uint32_t g = 0; g < G; ++g) nj_max = std::max(nj_max, group_off[g + 1] - group_off[g]);
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const double cutoff = std::log(PROB_CUTOFF);      // graph_processing.rs:146

    // ---- plan every ploidy once: kernel choice, grid, scratch per slot -------------------------------------------------------
    std::vector<PloidyPlan> plan(P + 1);
    const uint64_t fx_ctrl = fl::fx_ctrl_bytes(span_max + 1), fx_slot = fl::fx_slot_bytes(span_max + 1), fx_bytes = fx_ctrl + fx_slot;       // (+1: slack)
    const uint32_t mean_n = (uint32_t)(tot_reads / std::max<uint32_t>(1, n_jobs));
    for (uint32_t p = 1; p <= P; ++p) {
        PloidyPlan& q = plan[p];
        q.p = p; q.LM = p * B;
        const uint32_t LM = q.LM;
        if (LM > 65000) return fail(FLORIA_E_UNSUPPORTED, "ploidy*beam too large");
        // ploidy 1 has nothing to search: one state, one partition, every child passes (p_k - lse == 0 > ln 0.01), the
        // partition is "all reads in haplotype 0" (global_clustering.rs:74-134 with ploidy == 1) -> a memset.
        q.shortcut = p == 1 && !K.no_p1_shortcut;
        q.state_bytes = (uint64_t)LM * span_max * p * A * 8;
        // traceback records + the slab kernels' per-slab tables that live in HBM (window-exit hash terms; everything for wide beams)
        q.hist_stride = (((uint64_t)fl::beam_hist_off(n_max, LM, B) + LM + 64 + 1) & ~1ull) + fl::SLAB_DUMMY_WORDS + std::max<uint64_t>(4ull * LM * p, fl::wide_scratch_words(LM, p, any_q0, K.arith != 0));   // + the dummy words of beam_slab_kernel's branch-free tails
        const bool arith_slab = K.arith && K.beam_path != 1;      // the reference's running sums on the shared slabs (beam_slab_kernel<.., ARITH = true>)
        q.SL = fl::slab_lds_layout(LM, p, any_q0, arith_slab);
        q.WL = fl::wide_lds_layout(LM, p, any_q0);
        q.LY = fl::beam_lds_layout(LM);
        q.beam_spec = A == 2 && !any_q0 && B == 10 && p >= 2 && p <= 5 && !K.no_specialized;
        const uint32_t by_lds = std::max<uint32_t>(1, (uint32_t)((158 * 1024) / (q.SL.total + 256)));
        const uint32_t waves_per_simd = (uint32_t)fl::slab_waves(q.beam_spec ? (int)p : 0);
        q.beam_slots = ctx->user_slots ? ctx->user_slots : (uint32_t)ctx->n_cu * std::min<uint32_t>(4 * waves_per_simd, by_lds);
        q.beam_slots = std::min(q.beam_slots, nj_max);
        const uint64_t slab_code_bytes = (uint64_t)LM * p * ((span_max + 15u) & ~15u);              // beam_slab_kernel: one code byte per (slab, position)
        const bool fits32 = q.state_bytes + slab_code_bytes < 0xf0000000ull;
        // shared-slab kernels: register heap for ploidy*beam <= 63 (the CLI defaults give <= 50), LDS heap beyond; the generic kernel
        // (per-state slabs, any size) remains for beams whose slab tables fit neither
        const bool wide_ok = fits32 && LM * p <= (uint32_t)fl::WIDE_NS_MAX && LM < 65000 && q.WL.total <= 150 * 1024;
        // (narrow sums: the register-heap slab kernel keeps biallelic sums in 40 bits, beam_slab_kernel.h; a block of >= 65 536 reads goes down the wide / generic kernels)
        const bool narrow = A == 2 && !any_q0;
        q.slab = LM <= 63 && fits32 && LM * p <= (uint32_t)fl::SLAB_NS_MAX && q.SL.total <= 60 * 1024 && !(narrow && n_max >= 65536u);
        q.wide = !q.slab && wide_ok;
        switch (K.beam_path) {                               // dev/test knob
            case 1: q.wide = false; q.slab = false; break;                     // generic
            case 2: q.wide = false; break;                                     // slab (where it applies, else generic)
            case 3: q.wide = wide_ok; q.slab = q.slab && !wide_ok; break;      // wide (where it applies)
            default: break;
        }
        if (K.arith && q.slab && arith_slab) {               // the reference's running sums on the shared slabs: LDS (the terms of a step) allows two waves per SIMD
            q.wide = false;
            q.beam_slots = std::min<uint32_t>(nj_max, ctx->user_slots ? ctx->user_slots : (uint32_t)ctx->n_cu * std::min<uint32_t>(4 * fl::SLAB_WAVES_ARITH, by_lds));
        } else if (K.arith && q.wide && K.beam_path != 1) {  // ... on the shared slabs of a wide beam (beam_wide_kernel<.., ARITH = true>: one lane per live slab walks the cells)
            q.slab = false; q.beam_spec = false;
        } else if (K.arith) {                                // ... or the generic kernel, whose lanes walk a read's cells one by one (beams whose slab tables fit neither shared-slab kernel)
            q.slab = false; q.wide = false; q.beam_spec = false;
            q.LY = fl::beam_lds_layout(LM, p);
            q.beam_slots = std::min<uint32_t>(nj_max, ctx->user_slots ? ctx->user_slots : (uint32_t)ctx->n_cu * std::min<uint32_t>(16, std::max<uint32_t>(1, (uint32_t)((158 * 1024) / (q.LY.total + 256)))));
        }
        // (narrow sums: low words, high bytes and code bytes over one position-index space of NS * span_pad + SLAB_PAD_IDX entries, beam_slab_kernel.h)
        const uint64_t narrow_bytes = (((uint64_t)LM * p * ((span_max + 15u) & ~15u) + fl::SLAB_PAD_IDX) * 11ull + 255ull) & ~255ull;
        if (q.slab && narrow && narrow_bytes >= 0xf0000000ull) q.slab = false;
        if (q.slab) q.state_bytes = narrow ? narrow_bytes : q.state_bytes + slab_code_bytes;
        if (!q.shortcut) {
            if (q.LY.total > 160 * 1024 - 64) return fail(FLORIA_E_UNSUPPORTED, "ploidy*beam needs more LDS than a CU has");
            if (q.wide) q.beam_slots = std::min<uint32_t>(q.beam_slots, (uint32_t)ctx->n_cu * std::max<uint32_t>(1, (uint32_t)((158 * 1024) / (q.WL.total + 512))));
        }
        // optimise
        while (q.cand_cap < (uint64_t)n_max * std::max(1u, p - 1)) q.cand_cap <<= 1;
        uint32_t threads = mean_n >= 384 ? 1024 : (mean_n >= 96 ? 512 : 128);
        if (K.opt_threads == 1024 || K.opt_threads == 512 || K.opt_threads == 128) threads = K.opt_threads;
        const size_t moved_bytes = ((((size_t)n_max + 31) / 32) * 4 + 15) & ~(size_t)15;
        const size_t hist_bytes = (size_t)span_max * p * A * 8;
        const size_t meta_bytes = n_max <= (uint32_t)fl::OPT_META_MAX ? (((size_t)n_max * 12 + 15) & ~(size_t)15) : 0;
        q.hl = hist_bytes + (size_t)span_max * p + 32 + moved_bytes + meta_bytes <= 60 * 1024 && !K.opt_global;
        const size_t code_bytes = q.hl ? ((size_t)span_max * p + 15) & ~(size_t)15 : 0;       // one byte per (position, partition), see optimize_kernel.h
        q.opt_lds = moved_bytes + meta_bytes + (q.hl ? ((hist_bytes + 15) & ~(size_t)15) + code_bytes : 0);
        // where the ploidy-specialised instances apply (75-92 VGPRs), three 512-thread workgroups per CU beat one of 1024 threads
        if (A == 2 && q.hl && p <= 5 && threads == 1024 && !K.opt_threads && !K.no_specialized) threads = 512;
        q.threads = threads;
        q.opt_spec = A == 2 && q.hl && threads >= 512 && p <= 5 && !K.no_specialized && !K.arith;
        auto wg_per_cu = [&](size_t lds) { return std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)((156 * 1024) / (lds + 8 * 1024)), 2048 / threads)); };
        if (K.arith && !K.arith_hbm) {
            // the emulated position maps: control bytes (which every probe reads) + the claim words of a batched insertion, and the first-insertion keys as 32-bit words, in
            // LDS where that still leaves two workgroups per CU; the claim table as large as costs no workgroup (a table of fewer words than buckets makes false conflicts)
            const size_t cap = 68 * 1024;
            auto fx_sz = [&](uint32_t tags) { return (size_t)p * (2 * fx_ctrl + 4 * tags); };
            const size_t fk_sz = ((size_t)p * span_max * 4 + 15) & ~(size_t)15;
            const bool fk_ok = n_max < (1u << 20) && ctx->cur_len_max < 4096u;
            uint32_t tags = fl::FX_TAGS_MIN;
            if (q.opt_lds + fx_sz(tags) + 16 <= cap) {
                const bool fk = fk_ok && q.opt_lds + fx_sz(tags) + fk_sz + 16 <= cap;
                const size_t rest = q.opt_lds + (fk ? fk_sz : 0) + 16;
                const uint32_t tmax = std::min<uint32_t>(fl::FX_TAGS_MAX, K.fx_tags ? K.fx_tags : fl::fx_buckets_for(span_max + 1));
                while (tags * 2 <= tmax && rest + fx_sz(tags * 2) <= cap && wg_per_cu(rest + fx_sz(tags * 2)) == wg_per_cu(rest + fx_sz(fl::FX_TAGS_MIN))) tags *= 2;
                q.fx_lds_off = (uint32_t)q.opt_lds; q.opt_lds += fx_sz(tags); q.fx_tags = tags;
                if (fk) { q.fk_lds_off = (uint32_t)q.opt_lds; q.opt_lds += fk_sz; }
            } else if (fk_ok && q.opt_lds + fk_sz + 16 <= cap) { q.fk_lds_off = (uint32_t)q.opt_lds; q.opt_lds += fk_sz; }
        }
        q.opt_lds += 16;
        {   // the visiting order of the build / distance passes (optimize_kernel.h: u16 per read), where it costs no workgroup per CU
            const size_t pm_bytes = (((size_t)n_max * 2) + 15) & ~(size_t)15;
            if (!K.arith && !K.opt_block_order && meta_bytes && n_max <= 65535u && q.opt_lds + pm_bytes <= 60 * 1024 && wg_per_cu(q.opt_lds + pm_bytes) == wg_per_cu(q.opt_lds)) {
                q.pm_lds_off = (uint32_t)q.opt_lds; q.opt_lds += pm_bytes;
            }
        }
        uint32_t per_cu = wg_per_cu(q.opt_lds);
        per_cu = std::min<uint32_t>(per_cu, 8);
        q.opt_slots = std::min<uint32_t>((uint32_t)ctx->n_cu * per_cu, nj_max);
    }
    // ---- scratch: one fixed slice per lane, sized for the largest ploidy -----------------------------------------------------
    const uint64_t budget = (uint64_t)((double)(free_b + ctx->state_pool.cap + ctx->hist_pool.cap) * 0.6);
    for (;;) {
        uint64_t need = 0;
        for (uint32_t p = 1; p <= P; ++p) if (!plan[p].shortcut) need = std::max(need, (plan[p].state_bytes + plan[p].hist_stride * 4) * plan[p].beam_slots);
        if (need * n_lanes <= budget) break;
        bool shrunk = false;
        for (uint32_t p = 1; p <= P; ++p) if (!plan[p].shortcut && (plan[p].state_bytes + plan[p].hist_stride * 4) * plan[p].beam_slots == need && plan[p].beam_slots > 1) { plan[p].beam_slots /= 2; shrunk = true; }
        if (!shrunk) break;
    }
    uint64_t sl_state = 0, sl_hist = 0, sl_ohist = 0, sl_odist = 0, sl_ogain = 0, sl_okey = 0, sl_omoves = 0, sl_arith = 0;
    auto sort_cap_of = [&](uint32_t p) { uint64_t c = 1; while (c < (uint64_t)p * span_max) c <<= 1; return c; };
    auto up256 = [](uint64_t x) -> uint64_t { return (x + 255) & ~(uint64_t)255; };
    for (uint32_t p = 1; p <= P; ++p) {
        const PloidyPlan& q = plan[p];
        if (!q.shortcut) { sl_state = std::max(sl_state, up256(q.state_bytes * q.beam_slots)); sl_hist = std::max(sl_hist, up256(q.hist_stride * 4 * q.beam_slots)); }
        sl_ohist = std::max(sl_ohist, up256((uint64_t)q.opt_slots * span_max * p * A * 8));
        sl_odist = std::max(sl_odist, up256((uint64_t)q.opt_slots * n_max * p * 8));
        sl_ogain = std::max(sl_ogain, up256((uint64_t)q.opt_slots * q.cand_cap * 8));
        sl_okey = std::max(sl_okey, up256((uint64_t)q.opt_slots * q.cand_cap * 4));
        sl_omoves = std::max(sl_omoves, up256((uint64_t)q.opt_slots * n_max * 4));
        if (K.arith) sl_arith = std::max(sl_arith, up256((uint64_t)q.opt_slots * ((uint64_t)p * span_max * 8 + sort_cap_of(p) * 12 + (uint64_t)p * 2 * fx_bytes + 16 + (uint64_t)p * span_max * 8) + 64));
    }
    {   // (a pool that has to grow is freed first: hipFree synchronises the device, and nothing of this call is in flight yet)
        int rc = ctx->state_pool.ensure(sl_state * n_lanes); if (rc) return rc;
        rc = ctx->hist_pool.ensure(sl_hist * n_lanes); if (rc) return rc;
        rc = ctx->opt_hist.ensure(sl_ohist * n_lanes); if (rc) return rc;
        rc = ctx->opt_dist.ensure(sl_odist * n_lanes); if (rc) return rc;
        rc = ctx->opt_gain.ensure(sl_ogain * n_lanes); if (rc) return rc;
        rc = ctx->opt_key.ensure(sl_okey * n_lanes); if (rc) return rc;
        rc = ctx->opt_moves.ensure(sl_omoves * n_lanes); if (rc) return rc;
        if (K.arith) { rc = ctx->arith_pool.ensure(sl_arith * n_lanes); if (rc) return rc; }
    }
    // ---- streams and events: lane (g, j) runs on its own stream; lane (0, 0) is the context's main stream ---------------------
    hipStream_t ls[floria_hip_ctx::MAX_LANES];
    for (uint32_t l = 0; l < n_lanes; ++l) {
        if (l == 0) { ls[0] = ctx->stream; continue; }
        if (W > 1 && (l % W) >= (stages.front().size() > 1 ? 3u : 1u)) {      // ({1}{2}{3}{4..P}: the lanes of ploidy >= 5)        // a speculative stage's lanes of ploidy >= 4: lowest dispatch priority, so that the ploidies every
                                                            // block needs get the wave slots first and the stop rule is known before most of these jobs start
            if (!ctx->gstream_low[l]) {
                int least = 0, greatest = 0;
                (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
                if (hipStreamCreateWithPriority(&ctx->gstream_low[l], hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipStreamCreateWithFlags(&ctx->gstream_low[l], hipStreamNonBlocking)); }
            }
            if (!ctx->ev_join[l]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_join[l], hipEventDisableTiming));
            if (!ctx->ev_fork[l]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork[l], hipEventDisableTiming));
            ls[l] = ctx->gstream_low[l];
            continue;
        }
        if (!ctx->gstream[l]) HIPCHK(hipStreamCreateWithFlags(&ctx->gstream[l], hipStreamNonBlocking));
        if (!ctx->ev_join[l]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_join[l], hipEventDisableTiming));
        if (!ctx->ev_fork[l]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork[l], hipEventDisableTiming));
        ls[l] = ctx->gstream[l];
    }
    if (!ctx->ev_fork[0]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork[0], hipEventDisableTiming));
    const int t_phase = T.begin(K_PHASE);
    // ploidy 1 (always position 0 of the first stage) without a search: its beam partition is all zeros, for every group's reads
    if (plan[1].shortcut) { HIPCHK(hipMemsetAsync(d_beam_part, 0, tot_reads, ctx->stream)); p1_shortcut = true; }
    // fork: every group's first lane starts after what the main stream has queued so far (uploads, memsets, block_reads_kernel)
    if (G > 1) {
        HIPCHK(hipEventRecord(ctx->ev_fork[0], ctx->stream));
        for (uint32_t g = 1; g < G; ++g) HIPCHK(hipStreamWaitEvent(ls[g * W], ctx->ev_fork[0], 0));
    }
    // chunked inputs: group g's first lane also waits until chunk g's cells have arrived and been flattened
    if (chunk_ev) for (uint32_t g = 0; g < G; ++g) HIPCHK(hipStreamWaitEvent(ls[g * W], chunk_ev[g], 0));
    const uint64_t* H = ctx->d_hash.as<uint64_t>();
    // Per-block dataflow for the tail of a one-ploidy-per-stage plan (graph_processing.rs:132-252 runs a block's ploidies back to back and stops): few blocks reach
    // the last ploidy, so its beam launch is a handful of lone-wave chains as long as the longest block — behind the optimise launch of the ploidy below, which the
    // same long blocks keep waiting.  With tail_p != 0 that beam launch is queued on a stream of its own right behind the beam launch of tail_p - 1: it runs BESIDE
    // the optimise launch, a small persistent grid whose jobs each wait for their block's stop rule (tried[b] >= tail_p - 1 or blk_done[b], beam_slab_kernel.h), in
    // the order the optimise launch works through them; it writes its partitions into a plane of its own.  Nothing waits for this launch but the last optimise launch.
    uint32_t tail_p = 0;
    {
        const bool want = K.tail_overlap > 0;      // (measured on config 4: no gain — the optimise launch it hides is 1.2-1.7 ms when the chip is otherwise idle, and the waiting grid costs as much: profiles/r04_tail_overlap_ab.txt)
        if (want && W == 1 && stages.size() >= 3 && P >= 3 && stages.back().size() == 1 && stages.back()[0] == P && stages[stages.size() - 2][0] == P - 1
            && plan[P].slab && !plan[P].wide && !plan[P].shortcut && !plan[P - 1].shortcut && !K.arith && ctx->hw_queues >= 5 && G <= floria_hip_ctx::MAX_GROUPS) tail_p = P;
    }
    uint8_t* const tail_part = d_beam_part + (uint64_t)W * (tot_reads + 16);
    // one beam launch of group g: ploidy p on stream st (scratch slice `lane`), partitions into lane_part.  wait_tried != 0: the launch of the last ploidy that runs
    // beside the optimise launch of ploidy wait_tried (per-block dataflow, below)
    auto beam_launch = [&](const std::vector<uint32_t>& stage, uint32_t g, uint32_t nj, const uint32_t* gjobs, uint32_t p, uint32_t lane, hipStream_t st, uint32_t* gqueue,
                           uint8_t* lane_part, uint32_t wait_tried) -> int {
        const PloidyPlan& q = plan[p];
        const uint32_t slots_full = std::min(q.beam_slots, nj);
        fl::BeamArgs a{};
        a.bs = bs; a.job_block = gjobs; a.n_jobs = nj; a.ploidy = p; a.beam = B; a.span_max = span_max; a.n_max = n_max;
        a.queue_head = gqueue; a.blk_done = d_done; a.stop_at = stage.size() > 1 && !wait_tried ? d_stop : nullptr;
        a.tried = d_tried; a.wait_tried = wait_tried; a.wait_ticks = 2000000u;      // (20 ms: longer than any optimise launch takes)
        a.state_pool = (uint64_t*)(ctx->state_pool.as<char>() + sl_state * lane); a.state_stride = q.state_bytes;
        a.hist_pool = (uint32_t*)(ctx->hist_pool.as<char>() + sl_hist * lane); a.hist_stride = q.hist_stride;
        a.binom_tab = ctx->d_binom.as<double>(); a.binom_nmax = ctx->binom_nmax;
        a.eps = prm->epsilon; a.div_factor = DIV_FACTOR; a.cutoff = cutoff;
        a.ln_eps = (float)std::log(prm->epsilon); a.ln_1meps = (float)std::log(1.0 - prm->epsilon);
        a.Rq1 = H; a.Rp1 = H + ctx->hash_len; a.Rq2 = H + 2ull * ctx->hash_len; a.Rp2 = H + 3ull * ctx->hash_len;
        a.part_out = lane_part; a.job_margin = d_margin; a.max_ploidy = P; a.diag = d_diag; a.steps_done = d_steps;
        a.prof = (unsigned long long*)(d_diag + 4);
        a.no_bulk = K.no_bulk;
        a.cell_ord = ctx->cur_ord; a.cell_ord_off = ctx->cur_ord_off;
        auto big_lds = [&](const void* kern, uint32_t bytes) -> hipError_t {
            return bytes > 48 * 1024 ? hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) : hipSuccess;
        };
        const bool gated = !wait_tried && stage.size() > 1 && ((stage[0] <= 2 && p >= 4) || (stage[0] == 4 && p >= 5));      // (ungated ploidy 4 / 5 measured worse: 24.3 against 22.5 ms for the 250-contig shard)
        // ... and with a smaller persistent grid: the optimise workgroups of the lower ploidies (whose results decide which of these
        // jobs are needed at all) find room on the chip, and jobs that are dequeued later are dropped more often
        const uint32_t slots = wait_tried ? std::max<uint32_t>(1, std::min(slots_full, (uint32_t)ctx->n_cu * K.tail_waves)) : gated ? std::max<uint32_t>(1, std::min(slots_full, std::max<uint32_t>((uint32_t)ctx->n_cu, slots_full / K.spec_gate_div))) : slots_full;
        auto fire = [&](uint32_t slots) -> int {
            if (K.arith && q.slab) {
                const bool sp = a.stop_at != nullptr || wait_tried != 0;
                auto L = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(slots), dim3(64), q.SL.total, st, a); };
                if (any_q0) { if (sp) L(fl::beam_slab_kernel<A, true, 0, 0, true, true>); else L(fl::beam_slab_kernel<A, true, 0, 0, false, true>); }
                else if (q.beam_spec && p == 2) { if (sp) L(fl::beam_slab_kernel<2, false, 2, 10, true, true>); else L(fl::beam_slab_kernel<2, false, 2, 10, false, true>); }
                else if (q.beam_spec && p == 3) { if (sp) L(fl::beam_slab_kernel<2, false, 3, 10, true, true>); else L(fl::beam_slab_kernel<2, false, 3, 10, false, true>); }
                else if (q.beam_spec && p == 4) { if (sp) L(fl::beam_slab_kernel<2, false, 4, 10, true, true>); else L(fl::beam_slab_kernel<2, false, 4, 10, false, true>); }
                else if (q.beam_spec && p == 5) { if (sp) L(fl::beam_slab_kernel<2, false, 5, 10, true, true>); else L(fl::beam_slab_kernel<2, false, 5, 10, false, true>); }
                else if (sp) L(fl::beam_slab_kernel<A, false, 0, 0, true, true>);
                else L(fl::beam_slab_kernel<A, false, 0, 0, false, true>);
            } else if (K.arith && q.wide) {
                if (any_q0) { HIPCHK(big_lds((const void*)fl::beam_wide_kernel<A, true, true>, q.WL.total)); hipLaunchKernelGGL((fl::beam_wide_kernel<A, true, true>), dim3(slots), dim3(64), q.WL.total, st, a); }
                else { HIPCHK(big_lds((const void*)fl::beam_wide_kernel<A, false, true>, q.WL.total)); hipLaunchKernelGGL((fl::beam_wide_kernel<A, false, true>), dim3(slots), dim3(64), q.WL.total, st, a); }
            } else if (K.arith) {
                HIPCHK(big_lds((const void*)fl::beam_kernel<A, true>, q.LY.total));
                hipLaunchKernelGGL((fl::beam_kernel<A, true>), dim3(slots), dim3(64), q.LY.total, st, a);
            } else if (q.wide) {
                if (any_q0) { HIPCHK(big_lds((const void*)fl::beam_wide_kernel<A, true>, q.WL.total)); hipLaunchKernelGGL((fl::beam_wide_kernel<A, true>), dim3(slots), dim3(64), q.WL.total, st, a); }
                else { HIPCHK(big_lds((const void*)fl::beam_wide_kernel<A, false>, q.WL.total)); hipLaunchKernelGGL((fl::beam_wide_kernel<A, false>), dim3(slots), dim3(64), q.WL.total, st, a); }
            } else if (q.slab) {
                const bool sp = a.stop_at != nullptr || wait_tried != 0;      // speculative stage: the instances that can drop a job
                if (any_q0) { if (sp) hipLaunchKernelGGL((fl::beam_slab_kernel<A, true, 0, 0, true>), dim3(slots), dim3(64), q.SL.total, st, a);
                              else hipLaunchKernelGGL((fl::beam_slab_kernel<A, true>), dim3(slots), dim3(64), q.SL.total, st, a); }
                else if (q.beam_spec && p == 2) { if (sp) hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 2, 10, true>), dim3(slots), dim3(64), q.SL.total, st, a);
                                                  else hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 2, 10>), dim3(slots), dim3(64), q.SL.total, st, a); }
                else if (q.beam_spec && p == 3) { if (sp) hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 3, 10, true>), dim3(slots), dim3(64), q.SL.total, st, a);
                                                  else hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 3, 10>), dim3(slots), dim3(64), q.SL.total, st, a); }
                else if (q.beam_spec && p == 4) { if (sp) hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 4, 10, true>), dim3(slots), dim3(64), q.SL.total, st, a);
                                                  else hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 4, 10>), dim3(slots), dim3(64), q.SL.total, st, a); }
                else if (q.beam_spec && p == 5) { if (sp) hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 5, 10, true>), dim3(slots), dim3(64), q.SL.total, st, a);
                                                  else hipLaunchKernelGGL((fl::beam_slab_kernel<2, false, 5, 10>), dim3(slots), dim3(64), q.SL.total, st, a); }
                else if (sp) hipLaunchKernelGGL((fl::beam_slab_kernel<A, false, 0, 0, true>), dim3(slots), dim3(64), q.SL.total, st, a);
                else hipLaunchKernelGGL((fl::beam_slab_kernel<A, false>), dim3(slots), dim3(64), q.SL.total, st, a);
            } else {
                HIPCHK(big_lds((const void*)fl::beam_kernel<A>, q.LY.total));
                hipLaunchKernelGGL(fl::beam_kernel<A>, dim3(slots), dim3(64), q.LY.total, st, a);
            }
            return 0;
        };
        if (gated) {          // ploidies few blocks need start when ploidy 2 has left the chip: by the time their jobs run, the stop rule of most
                              // blocks is known and the jobs are dropped at dequeue or within 64 reads
            if (!ctx->ev_gate[g]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_gate[g], hipEventDisableTiming));
            HIPCHK(hipStreamWaitEvent(st, ctx->ev_gate[g], 0));
        }
        const int t = T.begin(K_BEAM, st);
        { const int frc = fire(slots); if (frc) return frc; }
        T.end(t);
        HIPCHK(hipGetLastError());
        ctx->timing.beam_launches++;
        if (!wait_tried && stage.size() > 1 && ((stage[0] <= 2 && p == 2) || (stage[0] == 4 && p == 4))) {          // (the beam search of ploidy 2 opens the gate; ploidy 3 measured worse)
            if (!ctx->ev_gate[g]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_gate[g], hipEventDisableTiming));
            HIPCHK(hipEventRecord(ctx->ev_gate[g], st));
        }
        return 0;
    };
    for (size_t si = 0; si < stages.size(); ++si) {
        const std::vector<uint32_t>& stage = stages[si];
        for (uint32_t g = 0; g < G; ++g) {
            const uint32_t nj = group_off[g + 1] - group_off[g];
            if (nj == 0) continue;
            const uint32_t* gjobs = d_jobs + group_off[g];
            hipStream_t s0 = ls[g * W];
            if (stage.size() > 1) {                          // the stage's extra lanes start after the group's previous stop rule
                HIPCHK(hipEventRecord(ctx->ev_fork[g * W], s0));
                for (uint32_t j = 1; j < stage.size(); ++j) HIPCHK(hipStreamWaitEvent(ls[g * W + j], ctx->ev_fork[g * W], 0));
            }
            for (uint32_t jj = 0; jj < stage.size(); ++jj) {
                const uint32_t j = jj;
                const uint32_t p = stage[j], lane = g * W + j;
                const PloidyPlan& q = plan[p];
                hipStream_t st = ls[lane];
                uint32_t* gqueue = d_queue + 2 * ((size_t)lane * P + (p - 1));     // [0] beam, [1] optimise: one counter pair per (lane, ploidy), zeroed once before the
                                                                                    // loop (a memset between persistent launches is a fill KERNEL that waits for wave slots)
                uint8_t* lane_part = d_beam_part + (uint64_t)j * (tot_reads + 16);
                // ---- beam search -----------------------------------------------------------------------------------------
                const bool tail_here = tail_p && p == tail_p;          // this ploidy's beam launch was queued with the stage below: join it
                if (tail_here) { HIPCHK(hipStreamWaitEvent(st, ctx->ev_tail_join[g], 0)); lane_part = tail_part; }
                else if (!q.shortcut) { const int brc = beam_launch(stage, g, nj, gjobs, p, lane, st, gqueue, lane_part, 0); if (brc) return brc; }
                if (tail_p && p + 1 == tail_p) {                         // the fork point: behind this ploidy's beam launch, before its optimise launch
                    if (!ctx->ev_tail[g]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_tail[g], hipEventDisableTiming));
                    HIPCHK(hipEventRecord(ctx->ev_tail[g], st));
                }
                // ---- optimise + MEC stats ------------------------------------------------------------------------------
                {
                    const uint32_t slots = std::min(q.opt_slots, nj);
                    const uint32_t threads = q.threads;
                    const size_t lds = q.opt_lds;
                    fl::OptArgs a{};
                    a.bs = bs; a.job_block = gjobs; a.n_jobs = nj; a.ploidy = p; a.max_ploidy = P; a.span_max = span_max; a.n_max = n_max;
                    a.queue_head = gqueue + 1; a.blk_done = d_done; a.eps = prm->epsilon;
                    a.part_in = lane_part; a.part_out = d_planes + (uint64_t)(p - 1) * tot_reads;
                    a.hist_pool = (uint64_t*)(ctx->opt_hist.as<char>() + sl_ohist * lane); a.dist_pool = (double*)(ctx->opt_dist.as<char>() + sl_odist * lane);
                    a.cand_gain_pool = (uint64_t*)(ctx->opt_gain.as<char>() + sl_ogain * lane); a.cand_key_pool = (uint32_t*)(ctx->opt_key.as<char>() + sl_okey * lane);
                    a.moves_pool = (uint32_t*)(ctx->opt_moves.as<char>() + sl_omoves * lane); a.cand_cap = q.cand_cap;
                    a.mec = d_mec; a.num_alleles = d_na; a.iters = d_iters;
                    a.prof = (unsigned long long*)(d_diag + 4);
                    a.fuse_select = stage.size() == 1 ? 1 : 0;
                    a.stopping_heuristic = prm->stopping_heuristic; a.mec_threshold = mec_threshold(prm, p);
                    a.blk_done_w = d_done; a.best_ploidy = d_best; a.tried = d_tried;
                    a.release_tried = tail_p != 0 ? 1u : 0u;
                    a.pm_lds_off = q.pm_lds_off;
                    // (a plan with ANY speculative stage publishes ready bits and stop_at from every stage: a later stage's stop rule compares with these MEC values)
                    if (K.arith) {
                        a.cell_ord = ctx->cur_ord; a.cell_ord_off = ctx->cur_ord_off;
                        char* base = ctx->arith_pool.as<char>() + sl_arith * lane;
                        a.sort_cap = sort_cap_of(p); a.fx_ctrl = fx_ctrl; a.fx_slot = fx_slot; a.fx_lds_off = q.fx_lds_off; a.fx_tags = q.fx_tags; a.fx_replay = K.arith_replay; a.fk_lds_off = q.fk_lds_off;
                        a.fk_pool = (uint64_t*)base; base += (uint64_t)slots * p * span_max * 8;
                        a.sk_pool = (uint64_t*)base; base += (uint64_t)slots * a.sort_cap * 8;
                        a.sp_pool = (uint32_t*)base; base += ((uint64_t)slots * a.sort_cap * 4 + 15) & ~(uint64_t)15;
                        a.fx_pool = (uint8_t*)base; base += (((uint64_t)slots * p * 2 * fx_bytes) + 15) & ~(uint64_t)15;
                        a.ol_pool = (uint32_t*)base;
                    }
                    if (W > 1) { a.stop_at = d_stop; a.ready = d_ready; for (uint32_t q2 = 2; q2 <= P; ++q2) a.thresholds[q2] = mec_threshold(prm, q2); }
                    int t = T.begin(K_OPT, st);
                    auto launch = [&](auto kern) -> hipError_t {
                        if (lds > 48 * 1024) { hipError_t e2 = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e2 != hipSuccess) return e2; }
                        hipLaunchKernelGGL(kern, dim3(slots), dim3(threads), lds, st, a);
                        return hipGetLastError();
                    };
                    hipError_t le;
                    // reference arithmetic, 512 threads, histogram in LDS, biallelic, ploidy <= 5 (every BASELINE config at the CLI's defaults): ploidy-specialised instances
                    // (round 6: the per-partition loops unroll, `pair / p` is a multiplication); four waves per SIMD (128 VGPRs) where LDS allows two workgroups per CU anyway
                    const bool arith_two_wg = K.arith && q.hl && threads == 512 && std::min<size_t>((156 * 1024) / (lds + 8 * 1024), 2048 / threads) <= 2 && !K.arith_ow6;
                    if (K.arith && q.hl && threads == 512 && A == 2 && p <= 5 && !K.no_specialized) {
                        if (arith_two_wg) le = p == 1 ? launch(fl::optimize_kernel<2, true, 512, 1, true, 4>) : p == 2 ? launch(fl::optimize_kernel<2, true, 512, 2, true, 4>) : p == 3 ? launch(fl::optimize_kernel<2, true, 512, 3, true, 4>)
                                             : p == 4 ? launch(fl::optimize_kernel<2, true, 512, 4, true, 4>) : launch(fl::optimize_kernel<2, true, 512, 5, true, 4>);
                        else le = p == 1 ? launch(fl::optimize_kernel<2, true, 512, 1, true>) : p == 2 ? launch(fl::optimize_kernel<2, true, 512, 2, true>) : p == 3 ? launch(fl::optimize_kernel<2, true, 512, 3, true>)
                                : p == 4 ? launch(fl::optimize_kernel<2, true, 512, 4, true>) : launch(fl::optimize_kernel<2, true, 512, 5, true>);
                    }
                    else if (arith_two_wg) le = launch(fl::optimize_kernel<A, true, 512, 0, true, 4>);      // (LDS allows two workgroups per CU: four waves per SIMD, 128 VGPRs)
                    else if (K.arith && q.hl) le = threads == 1024 ? launch(fl::optimize_kernel<A, true, 1024, 0, true>) : threads == 512 ? launch(fl::optimize_kernel<A, true, 512, 0, true>) : launch(fl::optimize_kernel<A, true, 128, 0, true>);
                    else if (K.arith) le = threads == 1024 ? launch(fl::optimize_kernel<A, false, 1024, 0, true>) : threads == 512 ? launch(fl::optimize_kernel<A, false, 512, 0, true>) : launch(fl::optimize_kernel<A, false, 128, 0, true>);
                    else if (q.opt_spec && threads == 1024)
                        le = p == 1 ? launch(fl::optimize_kernel<2, true, 1024, 1>) : p == 2 ? launch(fl::optimize_kernel<2, true, 1024, 2>) : p == 3 ? launch(fl::optimize_kernel<2, true, 1024, 3>)
                           : p == 4 ? launch(fl::optimize_kernel<2, true, 1024, 4>) : launch(fl::optimize_kernel<2, true, 1024, 5>);
                    else if (q.opt_spec)
                        le = p == 1 ? launch(fl::optimize_kernel<2, true, 512, 1>) : p == 2 ? launch(fl::optimize_kernel<2, true, 512, 2>) : p == 3 ? launch(fl::optimize_kernel<2, true, 512, 3>)
                           : p == 4 ? launch(fl::optimize_kernel<2, true, 512, 4>) : launch(fl::optimize_kernel<2, true, 512, 5>);
                    else if (q.hl) le = threads == 1024 ? launch(fl::optimize_kernel<A, true, 1024>) : threads == 512 ? launch(fl::optimize_kernel<A, true, 512>) : launch(fl::optimize_kernel<A, true, 128>);
                    else    le = threads == 1024 ? launch(fl::optimize_kernel<A, false, 1024>) : threads == 512 ? launch(fl::optimize_kernel<A, false, 512>) : launch(fl::optimize_kernel<A, false, 128>);
                    T.end(t);
                    if (le != hipSuccess) return fail(FLORIA_E_DEVICE, std::string("optimize_kernel launch: ") + hipGetErrorString(le));
                    ctx->timing.optimize_launches++;
                }
                if (tail_p && p + 1 == tail_p) {          // (queued AFTER the optimise launch it depends on: streams that share a hardware queue then still make progress)
                    if (!ctx->tstream[g]) HIPCHK(hipStreamCreateWithFlags(&ctx->tstream[g], hipStreamNonBlocking));
                    if (!ctx->ev_tail_join[g]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_tail_join[g], hipEventDisableTiming));
                    HIPCHK(hipStreamWaitEvent(ctx->tstream[g], ctx->ev_tail[g], 0));
                    const int brc = beam_launch(stages.back(), g, nj, gjobs, tail_p, lane, ctx->tstream[g], d_queue + 2 * ((size_t)lane * P + (tail_p - 1)), tail_part, p);
                    if (brc) return brc;
                    HIPCHK(hipEventRecord(ctx->ev_tail_join[g], ctx->tstream[g]));
                }
            }
            // join the stage's extra lanes, then the stop rule for the stage's ploidies in ascending order (graph_processing.rs:198-251)
            for (uint32_t j = 1; j < stage.size(); ++j) { HIPCHK(hipEventRecord(ctx->ev_join[g * W + j], ls[g * W + j])); HIPCHK(hipStreamWaitEvent(s0, ctx->ev_join[g * W + j], 0)); }
            for (uint32_t j = 0; j < stage.size() && stage.size() > 1; ++j) {                  // (a one-ploidy stage decided inside optimize_kernel)
                const uint32_t p = stage[j];
                fl::SelectArgs s{};
                s.job_block = gjobs; s.n_jobs = nj; s.ploidy = p; s.max_ploidy = P; s.stopping_heuristic = prm->stopping_heuristic; s.eps = prm->epsilon;
                s.mec_threshold = mec_threshold(prm, p);
                s.mec = d_mec; s.num_alleles = d_na; s.iters = d_iters; s.blk_done = d_done; s.best_ploidy = d_best; s.tried = d_tried;
                s.clear_from = (j + 1 == stage.size()) ? 1 : 0;      // last select of a speculative stage: forget the ploidies beyond `tried`
                s.stage_last = stage.back();
                int t = T.begin(K_SEL, s0);
                hipLaunchKernelGGL(fl::select_kernel, dim3((nj + 255) / 256), dim3(256), 0, s0, s);
                T.end(t);
                HIPCHK(hipGetLastError());
            }
        }
    }
    // join: the main stream continues after every group
    for (uint32_t g = 1; g < G; ++g) { HIPCHK(hipEventRecord(ctx->ev_join[g * W], ls[g * W])); HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_join[g * W], 0)); }
    T.end(t_phase);
    ctx->timing.streams = G;
    ctx->timing.stage_width = W;
    return 0;
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* floria_hip_last_error(void) { return g_err.c_str(); }

}  // extern "C"

// HIP reads GPU_MAX_HW_QUEUES once, when the runtime initialises (lazily, at the first API call); the launch plans of s1_core want 12 hardware queues (below).
// The library does NOT touch the environment on its own (rounds 2-4 did, from a constructor: a setenv() in a process that other threads may be reading the
// environment of): a host either exports GPU_MAX_HW_QUEUES=12 itself or calls floria_hip_init_env() once, from its main thread, before its first HIP
// call; a host that did neither is noticed (probe_hw_queues) and gets plans that stay within the queues it has, with one message on stderr.
extern "C" int floria_hip_init_env(void) { return setenv("GPU_MAX_HW_QUEUES", "12", 0) == 0 ? 0 : FLORIA_E_INVALID; }      // (never overrides a value the user has set)

namespace {
// How many streams of this process really execute side by side?  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the environment
// said otherwise BEFORE the runtime initialised — a host that touched HIP earlier cannot fix it afterwards), and streams that share a queue run one
// after the other.  The launch plans of s1_core put job groups and the ploidies of a speculative stage on separate streams that WAIT ON EACH OTHER'S
// EVENTS; with fewer queues than lanes they serialise (measured: 3-8x slower) instead of failing.  So the context measures it once: one wave that spins
// for 1 ms, alone and then on 12 fresh streams at once; 12 x (time alone) / (time together) is the number of lanes that ran concurrently.
__global__ void spin_kernel(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && threadIdx.x == 1234567) *sink = 1;
}
uint32_t probe_hw_queues(floria_hip_ctx* c) {
    constexpr int NS = 12;
    hipStream_t st[NS] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    uint32_t result = 0;
    bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    for (int i = 0; i < NS && ok; ++i) ok = hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) == hipSuccess;
    const unsigned long long ticks = 100000;         // 1 ms of the 100 MHz wall clock: long against the ~10 us the host needs per launch, so the 12 spins overlap if they can
    auto timed = [&](int n) -> float {
        float ms = 0.f;
        if (hipEventRecord(e0, c->stream) != hipSuccess) return -1.f;
        for (int i = 0; i < n; ++i) { (void)hipStreamWaitEvent(st[i], e0, 0); hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[i], ticks, (unsigned*)nullptr); }
        for (int i = 0; i < n; ++i) { (void)hipEventRecord(e1, st[i]); (void)hipStreamWaitEvent(c->stream, e1, 0); }
        if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
        return ms;
    };
    if (ok) {
        (void)timed(1);                              // (first launch: code object load)
        const float t1 = timed(1), tn = timed(NS);
        if (t1 > 0.f && tn > 0.f) result = (uint32_t)std::max(1.0f, std::min((float)NS, (float)NS * t1 / tn + 0.35f));
    }
    for (int i = 0; i < NS; ++i) if (st[i]) (void)hipStreamDestroy(st[i]);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipGetLastError();
    return result ? result : 4;                      // (probe failed: assume the runtime's default)
}
}  // namespace

extern "C" {
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
    if (c->d_w24.ensure(sizeof(w24)) || hipMemcpy(c->d_w24.p, w24, sizeof(w24), hipMemcpyHostToDevice) != hipSuccess) { floria_hip_destroy(c); return fail(FLORIA_E_DEVICE, "weight table upload failed"); }
    {   // development knobs: environment defaults, read once (floria_hip_set_option overrides them)
        Knobs& K = c->knobs;
        if (const char* v = getenv("FLORIA_HIP_GROUPS")) K.groups = (uint32_t)std::max(0, std::min<int>(atoi(v), floria_hip_ctx::MAX_GROUPS));
        if (const char* v = getenv("FLORIA_HIP_BEAM")) K.beam_path = !strcmp(v, "generic") ? 1 : !strcmp(v, "slab") ? 2 : !strcmp(v, "wide") ? 3 : 0;
        K.no_specialized = getenv("FLORIA_HIP_NO_SPECIALIZED") != nullptr;
        K.no_p1_shortcut = getenv("FLORIA_HIP_NO_P1_SHORTCUT") != nullptr;
        if (const char* v = getenv("FLORIA_HIP_OPT_THREADS")) { const int tv = atoi(v); if (tv == 1024 || tv == 512 || tv == 128) K.opt_threads = (uint32_t)tv; }
        K.opt_global = getenv("FLORIA_HIP_OPT_GLOBAL") != nullptr;
        K.opt_block_order = getenv("FLORIA_HIP_OPT_BLOCK_ORDER") != nullptr;
        if (const char* v = getenv("FLORIA_HIP_SPECULATE")) K.speculate = std::max(-1, std::min(3, atoi(v)));
        K.trace = getenv("FLORIA_HIP_TRACE") != nullptr;
        if (const char* v = getenv("FLORIA_HIP_TAIL_OVERLAP")) K.tail_overlap = std::max(-1, std::min(1, atoi(v)));
        if (const char* v = getenv("FLORIA_HIP_TAIL_WAVES")) K.tail_waves = (uint32_t)std::max(1, std::min(16, atoi(v)));
        if (const char* v = getenv("FLORIA_HIP_SPEC_GATE_DIV")) K.spec_gate_div = (uint32_t)std::max(1, std::min(16, atoi(v)));
        if (const char* v = getenv("FLORIA_HIP_STAGE_THREADS")) c->stage_threads = (uint32_t)std::max(1, std::min(16, atoi(v)));
        else c->stage_threads = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    }
    c->hw_queues = probe_hw_queues(c);
    if (const char* v = getenv("FLORIA_HIP_HW_QUEUES")) c->hw_queues = (uint32_t)std::max(1, std::min(64, atoi(v)));      // (measurements under a profiler, which slows the probe's launches down)
    if (c->knobs.trace) fprintf(stderr, "[floria_hip] device %d: %u streams run side by side\n", device, c->hw_queues);
    if (c->hw_queues < 5) {      // (the probe reads 6 with GPU_MAX_HW_QUEUES=12 - two rounds of its 12 spins - and 3-4 with HIP's default of 4 queues - three or four rounds)
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "floria_hip: only %u streams of this process run concurrently on device %d (GPU_MAX_HW_QUEUES was not set to 12 before HIP initialised?): "
                            "speculative ploidy stages are off and calls use at most two job groups; small batches will be slower.\n", c->hw_queues, device);
    }
    *out = c;
    return 0;
}

void floria_hip_destroy(floria_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    sync_all(c);
    for (DevBuf* b : {&c->d_binom, &c->d_hash, &c->d_w24, &c->state_pool, &c->hist_pool, &c->opt_hist, &c->opt_dist, &c->opt_gain, &c->opt_key, &c->opt_moves, &c->misc, &c->misc0, &c->graph_buf, &c->graph_hist, &c->graph_sort, &c->up_tmp, &c->arith_ord, &c->arith_scr, &c->arith_tab, &c->arith_pool}) b->release();
    for (Arena* a : c->arena_cache) { a->buf.release(); delete a; }
    c->stage.release();
    c->box.release();
    for (uint32_t g = 0; g < floria_hip_ctx::MAX_LANES; ++g) {
        if (c->gstream[g]) (void)hipStreamDestroy(c->gstream[g]);
        if (c->gstream_low[g]) (void)hipStreamDestroy(c->gstream_low[g]);
        if (g < floria_hip_ctx::MAX_GROUPS && c->ev_gate[g]) (void)hipEventDestroy(c->ev_gate[g]);
        if (g < floria_hip_ctx::MAX_GROUPS && c->ev_tail[g]) (void)hipEventDestroy(c->ev_tail[g]);
        if (g < floria_hip_ctx::MAX_GROUPS && c->ev_tail_join[g]) (void)hipEventDestroy(c->ev_tail_join[g]);
        if (g < floria_hip_ctx::MAX_GROUPS && c->tstream[g]) (void)hipStreamDestroy(c->tstream[g]);
        if (c->ev_join[g]) (void)hipEventDestroy(c->ev_join[g]);
        if (c->ev_fork[g]) (void)hipEventDestroy(c->ev_fork[g]);
    }
    if (c->ev_rids) (void)hipEventDestroy(c->ev_rids);
    for (hipEvent_t e : c->ev_chunk) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_copied) if (e) (void)hipEventDestroy(e);
    if (c->flat_stream) (void)hipStreamDestroy(c->flat_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

int floria_hip_set_slots(floria_hip_ctx* ctx, uint32_t beam_slots) {
    if (!ctx) return fail(FLORIA_E_INVALID, "null ctx");
    ctx->user_slots = beam_slots;
    return 0;
}

int floria_hip_set_option(floria_hip_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return fail(FLORIA_E_INVALID, "null argument");
    const std::string k(key);
    Knobs& K = ctx->knobs;
    if (k == "groups") K.groups = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, floria_hip_ctx::MAX_GROUPS));
    else if (k == "beam_path") { if (value < 0 || value > 3) return fail(FLORIA_E_INVALID, "beam_path: 0 auto | 1 generic | 2 slab | 3 wide"); K.beam_path = (uint32_t)value; }
    else if (k == "no_specialized") K.no_specialized = value != 0;
    else if (k == "no_p1_shortcut") K.no_p1_shortcut = value != 0;
    else if (k == "opt_threads") { if (value != 0 && value != 128 && value != 512 && value != 1024) return fail(FLORIA_E_INVALID, "opt_threads: 0 | 128 | 512 | 1024"); K.opt_threads = (uint32_t)value; }
    else if (k == "opt_global") K.opt_global = value != 0;
    else if (k == "spec_gate_div") K.spec_gate_div = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 16));
    else if (k == "speculate") { if (value < -1 || value > 3) return fail(FLORIA_E_INVALID, "speculate: -1 auto | 0 | 1 | 2 | 3"); K.speculate = (int32_t)value; }
    else if (k == "no_bulk") K.no_bulk = value != 0;
    else if (k == "opt_block_order") K.opt_block_order = value != 0;
    else if (k == "tail_overlap") { if (value < -1 || value > 1) return fail(FLORIA_E_INVALID, "tail_overlap: 0 | 1"); K.tail_overlap = (int32_t)value; }
    else if (k == "tail_waves") K.tail_waves = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 16));
    else if (k == "arith_hbm") K.arith_hbm = value != 0;
    else if (k == "arith_replay") { K.arith_replay = value != 0; ctx->ord_epoch = ~0ull; }      // (the cell orders are computed again, the long way or the short one)
    else if (k == "fx_tags") { if (value != 0 && (value < (int64_t)fl::FX_TAGS_MIN || value > (int64_t)fl::FX_TAGS_MAX || (value & (value - 1)))) return fail(FLORIA_E_INVALID, "fx_tags: 0 | a power of two in 128..1024"); K.fx_tags = (uint32_t)value; }
    else if (k == "arith_ow6") K.arith_ow6 = value != 0;
    else if (k == "s2_assign_only") K.s2_assign_only = value != 0;
    else if (k == "arith") { if (value < 0 || value > 1) return fail(FLORIA_E_INVALID, "arith: 0 canonical | 1 the reference's running sums"); K.arith = (uint32_t)value; }
    else if (k == "hw_queues") ctx->hw_queues = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 64));      // (tests: pretend the probe found this many)
    else if (k == "upload_chunks") K.upload_chunks = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, floria_hip_ctx::MAX_GROUPS));
    else if (k == "trace") K.trace = value != 0;
    else if (k == "reassign_path") { if (value < 0 || value > 2) return fail(FLORIA_E_INVALID, "reassign_path: 0 auto | 1 parallel | 2 chain"); K.reassign_path = (uint32_t)value; }
    else if (k == "slots") ctx->user_slots = (uint32_t)std::max<int64_t>(0, value);
    else if (k == "stage_threads") ctx->stage_threads = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 16));
    else return fail(FLORIA_E_INVALID, "unknown option '" + k + "'");
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
// Pinned host memory for hosts that marshal their `Vec<Frag>` straight into upload buffers: arrays that live here go to the
// device by DMA with no staging copy.
void* floria_hip_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); fail(FLORIA_E_NOMEM, "hipHostMalloc failed"); return nullptr; }
    return p;
}
void floria_hip_host_free(void* p) { if (p) (void)hipHostFree(p); }

}  // extern "C"

namespace {

bool is_pinned(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// One host->device transfer plan entry: `bytes` from `src` to `dst` (contiguous on both sides).
struct CopyRun { const char* src; char* dst; size_t bytes; };

int issue_copies(floria_hip_ctx* ctx, std::vector<CopyRun>& runs, uint64_t* pinned_bytes, uint64_t* staged_bytes);

Arena* arena_get(floria_hip_ctx* ctx, size_t bytes);
void arena_put(Arena* a);

int host_meta(const floria_hip_contig* cc) {
    floria_hip_contig* c = const_cast<floria_hip_contig*>(cc);
    Arena* a = c->arena;
    if (!a->host_meta) {
        a->h_first.resize(a->R + 1); a->h_last.resize(a->R + 1);
        if (a->R) {
            HIPCHK(hipMemcpy(a->h_first.data(), a->buf.as<char>() + a->off_first, 4 * a->R, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(a->h_last.data(), a->buf.as<char>() + a->off_last, 4 * a->R, hipMemcpyDeviceToHost));
        }
        a->host_meta = true;
    }
    c->h_first = a->h_first.data() + a->read_prefix[c->idx];
    c->h_last = a->h_last.data() + a->read_prefix[c->idx];
    return 0;
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// The plan of one batch upload: where every array of every contig lands in the arena, which host->device transfers carry it
// (arrays of consecutive contigs that are back to back in host memory merge into one transfer) and the flatten kernel's tables.
// The cell arrays (snp, allele, qual — 99 % of the bytes) are planned per CHUNK of consecutive contigs, so that a caller can
// start work on chunk g while chunk g+1 is still on the wire (floria_hip_phase_pileups_batch).
struct UploadPlan {
    uint32_t n = 0, n_chunks = 1;
    std::vector<uint64_t> rp, cp;                       // read / cell prefix per contig
    uint64_t R = 0, C = 0;
    Arena* A = nullptr;
    char *D = nullptr, *T = nullptr;                    // arena base, transient base
    size_t t_cd = 0, t_rp = 0, t_st = 0;
    std::vector<fl::UploadContig> ucd;
    std::vector<fl::UploadStatus> ust;
    std::vector<fl::PackedContig> upk;                  // packed uploads: what expand_kernel reads / writes (device table at t_pk)
    size_t t_pk = 0;
    bool packed = false;
    std::vector<CopyRun> small_runs;                    // read_off, first, last (packed: + bit_off) of every contig
    std::vector<std::vector<CopyRun>> chunk_runs;       // snp, allele, qual per chunk
    std::vector<uint32_t> chunk_first;                  // [n_chunks+1] contig boundaries
    std::vector<uint32_t> contig_chunk;                 // [n]
    std::vector<const uint32_t*> so_dev;                // [n] device copy of the contig's set_order (include/floria_hip.h), null where the host gave none
    bool all_pinned = false;
};

// (a batch is either CSR pileups or packed ones: `pk` non-null selects the compact wire form, expanded on the device)
int plan_upload(floria_hip_ctx* ctx, const floria_pileup* pileups, const floria_pileup_packed* pk, uint32_t n, uint32_t n_chunks, UploadPlan& P) {
    P.n = n; P.packed = pk != nullptr;
    ctx->upload_epoch++;                               // (device memory of contigs is about to be rewritten: cached cell orders no longer describe it)
    P.rp.assign(n + 1, 0); P.cp.assign(n + 1, 0);
    std::vector<floria_pileup> views;                   // packed: the fields both forms share, so that the code below reads one type
    if (pk) {
        views.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            const floria_pileup_packed* q = &pk[i];
            if (q->n_reads && (!q->read_off || !q->first || !q->last || !q->bit_off || !q->present || !q->allele2 || !q->qual)) return fail(FLORIA_E_INVALID, "null pileup field");
            views[i] = floria_pileup{q->read_off, nullptr, nullptr, q->qual, q->first, q->last, q->n_reads, q->set_order};
        }
        pileups = views.data();
    }
    for (uint32_t i = 0; i < n; ++i) {
        const floria_pileup* p = &pileups[i];
        if (!pk && p->n_reads && (!p->read_off || !p->snp || !p->allele || !p->qual || !p->first || !p->last)) return fail(FLORIA_E_INVALID, "null pileup field");
        const uint64_t nc = p->n_reads ? p->read_off[p->n_reads] : 0;
        if (nc >= (1ull << 32)) return fail(FLORIA_E_UNSUPPORTED, "more than 2^32 cells in one contig");
        P.rp[i + 1] = P.rp[i] + p->n_reads; P.cp[i + 1] = P.cp[i] + nc;
    }
    const uint64_t R = P.R = P.rp[n], C = P.C = P.cp[n];
    // chunks: consecutive contigs, balanced by cells
    n_chunks = std::max<uint32_t>(1, std::min<uint32_t>(n_chunks, n));
    P.n_chunks = n_chunks;
    P.chunk_first.assign(n_chunks + 1, n); P.chunk_first[0] = 0;
    P.contig_chunk.assign(n, 0);
    {
        // chunk weights: a small first chunk puts the GPU to work early, a small last chunk keeps the chain that starts last short
        std::vector<double> cum(n_chunks + 1, 0.0);
        for (uint32_t g = 0; g < n_chunks; ++g) {
            const double w = (n_chunks >= 2 && g == 0) ? 0.5 : 1.0;          // (a half-size first chunk measured best)
            cum[g + 1] = cum[g] + w;
        }
        uint32_t g = 0;
        for (uint32_t i = 0; i < n; ++i) {
            while (g + 1 < n_chunks && (double)P.cp[i] >= (double)C * cum[g + 1] / cum[n_chunks] && i > P.chunk_first[g]) P.chunk_first[++g] = i;
            P.contig_chunk[i] = g;
        }
        for (uint32_t h = g + 1; h < n_chunks; ++h) P.chunk_first[h] = n;          // (fewer non-empty chunks than asked for)
    }
    // ---- arena layout -------------------------------------------------------------------------------------------------------
    size_t cursor = 0;
    auto seg = [&](size_t bytes) { const size_t o = cursor; cursor += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_ro = seg(4 * (R + n)), o_first = seg(4 * R), o_last = seg(4 * R), o_snp = seg(4 * C + 16), o_aw = seg(4 * C + 16),      // 16-B tails: the
                 o_tw = seg(16 * R), o_meta = seg(32 * R);                                              // beam kernel's LDS-DMA moves cells in 16-B pieces
    bool any_so = false;
    for (uint32_t i = 0; i < n; ++i) any_so = any_so || (pileups[i].n_reads && pileups[i].set_order);
    const size_t o_so = any_so ? seg(4 * C + 16) : 0;                                                   // host-given set orders (optional, the reference-arithmetic mode reads them)
    Arena* A = P.A = arena_get(ctx, cursor + 256);
    if (!A) return FLORIA_E_NOMEM;
    A->n_contigs = n; A->R = R; A->C = C; A->off_ro = o_ro; A->off_first = o_first; A->off_last = o_last; A->read_prefix = P.rp; A->host_meta = false;
    char* D = P.D = A->buf.as<char>();
    // transient: raw allele / qual bytes, the kernel's contig table, prefix and status
    size_t c2 = 0;
    auto seg2 = [&](size_t bytes) { const size_t o = c2; c2 += (bytes + 255) & ~(size_t)255; return o; };
    const size_t t_al = seg2(C + 16), t_q = seg2(C + 16);
    P.t_cd = seg2(sizeof(fl::UploadContig) * n); P.t_rp = seg2(8 * (n + 1)); P.t_st = seg2(sizeof(fl::UploadStatus) * n);
    // packed: the compact arrays are transient too (bit offsets, presence bits, 2-bit alleles), one sub-range per contig — at the offsets
    // floria_hip_pack_pileups_batch gives them in host memory, so that a batch packed by it travels as one transfer per field and chunk
    std::vector<uint64_t> pb(n + 1, 0);                 // presence bytes prefix
    size_t t_bo = 0, t_pr = 0, t_a2 = 0;
    if (pk) {
        for (uint32_t i = 0; i < n; ++i) pb[i + 1] = pb[i] + (pk[i].n_reads ? ((uint64_t)pk[i].bit_off[pk[i].n_reads] + 7) / 8 : 0) + 1;
        t_bo = seg2(4 * (R + n)); t_pr = seg2(pb[n] + 16); t_a2 = seg2(C / 4 + n + 16); P.t_pk = seg2(sizeof(fl::PackedContig) * n);
    }
    if (int rc = ctx->up_tmp.ensure(c2 + 256)) { arena_put(A); P.A = nullptr; return rc; }
    char* T = P.T = ctx->up_tmp.as<char>();
    P.ucd.resize(n); P.ust.resize(n);
    auto add_run = [](std::vector<CopyRun>& runs, const void* src, char* dst, size_t bytes) {
        if (!bytes) return;
        // back to back on both sides — or separated by the same few padding bytes on both sides (the field-major layout of floria_hip_pack_pileups_batch
        // pads every contig's bit arrays by a byte): one transfer
        const ptrdiff_t gs = runs.empty() ? -1 : (const char*)src - (runs.back().src + runs.back().bytes), gd = runs.empty() ? -2 : dst - (runs.back().dst + runs.back().bytes);
        if (gs == gd && gs >= 0 && gs <= 64) runs.back().bytes += (size_t)gs + bytes;
        else runs.push_back({(const char*)src, dst, bytes});
    };
    for (int kind = 0; kind < 3; ++kind)
        for (uint32_t i = 0; i < n; ++i) {
            const floria_pileup* p = &pileups[i];
            if (!p->n_reads) continue;
            const uint64_t nr = p->n_reads;
            if (kind == 0) add_run(P.small_runs, p->read_off, D + o_ro + 4 * (P.rp[i] + i), 4 * (nr + 1));
            else if (kind == 1) add_run(P.small_runs, p->first, D + o_first + 4 * P.rp[i], 4 * nr);
            else add_run(P.small_runs, p->last, D + o_last + 4 * P.rp[i], 4 * nr);
        }
    if (pk) for (uint32_t i = 0; i < n; ++i) if (pk[i].n_reads) add_run(P.small_runs, pk[i].bit_off, T + t_bo + 4 * (P.rp[i] + i), 4 * ((size_t)pk[i].n_reads + 1));
    auto a2_off = [&](uint32_t i) { return (size_t)(P.cp[i] / 4 + i); };       // (every contig's 2-bit array starts on its own byte)
    P.chunk_runs.assign(n_chunks, {});
    for (uint32_t g = 0; g < n_chunks; ++g)
        for (int kind = 0; kind < 4; ++kind)
            for (uint32_t i = P.chunk_first[g]; i < P.chunk_first[g + 1]; ++i) {
                const floria_pileup* p = &pileups[i];
                if (!p->n_reads) continue;
                const uint64_t nc = P.cp[i + 1] - P.cp[i];
                if (kind == 3) { if (p->set_order) add_run(P.chunk_runs[g], p->set_order, D + o_so + 4 * P.cp[i], 4 * nc); continue; }
                if (pk) {
                    if (kind == 0) add_run(P.chunk_runs[g], pk[i].present, T + t_pr + pb[i], pb[i + 1] - pb[i] - 1);
                    else if (kind == 1) add_run(P.chunk_runs[g], pk[i].allele2, T + t_a2 + a2_off(i), (nc + 3) / 4);
                    else add_run(P.chunk_runs[g], p->qual, T + t_q + P.cp[i], nc);
                    continue;
                }
                if (kind == 0) add_run(P.chunk_runs[g], p->snp, D + o_snp + 4 * P.cp[i], 4 * nc);
                else if (kind == 1) add_run(P.chunk_runs[g], p->allele, T + t_al + P.cp[i], nc);
                else add_run(P.chunk_runs[g], p->qual, T + t_q + P.cp[i], nc);
            }
    for (uint32_t i = 0; i < n; ++i) {
        fl::UploadContig& u = P.ucd[i];
        u.read_off = (const uint32_t*)(D + o_ro + 4 * (P.rp[i] + i)); u.first = (const uint32_t*)(D + o_first + 4 * P.rp[i]); u.last = (const uint32_t*)(D + o_last + 4 * P.rp[i]);
        u.snp = (const uint32_t*)(D + o_snp + 4 * P.cp[i]); u.allele = (const uint8_t*)(T + t_al + P.cp[i]); u.qual = (const uint8_t*)(T + t_q + P.cp[i]);
        u.cell_aw = (uint32_t*)(D + o_aw + 4 * P.cp[i]); u.tw = (uint64_t*)(D + o_tw + 16 * P.rp[i]); u.meta = (uint32_t*)(D + o_meta + 32 * P.rp[i]);
        u.n_reads = pileups[i].n_reads; u.n_cells = (uint32_t)(P.cp[i + 1] - P.cp[i]);
        P.ust[i] = fl::UploadStatus{~0ull, 0, 0, 0, 0};
    }
    P.so_dev.assign(n, nullptr);
    for (uint32_t i = 0; i < n; ++i) if (pileups[i].n_reads && pileups[i].set_order) P.so_dev[i] = (const uint32_t*)(D + o_so + 4 * P.cp[i]);
    if (pk) {
        P.upk.resize(n);
        for (uint32_t i = 0; i < n; ++i) {
            fl::PackedContig& q = P.upk[i];
            q.bit_off = (const uint32_t*)(T + t_bo + 4 * (P.rp[i] + i)); q.present = (const uint8_t*)(T + t_pr + pb[i]); q.allele2 = (const uint8_t*)(T + t_a2 + a2_off(i));
            q.snp = (uint32_t*)(D + o_snp + 4 * P.cp[i]); q.allele = (uint8_t*)(T + t_al + P.cp[i]);
            q.present_bytes = pk[i].n_reads ? ((uint64_t)pk[i].bit_off[pk[i].n_reads] + 7) / 8 : 0;
        }
    }
    P.all_pinned = true;
    auto pinned_run = [](const CopyRun& r) { return r.bytes < 4096 || (is_pinned(r.src) && is_pinned(r.src + r.bytes - 1)); };
    for (auto& r : P.small_runs) P.all_pinned = P.all_pinned && pinned_run(r);
    for (auto& v : P.chunk_runs) for (auto& r : v) P.all_pinned = P.all_pinned && pinned_run(r);
    return 0;
}

// the flatten kernel's tables (contig table, read prefix, fresh status words) — before any flatten launch
hipError_t issue_tables(floria_hip_ctx* ctx, UploadPlan& P, hipStream_t st) {
    hipError_t e = hipMemcpyAsync(P.T + P.t_cd, P.ucd.data(), sizeof(fl::UploadContig) * P.n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(P.T + P.t_rp, P.rp.data(), 8 * (P.n + 1), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(P.T + P.t_st, P.ust.data(), sizeof(fl::UploadStatus) * P.n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && P.packed) e = hipMemcpyAsync(P.T + P.t_pk, P.upk.data(), sizeof(fl::PackedContig) * P.n, hipMemcpyHostToDevice, st);
    return e;
}
// validate + flatten the reads of chunk g (its cells and the small arrays must have arrived on `st`)
hipError_t launch_flatten(floria_hip_ctx* ctx, UploadPlan& P, uint32_t g, hipStream_t st) {
    const uint32_t c0 = P.chunk_first[g], c1 = P.chunk_first[g + 1];
    const uint64_t nr = P.rp[c1] - P.rp[c0];
    if (!nr) return hipSuccess;
    fl::UploadArgs a{};
    a.contigs = (const fl::UploadContig*)(P.T + P.t_cd) + c0; a.read_prefix = (const uint64_t*)(P.T + P.t_rp) + c0; a.status = (fl::UploadStatus*)(P.T + P.t_st) + c0;
    a.w24 = ctx->d_w24.as<uint32_t>(); a.Rq1 = ctx->d_hash.as<uint64_t>(); a.Rq2 = ctx->d_hash.as<uint64_t>() + 2ull * ctx->hash_len;
    a.n_contigs = c1 - c0; a.read_base = P.rp[c0]; a.n_reads_total = nr;
    const unsigned wgs = (unsigned)((nr + fl::UP_READS_PER_WG - 1) / fl::UP_READS_PER_WG);
    if (P.packed) hipLaunchKernelGGL(fl::expand_kernel, dim3(wgs), dim3(64), 0, st, a, (const fl::PackedContig*)(P.T + P.t_pk) + c0);     // compact wire form -> CSR, then as usual
    hipLaunchKernelGGL(fl::flatten_kernel, dim3(wgs), dim3(64), 0, st, a);
    return hipGetLastError();
}
// status words -> error (if any), else the contig handles
int finish_upload(floria_hip_ctx* ctx, UploadPlan& P, floria_hip_contig** out) {
    const uint32_t n = P.n;
    for (uint32_t i = 0; i < n; ++i) if (P.ust[i].err != ~0ull) {
        const std::string r = std::to_string((unsigned long long)(P.ust[i].err >> 8)), ci = n > 1 ? " (contig " + std::to_string(i) + " of the batch)" : std::string();
        switch ((uint32_t)(P.ust[i].err & 0xff)) {
            case fl::UP_PACKED:        return fail(FLORIA_E_INVALID, "packed pileup: the presence bits of read " + r + " do not match its first / last / cell count" + ci);
            case fl::UP_NO_CELLS:      return fail(FLORIA_E_INVALID, "read " + r + " has no cells (or read_off is not monotone)" + ci);
            case fl::UP_FIRST_LAST:    return fail(FLORIA_E_INVALID, "first/last of read " + r + " do not match its cells" + ci);
            case fl::UP_ONE_BASED:     return fail(FLORIA_E_INVALID, "SNP positions are 1-based" + ci);
            case fl::UP_NOT_ASCENDING: return fail(FLORIA_E_INVALID, "cells of read " + r + " not strictly ascending" + ci);
            case fl::UP_ALLELE:        return fail(FLORIA_E_UNSUPPORTED, "allele index > 3 (read " + r + ")" + ci);
            default:                   return fail(FLORIA_E_INVALID, "reads not sorted by Frag::cmp at read " + r + ci);
        }
    }
    if (!out) return 0;
    for (uint32_t i = 0; i < n; ++i) {
        floria_hip_contig* c = new floria_hip_contig();
        c->ctx = ctx; c->arena = P.A; c->idx = i; c->n_reads = P.ucd[i].n_reads; c->n_cells = P.cp[i + 1] - P.cp[i];
        c->max_len = P.ust[i].max_len; c->n_alleles = P.ust[i].max_allele >= 2 ? 4 : 2; c->has_q0 = P.ust[i].has_q0 != 0;
        c->dev.read_off = P.ucd[i].read_off; c->dev.first = P.ucd[i].first; c->dev.last = P.ucd[i].last; c->dev.cell_snp = P.ucd[i].snp;
        c->dev.cell_aw = P.ucd[i].cell_aw; c->dev.tw = P.ucd[i].tw; c->dev.meta = P.ucd[i].meta; c->dev.n_reads = P.ucd[i].n_reads;
        c->dev.set_order = P.so_dev[i];
        out[i] = c;
    }
    P.A->refs = n;
    return 0;
}

}  // namespace

extern "C" {

// Upload a batch of contigs: plan the arena, DMA the raw arrays (consecutive contigs whose arrays are back to back in host
// memory travel as one transfer), validate + flatten on the device (upload_kernel.h), read back the per-contig status.
static int upload_batch_impl(floria_hip_ctx* ctx, const floria_pileup* pileups, const floria_pileup_packed* pk, uint32_t n, floria_hip_contig** out) {
    if (!ctx || !out || (n && !pileups && !pk)) return fail(FLORIA_E_INVALID, "null argument");
    for (uint32_t i = 0; i < n; ++i) out[i] = nullptr;
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(ctx->device));
    UploadPlan P;
    int rc = plan_upload(ctx, pileups, pk, n, 1, P);
    if (rc) return rc;
    EventTimer Tm(ctx->stream);
    const int th = Tm.begin(K_H2D);
    uint64_t pinned_b = 0, staged_b = 0;
    std::vector<CopyRun> runs = P.small_runs;
    runs.insert(runs.end(), P.chunk_runs[0].begin(), P.chunk_runs[0].end());
    rc = issue_copies(ctx, runs, &pinned_b, &staged_b);
    hipError_t e = rc ? hipSuccess : issue_tables(ctx, P, ctx->stream);
    Tm.end(th);
    if (!rc && e == hipSuccess) { const int tk = Tm.begin(K_SEL); e = launch_flatten(ctx, P, 0, ctx->stream); Tm.end(tk); }
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(P.ust.data(), P.T + P.t_st, sizeof(fl::UploadStatus) * n, hipMemcpyDeviceToHost, ctx->stream);
    if (!rc && e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (!rc && e != hipSuccess) rc = fail(FLORIA_E_DEVICE, std::string("contig upload: ") + hipGetErrorString(e));
    if (!rc) rc = finish_upload(ctx, P, out);
    if (rc) { sync_all(ctx); arena_put(P.A); return rc; }
    ctx->timing = floria_timing{};
    ctx->timing.h2d_ms = Tm.sum(K_H2D); ctx->timing.select_ms = Tm.sum(K_SEL); ctx->timing.total_ms = Tm.span();
    ctx->timing.upload_pinned_bytes = pinned_b; ctx->timing.upload_staged_bytes = staged_b;
    return 0;
}
int floria_hip_contig_upload_batch(floria_hip_ctx* ctx, const floria_pileup* pileups, uint32_t n, floria_hip_contig** out) {
    return upload_batch_impl(ctx, pileups, nullptr, n, out);
}
int floria_hip_contig_upload_batch_packed(floria_hip_ctx* ctx, const floria_pileup_packed* pileups, uint32_t n, floria_hip_contig** out) {
    return upload_batch_impl(ctx, nullptr, pileups, n, out);
}

// The compact wire form of CSR pileups (include/floria_hip.h: floria_pileup_packed), host side.  A batch is laid out FIELD BY FIELD (every
// contig's read_off, then every bit_off, ...) at exactly the per-contig offsets plan_upload gives the device copies, so that an upload of the whole
// batch is one transfer per field (and per chunk) instead of seven per contig.
namespace {
struct PackLayout { std::vector<uint64_t> rp, cp, pb; size_t o_ro, o_bo, o_fi, o_la, o_pr, o_a2, o_qu, total; };
bool pack_layout(const floria_pileup* in, uint32_t n, PackLayout& Y) {
    Y.rp.assign(n + 1, 0); Y.cp.assign(n + 1, 0); Y.pb.assign(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const floria_pileup* p = &in[i];
        const uint64_t R = p->n_reads;
        if (R && (!p->read_off || !p->first || !p->last || !p->snp || !p->allele || !p->qual)) return false;
        uint64_t bits = 0;
        // read_off sizes every buffer (through read_off[R]) AND bounds the packing loop below, which runs before any device-side validation: it must be a
        // CSR offset array — 0 first, strictly ascending (a read without cells is refused by the CSR upload as well)
        if (R && p->read_off[0] != 0) return false;
        for (uint64_t r = 0; r < R; ++r) {
            if (p->last[r] < p->first[r] || p->read_off[r + 1] <= p->read_off[r]) return false;
            bits += (uint64_t)p->last[r] - p->first[r] + 1;
        }
        if (bits >= (1ull << 32)) return false;
        Y.rp[i + 1] = Y.rp[i] + R; Y.cp[i + 1] = Y.cp[i] + (R ? p->read_off[R] : 0); Y.pb[i + 1] = Y.pb[i] + (bits + 7) / 8 + 1;
    }
    size_t cur = 0;
    auto seg = [&](uint64_t bytes) { const size_t o = cur; cur += (size_t)((bytes + 63) & ~(uint64_t)63); return o; };
    const uint64_t R = Y.rp[n], C = Y.cp[n];
    Y.o_ro = seg(4 * (R + n)); Y.o_bo = seg(4 * (R + n)); Y.o_fi = seg(4 * R); Y.o_la = seg(4 * R);
    Y.o_pr = seg(Y.pb[n] + 16); Y.o_a2 = seg(C / 4 + n + 16); Y.o_qu = seg(C + 16);
    Y.total = cur;
    return true;
}
}  // namespace
size_t floria_hip_pack_bytes_batch(const floria_pileup* in, uint32_t n) {
    PackLayout Y;
    if (!in || !n || !pack_layout(in, n, Y)) return 0;
    return Y.total;
}
int floria_hip_pack_pileups_batch(const floria_pileup* in, uint32_t n, void* buf, size_t buf_bytes, floria_pileup_packed* out) {
    if ((n && (!in || !out)) || (!buf && buf_bytes)) return fail(FLORIA_E_INVALID, "null argument");
    if (!n) return 0;
    PackLayout Y;
    if (!pack_layout(in, n, Y)) return fail(FLORIA_E_INVALID, "pileup cannot be packed (null field, read_off not a strictly ascending offset array from 0, last < first, or spans of 2^32 bits and more)");
    if (buf_bytes < Y.total) return fail(FLORIA_E_INVALID, "pack buffer too small (floria_hip_pack_bytes_batch)");
    char* B = (char*)buf;
    memset(B + Y.o_pr, 0, Y.pb[n] + 16); memset(B + Y.o_a2, 0, Y.cp[n] / 4 + n + 16);
    for (uint32_t i = 0; i < n; ++i) {
        const floria_pileup* p = &in[i];
        const uint64_t R = p->n_reads, C = R ? p->read_off[R] : 0;
        uint32_t* ro = (uint32_t*)(B + Y.o_ro + 4 * (Y.rp[i] + i)); uint32_t* bo = (uint32_t*)(B + Y.o_bo + 4 * (Y.rp[i] + i));
        uint32_t* fi = (uint32_t*)(B + Y.o_fi + 4 * Y.rp[i]);       uint32_t* la = (uint32_t*)(B + Y.o_la + 4 * Y.rp[i]);
        uint8_t* pr = (uint8_t*)(B + Y.o_pr + Y.pb[i]); uint8_t* a2 = (uint8_t*)(B + Y.o_a2 + Y.cp[i] / 4 + i); uint8_t* qu = (uint8_t*)(B + Y.o_qu + Y.cp[i]);
        uint64_t bit = 0;
        for (uint64_t r = 0; r < R; ++r) {
            ro[r] = p->read_off[r]; fi[r] = p->first[r]; la[r] = p->last[r]; bo[r] = (uint32_t)bit;
            const uint32_t F = p->first[r], span = p->last[r] - F + 1;
            for (uint32_t c = p->read_off[r]; c < p->read_off[r + 1]; ++c) {
                const uint32_t sp = p->snp[c];
                if (sp < F || sp - F >= span || p->allele[c] > 3) {
                    const std::string where = " (read " + std::to_string((unsigned long long)r) + (n > 1 ? ", contig " + std::to_string(i) + " of the batch)" : ")");
                    if (p->allele[c] > 3 && !(sp < F || sp - F >= span)) return fail(FLORIA_E_UNSUPPORTED, "allele index > 3" + where);
                    return fail(FLORIA_E_INVALID, "cell outside [first, last]" + where);
                }
                const uint64_t bi = bit + (sp - F);
                pr[bi >> 3] |= (uint8_t)(1u << (bi & 7));
                a2[c >> 2] |= (uint8_t)(p->allele[c] << (2 * (c & 3)));
            }
            bit += span;
        }
        ro[R] = (uint32_t)C; bo[R] = (uint32_t)bit;
        if (C) memcpy(qu, p->qual, C);
        out[i] = floria_pileup_packed{ro, fi, la, bo, pr, a2, qu, p->n_reads, p->set_order};      // (the optional set order is not repacked: the packed pileup points at the caller's array)
    }
    return 0;
}
size_t floria_hip_pack_bytes(const floria_pileup* in) { return floria_hip_pack_bytes_batch(in, in ? 1 : 0); }
int floria_hip_pack_pileup(const floria_pileup* in, void* buf, size_t buf_bytes, floria_pileup_packed* out) {
    if (!in || !out) return fail(FLORIA_E_INVALID, "null argument");
    return floria_hip_pack_pileups_batch(in, 1, buf, buf_bytes, out);
}



int floria_hip_contig_upload(floria_hip_ctx* ctx, const floria_pileup* p, floria_hip_contig** out) {
    if (!ctx || !out || !p) return fail(FLORIA_E_INVALID, p ? "null argument" : "null pileup");
    return floria_hip_contig_upload_batch(ctx, p, 1, out);
}
void floria_hip_contig_free(floria_hip_contig* c) {
    if (!c) return;
    if (c->arena && --c->arena->refs == 0) arena_put(c->arena);
    delete c;
}

// Diagnostic: copy one resident array of a contig back to the host (tests compare the device-side flatten with the formulas).
int floria_hip_contig_download(const floria_hip_contig* c, int field, void* dst, size_t bytes) {
    if (!c || (!dst && bytes)) return fail(FLORIA_E_INVALID, "null argument");
    const void* src = nullptr;
    size_t have = 0;
    switch (field) {
        case FLORIA_FIELD_READ_OFF: src = c->dev.read_off; have = 4ull * (c->n_reads + (c->n_reads ? 1 : 0)); break;
        case FLORIA_FIELD_FIRST:    src = c->dev.first;    have = 4ull * c->n_reads; break;
        case FLORIA_FIELD_LAST:     src = c->dev.last;     have = 4ull * c->n_reads; break;
        case FLORIA_FIELD_SNP:      src = c->dev.cell_snp; have = 4ull * c->n_cells; break;
        case FLORIA_FIELD_CELL_AW:  src = c->dev.cell_aw;  have = 4ull * c->n_cells; break;
        case FLORIA_FIELD_TW:       src = c->dev.tw;       have = 16ull * c->n_reads; break;
        case FLORIA_FIELD_META:     src = c->dev.meta;     have = 32ull * c->n_reads; break;
        default: return fail(FLORIA_E_INVALID, "unknown field");
    }
    if (bytes > have) return fail(FLORIA_E_INVALID, "field is smaller than the requested size");
    HIPCHK(hipSetDevice(c->ctx->device));
    if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"

namespace {

Arena* arena_get(floria_hip_ctx* ctx, size_t bytes) {
    Arena* a = nullptr;
    // best fit among the cached arenas, else grow the largest, else a new one
    size_t best = (size_t)-1;
    for (size_t i = 0; i < ctx->arena_cache.size(); ++i) if (ctx->arena_cache[i]->buf.cap >= bytes && (best == (size_t)-1 || ctx->arena_cache[i]->buf.cap < ctx->arena_cache[best]->buf.cap)) best = i;
    if (best == (size_t)-1 && !ctx->arena_cache.empty()) { best = 0; for (size_t i = 1; i < ctx->arena_cache.size(); ++i) if (ctx->arena_cache[i]->buf.cap > ctx->arena_cache[best]->buf.cap) best = i; }
    if (best != (size_t)-1) { a = ctx->arena_cache[best]; ctx->arena_cache.erase(ctx->arena_cache.begin() + best); }
    else { a = new Arena(); a->ctx = ctx; }
    if (a->buf.ensure(bytes)) { a->buf.release(); delete a; return nullptr; }
    a->refs = 0; a->host_meta = false;
    return a;
}
void arena_put(Arena* a) {
    floria_hip_ctx* ctx = a->ctx;
    a->h_first.clear(); a->h_last.clear(); a->host_meta = false; a->refs = 0;
    if (ctx->arena_cache.size() < 4) ctx->arena_cache.push_back(a);
    else { a->buf.release(); delete a; }
}

int issue_copies(floria_hip_ctx* ctx, std::vector<CopyRun>& runs, uint64_t* pinned_bytes, uint64_t* staged_bytes) {
    std::vector<CopyRun> staged;          // pageable runs, cut into SEG pieces
    for (const CopyRun& r : runs) {
        const bool pin = r.bytes >= 4096 && is_pinned(r.src) && is_pinned(r.src + r.bytes - 1);
        if (pin) {
            *pinned_bytes += r.bytes;
            HIPCHK(hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyHostToDevice, ctx->stream));
        } else {
            *staged_bytes += r.bytes;
            for (size_t o = 0; o < r.bytes; o += StagePool::SEG) staged.push_back({r.src + o, r.dst + o, std::min(StagePool::SEG, r.bytes - o)});
        }
    }
    if (staged.empty()) return 0;
    size_t tot = 0;
    for (auto& r : staged) tot += r.bytes;
    if (tot < (1u << 20)) {                  // small: let the runtime stage it
        for (auto& r : staged) HIPCHK(hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyHostToDevice, ctx->stream));
        return 0;
    }
    if (int rc = ctx->stage.init()) return rc;
    StagePool& SP = ctx->stage;
    if (!ctx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->ev_rids) HIPCHK(hipEventCreateWithFlags(&ctx->ev_rids, hipEventDisableTiming));
    // Worker threads only memcpy (pageable source -> pinned segment); THIS thread makes every HIP call: it issues the DMA of
    // segment k as soon as it is filled — alternating between two streams, so the completion latency of one copy hides behind
    // the next — and hands a staging buffer back to the fillers once the DMA that read it has completed.
    const uint32_t nseg = (uint32_t)staged.size();
    const uint32_t nthreads = std::max(1u, std::min<uint32_t>({ctx->stage_threads, nseg, 16u}));
    std::vector<std::atomic<int>> filled(nseg);
    for (auto& f : filled) f.store(0);
    std::atomic<uint32_t> next{0}, freed{StagePool::NBUF};     // segments < freed may be filled (their buffer is free)
    std::atomic<int> failed{0};
    auto worker = [&]() {
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= nseg) return;
            while (freed.load(std::memory_order_acquire) <= k) { if (failed.load()) return; std::this_thread::yield(); }
            memcpy(SP.buf[k % StagePool::NBUF], staged[k].src, staged[k].bytes);
            filled[k].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nthreads; ++t) th.emplace_back(worker);
    hipStream_t cs[2] = {ctx->stream, ctx->copy_stream};
    hipError_t e = hipEventRecord(ctx->ev_rids, ctx->stream);                       // the second stream starts after what the first has queued
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, ctx->ev_rids, 0);
    uint32_t done = 0;                                                              // DMAs known complete
    for (uint32_t k = 0; k < nseg && e == hipSuccess; ++k) {
        while (!filled[k].load(std::memory_order_acquire)) {
            // while waiting for the fillers, retire completed DMAs so that their buffers can be refilled
            if (done < k && hipEventQuery(SP.ev[done % StagePool::NBUF]) == hipSuccess) { ++done; freed.store(done + StagePool::NBUF, std::memory_order_release); }
            else std::this_thread::yield();
        }
        e = hipMemcpyAsync(staged[k].dst, SP.buf[k % StagePool::NBUF], staged[k].bytes, hipMemcpyHostToDevice, cs[k & 1]);
        if (e == hipSuccess) e = hipEventRecord(SP.ev[k % StagePool::NBUF], cs[k & 1]);
        // the fillers may be blocked on a buffer whose DMA is still running: wait for the oldest one when the ring is exhausted
        while (e == hipSuccess && done + StagePool::NBUF <= k + 1 && k + 1 < nseg && freed.load() <= k + 1) {
            e = hipEventSynchronize(SP.ev[done % StagePool::NBUF]);
            ++done; freed.store(done + StagePool::NBUF, std::memory_order_release);
        }
    }
    if (e != hipSuccess) failed.store(1);
    freed.store(0xffffffffu);
    for (auto& t : th) t.join();
    (void)hipGetLastError();                                                        // hipEventQuery's hipErrorNotReady is not an error
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_rids, ctx->copy_stream);        // the main stream continues after both
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ev_rids, 0);
    if (e != hipSuccess) { sync_all(ctx); return fail(FLORIA_E_DEVICE, std::string("staged upload failed: ") + hipGetErrorString(e)); }
    return 0;
}

}  // namespace

extern "C" {

// ---- S1 --------------------------------------------------------------------------------------------------------
}  // extern "C"

namespace {

// Reference-arithmetic mode: the cells of every read of the given contigs in the iteration order of its position set (cell_order_kernel), for
// the call in flight: ctx->cur_ord[ctx->cur_ord_off[c] + read_off[r] + x] = {SNP, allele << 28 | weight} of the x-th position of read r's set.
// Computed by two launches (round 5: ≈ 130 ms for the 331 M cells of config 4 with the one-thread-per-read table emulation alone) and kept until the context uploads contigs again: S2 after S1, or the next S1 call over the same resident batch, reuses it.
struct OrderPlan {                        // host side of the cell orders of one set of contigs
    std::vector<uint64_t> pre;            // reads before contig c [n+1] | cells before contig c [n] | [1 pad] | status word of the host-given set orders
    uint64_t cells = 0, R_all = 0, sig = 0;
    uint32_t n_contigs = 0;
};
void order_plan(const std::vector<fl::ContigDev>& cdev, const std::vector<uint64_t>& n_cells, OrderPlan& O) {
    const uint32_t n_contigs = O.n_contigs = (uint32_t)cdev.size();
    O.pre.assign(2 * (size_t)n_contigs + 3, 0);
    uint64_t cells = 0;
    for (uint32_t i = 0; i < n_contigs; ++i) { O.pre[i + 1] = O.pre[i] + cdev[i].n_reads; O.pre[n_contigs + 1 + i] = cells; cells += n_cells[i]; }
    O.cells = cells; O.R_all = O.pre[n_contigs];
    uint64_t sig = 1469598103934665603ull;                                    // FNV-1a over what identifies the contigs: their device arrays and sizes
    auto mix = [&](uint64_t v) { for (int b = 0; b < 8; ++b) { sig ^= (v >> (8 * b)) & 0xff; sig *= 1099511628211ull; } };
    for (uint32_t i = 0; i < n_contigs; ++i) { mix((uint64_t)(uintptr_t)cdev[i].cell_snp); mix((uint64_t)(uintptr_t)cdev[i].read_off); mix(cdev[i].n_reads); mix(n_cells[i]); mix((uint64_t)(uintptr_t)cdev[i].set_order); }
    O.sig = sig;
}
// the two launches for the reads [r0, r1) of the contigs [c0, c1) on `st` (their cells must have been flattened on that stream, or before it)
int order_launch(floria_hip_ctx* ctx, const OrderPlan& O, const fl::ContigDev* d_contigs, uint32_t c0, uint32_t c1, uint64_t* todo, uint8_t* scratch, uint64_t scratch_bytes, uint32_t len_max,
                 const uint32_t* status_max_len, hipStream_t st) {
    const uint64_t r0 = O.pre[c0], nr = O.pre[c1] - r0;
    if (!nr) return 0;
    fl::CellOrderArgs oa{};
    oa.todo = todo; oa.replay_all = ctx->knobs.arith_replay;
    oa.bad = ctx->arith_tab.as<uint64_t>() + 2 * (size_t)O.n_contigs + 2;
    oa.contigs = d_contigs + c0; oa.read_prefix = ctx->arith_tab.as<uint64_t>() + c0; oa.cell_prefix = ctx->arith_tab.as<uint64_t>() + O.n_contigs + 1 + c0;
    oa.n_contigs = c1 - c0; oa.n_reads = nr; oa.read_base = r0; oa.ord = ctx->arith_ord.as<uint2>(); oa.scratch = scratch; oa.scratch_bytes = scratch_bytes;
    oa.len_max = len_max; oa.status_max_len = status_max_len; oa.status_stride = (uint32_t)(sizeof(fl::UploadStatus) / 4);
    uint64_t nth = std::min<uint64_t>((nr + 255) & ~255ull, 131072);
    if (len_max) {
        oa.ctrl_bytes = fl::fx_ctrl_bytes(len_max); oa.slot_bytes = fl::fx_slot_bytes(len_max);
        nth = std::max<uint64_t>(256, std::min<uint64_t>(nth, (scratch_bytes / (3 * (oa.ctrl_bytes + oa.slot_bytes))) & ~255ull));
    } else nth = std::min<uint64_t>(nth, 16384);
    // a wavefront per read where the home-bucket rule applies or the host gave the order (arith_kernel.h), then one thread per read for the rest (none on BASELINE's configs)
    hipLaunchKernelGGL(fl::cell_order_direct_kernel, dim3((uint32_t)std::min<uint64_t>((nr + 3) / 4, (uint64_t)ctx->n_cu * 16)), dim3(256), 0, st, oa);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(fl::cell_order_kernel, dim3((uint32_t)(nth / 256)), dim3(256), 0, st, oa);
    HIPCHK(hipGetLastError());
    return 0;
}
int order_bad_read(const OrderPlan& O, uint64_t bad) {
    if (!bad) return 0;
    uint32_t ci = 0;
    const uint64_t gr = ~bad;
    while (ci + 1 < O.n_contigs && O.pre[ci + 1] <= gr) ++ci;
    return fail(FLORIA_E_INVALID, "set_order of read " + std::to_string((unsigned long long)(gr - O.pre[ci])) + (O.n_contigs > 1 ? " (contig " + std::to_string(ci) + " of the batch)" : std::string())
                                  + " is not a permutation of the indices of its cells");
}
int cell_orders(floria_hip_ctx* ctx, const fl::ContigDev* d_contigs, const std::vector<fl::ContigDev>& cdev, const std::vector<uint64_t>& n_cells, uint32_t len_max) {
    const uint32_t n_contigs = (uint32_t)cdev.size();
    OrderPlan O;
    order_plan(cdev, n_cells, O);
    if (ctx->ord_epoch == ctx->upload_epoch && ctx->ord_sig == O.sig && ctx->arith_ord.p && ctx->arith_tab.p) {
        ctx->cur_ord = ctx->arith_ord.as<uint2>(); ctx->cur_ord_off = ctx->arith_tab.as<uint64_t>() + n_contigs + 1;
        return 0;
    }
    int rc = ctx->arith_tab.ensure(O.pre.size() * 8); if (rc) return rc;
    rc = ctx->arith_ord.ensure(8 * O.cells + 16); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(ctx->arith_tab.p, O.pre.data(), O.pre.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));                                // (`pre` is pageable and local)
    if (O.R_all) {
        const uint64_t tb = fl::fx_ctrl_bytes(std::max(1u, len_max)) + fl::fx_slot_bytes(std::max(1u, len_max));
        uint64_t nth = std::min<uint64_t>((O.R_all + 255) & ~255ull, 131072);
        nth = std::max<uint64_t>(256, std::min<uint64_t>(nth, ((2ull << 30) / (3 * tb)) & ~255ull));
        const uint64_t todo_bytes = (8 * (O.R_all + 1) + 255) & ~255ull;          // [count | reads for the table emulation] in front of the emulation's scratch
        rc = ctx->arith_scr.ensure(todo_bytes + 3 * tb * nth); if (rc) return rc;
        HIPCHK(hipMemsetAsync(ctx->arith_scr.p, 0, 8, ctx->stream));
        rc = order_launch(ctx, O, d_contigs, 0, n_contigs, ctx->arith_scr.as<uint64_t>(), ctx->arith_scr.as<uint8_t>() + todo_bytes, 3 * tb * nth, std::max(1u, len_max), nullptr, ctx->stream);
        if (rc) return rc;
        bool any_given = false;
        for (uint32_t i = 0; i < n_contigs; ++i) any_given = any_given || cdev[i].set_order != nullptr;
        if (any_given) {                                                       // a host-given order that is not a permutation of a read's cells: refuse the call
            uint64_t bad = 0;
            HIPCHK(hipMemcpyAsync(&bad, ctx->arith_tab.as<uint64_t>() + 2 * (size_t)n_contigs + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (int brc = order_bad_read(O, bad)) return brc;
        }
    }
    ctx->cur_ord = ctx->arith_ord.as<uint2>(); ctx->cur_ord_off = ctx->arith_tab.as<uint64_t>() + n_contigs + 1;
    ctx->ord_epoch = ctx->upload_epoch; ctx->ord_sig = O.sig;
    return 0;
}

// What S1 needs to know about its contigs.  With `chunk_ev` the cell arrays of chunk g (contig_chunk[ci] == g) are still on
// the wire: they are complete once chunk_ev[g] has fired, and the blocks of chunk g form job group g whose stream waits for it
// (read_off / first / last of every contig are already ordered before the context's main stream).
struct S1Contigs {
    std::vector<fl::ContigDev> cdev;
    std::vector<uint64_t> n_cells;     // per contig (reference-arithmetic mode: the layout of the cell orders)
    uint32_t len_max = 1, nall = 2;
    bool any_q0 = false;
    const uint32_t* contig_chunk = nullptr;
    uint32_t n_chunks = 0;
    hipEvent_t* chunk_ev = nullptr;
    bool orders_in_flight = false; // reference-arithmetic mode, pipelined upload: every chunk's cell orders are computed behind its flatten launch, ahead of its event
    uint32_t chunk_groups = 0;     // job groups the chunks are merged into (0 = one per chunk)
    bool ends_apart = false;       // with three groups: first chunk | middle chunks | last chunk
};

// host mailbox -> device array, by a kernel on the call's own stream (no copy engine involved)
__global__ void box_copy_kernel(uint32_t* dst, const uint32_t* src, uint32_t n_words) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

struct Trace {
    bool on; std::chrono::steady_clock::time_point t0;
    explicit Trace(bool o) : on(o), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) const { if (on) fprintf(stderr, "[floria_hip trace] %8.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what); }
};

int s1_core(floria_hip_ctx* ctx, const S1Contigs& SC, const uint32_t* blk_contig, const uint32_t* blk_start, const uint32_t* blk_end,
            uint32_t n_blocks, const floria_params* prm, floria_block_result** out) {
    const uint32_t n_contigs = (uint32_t)SC.cdev.size();
    const Trace TR(ctx->knobs.trace);
    TR.mark("s1_core enter");
    if (prm->max_ploidy < 1 || prm->max_ploidy > FLORIA_MAX_PLOIDY) return fail(FLORIA_E_INVALID, "max_ploidy must be in 1..16");
    if (prm->beam < 1) return fail(FLORIA_E_INVALID, "beam (max_number_solns) must be >= 1");
    if (!(prm->epsilon > 0.0 && prm->epsilon < 1.0)) return fail(FLORIA_E_INVALID, "epsilon must be in (0,1)");
    ctx->timing = floria_timing{};
    ctx->batch_token = 0;
    const uint32_t P = prm->max_ploidy;

    // ---- block read lists: find_reads_in_interval on the device (blocks_kernel.h) -------------------------------
    std::vector<uint32_t> bc(n_blocks, 0);
    uint32_t n_max = 1, span_max = 1;
    const uint32_t len_max = SC.len_max, nall = SC.nall;
    const bool any_q0 = SC.any_q0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t ci = blk_contig ? blk_contig[b] : 0;
        if (ci >= n_contigs) return fail(FLORIA_E_INVALID, "blk_contig out of range");
        bc[b] = ci;
    }
    const std::vector<fl::ContigDev>& cdev = SC.cdev;
    struct Seg { size_t off, bytes; };
    size_t cursor = 0;
    auto seg = [&](size_t bytes) { Seg s{cursor, bytes}; cursor += (bytes + 255) & ~(size_t)255; return s; };
    const Seg s_cdev = seg(sizeof(fl::ContigDev) * std::max(1u, n_contigs)), s_bc = seg(4ull * n_blocks + 4), s_bs = seg(4ull * n_blocks + 4),
              s_be = seg(4ull * n_blocks + 4), s_p0 = seg(4ull * n_blocks + 4), s_sp = seg(4ull * n_blocks + 4), s_cnt = seg(4ull * n_blocks + 4),
              s_bytes = seg(8ull * n_blocks + 8), s_roff = seg(8ull * (n_blocks + 1));
    int rc = ctx->misc0.ensure(cursor + 256); if (rc) return rc;
    char* M0 = ctx->misc0.as<char>();
    EventTimer T(ctx->stream);
    // the call's small transfers go through the context's host mailbox (HostBox): up by a copy kernel on the main stream, down by the kernel's own stores
    static_assert(sizeof(fl::ContigDev) % 4 == 0, "box_copy_kernel moves 4-byte words");
    size_t box_cur = 0;
    auto box_take = [&](size_t bytes) { const size_t o = box_cur; box_cur += (bytes + 63) & ~(size_t)63; return o; };
    const size_t bx_cdev = box_take(sizeof(fl::ContigDev) * n_contigs), bx_bc = box_take(4ull * n_blocks), bx_bs = box_take(4ull * n_blocks), bx_be = box_take(4ull * n_blocks),
                 bx_cnt = box_take(4ull * n_blocks), bx_span = box_take(4ull * n_blocks), bx_bytes = box_take(8ull * n_blocks), bx_roff = box_take(8ull * (n_blocks + 1)),
                 bx_jobs = box_take(4ull * n_blocks);
    rc = ctx->box.ensure(box_cur + 64); if (rc) return rc;
    auto box_up = [&](void* dst, const void* src, size_t bytes, size_t at) -> hipError_t {
        if (!bytes) return hipSuccess;
        memcpy(ctx->box.h + at, src, bytes);
        const uint32_t words = (uint32_t)(bytes / 4);
        hipLaunchKernelGGL(box_copy_kernel, dim3(std::max(1u, std::min(256u, (words + 255) / 256))), dim3(256), 0, ctx->stream, (uint32_t*)dst, (const uint32_t*)(ctx->box.d + at), words);
        return hipGetLastError();
    };
    int th = T.begin(K_H2D);
    HIPCHK(box_up(M0 + s_cdev.off, cdev.data(), sizeof(fl::ContigDev) * n_contigs, bx_cdev));
    if (n_blocks) {
        HIPCHK(box_up(M0 + s_bc.off, bc.data(), 4ull * n_blocks, bx_bc));
        HIPCHK(box_up(M0 + s_bs.off, blk_start, 4ull * n_blocks, bx_bs));
        HIPCHK(box_up(M0 + s_be.off, blk_end, 4ull * n_blocks, bx_be));
    }
    T.end(th);
    fl::ScanArgs sa{};
    sa.contigs = (const fl::ContigDev*)(M0 + s_cdev.off);
    sa.blk_contig = (const uint32_t*)(M0 + s_bc.off); sa.blk_start = (const uint32_t*)(M0 + s_bs.off); sa.blk_end = (const uint32_t*)(M0 + s_be.off);
    sa.n_blocks = n_blocks; sa.max_ploidy = P;
    sa.cnt = (uint32_t*)(M0 + s_cnt.off); sa.pos0 = (uint32_t*)(M0 + s_p0.off); sa.span = (uint32_t*)(M0 + s_sp.off); sa.bytes = (uint64_t*)(M0 + s_bytes.off);
    sa.h_cnt = (uint32_t*)(ctx->box.d + bx_cnt); sa.h_span = (uint32_t*)(ctx->box.d + bx_span); sa.h_bytes = (uint64_t*)(ctx->box.d + bx_bytes);
    std::vector<uint32_t> cnt(n_blocks, 0), span(n_blocks, 0);
    std::vector<uint64_t> blk_bytes(n_blocks, 0), roff(n_blocks + 1, 0);
    if (n_blocks) {
        int tk = T.begin(K_SEL);
        hipLaunchKernelGGL(fl::block_reads_kernel<false>, dim3(n_blocks), dim3(64), 0, ctx->stream, sa);
        T.end(tk);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));           // (the kernel wrote its counts into the mailbox)
        memcpy(cnt.data(), ctx->box.h + bx_cnt, 4ull * n_blocks);
        memcpy(span.data(), ctx->box.h + bx_span, 4ull * n_blocks);
        memcpy(blk_bytes.data(), ctx->box.h + bx_bytes, 8ull * n_blocks);
    }
    TR.mark("block counts on the host");
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
    TR.mark("jobs sorted");
    // job groups (longest-first inside each group).  Resident inputs: dealt round-robin so every group sees the same size mix.
    // Chunked inputs (cells still arriving): group g = the blocks of chunk g, whose stream waits for the chunk's event.
    const bool chunked = SC.chunk_ev != nullptr && SC.n_chunks > 1;
    // (measured, config 4, resident: one group wins while a stage has less than three rounds of jobs — 750 / 1000 / 1500 contigs: 55 / 61.5 / 80.5 ms against
    // 58.7 / 68 / 82 ms with two groups — two groups win from there: 2000 contigs 96.4 against 99.4 ms.  End of round 4 (profiles/r04_groups_ab.txt), ms per
    // step with 1 / 2 / 3 / 4 groups: 7 249 blocks 51.6 / 55.2 / 51.4 / -, 10 875 blocks 67.8 / 66.7 / 65.3 / -, 14 503 blocks 81.6 / 78.8 / 75.9 or 79.3 / 78.4-79.1:
    // three groups are 3.7 % faster in five runs of seven and level in the other two (the three chains interleave in one of two patterns); the pipelined call,
    // which has its own three chunk groups, does not change.  Round 5 (profiles/r05_groups_ab.txt; box A, three interleaved runs): 2 groups 79.5 / 79.6 / 80.2,
    // 3 groups 77.3 / 76.1 / 77.2, 4 groups with half-size grids 78.9 / 79.3 / 79.2, an equal share of the wave slots per group (3 x 1 366) 81.8-82.2; box B, in the
    // default bench flow and resident-only: 2 groups 79.9-80.3 (six runs), 3 groups 80.3 / 80.7 / 79.8 / 79.8 / 77.5 / 78.3; 1500 contigs 64.5 against 67.8 (2) and
    // 68.6 (1).  Three groups land on the level of two or 3 % below it, never above by more than the noise -> three from 40 x CUs blocks on (round 4 kept two for
    // the per-launch roofline fraction, which is not what a step costs).  Unequal groups (dealing weights 10:8:6 .. 14:8:2) do not pin the good pattern (78.4-80.9).
    uint32_t G = ctx->knobs.groups ? ctx->knobs.groups : (jobs.size() >= (size_t)ctx->n_cu * 40 ? 3 : 1);
    G = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(G, floria_hip_ctx::MAX_GROUPS), (uint32_t)(jobs.size() / 1024)));
    if (ctx->hw_queues < 5 && !ctx->knobs.groups) G = std::min<uint32_t>(G, 2);
    // chunked: consecutive chunks may share a job group (SC.chunk_groups), which then starts when its LAST chunk has landed
    std::vector<uint32_t> chunk_group;
    hipEvent_t group_ev[floria_hip_ctx::MAX_GROUPS] = {};
    if (chunked) {
        const uint32_t nc = std::min<uint32_t>(SC.n_chunks, floria_hip_ctx::MAX_GROUPS);
        G = SC.chunk_groups ? std::min<uint32_t>(SC.chunk_groups, nc) : nc;
        if (ctx->hw_queues < 5) G = std::min<uint32_t>(G, 2);          // (groups on shared hardware queues serialise: two at most)
        chunk_group.resize(SC.n_chunks);
        for (uint32_t c = 0; c < SC.n_chunks; ++c) {
            const uint32_t cc = std::min(c, nc - 1);
            // ends_apart: the first chunk is a group of its own (it starts while the rest is still arriving and is too small to own every wave slot), so is the last
            chunk_group[c] = SC.ends_apart && G == 3 && nc >= 3 ? (cc == 0 ? 0u : cc == nc - 1 ? 2u : 1u) : std::min<uint32_t>((uint32_t)((uint64_t)cc * G / nc), G - 1);
            group_ev[chunk_group[c]] = SC.chunk_ev[cc];
        }
    }
    std::vector<uint32_t> group_off(G + 1, 0);
    if (chunked) {
        std::vector<uint32_t> dealt; dealt.reserve(jobs.size());
        for (uint32_t g = 0; g < G; ++g) { for (uint32_t b : jobs) if (chunk_group[SC.contig_chunk[bc[b]]] == g) dealt.push_back(b); group_off[g + 1] = (uint32_t)dealt.size(); }
        jobs.swap(dealt);
    } else if (G > 1) {
        std::vector<uint32_t> dealt; dealt.reserve(jobs.size());
        for (uint32_t g = 0; g < G; ++g) { for (size_t j = g; j < jobs.size(); j += G) dealt.push_back(jobs[j]); group_off[g + 1] = (uint32_t)dealt.size(); }
        jobs.swap(dealt);
    } else group_off[1] = (uint32_t)jobs.size();
    // ploidy stages (run_phase): one ploidy per stage unless the batch is too small to fill the chip with (block, ploidy) jobs
    std::vector<std::vector<uint32_t>> stages;
    {
        int spec = ctx->knobs.speculate;
        const bool slab_path = P * prm->beam <= 63 && !ctx->knobs.beam_path;          // the wide-beam kernels own whole CUs: nothing to gain there
        // (measured on config-4 shards, resident, with the gated high ploidies and the early stop-rule flags of run_phase; ms for speculate = 0 / 1 / 2:
        //   915 blocks 37.6 / 21.3 / 24.3   1822: 40.4 / 24.8 / 27.1   2724: 44.1 / 32.2 / 32.7   3636: 48.0 / 46.5 / 36.8   4543: 50.9 / 57.7 / 44.9
        //   5455: 55.5 / 69.1 / 50.9   6348: 58.7 / 80.8 / 56.9   7249: 61.6 / - / 62.5   14503: 98.6 / 141-155 / 115.5)
        // round 4, faster beam steps (scripts/shard_sweep.py, profiles/r04_shard_sweep.txt; ms for speculate = 0 / 1 / 2):
        //   915: 32.7 / 19.8 / 26.6   1822: 36.1 / 23.0 / 29.5   2724: 40.6 / 31.3 / 34.9   3636: 43.2 / 36.1 / 39.5   4543: 46.6 / 42.2 / 44.2
        //   5455: 50.4 / 50.3 / 48.9   6348: 52.7 / 56.0 / 53.3   7249: 56.8 / 63.7 / 58.1
        // -> every ploidy at once up to 18 x CUs blocks, {1,2,3} then {4..P} up to 23 x CUs, one ploidy per stage above
        if (spec < 0) spec = !(slab_path && P >= 3) ? 0 : jobs.size() <= (size_t)ctx->n_cu * 18 ? 1 : (jobs.size() <= (size_t)ctx->n_cu * 23 && P >= 4) ? 2 : 0;
        if (P * G > floria_hip_ctx::MAX_LANES || P < 3) spec = 0;
        // a speculative stage's lanes wait on each other's events: with more lanes than hardware queues (GPU_MAX_HW_QUEUES, 12 in our hosts, minus the main,
        // copy and flatten streams) they share queues and the gates serialise the stage (measured: 250 contigs in 3 / 4 / 5 chunk groups 94 / 141 / 187 ms against 32)
        // (floria_hip_create measured how many streams really run side by side: hw_queues; main, copy and flatten streams take up to three of them)
        // With GPU_MAX_HW_QUEUES=12 the probe sees 6 spinning kernels side by side on an MI355X, and stages of up to 10 lanes measured fine (round 2: 250
        // contigs 24-26 ms against 38 without); with the runtime's default of 4 queues it sees 4 or fewer, and there the gated lanes serialise (3-8x).
        const uint32_t lanes_ok = ctx->hw_queues >= 5 ? 10u : (ctx->hw_queues > 1 ? ctx->hw_queues - 1 : 1u);
        if (ctx->knobs.speculate < 0 && spec) { const uint32_t w = spec == 1 ? P : std::max<uint32_t>(std::min(3u, P), P > 3 ? P - 3 : 0); if (w * G > lanes_ok) spec = 0; }
        if (spec == 3 && P >= 5) {          // {1}{2}{3}{4..P}: only the ploidies few blocks reach run side by side
            for (uint32_t p = 1; p <= 3; ++p) stages.push_back({p});
            stages.emplace_back(); for (uint32_t p = 4; p <= P; ++p) stages.back().push_back(p);
        }
        else if (spec == 1) { stages.emplace_back(); for (uint32_t p = 1; p <= P; ++p) stages.back().push_back(p); }
        else if (spec == 2) { stages.emplace_back(); for (uint32_t p = 1; p <= std::min(3u, P); ++p) stages.back().push_back(p);
                              if (P > 3) { stages.emplace_back(); for (uint32_t p = 4; p <= P; ++p) stages.back().push_back(p); } }
        else for (uint32_t p = 1; p <= P; ++p) stages.push_back({p});
        if (spec == 3 && P < 5) { stages.clear(); for (uint32_t p = 1; p <= P; ++p) stages.push_back({p}); }
    }
    uint32_t stage_w = 1;
    for (auto& st : stages) stage_w = std::max<uint32_t>(stage_w, (uint32_t)st.size());

    rc = ensure_binom(ctx, prm->epsilon, len_max); if (rc) return rc;
    if (span_max > fl::HASH_M) return fail(FLORIA_E_UNSUPPORTED, "a block's reads span more than 65536 SNPs");
    rc = ensure_hash(ctx, span_max * nall); if (rc) return rc;

    // ---- device staging of the per-call arrays ----------------------------------------------------------------------
    cursor = 0;
    const Seg s_rids = seg(4ull * tot + 4), s_jobs = seg(4ull * jobs.size() + 4), s_planes = seg((uint64_t)P * tot + 16), s_bpart = seg((uint64_t)(stage_w + 1) * (tot + 16)),        // (+1: the plane of a last-stage beam launch that runs beside the optimise launch below it)
              s_margin = seg(8ull * n_blocks * P + 16),
              // zero-initialised, contiguous (ONE memset): partition output, mec / num_alleles / iters, stop-rule state, queue counters, diagnostics
              s_out = seg(tot + 16), s_mec = seg(8ull * n_blocks * P + 8), s_na = seg(8ull * n_blocks * P + 8), s_it = seg(4ull * n_blocks * P + 4),
              s_done = seg(n_blocks + 4), s_best = seg(4ull * n_blocks + 4), s_tried = seg(4ull * n_blocks + 4), s_ready = seg(4ull * n_blocks + 4),
              s_q = seg(8ull * floria_hip_ctx::MAX_LANES * FLORIA_MAX_PLOIDY + 16), s_diag = seg(16 + 8 * 64), s_steps = seg(16);
    const size_t zero_bytes = cursor - s_out.off;
    const Seg s_stop = seg(4ull * n_blocks + 4);             // speculative stages: smallest ploidy at which the stop rule is known to break, 0xffffffff = unknown
    rc = ctx->misc.ensure(cursor + 256); if (rc) return rc;
    char* M = ctx->misc.as<char>();
    th = T.begin(K_H2D);
    HIPCHK(box_up(M0 + s_roff.off, roff.data(), 8ull * (n_blocks + 1), bx_roff));
    if (!jobs.empty()) HIPCHK(box_up(M + s_jobs.off, jobs.data(), 4ull * jobs.size(), bx_jobs));
    HIPCHK(hipMemsetAsync(M + s_out.off, 0, zero_bytes, ctx->stream));
    if (stage_w > 1) HIPCHK(hipMemsetAsync(M + s_stop.off, 0xff, s_stop.bytes, ctx->stream));
    const double inf = std::numeric_limits<double>::infinity();
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

    // ---- reference-arithmetic mode: the iteration order of every read's position set (arith_kernel.h) --------------------------------
    ctx->cur_ord = nullptr; ctx->cur_ord_off = nullptr; ctx->cur_len_max = len_max;
    if (ctx->knobs.arith && n_contigs && SC.orders_in_flight) {
        ctx->cur_ord = ctx->arith_ord.as<uint2>(); ctx->cur_ord_off = ctx->arith_tab.as<uint64_t>() + n_contigs + 1;
    } else if (ctx->knobs.arith && n_contigs) {
        if (SC.n_cells.size() != n_contigs || chunked) return fail(FLORIA_E_INVALID, "internal: the reference-arithmetic mode needs resident contigs");
        int tk = T.begin(K_SEL);
        rc = cell_orders(ctx, bs.contigs, cdev, SC.n_cells, len_max);
        T.end(tk);
        if (rc) return rc;
    }

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
             (uint8_t*)(M + s_bpart.off), stages, chunked ? group_ev : nullptr, (double*)(M + s_mec.off), (double*)(M + s_na.off), (uint32_t*)(M + s_it.off),
             (uint8_t*)(M + s_done.off), (uint32_t*)(M + s_best.off), (uint32_t*)(M + s_tried.off), (uint32_t*)(M + s_q.off),
             (double*)(M + s_margin.off), (uint32_t*)(M + s_diag.off), (unsigned long long*)(M + s_steps.off), T, p1_shortcut,
             (uint32_t*)(M + s_stop.off), (uint32_t*)(M + s_ready.off));
    if (rc) { sync_all(ctx); return rc; }
    TR.mark("every launch queued");
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
    std::vector<double> job_margin((size_t)n_blocks * P + 1, inf);
    if (e == hipSuccess && n_blocks) e = hipMemcpyAsync(job_margin.data(), M + s_margin.off, 8ull * n_blocks * P, hipMemcpyDeviceToHost, ctx->stream);
    T.end(td);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && ctx->copy_stream) e = hipStreamSynchronize(ctx->copy_stream);
    if (e != hipSuccess) { sync_all(ctx); return fail(FLORIA_E_DEVICE, std::string("phase_blocks: ") + hipGetErrorString(e)); }
    TR.mark("results on the host");
    if (diag[1]) { return fail(FLORIA_E_DEVICE, "internal: beam slab free-list underflow"); }
#ifdef FLORIA_PROF
    { unsigned long long prof[64]; (void)hipMemcpy(prof, M + s_diag.off + 16, sizeof(prof), hipMemcpyDeviceToHost); fprintf(stderr, "[prof]"); for (int i = 0; i < 64; ++i) fprintf(stderr, " %d:%.1fM", i, prof[i] / 1e6); fprintf(stderr, "\n"); }
#endif
    memcpy(R->read_off, roff.data(), 8ull * (n_blocks + 1));
    // the pruning decisions of the (block, ploidy) jobs the reference runs, i.e. ploidy <= ploidies_tried (a speculative stage may have run more)
    for (uint32_t b = 0; b < n_blocks; ++b)
        for (uint32_t p = (p1_shortcut ? 2 : 1); p <= R->ploidies_tried[b]; ++p) margin = std::min(margin, job_margin[(size_t)b * P + p - 1]);
    if (p1_shortcut && !jobs.empty()) margin = std::min(margin, std::fabs(0.0 - std::log(PROB_CUTOFF)));   // the ploidy-1 decisions: p_k - lse == 0
    R->min_prune_margin = margin;
    ctx->timing.beam_ms = T.sum(K_BEAM); ctx->timing.optimize_ms = T.sum(K_OPT); ctx->timing.select_ms = T.sum(K_SEL);
    ctx->timing.beam_union_ms = T.union_ms(K_BEAM); ctx->timing.optimize_union_ms = T.union_ms(K_OPT);
    ctx->timing.h2d_ms = T.sum(K_H2D); ctx->timing.d2h_ms = T.sum(K_D2H); ctx->timing.total_ms = T.span(); ctx->timing.phase_ms = T.sum(K_PHASE);
    ctx->timing.algorithmic_bytes = algo_bytes; ctx->timing.beam_steps = steps;
    std::vector<uint32_t> stage_first(P + 2, 0);       // first ploidy of the stage that holds ploidy p
    for (auto& st : stages) for (uint32_t p : st) stage_first[p] = st.front();
    for (uint32_t b = 0; b < n_blocks; ++b) {          // a beam launch for ploidy p phases the blocks still active when p's stage starts (ploidy 1 needs no launch)
        const uint32_t tb = R->ploidies_tried[b];
        uint32_t launches_b = 0;
        for (uint32_t p = 1; p <= P && tb; ++p) if (tb >= stage_first[p] && !(p == 1 && p1_shortcut)) launches_b++;
        ctx->timing.beam_launch_bytes += blk_bytes[b] * launches_b; ctx->timing.jobs += tb;
    }
    ctx->batch_token = R->batch_token = ++ctx->token_counter;
    ctx->last_bs = bs; ctx->last_part = (const uint8_t*)(M + s_out.off); ctx->last_best = (const uint32_t*)(M + s_best.off); ctx->last_nall = nall;
    ctx->last_bc = bc; ctx->last_start.assign(blk_start, blk_start + n_blocks); ctx->last_end.assign(blk_end, blk_end + n_blocks);
    guard.r = nullptr;
    *out = R;
    return 0;
}

}  // namespace

extern "C" {

int floria_hip_phase_blocks_batch(floria_hip_ctx* ctx, const floria_hip_contig* const* contigs, uint32_t n_contigs,
                                  const uint32_t* blk_contig, const uint32_t* blk_start, const uint32_t* blk_end,
                                  uint32_t n_blocks, const floria_params* prm, floria_block_result** out) {
    if (!ctx || !out || !prm || (n_blocks && (!blk_start || !blk_end || !contigs))) return fail(FLORIA_E_INVALID, "null argument");
    *out = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    S1Contigs SC;
    SC.cdev.resize(n_contigs);
    SC.n_cells.assign(n_contigs, 0);
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t ci = blk_contig ? blk_contig[b] : 0;
        if (ci >= n_contigs || !contigs[ci]) return fail(FLORIA_E_INVALID, "blk_contig out of range");
    }
    for (uint32_t i = 0; i < n_contigs; ++i) if (contigs[i]) {
        if (contigs[i]->ctx != ctx) return fail(FLORIA_E_INVALID, "contig belongs to another context");
        SC.cdev[i] = contigs[i]->dev;
        SC.n_cells[i] = contigs[i]->n_cells;
        SC.len_max = std::max(SC.len_max, contigs[i]->max_len);
        SC.nall = std::max(SC.nall, contigs[i]->n_alleles);
        SC.any_q0 = SC.any_q0 || contigs[i]->has_q0;
    }
    return s1_core(ctx, SC, blk_contig, blk_start, blk_end, n_blocks, prm, out);
}

// S1 straight from host pileups: floria_hip_contig_upload_batch + floria_hip_phase_blocks_batch as ONE pipelined call.  The
// small arrays (read_off, first, last) go first, so find_reads_in_interval and the launch plan are made while the cell arrays
// are still on the wire; the cells travel in chunks of consecutive contigs on the copy stream, each followed by its validate +
// flatten launch, and the blocks of chunk g form job group g, whose kernels start when the chunk has landed.  The PCIe time
// of all but the first chunk hides behind the kernels of the earlier ones.  The launch plan is made for biallelic pileups
// without q = 0 cells (known only once every chunk has been validated); a batch that turns out otherwise is phased again from
// its — by then resident — contigs with the matching kernels, so results never depend on the route.
static int phase_pileups_impl(floria_hip_ctx* ctx, const floria_pileup* pileups, const floria_pileup_packed* pk, uint32_t n_contigs,
                              const uint32_t* blk_contig, const uint32_t* blk_start, const uint32_t* blk_end, uint32_t n_blocks,
                              const floria_params* prm, floria_block_result** out, floria_hip_contig** keep) {
    if (!ctx || !out || !prm || (n_contigs && !pileups && !pk) || (n_blocks && (!blk_start || !blk_end))) return fail(FLORIA_E_INVALID, "null argument");
    *out = nullptr;
    if (keep) for (uint32_t i = 0; i < n_contigs; ++i) keep[i] = nullptr;
    HIPCHK(hipSetDevice(ctx->device));
    for (uint32_t b = 0; b < n_blocks; ++b) if ((blk_contig ? blk_contig[b] : 0) >= n_contigs) return fail(FLORIA_E_INVALID, "blk_contig out of range");
    if (n_contigs == 0) { S1Contigs SC; return s1_core(ctx, SC, blk_contig, blk_start, blk_end, n_blocks, prm, out); }
    uint64_t cells = 0;
    auto n_reads_of = [&](uint32_t i) { return pk ? pk[i].n_reads : pileups[i].n_reads; };
    for (uint32_t i = 0; i < n_contigs; ++i) { const uint32_t* ro = pk ? pk[i].read_off : pileups[i].read_off; if (n_reads_of(i) && ro) cells += ro[n_reads_of(i)]; }
    // auto: ~0.5 GB of host pileup per chunk, at most 5 (measured on BASELINE config 4, 2.65 GB: 5 chunks, the first one half-size, are best)
    uint32_t want_chunks = ctx->knobs.upload_chunks ? ctx->knobs.upload_chunks : (uint32_t)std::max<uint64_t>(cells * 6 >= (160ull << 20) ? 2 : 1, std::min<uint64_t>(5, cells * 6 / (200ull << 20)));
    // reference-arithmetic mode: a job group's chain of stages is longer (fewer wave slots, the sequential folds), so a group that starts late ends the call late: two chunks
    // (measured on config 4 at -e 0.04, ms per call from packed pinned memory, 1 / 2 / 3 / 5 chunks: 122.2 / 117.7 / 120.5 / 123.4; resident 98; profiles/r06_arith_pipe_timing.txt)
    if (ctx->knobs.arith && !ctx->knobs.upload_chunks) want_chunks = std::min<uint32_t>(want_chunks, 2);
    // (measured, config 4, H2D-inclusive: 500 contigs / 0.66 GB: 2 chunks 60.5 ms, 3-4 chunks 56.5; 1000 contigs / 1.33 GB: 2 chunks 86 ms, 4-5 chunks 77; 2000 contigs: 5 chunks)
    // batches small enough for speculative ploidy stages (s1_core) keep chunk groups x stage width within the hardware queues
    UploadPlan UP;
    uint32_t chunk_cap = floria_hip_ctx::MAX_GROUPS;
    if (!ctx->knobs.upload_chunks && ctx->knobs.speculate < 0 && prm->max_ploidy >= 3) {
        if (n_blocks <= (uint32_t)ctx->n_cu * 18) chunk_cap = std::max<uint32_t>(1, 10 / prm->max_ploidy);          // (the thresholds of s1_core's stage plan)
        else if (n_blocks <= (uint32_t)ctx->n_cu * 23 && prm->max_ploidy >= 4) chunk_cap = std::max<uint32_t>(1, 10 / std::max<uint32_t>(3, prm->max_ploidy - 3));
    }
    int rc = plan_upload(ctx, pileups, pk, n_contigs, std::min<uint32_t>(std::min(want_chunks, chunk_cap), floria_hip_ctx::MAX_GROUPS), UP);
    if (rc) return rc;
    std::vector<floria_hip_contig*> handles(n_contigs, nullptr);
    auto drop = [&](int code) { sync_all(ctx); for (auto* h : handles) if (h) { h->arena = nullptr; delete h; } arena_put(UP.A); return code; };
    bool pipelined = UP.all_pinned && UP.n_chunks > 1;
    const bool arith_pipe = pipelined && ctx->knobs.arith != 0;      // reference-arithmetic mode: every chunk's cell orders are computed behind its flatten launch (round 6; until then this mode uploaded the whole batch first)
    OrderPlan OP;
    std::vector<fl::ContigDev> pipe_cdev;
    std::vector<uint64_t> pipe_cells;
    const uint64_t order_scratch = 64ull << 20;
    uint64_t pinned_b = 0, staged_b = 0;
    floria_block_result* R = nullptr;
    if (pipelined) {
        if (!ctx->copy_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        if (!ctx->flat_stream) {                    // highest priority: when wave slots free up, a waiting flatten launch goes first
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (hipStreamCreateWithPriority(&ctx->flat_stream, hipStreamNonBlocking, hi) != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipStreamCreateWithFlags(&ctx->flat_stream, hipStreamNonBlocking)); }
        }
        for (uint32_t g = 0; g <= UP.n_chunks; ++g) {
            if (!ctx->ev_chunk[g]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_chunk[g], hipEventDisableTiming));
            if (!ctx->ev_copied[g]) HIPCHK(hipEventCreateWithFlags(&ctx->ev_copied[g], hipEventDisableTiming));
        }
        rc = issue_copies(ctx, UP.small_runs, &pinned_b, &staged_b);                       // main stream: read_off / first / last + tables
        hipError_t e = rc ? hipSuccess : issue_tables(ctx, UP, ctx->stream);
        pipe_cdev.resize(n_contigs); pipe_cells.resize(n_contigs);
        for (uint32_t i = 0; i < n_contigs; ++i) {
            fl::ContigDev& d = pipe_cdev[i];
            d.read_off = UP.ucd[i].read_off; d.first = UP.ucd[i].first; d.last = UP.ucd[i].last; d.cell_snp = UP.ucd[i].snp;
            d.cell_aw = UP.ucd[i].cell_aw; d.tw = UP.ucd[i].tw; d.meta = UP.ucd[i].meta; d.n_reads = n_reads_of(i);
            d.set_order = UP.so_dev[i];
            pipe_cells[i] = UP.cp[i + 1] - UP.cp[i];
        }
        size_t ord_cdev_off = 0, ord_todo_words = 0;
        if (arith_pipe && !rc && e == hipSuccess) {
            // the order kernels' tables: prefix sums + status word | the contigs' device arrays; the todo lists of the chunks (one count word each) + the emulation's scratch
            order_plan(pipe_cdev, pipe_cells, OP);
            ord_cdev_off = (OP.pre.size() * 8 + 255) & ~(size_t)255;
            ord_todo_words = OP.R_all + UP.n_chunks;
            rc = ctx->arith_tab.ensure(ord_cdev_off + sizeof(fl::ContigDev) * n_contigs);
            if (!rc) rc = ctx->arith_ord.ensure(8 * OP.cells + 16);
            if (!rc) rc = ctx->arith_scr.ensure(((8 * ord_todo_words + 255) & ~(size_t)255) + order_scratch);
            if (!rc) e = hipMemcpyAsync(ctx->arith_tab.p, OP.pre.data(), OP.pre.size() * 8, hipMemcpyHostToDevice, ctx->stream);       // (OP / pipe_cdev live until this call returns, behind a sync)
            if (!rc && e == hipSuccess) e = hipMemcpyAsync(ctx->arith_tab.as<char>() + ord_cdev_off, pipe_cdev.data(), sizeof(fl::ContigDev) * n_contigs, hipMemcpyHostToDevice, ctx->stream);
            for (uint32_t g = 0; g < UP.n_chunks && !rc && e == hipSuccess; ++g) e = hipMemsetAsync(ctx->arith_scr.as<uint64_t>() + OP.pre[UP.chunk_first[g]] + g, 0, 8, ctx->stream);      // the chunks' todo counters
            ctx->ord_epoch = ~0ull;                                                            // (the cached orders are being overwritten)
        }
        if (!rc && e == hipSuccess) e = hipEventRecord(ctx->ev_chunk[UP.n_chunks], ctx->stream);
        if (!rc && e == hipSuccess) e = hipStreamWaitEvent(ctx->flat_stream, ctx->ev_chunk[UP.n_chunks], 0);
        for (uint32_t g = 0; g < UP.n_chunks && !rc && e == hipSuccess; ++g) {             // copy stream: the cells of chunk g, back to back with chunk g+1;
            for (const CopyRun& r : UP.chunk_runs[g]) { pinned_b += r.bytes; if (e == hipSuccess) e = hipMemcpyAsync(r.dst, r.src, r.bytes, hipMemcpyHostToDevice, ctx->copy_stream); }
            if (e == hipSuccess) e = hipEventRecord(ctx->ev_copied[g], ctx->copy_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->flat_stream, ctx->ev_copied[g], 0);       // flatten stream: validate + flatten chunk g
            if (e == hipSuccess) e = launch_flatten(ctx, UP, g, ctx->flat_stream);
            if (e == hipSuccess && arith_pipe) {                                            // ... and the iteration orders of its reads' position sets (arith_kernel.h)
                const uint32_t c0 = UP.chunk_first[g], c1 = UP.chunk_first[g + 1];
                const int orc = order_launch(ctx, OP, (const fl::ContigDev*)(ctx->arith_tab.as<char>() + ord_cdev_off), c0, c1, ctx->arith_scr.as<uint64_t>() + OP.pre[c0] + g,
                                             ctx->arith_scr.as<uint8_t>() + ((8 * ord_todo_words + 255) & ~(size_t)255), order_scratch, 0,
                                             &((const fl::UploadStatus*)(UP.T + UP.t_st) + c0)->max_len, ctx->flat_stream);
                if (orc) rc = orc;
            }
            if (e == hipSuccess) e = hipEventRecord(ctx->ev_chunk[g], ctx->flat_stream);
        }
        if (!rc && e != hipSuccess) rc = fail(FLORIA_E_DEVICE, std::string("pipelined upload: ") + hipGetErrorString(e));
        if (rc) return drop(rc);
        S1Contigs SC;
        SC.cdev = pipe_cdev;
        if (arith_pipe) { SC.n_cells = pipe_cells; SC.orders_in_flight = true; }
        SC.len_max = BINOM_NMAX_CAP; SC.nall = 2; SC.any_q0 = false;                      // optimistic plan, verified below
        SC.contig_chunk = UP.contig_chunk.data(); SC.n_chunks = UP.n_chunks; SC.chunk_ev = ctx->ev_chunk;
        // the compact wire form is on the device in a tenth of the step: what counts then is that the expand / flatten launches of the later chunks are
        // not starved by persistent grids that already own every wave slot.  Measured on config 4 (5 chunks, three runs each, ms per step; chunk -> group):
        // 0|111|2 106.7, 0|11|22 107.7, 00|11|2 108.0, 000|11 108.9, 00|111 108.9, 0|1111 109.3, a group per chunk 109.6, one group 109.4
        if (pk) { SC.chunk_groups = 3; SC.ends_apart = true; }
        rc = s1_core(ctx, SC, blk_contig, blk_start, blk_end, n_blocks, prm, &R);
        if (rc) return drop(rc);
        const floria_timing tm = ctx->timing;
        hipError_t e2 = hipMemcpy(UP.ust.data(), UP.T + UP.t_st, sizeof(fl::UploadStatus) * n_contigs, hipMemcpyDeviceToHost);
        if (e2 != hipSuccess) { floria_hip_block_result_free(R); return drop(fail(FLORIA_E_DEVICE, std::string("upload status: ") + hipGetErrorString(e2))); }
        rc = finish_upload(ctx, UP, handles.data());
        if (!rc && arith_pipe) {
            uint64_t bad = 0;
            e2 = hipMemcpy(&bad, ctx->arith_tab.as<uint64_t>() + 2 * (size_t)n_contigs + 2, 8, hipMemcpyDeviceToHost);
            if (e2 != hipSuccess) rc = fail(FLORIA_E_DEVICE, std::string("set-order status: ") + hipGetErrorString(e2));
            else rc = order_bad_read(OP, bad);
            if (!rc) { ctx->ord_epoch = ctx->upload_epoch; ctx->ord_sig = OP.sig; }          // S2 / the next S1 call over these contigs reuses the orders
            if (rc) { floria_hip_block_result_free(R); for (auto* h : handles) floria_hip_contig_free(h); return rc; }      // (the handles own the arena by now)
        }
        if (rc) { floria_hip_block_result_free(R); for (auto*& h : handles) h = nullptr; return drop(rc); }
        bool plan_ok = true;
        for (auto* h : handles) plan_ok = plan_ok && h->n_alleles == 2 && !h->has_q0;
        if (!plan_ok) {                                                                   // rare: phase again with the kernels this batch needs
            floria_hip_block_result_free(R); R = nullptr;
            rc = floria_hip_phase_blocks_batch(ctx, handles.data(), n_contigs, blk_contig, blk_start, blk_end, n_blocks, prm, &R);
            if (rc) { for (auto* h : handles) floria_hip_contig_free(h); return rc; }
        } else ctx->timing = tm;
    } else {
        arena_put(UP.A);                                                                  // (back to the cache: the plain upload takes it from there)
        rc = upload_batch_impl(ctx, pileups, pk, n_contigs, handles.data());
        if (rc) return rc;
        pinned_b = ctx->timing.upload_pinned_bytes; staged_b = ctx->timing.upload_staged_bytes;
        rc = floria_hip_phase_blocks_batch(ctx, handles.data(), n_contigs, blk_contig, blk_start, blk_end, n_blocks, prm, &R);
        if (rc) { for (auto* h : handles) floria_hip_contig_free(h); return rc; }
    }
    ctx->timing.upload_pinned_bytes = pinned_b; ctx->timing.upload_staged_bytes = staged_b; ctx->timing.upload_chunks = pipelined ? UP.n_chunks : 1;
    if (keep) for (uint32_t i = 0; i < n_contigs; ++i) keep[i] = handles[i];
    else { for (auto* h : handles) floria_hip_contig_free(h); ctx->batch_token = 0; R->batch_token = 0; }     // nothing stays resident: no hap graph for this batch
    *out = R;
    return 0;
}

int floria_hip_phase_pileups_batch(floria_hip_ctx* ctx, const floria_pileup* pileups, uint32_t n_contigs,
                                   const uint32_t* blk_contig, const uint32_t* blk_start, const uint32_t* blk_end, uint32_t n_blocks,
                                   const floria_params* prm, floria_block_result** out, floria_hip_contig** keep) {
    return phase_pileups_impl(ctx, pileups, nullptr, n_contigs, blk_contig, blk_start, blk_end, n_blocks, prm, out, keep);
}
int floria_hip_phase_pileups_batch_packed(floria_hip_ctx* ctx, const floria_pileup_packed* pileups, uint32_t n_contigs,
                                          const uint32_t* blk_contig, const uint32_t* blk_start, const uint32_t* blk_end, uint32_t n_blocks,
                                          const floria_params* prm, floria_block_result** out, floria_hip_contig** keep) {
    return phase_pileups_impl(ctx, nullptr, pileups, n_contigs, blk_contig, blk_start, blk_end, n_blocks, prm, out, keep);
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

// ---- self-test of an assumption the kernels make about the hardware: the f32 screen of stable_binom_cdf_p_rev (beam_slab_kernel.h: binom_screen_f32, hardware rcp / log2)
// against the host-libm table, over every (n, k) with n <= n_max.  *max_err_per_n receives max |screen - table| / n; the screen's tolerance assumes <= BINOM_SCREEN_C = 2e-5.
int floria_hip_selftest(floria_hip_ctx* ctx, double epsilon, uint32_t n_max, double* max_err_per_n) {
    if (!ctx || !max_err_per_n || !(epsilon > 0.0 && epsilon < 1.0) || n_max == 0 || n_max > 4096) return fail(FLORIA_E_INVALID, "selftest: context, 0 < epsilon < 1, 1 <= n_max <= 4096");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->batch_token = 0;
    int rc = ensure_binom(ctx, epsilon, n_max); if (rc) return rc;
    rc = ctx->misc.ensure(256); if (rc) return rc;
    double* d_out = ctx->misc.as<double>();
    HIPCHK(hipMemsetAsync(d_out, 0, 8, ctx->stream));
    hipLaunchKernelGGL(fl::binom_screen_selftest_kernel, dim3((n_max + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_binom.as<double>(), n_max,
                       (float)std::log(epsilon), (float)std::log(1.0 - epsilon), (float)epsilon, (float)(1.0 / DIV_FACTOR), d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(max_err_per_n, d_out, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- alignment::realign (alignment.rs:7-64) for the windows the host could not decide --------------------------------------------------------
int floria_hip_realign(floria_hip_ctx* ctx, const uint8_t* read_windows, const uint8_t* ref_windows, const uint8_t* alleles, const uint8_t* n_alleles,
                       uint64_t n, uint8_t* best, int32_t* score) {
    if (!ctx || (n && (!read_windows || !ref_windows || !alleles || !n_alleles || !best))) return fail(FLORIA_E_INVALID, "null argument");
    if (n == 0) return 0;
    for (uint64_t i = 0; i < n; ++i) if (n_alleles[i] == 0 || n_alleles[i] > FLORIA_MAX_ALLELES) return fail(FLORIA_E_INVALID, "n_alleles must be 1..FLORIA_MAX_ALLELES");
    HIPCHK(hipSetDevice(ctx->device));
    ctx->batch_token = 0;
    struct Seg { size_t off, bytes; };
    size_t cursor = 0;
    auto seg = [&](size_t bytes) { Seg sg{cursor, bytes}; cursor += (bytes + 255) & ~(size_t)255; return sg; };
    const Seg s_q = seg(32 * n), s_r = seg(32 * n), s_a = seg((size_t)FLORIA_MAX_ALLELES * n), s_n = seg(n), s_b = seg(n), s_s = seg(4 * n);
    int rc = ctx->misc.ensure(cursor + 256); if (rc) return rc;
    char* M = ctx->misc.as<char>();
    HIPCHK(hipMemcpyAsync(M + s_q.off, read_windows, 32 * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(M + s_r.off, ref_windows, 32 * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(M + s_a.off, alleles, (size_t)FLORIA_MAX_ALLELES * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(M + s_n.off, n_alleles, n, hipMemcpyHostToDevice, ctx->stream));
    fl::RealignArgs a{};
    a.q = (const uint8_t*)(M + s_q.off); a.r = (const uint8_t*)(M + s_r.off); a.alleles = (const uint8_t*)(M + s_a.off); a.n_alleles = (const uint8_t*)(M + s_n.off);
    a.best = (uint8_t*)(M + s_b.off); a.score = score ? (int32_t*)(M + s_s.off) : nullptr; a.n = n;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 3) / 4, (uint64_t)ctx->n_cu * 32);      // 4 windows per workgroup, grid-stride beyond 8 workgroups per CU
    hipLaunchKernelGGL(fl::realign_kernel, dim3(grid), dim3(256), 0, ctx->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(best, M + s_b.off, n, hipMemcpyDeviceToHost, ctx->stream));
    if (score) HIPCHK(hipMemcpyAsync(score, M + s_s.off, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
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
        if (int rc0 = host_meta(contigs[c])) return rc0;
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
        if (int rc0 = host_meta(contigs[i])) return rc0;
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
    std::vector<uint32_t> r2g_all, gpos0_all, multi_all;
    std::vector<uint64_t> multi_off_all;
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
        multi_off_all.push_back(multi_all.size());
        {                                                   // visiting indices of the reads that have a choice (more than one candidate group)
            const uint64_t nv = read_order ? order_off[ci + 1] - order_off[ci] : N;
            for (uint64_t v = 0; v < nv; ++v) { const uint32_t r = read_order ? read_order[order_off[ci] + v] : (uint32_t)v; if (off[r + 1] - off[r] > 1) multi_all.push_back((uint32_t)v); }
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
                  s_oo = seg(8ull * (n_contigs + 1)), s_mu = seg(4ull * multi_all.size() + 4), s_mo = seg(8ull * (n_contigs + 1)), s_ls = seg(4ull * n_contigs + 4);
        multi_off_all.push_back(multi_all.size());
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
        HIPCHK(h2d(s_mu, multi_all.data(), 4ull * multi_all.size())); HIPCHK(h2d(s_mo, multi_off_all.data(), 8ull * (n_contigs + 1)));
        T.end(th);
        fl::ReassignArgs a{};
        a.contigs = (const fl::ContigDev*)(M + s_cd.off); a.n_contigs = n_contigs;
        a.r2g_off_base = (const uint64_t*)(M + s_rob.off); a.r2g_off = (const uint64_t*)(M + s_ro.off);
        a.r2g_base = (const uint64_t*)(M + s_rb.off); a.r2g = (const uint32_t*)(M + s_r2g.off);
        a.grp_base = (const uint64_t*)(M + s_gb.off); a.grp_hist_off = (const uint64_t*)(M + s_ho.off); a.grp_pos0 = (const uint32_t*)(M + s_p0.off);
        a.hist = (uint64_t*)(M + s_hist.off); a.assign = (int32_t*)(M + s_as.off); a.assign_base = (const uint64_t*)(M + s_ab.off);
        a.eps = epsilon; a.queue_head = (uint32_t*)(M + s_q.off);
        a.order = read_order ? (const uint32_t*)(M + s_ord.off) : nullptr; a.order_off = (const uint64_t*)(M + s_oo.off);
        a.multi = (const uint32_t*)(M + s_mu.off); a.multi_off = (const uint64_t*)(M + s_mo.off);
        const bool arith = ctx->knobs.arith != 0;
        if (arith) {                                // the reference's running sums: every read's cells in the order of its position set
            std::vector<uint64_t> ncells(n_contigs); uint32_t len_max = 1;
            for (uint32_t ci = 0; ci < n_contigs; ++ci) { ncells[ci] = contigs[ci]->n_cells; len_max = std::max(len_max, contigs[ci]->max_len); }
            rc = cell_orders(ctx, a.contigs, cdev, ncells, len_max); if (rc) return rc;
            a.cell_ord = ctx->cur_ord; a.cell_ord_off = ctx->cur_ord_off;
        }
        // contigs where at least 1 read in 8 has a choice take the latency-optimised chain kernel, the others the parallel one
        std::vector<uint32_t> list_par, list_chain;
        for (uint32_t ci = 0; ci < n_contigs; ++ci) {
            const uint64_t nm = multi_off_all[ci + 1] - multi_off_all[ci];
            const uint64_t nv = read_order ? order_off[ci + 1] - order_off[ci] : contigs[ci]->n_reads;
            const bool chain = !arith && (ctx->knobs.reassign_path == 2 || (ctx->knobs.reassign_path == 0 && nm * 8 >= nv && nm > 0));
            (chain ? list_chain : list_par).push_back(ci);
        }
        std::vector<uint32_t> lists(list_par);
        lists.insert(lists.end(), list_chain.begin(), list_chain.end());
        HIPCHK(hipMemcpyAsync(M + s_ls.off, lists.data(), 4ull * lists.size(), hipMemcpyHostToDevice, ctx->stream));
        int tk = T.begin(K_REASSIGN);
        if (!list_par.empty()) {
            a.list = (const uint32_t*)(M + s_ls.off); a.n_list = (uint32_t)list_par.size(); a.queue_head = (uint32_t*)(M + s_q.off);
            const uint32_t grid = std::min<uint32_t>(a.n_list, (uint32_t)ctx->n_cu * 8);
            if (arith) { if (A == 2) hipLaunchKernelGGL((fl::reassign_kernel<2, true>), dim3(grid), dim3(fl::REASSIGN_THREADS), 0, ctx->stream, a);
                         else hipLaunchKernelGGL((fl::reassign_kernel<4, true>), dim3(grid), dim3(fl::REASSIGN_THREADS), 0, ctx->stream, a); }
            else if (A == 2) hipLaunchKernelGGL(fl::reassign_kernel<2>, dim3(grid), dim3(fl::REASSIGN_THREADS), 0, ctx->stream, a);
            else hipLaunchKernelGGL(fl::reassign_kernel<4>, dim3(grid), dim3(fl::REASSIGN_THREADS), 0, ctx->stream, a);
        }
        if (!list_chain.empty()) {
            a.list = (const uint32_t*)(M + s_ls.off) + list_par.size(); a.n_list = (uint32_t)list_chain.size(); a.queue_head = (uint32_t*)(M + s_q.off) + 1;
            const uint32_t grid = std::min<uint32_t>(a.n_list, (uint32_t)ctx->n_cu * 16);
            if (A == 2) hipLaunchKernelGGL(fl::reassign_chain_kernel<2>, dim3(grid), dim3(64), 0, ctx->stream, a);
            else hipLaunchKernelGGL(fl::reassign_chain_kernel<4>, dim3(grid), dim3(64), 0, ctx->stream, a);
        }
        ctx->timing.jobs = multi_all.size();                        // reads that had a choice (the sequential part of S2)
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
        // Which read a split drops depends on the iteration order of the reference's FxHashSet among reads that share a first_position (:34-35, :62-63; measured:
        // every short-read contig of scripts/a14_sensitivity.py changes with it, no long-read one).  Here: ascending counter_id.  A host that wants its own sets'
        // order sets "s2_assign_only": it gets the haplogroups as re-inserted and runs these two cheap integer steps on its own sets.
        if (!ctx->knobs.s2_assign_only) {
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
        if (!ctx->knobs.s2_assign_only) std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return ranges[x] < ranges[y]; });
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
