// floria_oracle.cpp — CPU restatement of floria's per-block read->haplotype clustering path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; nothing under floria_amd/ (the product) may.
//
// PARITY UNPINNED: the reference (Rust) cannot be built in this environment (no rustc/cargo, no
// vendored crates) and ships no tests, golden vectors or fixtures for this path.
// This file follows the reference source function by function (citations are file:line under
// /root/reference/src) and is pinned only by hand-derived known-answer tests (tests/test_kat.py).
//
// Faithful-mode data structures: haplotypes are hash maps position -> allele counts that are
// deep-cloned per beam child and deep-compared for duplicate suppression, like the reference's
// `Haplotype = FxHashMap<SnpPosition, FxHashMap<Genotype, GenotypeCount>>` (types_structs.rs:15).
//
// Two documented canonicalisations (DESIGN.md "Oracle"):
//  (1) Weighted sums.  Every quality weight w(q) (utils_frags.rs:702-711) is an exact multiple of
//      2^-24, so `same`/`diff`/histogram sums are carried as Q24 integers plus a count m of
//      "+epsilon" terms and converted once: value = Q*2^-24 + m*epsilon.  For dyadic epsilon this is
//      bit-identical to the reference's f64 running sums in any order; for other epsilon the
//      reference's own sums depend on FxHash iteration order at the ulp level.
//  (2) Iteration order.  At the three sites where the reference's *result* depends on
//      FxHashSet/FxHashMap iteration order (local_clustering.rs:304, part_block_manip.rs:195-222,
//      part_block_manip.rs:34-35,62-63) this restatement iterates in ascending counter_id.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../include/floria_hip.h"

namespace {

// ---- constants.rs ------------------------------------------------------------------------------
constexpr int    NUM_ITER_OPTIMIZE = 20;      // constants.rs:3
constexpr double DIV_FACTOR        = 0.25;    // constants.rs:5
constexpr double PROB_CUTOFF       = 0.01;    // constants.rs:6
// USE_QUAL_SCORES = true (:15), SEPARATE_BROKEN_HAPLOGROUPS = true (:17), WEIRD_SPLIT = false (:18),
// MERGE_SIMILAR_HAPLOGROUPS = false (:16)

thread_local std::string g_err;
// Sensitivity knob (tests only): 0 = canonical ascending counter_id at the order-dependent sites, 1 = DEscending.
// The reference's FxHash order is some third permutation; comparing 0 vs 1 measures how often the order matters at all.
std::atomic<int> g_order_mode{0};
// Arithmetic mode.  0 (canonical, what the HIP path computes): every weighted sum is carried as (Q24 integer, #epsilon) and turned
// into f64 once, value = Q * 2^-24 + m * eps.  1 (the reference's own arithmetic): running f64 sums in the order the
// reference adds the terms, i.e. the iteration order of its hash containers (emulated with FxSet below) — `diff += epsilon`
// interleaved with `diff += w` over a read's positions (utils_frags.rs:32-75), error_vec per partition summed over the partitions
// (global_clustering.rs:181-208), `errors +=` over a haplotype's positions (local_clustering.rs:218-260).  The two agree bit for bit
// when eps is dyadic; mode 1 exists to COUNT how often they part at the reference's non-dyadic operating points.
std::atomic<int> g_arith_mode{0};

// ---- mode 2: FxHashSet<&Frag> as the reference's binary lays it out -------------------------------------------------------------
// The three order-dependent sites iterate hash sets / maps keyed by &Frag, whose Hash is `counter_id.hash()` (types_structs.rs:
// 101-106).  fxhash 0.2.1 (Cargo.lock) hashes one usize word to  id * 0x517cc1b727220a95 ; the table is the hashbrown SwissTable
// bundled in rustc's std (0.14.x for rustc 1.80, the README's minimum; x86-64 => 16-byte SSE2 groups).  Restated from the crates
// as published — it cannot be checked against a real build here (no Rust toolchain), so mode 2 is an ORACLE MODE for measuring
// how much the unknown order matters, not a parity claim:
//   buckets: power of two, capacity -> buckets: < 4 -> 4, < 8 -> 8, else next_pow2(cap * 8 / 7); usable = mask if mask < 8 else buckets / 8 * 7
//   ctrl bytes: EMPTY 0xFF, DELETED 0x80, full = top 7 bits of the hash; buckets + 16 of them, the first 16 mirrored at the end
//   probe: start hash & mask, groups of 16 at triangular strides; insert = first EMPTY / DELETED in probe order (small-table wrap fix)
//   insert first reserves 1: when growth_left == 0 -> rehash in place if items + 1 <= usable / 2, else resize to max(items + 1, usable + 1)
//   remove -> EMPTY when the run of non-EMPTY bytes around the slot is shorter than a group, else DELETED (tombstone)
//   resize re-inserts in ascending old bucket order; clone copies the layout; ITERATION = ascending bucket index
struct FxSet {
    static constexpr size_t W = 16;
    static constexpr uint8_t EMPTY = 0xFF, DELETED = 0x80;
    std::vector<uint8_t> ctrl;
    std::vector<uint64_t> slot;
    size_t buckets = 0, items = 0, growth_left = 0;
    static uint64_t hash_of(uint64_t k) { return k * 0x517cc1b727220a95ull; }
    static uint8_t h2(uint64_t h) { return (uint8_t)(h >> 57); }
    size_t mask() const { return buckets - 1; }
    static size_t cap_of(size_t nb) { return nb == 0 ? 0 : (nb - 1 < 8 ? nb - 1 : nb / 8 * 7); }
    static size_t buckets_for(size_t cap) {
        if (cap < 8) return cap < 4 ? 4 : 8;
        size_t adj = cap * 8 / 7, b = 1;
        while (b < adj) b <<= 1;
        return b;
    }
    void alloc(size_t nb) { buckets = nb; ctrl.assign(nb + W, EMPTY); slot.assign(nb, 0); growth_left = cap_of(nb); items = 0; }
    void set_ctrl(size_t i, uint8_t c) { ctrl[i] = c; ctrl[((i - W) & mask()) + W] = c; }
    // lowest index in the 16-byte group at pos whose byte satisfies pred, or W
    template <class F> size_t group_first(size_t pos, F pred) const { for (size_t b = 0; b < W; ++b) if (pred(ctrl[pos + b])) return b; return W; }
    size_t find_insert_slot(uint64_t h) const {
        size_t pos = (size_t)h & mask(), stride = 0;
        for (;;) {
            const size_t b = group_first(pos, [](uint8_t c) { return (c & 0x80) != 0; });
            if (b < W) {
                size_t idx = (pos + b) & mask();
                if ((ctrl[idx] & 0x80) == 0) idx = group_first(0, [](uint8_t c) { return (c & 0x80) != 0; });   // table smaller than a group: the hit was in the mirror/padding
                return idx;
            }
            stride += W; pos = (pos + stride) & mask();
        }
    }
    long find(uint64_t key) const {
        if (buckets == 0) return -1;
        const uint64_t h = hash_of(key);
        size_t pos = (size_t)h & mask(), stride = 0;
        for (;;) {
            bool any_empty = false;
            for (size_t b = 0; b < W; ++b) {
                const uint8_t c = ctrl[pos + b];
                if (c == h2(h) && slot[(pos + b) & mask()] == key) return (long)((pos + b) & mask());
                if (c == EMPTY) any_empty = true;
            }
            if (any_empty) return -1;
            stride += W; pos = (pos + stride) & mask();
        }
    }
    void resize(size_t capacity) {
        FxSet n; n.alloc(buckets_for(capacity));
        for (size_t i = 0; i < buckets; ++i) if ((ctrl[i] & 0x80) == 0) {
            const uint64_t h = hash_of(slot[i]);
            const size_t idx = n.find_insert_slot(h);
            n.set_ctrl(idx, h2(h)); n.slot[idx] = slot[i];
        }
        n.items = items; n.growth_left = cap_of(n.buckets) - items;
        *this = std::move(n);
    }
    void rehash_in_place() {
        for (size_t i = 0; i < buckets; ++i) ctrl[i] = (ctrl[i] & 0x80) ? EMPTY : DELETED;           // special -> EMPTY, full -> DELETED
        if (buckets < W) { for (size_t i = buckets; i < W; ++i) ctrl[i] = EMPTY; for (size_t i = 0; i < buckets; ++i) ctrl[W + i] = ctrl[i]; }
        else for (size_t i = 0; i < W; ++i) ctrl[buckets + i] = ctrl[i];
        for (size_t i = 0; i < buckets; ++i) {
            if (ctrl[i] != DELETED) continue;
            for (;;) {
                const uint64_t h = hash_of(slot[i]);
                const size_t ni = find_insert_slot(h);
                const size_t start = (size_t)h & mask();
                auto probe_index = [&](size_t pos) { return ((pos - start) & mask()) / W; };
                if (probe_index(i) == probe_index(ni)) { set_ctrl(i, h2(h)); break; }
                const uint8_t prev = ctrl[ni];
                set_ctrl(ni, h2(h));
                if (prev == EMPTY) { set_ctrl(i, EMPTY); slot[ni] = slot[i]; break; }
                std::swap(slot[i], slot[ni]);                                                           // prev == DELETED: carry on with the element swapped in
            }
        }
        growth_left = cap_of(buckets) - items;
    }
    void reserve(size_t additional) {                                                                 // RawTable::reserve -> reserve_rehash
        if (additional <= growth_left) return;
        const size_t new_items = items + additional, full = cap_of(buckets);
        if (buckets != 0 && new_items <= full / 2) rehash_in_place(); else resize(std::max(new_items, full + 1));
    }
    bool insert(uint64_t key) {
        if (growth_left == 0) {                                                                       // reserve(1)
            const size_t new_items = items + 1, full = cap_of(buckets);
            if (buckets != 0 && new_items <= full / 2) rehash_in_place(); else resize(std::max(new_items, full + 1));
        }
        if (find(key) >= 0) return false;
        const uint64_t h = hash_of(key);
        const size_t idx = find_insert_slot(h);
        if (ctrl[idx] == EMPTY) --growth_left;
        set_ctrl(idx, h2(h)); slot[idx] = key; ++items;
        return true;
    }
    // HashMap::entry(key).or_insert(..) (std's rustc_entry): the key is looked up FIRST and room for one more is reserved only on the Vacant
    // path, so touching a key that is already there never grows a full table — unlike insert above, which reserves before it looks
    // (hashbrown's find_or_find_insert_slot).  set_to_seq_dict fills its position map this way (utils_frags.rs:165).
    bool entry_insert(uint64_t key) {
        if (find(key) >= 0) return false;
        reserve(1);
        const uint64_t h = hash_of(key);
        const size_t idx = find_insert_slot(h);
        if (ctrl[idx] == EMPTY) --growth_left;
        set_ctrl(idx, h2(h)); slot[idx] = key; ++items;
        return true;
    }
    bool remove(uint64_t key) {
        const long f = find(key);
        if (f < 0) return false;
        const size_t idx = (size_t)f, before = (idx - W) & mask();
        size_t lead = 0, trail = 0;                      // EMPTY-free run just before / from the slot: leading zeros of match_empty(before), trailing zeros of match_empty(idx)
        for (size_t b = W; b-- > 0;) { if (ctrl[before + b] == EMPTY) break; ++lead; }
        for (size_t b = 0; b < W; ++b) { if (ctrl[idx + b] == EMPTY) break; ++trail; }
        if (lead + trail >= W) set_ctrl(idx, DELETED); else { set_ctrl(idx, EMPTY); ++growth_left; }
        --items;
        return true;
    }
    template <class F> void for_each(F f) const { for (size_t i = 0; i < buckets; ++i) if ((ctrl[i] & 0x80) == 0) f(slot[i]); }
    std::vector<uint32_t> order() const { std::vector<uint32_t> v; for_each([&](uint64_t k) { v.push_back((uint32_t)k); }); return v; }
};

// ---- phred_scale (utils_frags.rs:702-711) ---------------------------------------------------------
// prob = 1f32 - 10f32.powf(q as f32 / -10.), widened to f64.  Always k * 2^-24 (checked below).
struct WeightLut {
    uint32_t q24[256];
    WeightLut() {
        for (int q = 0; q < 256; ++q) {
            float x    = (float)q / -10.0f;
            float prob = 1.0f - powf(10.0f, x);
            double s   = (double)prob * 16777216.0;
            uint32_t k = (uint32_t)s;
            if ((double)k != s) { fprintf(stderr, "oracle: w(%d) is not a multiple of 2^-24\n", q); abort(); }
            q24[q] = k;
        }
    }
};
const WeightLut g_w;

// value = Q*2^-24 + m*eps  (canonicalisation (1))
inline double qm_to_f64(uint64_t q, uint64_t m, double eps) {
    return (double)q * 0x1p-24 + (double)m * eps;
}

// ---- Frag (types_structs.rs:68-112) as a view into the CSR pileup -----------------------------------
struct Pile {
    const floria_pileup* p;
    const uint32_t* order = nullptr;      // arithmetic mode 1: order[beg(r) .. end(r)) = the read's cells in the iteration order of Frag.positions
    uint32_t n() const { return p->n_reads; }
    uint32_t beg(uint32_t r) const { return p->read_off[r]; }
    uint32_t end(uint32_t r) const { return p->read_off[r + 1]; }
};

// ---- Haplotype: pos -> {allele -> count} (types_structs.rs:15) ----------------------------------------
struct Site {
    uint64_t q[FLORIA_MAX_ALLELES];   // Q24 sum (or unit count when !use_qual)
    uint8_t  present;                 // bitmask of allele keys that exist in the inner map
    bool operator==(const Site& o) const {
        if (present != o.present) return false;
        for (int a = 0; a < FLORIA_MAX_ALLELES; ++a)
            if (((present >> a) & 1) && q[a] != o.q[a]) return false;
        return true;
    }
};

// open-addressing hash map u32 -> Site (FxHash multiplier; linear probing; tombstone-free because
// the only removal pattern is "drop every key < x", done by rebuild)
class Hap {
public:
    Hap() : mask_(0), len_(0) {}
    size_t len() const { return len_; }
    Site* find(uint32_t pos) {
        if (!len_) return nullptr;
        size_t i = slot(pos);
        while (keys_[i] != EMPTY) {
            if (keys_[i] == pos) return &vals_[i];
            i = (i + 1) & mask_;
        }
        return nullptr;
    }
    const Site* find(uint32_t pos) const { return const_cast<Hap*>(this)->find(pos); }
    Site& entry(uint32_t pos) {              // entry(pos).or_insert(default)
        if ((len_ + 1) * 8 > (mask_ + 1) * 7 || keys_.empty()) grow();
        size_t i = slot(pos);
        while (keys_[i] != EMPTY) {
            if (keys_[i] == pos) return vals_[i];
            i = (i + 1) & mask_;
        }
        keys_[i] = pos;
        memset(&vals_[i], 0, sizeof(Site));
        ++len_;
        return vals_[i];
    }
    template <class F> void for_each(F f) const {
        for (size_t i = 0; i < keys_.size(); ++i)
            if (keys_[i] != EMPTY) f(keys_[i], vals_[i]);
    }
    // block_vec[i].remove(pos) for every pos < cut (types_structs.rs:346-360)
    void drop_below(uint32_t cut) {
        bool any = false;
        for (size_t i = 0; i < keys_.size(); ++i)
            if (keys_[i] != EMPTY && keys_[i] < cut) { any = true; break; }
        if (!any) return;
        std::vector<uint32_t> k; std::vector<Site> v;
        k.swap(keys_); v.swap(vals_);
        size_t cap = k.size();
        keys_.assign(cap, EMPTY); vals_.resize(cap); len_ = 0;
        for (size_t i = 0; i < cap; ++i)
            if (k[i] != EMPTY && k[i] >= cut) { entry(k[i]) = v[i]; }
    }
    bool operator==(const Hap& o) const {     // deep map equality (global_clustering.rs:124)
        if (len_ != o.len_) return false;
        bool eq = true;
        for_each([&](uint32_t pos, const Site& s) {
            if (!eq) return;
            const Site* t = o.find(pos);
            if (!t || !(*t == s)) eq = false;
        });
        return eq;
    }
private:
    static constexpr uint32_t EMPTY = 0xffffffffu;
    size_t slot(uint32_t pos) const { return (size_t)(((uint64_t)pos * 0x517cc1b727220a95ull) >> 32) & mask_; }
    void grow() {
        size_t ncap = keys_.empty() ? 8 : keys_.size() * 2;
        std::vector<uint32_t> k; std::vector<Site> v;
        k.swap(keys_); v.swap(vals_);
        keys_.assign(ncap, EMPTY); vals_.resize(ncap); mask_ = ncap - 1; len_ = 0;
        for (size_t i = 0; i < k.size(); ++i)
            if (k[i] != EMPTY) entry(k[i]) = v[i];
    }
    std::vector<uint32_t> keys_;
    std::vector<Site>     vals_;
    size_t mask_, len_;
};

struct HapBlock {                              // types_structs.rs:253-256
    std::vector<Hap> blocks;
    bool operator==(const HapBlock& o) const { return blocks == o.blocks; }
};

// sites.entry(allele).or_insert(0.) += w   (utils_frags.rs:166-172, types_structs.rs:368-373)
inline void site_add(Site& s, uint8_t allele, uint64_t w) {
    s.present |= (uint8_t)(1u << allele);
    s.q[allele] += w;
}

// ---- set_to_seq_dict / hap_block_from_partition (utils_frags.rs:160-184) ------------------------------
Hap set_to_seq_dict(const Pile& P, const std::vector<uint32_t>& frag_set, bool use_phred) {
    Hap h;
    for (uint32_t r : frag_set)
        for (uint32_t c = P.beg(r); c < P.end(r); ++c)
            site_add(h.entry(P.p->snp[c]), P.p->allele[c], use_phred ? g_w.q24[P.p->qual[c]] : 1);
    return h;
}
HapBlock hap_block_from_partition(const Pile& P, const std::vector<std::vector<uint32_t>>& part, bool use_qual) {
    HapBlock b;
    for (auto& s : part) b.blocks.push_back(set_to_seq_dict(P, s, use_qual));
    return b;
}

// ---- distance_read_haplo_epsilon_empty (utils_frags.rs:32-75) ------------------------------------------
struct SD { uint64_t same, diff, m; double same_f, diff_f; };   // same = Q24, diff = Q24 + m*eps; *_f: the running f64 sums (arithmetic mode 1)
inline SD distance_read_haplo_epsilon_empty(const Pile& P, uint32_t r, const Hap& hap, double epsilon = 0.0) {
    SD d{0, 0, 0, 0.0, 0.0};
    const bool running = P.order != nullptr;                 // (arithmetic mode 1; mode 0 — the timed CPU baseline — pays nothing for it)
    if (!running) {
        for (uint32_t c = P.beg(r); c < P.end(r); ++c) {
            const Site* s = hap.find(P.p->snp[c]);
            uint64_t mx = 0;                                   // :36-44 empty_pos <=> no non-zero count
            if (s) for (int a = 0; a < FLORIA_MAX_ALLELES; ++a) if ((s->present >> a) & 1) mx = std::max(mx, s->q[a]);
            if (mx == 0) { d.m += 1; continue; }               // :45-48 diff += epsilon
            uint8_t a = P.p->allele[c];
            uint64_t w = g_w.q24[P.p->qual[c]];
            bool has = (s->present >> a) & 1;                  // :52-71  same if the read's allele is (tied for) the consensus, else diff
            if (has && s->q[a] == mx) d.same += w; else d.diff += w;
        }
        return d;
    }
    for (uint32_t cc = P.beg(r); cc < P.end(r); ++cc) {
        const uint32_t c = P.order[cc];                      // `for pos in r.positions.iter()` (:35): an FxHashSet
        const Site* s = hap.find(P.p->snp[c]);
        uint64_t mx = 0;                                   // :36-44 empty_pos <=> no non-zero count
        if (s) for (int a = 0; a < FLORIA_MAX_ALLELES; ++a) if ((s->present >> a) & 1) mx = std::max(mx, s->q[a]);
        if (mx == 0) { d.m += 1; d.diff_f += epsilon; continue; }               // :45-48 diff += epsilon
        uint8_t a = P.p->allele[c];
        uint64_t w = g_w.q24[P.p->qual[c]];
        // :52-71  same if the read's allele is (tied for) the consensus, else diff
        bool has = (s->present >> a) & 1;
        if (has && s->q[a] == mx) { d.same += w; d.same_f += (double)w * 0x1p-24; } else { d.diff += w; d.diff_f += (double)w * 0x1p-24; }
    }
    return d;
}
// the two f64 values the reference goes on with
inline double sd_same(const SD& d, double eps) { return g_arith_mode.load(std::memory_order_relaxed) >= 1 ? d.same_f : qm_to_f64(d.same, 0, eps); }
inline double sd_diff(const SD& d, double eps) { return g_arith_mode.load(std::memory_order_relaxed) >= 1 ? d.diff_f : qm_to_f64(d.diff, d.m, eps); }

// Frag.positions (file_reader.rs:729-733) = seq_dict.keys().collect::<FxHashSet<_>>(): seq_dict is an FxHashMap filled in ascending
// order by the CIGAR walk (:661-727, growing as it goes); `collect` reserves room for all keys at once and inserts them in the map's
// iteration order; iterating the set then walks ITS buckets.  -> for every read the permutation of its cells in that order.
// (Mates and supplementary alignments merged by combine_frags extend the first alignment's set by the second's, :540-542, and --ignore-monomorphic
// removes positions from the set, utils_frags.rs:745-755: a pileup that carries such fragments says what its sets iterate like in `set_order`
// (include/floria_hip.h), which is then used as given — positions_order below restates how such a set comes about.)
inline std::vector<uint32_t> build_cell_order(const floria_pileup* p) {
    std::vector<uint32_t> ord(p->read_off[p->n_reads]);
    if (g_arith_mode.load() == 2) {                            // mode 2: running sums with every container iterated in ASCENDING key order (tests/test_py_restatement.py)
        for (uint32_t c = 0; c < (uint32_t)ord.size(); ++c) ord[c] = c;
        return ord;
    }
    if (p->set_order) {                                        // the host's own iteration order (validated as a permutation per read by validate())
        for (uint32_t r = 0; r < p->n_reads; ++r) for (uint32_t c = p->read_off[r]; c < p->read_off[r + 1]; ++c) ord[c] = p->read_off[r] + p->set_order[c];
        return ord;
    }
    for (uint32_t r = 0; r < p->n_reads; ++r) {
        const uint32_t b = p->read_off[r], e = p->read_off[r + 1];
        FxSet seq_dict, positions;
        for (uint32_t c = b; c < e; ++c) seq_dict.insert(p->snp[c]);
        positions.reserve(e - b);
        seq_dict.for_each([&](uint64_t pos) { positions.insert(pos); });
        uint32_t k = b;
        positions.for_each([&](uint64_t pos) { ord[k++] = (uint32_t)(std::lower_bound(p->snp + b, p->snp + e, (uint32_t)pos) - p->snp); });
    }
    return ord;
}

// ---- stable_binom_cdf_p_rev / log_sum_exp (utils_frags.rs:211-258) ---------------------------------------
double stable_binom_cdf_p_rev(uint64_t n, uint64_t k, double p, double div_factor) {
    if (n == 0) return 0.0;
    double n64 = (double)n, k64 = (double)k;
    double a = k64 / n64;
    if (a == 1.0) a = 0.9999999;
    if (a == 0.0) a = 0.0000001;
    double rel_ent = a * std::log(a / p) + (1.0 - a) * std::log((1.0 - a) / (1.0 - p));
    if (a < p) rel_ent = -rel_ent;
    return -1.0 * n64 / div_factor * rel_ent;
}
double log_sum_exp(const std::vector<double>& probs) {
    double mx = std::numeric_limits<double>::quiet_NaN();
    for (double v : probs) mx = std::isnan(mx) ? v : (std::isnan(v) ? mx : std::max(mx, v));   // f64::max ignores NaN
    double sum = 0.0;
    for (double v : probs) sum += std::exp(v - mx);
    return mx + std::log(sum);
}
inline uint64_t f64_as_usize(double x) {           // Rust `as usize`: truncating, saturating, NaN -> 0
    if (!(x > 0.0)) return 0;
    if (x >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)x;
}

// ---- std::collections::BinaryHeap on (score, payload) ------------------------------
// Ordering of (Rc<SearchNode>, HapBlock): SearchNode::cmp = score (types_structs.rs:127-131), then
// HapBlock::cmp = blocks.len() (:258-262, always equal) => ties compare Equal.
struct HeapItem {
    double   score;
    int      node;       // index into the node arena
    HapBlock block;
};
struct BinaryHeap {
    std::vector<HeapItem> data;
    static bool le(const HeapItem& a, const HeapItem& b) { return a.score <= b.score; }
    static bool lt(const HeapItem& a, const HeapItem& b) { return a.score <  b.score; }
    static bool ge(const HeapItem& a, const HeapItem& b) { return a.score >= b.score; }
    size_t len() const { return data.size(); }
    size_t sift_up(size_t start, size_t pos) {
        HeapItem hole = std::move(data[pos]);
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (le(hole, data[parent])) break;
            data[pos] = std::move(data[parent]);
            pos = parent;
        }
        data[pos] = std::move(hole);
        return pos;
    }
    void push(HeapItem&& it) {
        size_t old_len = data.size();
        data.push_back(std::move(it));
        sift_up(0, old_len);
    }
    void sift_down_to_bottom(size_t pos) {
        size_t end = data.size(), start = pos;
        HeapItem hole = std::move(data[pos]);
        size_t child = 2 * pos + 1;
        while (child <= (end >= 2 ? end - 2 : 0)) {
            child += le(data[child], data[child + 1]) ? 1 : 0;
            data[pos] = std::move(data[child]);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { data[pos] = std::move(data[child]); pos = child; }
        data[pos] = std::move(hole);
        sift_up(start, pos);
    }
    void pop() {                                  // discards the max (worst MEC) element
        HeapItem item = std::move(data.back());
        data.pop_back();
        if (!data.empty()) { std::swap(item, data[0]); sift_down_to_bottom(0); }
    }
    void sift_down_range(size_t pos, size_t end) {
        HeapItem hole = std::move(data[pos]);
        size_t child = 2 * pos + 1;
        while (child <= (end >= 2 ? end - 2 : 0)) {
            child += le(data[child], data[child + 1]) ? 1 : 0;
            if (ge(hole, data[child])) { data[pos] = std::move(hole); return; }
            data[pos] = std::move(data[child]);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1 && lt(hole, data[child])) { data[pos] = std::move(data[child]); pos = child; }
        data[pos] = std::move(hole);
    }
    void into_sorted_vec() {
        size_t end = data.size();
        while (end > 1) { --end; std::swap(data[0], data[end]); sift_down_range(0, end); }
    }
};

// ---- SearchNode (types_structs.rs:114-125) -------------------------------------------------------------
struct SearchNode {
    uint32_t read; int part; int parent;
    uint64_t diff_q, diff_m;       // sum over partitions of error_vec[k].1 ("mec", global_clustering.rs:202)
    double   score;
    double   errv[FLORIA_MAX_PLOIDY];      // arithmetic mode 1: error_vec[k].1 as running f64 sums (:196-200)
};

// ---- build_truncated_hap_block (types_structs.rs:326-376); broken_blocks bookkeeping (:340-366) is dead
// w.r.t. results because WEIRD_SPLIT=false (graph_processing.rs:166-181) -------------------------------------
HapBlock build_truncated_hap_block(const Pile& P, const HapBlock& block, uint32_t r, int part, uint32_t current_startpos) {
    HapBlock nb; nb.blocks = block.blocks;                       // manual deepcopy :337
    for (auto& h : nb.blocks) h.drop_below(current_startpos);    // :356-358
    for (uint32_t c = P.beg(r); c < P.end(r); ++c)               // :368-373
        site_add(nb.blocks[part].entry(P.p->snp[c]), P.p->allele[c], g_w.q24[P.p->qual[c]]);
    return nb;
}

// ---- beam_search_phasing (global_clustering.rs:10-179) with read_to_node_value (:181-208) --------------------
// clique = vec![FxHashSet::default(); ploidy] (graph_processing.rs:141) => frag_in_clique is always false.
void beam_search_phasing(const Pile& P, const std::vector<uint32_t>& all_reads, int ploidy, double epsilon,
                         double div_factor, double cutoff_value, size_t max_number_solns,
                         std::vector<std::vector<uint32_t>>& partition, double* min_margin, std::vector<FxSet>* shadow = nullptr) {
    partition.assign(ploidy, {});
    if (all_reads.empty()) return;                                                    // :24-26
    std::vector<SearchNode> arena;
    { SearchNode root{}; root.read = all_reads[0]; root.part = -1; root.parent = -1; arena.push_back(root); }                     // first_node :34-44
    BinaryHeap heap;
    { HeapItem it; it.score = 0.0; it.node = 0; it.block.blocks.assign(ploidy, Hap()); heap.push(std::move(it)); }
    std::vector<double> p_value_list(ploidy);
    std::vector<SD> sd(ploidy);
    for (size_t i = 0; i < all_reads.size(); ++i) {
        size_t max_num_soln_mut = max_number_solns;
        if (i < 25) max_num_soln_mut = (size_t)ploidy * max_number_solns;              // :50-53
        BinaryHeap next;
        uint32_t frag = all_reads[i];
        uint32_t current_startpos = P.p->first[frag];
        for (size_t h = 0; h < heap.data.size(); ++h) {                                // heap.iter() = Vec order :71
            const HeapItem& cur = heap.data[h];
            for (int k = 0; k < ploidy; ++k) {                                         // :74-91
                sd[k] = distance_read_haplo_epsilon_empty(P, frag, cur.block.blocks[k], epsilon);
                double same = sd_same(sd[k], epsilon);
                double diff = sd_diff(sd[k], epsilon);
                p_value_list[k] = 1.0 * stable_binom_cdf_p_rev(f64_as_usize(same + diff), f64_as_usize(diff), epsilon, div_factor);
            }
            double lse = log_sum_exp(p_value_list);                                    // :93
            for (int j = 0; j < ploidy; ++j) {
                double margin = (p_value_list[j] - lse) - cutoff_value;
                if (min_margin && std::fabs(margin) < *min_margin) *min_margin = std::fabs(margin);
                if (p_value_list[j] - lse > cutoff_value) {                            // :98
                    // read_to_node_value :181-208 (recomputes the same distance :193)
                    const SearchNode& pn = arena[cur.node];
                    SearchNode nn = pn;
                    nn.read = frag; nn.part = j; nn.parent = cur.node;
                    nn.diff_q = pn.diff_q + sd[j].diff;
                    nn.diff_m = pn.diff_m + sd[j].m;
                    nn.score  = qm_to_f64(nn.diff_q, nn.diff_m, epsilon);               // new_node_score = -(-mec) :105
                    if (g_arith_mode.load(std::memory_order_relaxed) >= 1) {
                        nn.errv[j] = pn.errv[j] + sd[j].diff_f;                         // node.error_vec[i].1 + diff :198
                        double mec = 0.0;                                               // new_error_vec.iter().map(|x| x.1).sum() :202
                        for (int k = 0; k < ploidy; ++k) mec += nn.errv[k];
                        nn.score = mec;
                    }
                    HapBlock new_block = build_truncated_hap_block(P, cur.block, frag, j, current_startpos);   // :118
                    bool project_exists = false;                                       // :122-127
                    for (const HeapItem& e : next.data)
                        if (e.block == new_block && e.score >= nn.score) project_exists = true;
                    if (!project_exists) {
                        arena.push_back(nn);
                        HeapItem it; it.score = nn.score; it.node = (int)arena.size() - 1; it.block = std::move(new_block);
                        next.push(std::move(it));                                      // :130
                        if (next.len() > max_num_soln_mut) next.pop();                 // :132-134
                    }
                }
            }
        }
        heap.data.swap(next.data);                                                     // :145
    }
    heap.into_sorted_vec();                                                            // :149
    int np = heap.data[0].node;                                                        // :150
    if (shadow) shadow->assign(ploidy, FxSet());
    while (arena[np].parent >= 0) {                                                    // :155-176
        partition[arena[np].part].push_back(arena[np].read);
        if (shadow) (*shadow)[arena[np].part].insert(arena[np].read);                  // the reference's sets, in traceback order (:168)
        np = arena[np].parent;
    }
    for (auto& s : partition) std::sort(s.begin(), s.end());                            // canonical set order
}

// ---- get_mec_stats_epsilon (local_clustering.rs:218-260, use_gaps=true) and _no_phred (:187-215) ---------
struct QM { uint64_t bases, errors, m; double errors_f; };       // errors_f: the running f64 sum (arithmetic mode 1)
// part_order (arithmetic mode 1): the reads of every partition in the iteration order of its FxHashSet.  set_to_seq_dict
// (utils_frags.rs:160-176) fills the haplotype's FxHashMap position by position in that order (reads x their own position order), and
// `for seq_dict in hap.values()` (:227) walks the map's buckets: the order in which `errors` receives its terms.
std::vector<QM> mec_stats_of_block(const HapBlock& hb, uint64_t one, const Pile* P = nullptr,
                                   const std::vector<std::vector<uint32_t>>* part_order = nullptr, double epsilon = 0.0) {
    std::vector<QM> v;
    const double scale = one == 1 ? 1.0 : 0x1p-24;
    for (size_t pi = 0; pi < hb.blocks.size(); ++pi) {
        const Hap& hap = hb.blocks[pi];
        QM s{0, 0, 0, 0.0};
        auto site_terms = [&](const Site& site, bool running) {
            if (!site.present) return;                           // allele_counts.is_empty() -> continue
            uint64_t mx = 0, tot = 0;
            for (int a = 0; a < FLORIA_MAX_ALLELES; ++a) if ((site.present >> a) & 1) { mx = std::max(mx, site.q[a]); tot += site.q[a]; }
            if (!running) {
                s.bases += mx;                                       // cons_bases = last after sort
                s.errors += tot - mx;                                // all but the last
                if (mx <= one) s.m += 1;                             // cons_bases <= 1. -> errors += epsilon
            } else {
                uint64_t vals[FLORIA_MAX_ALLELES]; int nv = 0;
                for (int a = 0; a < FLORIA_MAX_ALLELES; ++a) if ((site.present >> a) & 1) vals[nv++] = site.q[a];
                std::sort(vals, vals + nv);                          // allele_counts.sort_by(count) (:244); equal counts add equal terms
                for (int i = 0; i + 1 < nv; ++i) s.errors_f += (double)vals[i] * scale;      // :248-250
                if (mx <= one) s.errors_f += epsilon;                // :251-253
            }
        };
        hap.for_each([&](uint32_t, const Site& site) { site_terms(site, false); });
        if (P && P->order && part_order) {
            FxSet pos_map;
            for (uint32_t r : (*part_order)[pi])
                for (uint32_t cc = P->beg(r); cc < P->end(r); ++cc) pos_map.entry_insert(P->p->snp[P->order[cc]]);      // hap_map.entry(*pos).or_insert(..) (utils_frags.rs:165)
            if (g_arith_mode.load(std::memory_order_relaxed) == 2) {       // ascending positions
                std::vector<uint32_t> ps;
                pos_map.for_each([&](uint64_t pos) { ps.push_back((uint32_t)pos); });
                std::sort(ps.begin(), ps.end());
                for (uint32_t pos : ps) { const Site* site = hap.find(pos); if (site) site_terms(*site, true); }
            } else
            pos_map.for_each([&](uint64_t pos) { const Site* site = hap.find((uint32_t)pos); if (site) site_terms(*site, true); });
        }
        v.push_back(s);
    }
    return v;
}
inline std::vector<std::vector<uint32_t>> set_orders(const std::vector<std::vector<uint32_t>>& partition, const std::vector<FxSet>* shadow) {
    if (!shadow) return partition;
    std::vector<std::vector<uint32_t>> o;
    for (const FxSet& f : *shadow) o.push_back(f.order());
    return o;
}

// ---- opt_iterate (local_clustering.rs:292-358) -------------------------------------------------------------
struct Move { double gain; int i; uint32_t read; int j; };
std::vector<std::vector<uint32_t>> opt_iterate(const Pile& P, const std::vector<std::vector<uint32_t>>& partition,
                                               const HapBlock& hap_block, double epsilon, const std::vector<FxSet>* shadow = nullptr,
                                               std::vector<FxSet>* new_shadow = nullptr) {
    int ploidy = (int)partition.size();
    std::vector<Move> best_moves;
    for (int i = 0; i < ploidy; ++i) {
        if (partition[i].size() <= 1) continue;                                          // :300-302
        std::vector<uint32_t> order_i(partition[i]);
        if (shadow) order_i = (*shadow)[i].order();                                      // mode 2: the emulated FxHashSet's bucket order (:304)
        else if (g_order_mode.load() == 1) std::reverse(order_i.begin(), order_i.end());
        for (uint32_t read : order_i) {                                                  // canonical: ascending id (2)
            SD own = distance_read_haplo_epsilon_empty(P, read, hap_block.blocks[i], epsilon);
            double errors_read = sd_diff(own, epsilon);
            for (int j = 0; j < ploidy; ++j) {
                if (j == i) continue;
                SD oth = distance_read_haplo_epsilon_empty(P, read, hap_block.blocks[j], epsilon);
                double read_errors_movej = sd_diff(oth, epsilon);
                double diff_score = errors_read - read_errors_movej;                     // :320
                if (diff_score > 0.0) best_moves.push_back(Move{diff_score, i, read, j});
            }
        }
    }
    std::vector<std::vector<uint32_t>> new_part = partition;                               // :329
    std::vector<size_t> sizes(ploidy);
    for (int i = 0; i < ploidy; ++i) sizes[i] = partition[i].size();
    std::stable_sort(best_moves.begin(), best_moves.end(), [](const Move& a, const Move& b) { return a.gain > b.gain; });   // :330
    size_t number_of_moves = best_moves.size() / 10;                                       // :336
    if (number_of_moves == 0 && !best_moves.empty()) number_of_moves = best_moves.size() / 3 + 1;
    std::vector<uint8_t> moved(P.n(), 0);   // moved_reads
    std::vector<std::pair<uint32_t, std::pair<int, int>>> applied;
    for (size_t mv_num = 0; mv_num < best_moves.size(); ++mv_num) {                         // :342-356
        const Move& mv = best_moves[mv_num];
        if (moved[mv.read]) continue;
        if (sizes[mv.i] == 1) continue;
        applied.push_back({mv.read, {mv.i, mv.j}});
        sizes[mv.j] += 1; sizes[mv.i] -= 1;
        moved[mv.read] = 1;
        if (mv_num > number_of_moves) break;
    }
    if (shadow && new_shadow) {                                                            // new_part = partition.clone(); insert then remove (:329,349-350)
        *new_shadow = *shadow;
        for (auto& a : applied) { (*new_shadow)[a.second.second].insert(a.first); (*new_shadow)[a.second.first].remove(a.first); }
    }
    for (auto& a : applied) {
        auto& src = new_part[a.second.first];
        src.erase(std::lower_bound(src.begin(), src.end(), a.first));
        auto& dst = new_part[a.second.second];
        dst.insert(std::lower_bound(dst.begin(), dst.end(), a.first), a.first);
    }
    return new_part;
}

// ---- optimize_clustering (local_clustering.rs:71-130) -------------------------------------------------------
double mec_score_of(const std::vector<QM>& v, double epsilon) {
    double s = 0.0;                                   // binom_vec.iter().map(|x| x.1).sum()
    const bool running = g_arith_mode.load(std::memory_order_relaxed) >= 1;
    for (const QM& x : v) s += running ? x.errors_f : qm_to_f64(x.errors, x.m, epsilon);
    return s * -1.0;
}
std::vector<std::vector<uint32_t>> optimize_clustering(const Pile& P, std::vector<std::vector<uint32_t>> partition,
                                                       double epsilon, int max_iters, int* iters_done, std::vector<FxSet>* shadow = nullptr) {
    bool not_empty = false;
    for (auto& p : partition) if (!p.empty()) not_empty = true;
    if (iters_done) *iters_done = 0;
    if (!not_empty) return partition;                                                    // :82-85
    HapBlock prev_hap_block = hap_block_from_partition(P, partition, true);
    const bool running = g_arith_mode.load(std::memory_order_relaxed) >= 1 && P.order;
    auto stats_of = [&](const HapBlock& hb, const std::vector<std::vector<uint32_t>>& part, const std::vector<FxSet>* sh) {
        if (!running) return mec_stats_of_block(hb, 1u << 24);
        const auto ord = set_orders(part, sh);
        return mec_stats_of_block(hb, 1u << 24, &P, &ord, epsilon);
    };
    double prev_score = mec_score_of(stats_of(prev_hap_block, partition, shadow), epsilon);   // :97-99
    std::vector<std::vector<uint32_t>> best_part = std::move(partition);
    for (int i = 0; i < max_iters; ++i) {                                                // :105-127
        std::vector<FxSet> new_shadow;
        auto new_part = opt_iterate(P, best_part, prev_hap_block, epsilon, shadow, shadow ? &new_shadow : nullptr);
        HapBlock new_block = hap_block_from_partition(P, new_part, true);
        double new_score = mec_score_of(stats_of(new_block, new_part, shadow ? &new_shadow : nullptr), epsilon);
        if (iters_done) *iters_done = i + 1;
        if (new_score > prev_score) { prev_score = new_score; best_part = std::move(new_part); prev_hap_block = std::move(new_block); if (shadow) *shadow = std::move(new_shadow); }
        else return best_part;
    }
    return best_part;
}

// ---- find_reads_in_interval (local_clustering.rs:12-59, max_num_reads = usize::MAX) --------------------------
std::vector<uint32_t> find_reads_in_interval(const Pile& P, uint32_t start, uint32_t end) {
    std::vector<uint32_t> out;
    for (uint32_t r = 0; r < P.n(); ++r) {
        if (P.p->last[r] < start) continue;                        // :36-38
        if (P.p->first[r] > end) break;                            // :39-41
        if (P.p->last[r] - P.p->first[r] > 10000) continue;        // :44-46
        out.push_back(r);
    }
    return out;   // ascending counter_id == Frag::cmp order (reads.sort(), graph_processing.rs:139)
}

// ---- get_local_hap_blocks (graph_processing.rs:103-304) -------------------------------------------------------
struct BlockOut { uint32_t best_ploidy = 0, tried = 0; std::vector<uint32_t> reads; std::vector<uint8_t> part; std::vector<double> mec; double min_margin;
                  std::vector<uint32_t> set_order; };     // mode 2: read ids partition by partition, each in its emulated set's iteration order

void get_local_hap_blocks(const Pile& P, uint32_t start, uint32_t end, const floria_params& o, BlockOut& out) {
    const int max_ploidy = (int)o.max_ploidy;
    const double epsilon = o.epsilon;
    out.mec.assign(max_ploidy, 0.0);
    out.min_margin = std::numeric_limits<double>::infinity();
    std::vector<double> expected_errors_ref;
    std::vector<std::vector<std::vector<uint32_t>>> parts_vector;
    std::vector<std::vector<FxSet>> shadow_vector;
    std::vector<uint32_t> reads = find_reads_in_interval(P, start, end);                  // :121-126
    out.reads = reads;
    if (reads.empty()) { out.best_ploidy = 0; return; }                                   // :129-131 -> None
    int best_ploidy = 1;
    const double cutoff = std::log(PROB_CUTOFF);                                          // :146
    for (int ploidy = 1; ploidy <= max_ploidy; ++ploidy) {                                // :132
        best_ploidy = ploidy;
        out.tried = ploidy;
        double num_alleles = 0.0;
        std::vector<std::vector<uint32_t>> part;
        std::vector<FxSet> shadow;
        const bool emu = g_order_mode.load() == 2;
        beam_search_phasing(P, reads, ploidy, epsilon, DIV_FACTOR, cutoff, o.beam, part, &out.min_margin, emu ? &shadow : nullptr);   // :140-151
        auto optimized_part = optimize_clustering(P, std::move(part), epsilon, NUM_ITER_OPTIMIZE, nullptr, emu ? &shadow : nullptr);  // :153-154
        if (emu) shadow_vector.push_back(shadow);
        HapBlock np = hap_block_from_partition(P, optimized_part, false);                   // :156 (_no_phred)
        const bool running = g_arith_mode.load(std::memory_order_relaxed) >= 1 && P.order;
        const auto final_orders = running ? set_orders(optimized_part, emu ? &shadow : nullptr) : std::vector<std::vector<uint32_t>>();
        for (const QM& s : (running ? mec_stats_of_block(np, 1, &P, &final_orders, epsilon) : mec_stats_of_block(np, 1))) {   // :158-162
            double good = (double)s.bases;
            double bad  = running ? s.errors_f : (double)s.errors + (double)s.m * epsilon;
            out.mec[ploidy - 1] += bad;
            num_alleles += good;
            num_alleles += bad;
        }
        parts_vector.push_back(optimized_part);
        expected_errors_ref.push_back(num_alleles * epsilon);                               // :196
        if (ploidy > 1) {                                                                   // :198-246
            double mec_threshold;
            if (o.ploidy_sensitivity == 1)      mec_threshold = 1.0 / (1.0 - epsilon) / (1.0 + 1.0 / (std::pow((double)ploidy, 0.50) + 1.00));
            else if (o.ploidy_sensitivity == 2) mec_threshold = 1.0 / (1.0 - epsilon) / (1.0 + 1.0 / (std::pow((double)ploidy, 1.00) + 1. / 3.));
            else                                mec_threshold = 1.0 / (1.0 - epsilon) / (1.0 + 1.0 / (std::pow((double)ploidy, 1.00) + 1.00));
            if ((out.mec[ploidy - 1] / out.mec[ploidy - 2]) < mec_threshold) { /* do nothing */ }
            else if (o.stopping_heuristic) { best_ploidy -= 1; break; }
            if (out.mec[ploidy - 1] < expected_errors_ref[ploidy - 1]) break;
        } else {
            if (out.mec[ploidy - 1] < expected_errors_ref[ploidy - 1]) break;               // :247-250
        }
    }
    out.best_ploidy = best_ploidy;
    if (!shadow_vector.empty()) {                 // mode 2: the iteration order of every final partition set (HapNode.frag_set, types_structs.rs:156)
        out.set_order.clear();
        for (const FxSet& fs : shadow_vector[best_ploidy - 1]) { const auto o2 = fs.order(); out.set_order.insert(out.set_order.end(), o2.begin(), o2.end()); }
    }
    const auto& bp = parts_vector[best_ploidy - 1];                                         // :268
    out.part.assign(reads.size(), 0);
    for (int k = 0; k < (int)bp.size(); ++k)
        for (uint32_t r : bp[k]) {
            size_t idx = std::lower_bound(reads.begin(), reads.end(), r) - reads.begin();
            out.part[idx] = (uint8_t)k;
        }
}

// ---- get_range_with_lengths (utils_frags.rs:405-463) ----------------------------------------------------------
int range_with_lengths(const uint64_t* g, uint32_t n, uint64_t block_length, uint64_t overlap_len, double minimal_density,
                       std::vector<std::pair<uint32_t, uint32_t>>& ret) {
    ret.clear();
    if (n == 0) { g_err = "empty snp_to_genome_pos (reference panics on index 0)"; return FLORIA_E_INVALID; }
    uint64_t cum_pos = 0, last_pos = g[0];
    uint32_t left_endpoint = 0, new_left_end = 0;
    bool hit_new_left = false;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t pos = g[i];
        if (i == n - 1) { ret.push_back({left_endpoint, i}); break; }
        if (pos < last_pos) { g_err = "VCF malformed. Positions are not increasing"; return FLORIA_E_INVALID; }
        cum_pos += pos - last_pos;
        last_pos = pos;
        if (cum_pos > block_length - overlap_len && !hit_new_left) { new_left_end = i; hit_new_left = true; }
        if (cum_pos > block_length) {
            cum_pos = 0;
            double snp_density = (double)(i - left_endpoint) / (double)block_length;
            if (snp_density > minimal_density) ret.push_back({left_endpoint, i - 1});
            if (g[new_left_end] + block_length < g[new_left_end + 1]) left_endpoint = new_left_end;
            else left_endpoint = new_left_end + 1;
            last_pos = g[left_endpoint];
            hit_new_left = false;
        }
    }
    for (auto& x : ret) { x.first += 1; x.second += 1; }
    return 0;
}

// ---- process_reads_for_final_parts (part_block_manip.rs:174-274), reassign_short=false -----------------------------
void add_read_to_block(const Pile& P, Hap& h, uint32_t r) {                              // utils_frags.rs:465-474
    for (uint32_t c = P.beg(r); c < P.end(r); ++c)
        site_add(h.entry(P.p->snp[c]), P.p->allele[c], g_w.q24[P.p->qual[c]]);
}
void remove_read_from_block(const Pile& P, Hap& h, uint32_t r) {                         // utils_frags.rs:476-490
    for (uint32_t c = P.beg(r); c < P.end(r); ++c) {
        Site& s = h.entry(P.p->snp[c]);
        uint8_t a = P.p->allele[c];
        if (!((s.present >> a) & 1)) { s.present |= (uint8_t)(1u << a); s.q[a] = 0; }    // or_insert(0.)
        if (s.q[a] != 0) s.q[a] -= g_w.q24[P.p->qual[c]];
        if (s.q[a] == 0) { s.present &= (uint8_t)~(1u << a); }                           // <= 0 -> remove key (exact arithmetic: never negative)
    }
}

// Which read a split DROPS (:69-84: the first read behind a coverage gap trips the split and is inserted nowhere) depends on the order of the reads that share
// its first_position: the reference iterates an FxHashSet and stable-sorts by first_position only (:34-35, :62-63).  Test-only modes to measure how much that
// matters: 0 = ascending counter_id (the canonical order, what the product does), 1 = descending counter_id among equal first positions,
// 2 = the iteration order of an FxHashSet filled in the order process_reads_for_final_parts re-inserted the reads (:219 — an approximation: the reference's set
// still has the buckets and tombstones of the haplogroup it was before its reads were removed, :195-200, which the seam does not carry).
std::atomic<int> g_a14_tie_mode{0};
uint64_t g_a14_dropped = 0;        // reads dropped by splits since the last reset (diagnostic)
void separate_broken_haplogroups(const Pile& P, std::vector<std::vector<uint32_t>>& parts_in,
                                 std::vector<std::pair<uint32_t, uint32_t>>& ranges, uint64_t* n_dropped = nullptr) {     // part_block_manip.rs:27-98
    // parts_in[i]: the reads of haplogroup i in the order they were inserted; `parts` = the order the reference's loops see: set order, stable-sorted by first_position
    const int tie_mode = g_a14_tie_mode.load();
    std::vector<std::vector<uint32_t>> parts(parts_in.size());
    for (size_t i = 0; i < parts_in.size(); ++i) {
        std::vector<uint32_t> v = parts_in[i];
        if (tie_mode == 2) { FxSet fs; for (uint32_t r : v) fs.insert(r); v = fs.order(); }
        else { std::sort(v.begin(), v.end()); if (tie_mode == 1) std::reverse(v.begin(), v.end()); }
        std::stable_sort(v.begin(), v.end(), [&](uint32_t a, uint32_t b) { return P.p->first[a] < P.p->first[b]; });
        parts[i] = std::move(v);
        std::sort(parts_in[i].begin(), parts_in[i].end());
    }
    std::vector<std::pair<size_t, std::vector<uint32_t>>> all_breaks;
    for (size_t i = 0; i < ranges.size(); ++i) {
        uint32_t current_lastest_pos = 0;
        std::vector<uint32_t> breaks;
        for (uint32_t r : parts[i]) {
            if (current_lastest_pos != 0 && P.p->first[r] > current_lastest_pos)
                if (current_lastest_pos >= ranges[i].first && current_lastest_pos < ranges[i].second) breaks.push_back(current_lastest_pos);
            if (P.p->last[r] > current_lastest_pos) current_lastest_pos = P.p->last[r];
        }
        if (!breaks.empty()) all_breaks.push_back({i, breaks});
    }
    std::vector<std::vector<uint32_t>> new_parts;
    std::vector<std::pair<uint32_t, uint32_t>> new_ranges;
    for (auto& bi : all_breaks) {
        size_t spot_index = 0;
        auto& break_spots = bi.second;
        uint32_t break_start = ranges[bi.first].first;
        uint32_t end_spot = break_spots[spot_index];
        std::vector<uint32_t> new_part;
        for (uint32_t r : parts[bi.first]) {
            if (P.p->last[r] <= end_spot) new_part.push_back(r);
            else {                                                   // :69-84 — the read that trips the split is dropped
                if (n_dropped) ++*n_dropped;
                new_parts.push_back(new_part);
                new_ranges.push_back({break_start, end_spot});
                break_start = end_spot + 1;
                spot_index += 1;
                if (spot_index != break_spots.size()) end_spot = break_spots[spot_index];
                else end_spot = UINT32_MAX;
                new_part.clear();
            }
        }
        new_parts.push_back(new_part);
        new_ranges.push_back({break_start, ranges[bi.first].second});
    }
    for (auto& bi : all_breaks) parts_in[bi.first].clear();
    for (size_t i = 0; i < new_parts.size(); ++i) { std::sort(new_parts[i].begin(), new_parts[i].end()); parts_in.push_back(new_parts[i]); ranges.push_back(new_ranges[i]); }
}

void process_reads_for_final_parts(const Pile& P, std::vector<std::vector<uint32_t>>& parts,
                                   std::vector<std::pair<uint32_t, uint32_t>>& ranges, double epsilon,
                                   const uint32_t* read_order = nullptr, uint32_t n_order = 0) {
    HapBlock block = hap_block_from_partition(P, parts, true);                              // :184
    std::vector<std::vector<uint32_t>> read_to_parts(P.n());                                 // :185-193
    for (size_t i = 0; i < parts.size(); ++i)
        for (uint32_t r : parts[i]) {
            auto& v = read_to_parts[r];
            if (std::find(v.begin(), v.end(), (uint32_t)i) == v.end()) v.push_back((uint32_t)i);
        }
    for (uint32_t r = 0; r < P.n(); ++r)                                                     // :195-200
        for (uint32_t id : read_to_parts[r]) remove_read_from_block(P, block.blocks[id], r);
    for (auto& p : parts) p.clear();
    const uint32_t n_visit = read_order ? n_order : P.n();
    for (uint32_t rr = 0; rr < n_visit; ++rr) {                                               // :203-222, canonical order (2) unless the caller gives the map's order
        const uint32_t r = read_order ? read_order[rr] : (g_order_mode.load() ? P.n() - 1 - rr : rr);
        if (read_to_parts[r].empty()) continue;
        bool have = false; double bd = 0, bs = 0; uint32_t bid = 0;
        for (uint32_t id : read_to_parts[r]) {
            SD d = distance_read_haplo_epsilon_empty(P, r, block.blocks[id], epsilon);
            double key_d = sd_diff(d, epsilon) + 1.;                                         // (diff + 1., id, same)
            double key_s = sd_same(d, epsilon);
            bool less = !have || key_d < bd || (key_d == bd && (id < bid || (id == bid && key_s < bs)));
            if (less) { have = true; bd = key_d; bs = key_s; bid = id; }
        }
        parts[bid].push_back(r);
        add_read_to_block(P, block.blocks[bid], r);
    }
    if (g_a14_tie_mode.load() < 0) { for (auto& pp : parts) std::sort(pp.begin(), pp.end()); return; }      // (tests: the haplogroups as re-inserted, before :231-233 and sort_parts)
    separate_broken_haplogroups(P, parts, ranges, &g_a14_dropped);                            // :231-233 (sorts every haplogroup by read id on its way out)
    // sort_parts :276-288 — stable sort by range
    std::vector<size_t> idx(parts.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ranges[a] < ranges[b]; });
    std::vector<std::vector<uint32_t>> np; std::vector<std::pair<uint32_t, uint32_t>> nr;
    for (size_t i : idx) { np.push_back(parts[i]); nr.push_back(ranges[i]); }
    parts.swap(np); ranges.swap(nr);
}

// ---- HapNode::new (types_structs.rs:168-209) and update_hap_graph (graph_processing.rs:22-100) -----------------------
// hap_map: phred histogram of the node's reads restricted to the block's SNP endpoints (:172-177);
// cov = allele counts sorted ascending, element [len*2/3] (:187-193).
struct NodeO { std::vector<uint32_t> frags; Hap hap_map; double cov; };
NodeO hap_node_new(const Pile& P, const std::vector<uint32_t>& frag_set, uint32_t lo, uint32_t hi) {
    NodeO n; n.frags = frag_set;
    for (uint32_t r : frag_set)
        for (uint32_t c = P.beg(r); c < P.end(r); ++c) {
            const uint32_t pos = P.p->snp[c];
            if (pos <= hi && pos >= lo) site_add(n.hap_map.entry(pos), P.p->allele[c], g_w.q24[P.p->qual[c]]);
        }
    std::vector<uint64_t> counts;
    n.hap_map.for_each([&](uint32_t, const Site& s) { for (int a = 0; a < FLORIA_MAX_ALLELES; ++a) if ((s.present >> a) & 1) counts.push_back(s.q[a]); });
    std::sort(counts.begin(), counts.end());
    n.cov = counts.empty() ? 0.0 : (double)counts[counts.size() * 2 / 3] * 0x1p-24;
    return n;
}
// distance_read_haplo (utils_frags.rs:77-108): only `diff` is consumed by update_hap_graph (:35 `_same`); a tie between the
// read's allele and the consensus never counts as diff (:95-101); result is rounded (:107)
uint64_t distance_read_haplo_diff(const Pile& P, uint32_t r, const Hap& hap) {
    uint64_t diff = 0;
    for (uint32_t c = P.beg(r); c < P.end(r); ++c) {
        const Site* s = hap.find(P.p->snp[c]);
        if (!s) continue;                                               // :80-82
        uint64_t mx = 0;
        for (int a = 0; a < FLORIA_MAX_ALLELES; ++a) if ((s->present >> a) & 1) mx = std::max(mx, s->q[a]);
        const uint8_t a = P.p->allele[c];
        const bool has = (s->present >> a) & 1;
        if (has && s->q[a] == mx) continue;                             // consensus or tied with it
        diff += g_w.q24[P.p->qual[c]];
    }
    return (diff + (1ull << 23)) >> 24;                                 // f64::round, non-negative
}

int validate(const floria_pileup* p) {
    if (!p || (p->n_reads && (!p->read_off || !p->snp || !p->allele || !p->qual || !p->first || !p->last))) { g_err = "null pileup field"; return FLORIA_E_INVALID; }
    for (uint32_t r = 0; r < p->n_reads; ++r) {
        uint32_t b = p->read_off[r], e = p->read_off[r + 1];
        if (e <= b) { g_err = "read with no cells"; return FLORIA_E_INVALID; }
        if (p->snp[b] != p->first[r] || p->snp[e - 1] != p->last[r]) { g_err = "first/last do not match cells"; return FLORIA_E_INVALID; }
        for (uint32_t c = b; c < e; ++c) {
            if (c > b && p->snp[c] <= p->snp[c - 1]) { g_err = "cells not strictly ascending"; return FLORIA_E_INVALID; }
            if (p->allele[c] >= FLORIA_MAX_ALLELES) { g_err = "allele index > 3"; return FLORIA_E_UNSUPPORTED; }
        }
        if (r > 0) {   // Frag::cmp order (types_structs.rs:87-93)
            bool ok = p->first[r - 1] < p->first[r] || (p->first[r - 1] == p->first[r] && p->last[r - 1] >= p->last[r]);
            if (!ok) { g_err = "reads not sorted by Frag::cmp"; return FLORIA_E_INVALID; }
        }
        if (p->set_order) {
            std::vector<uint8_t> seen(e - b, 0);
            for (uint32_t c = b; c < e; ++c) {
                const uint32_t x = p->set_order[c];
                if (x >= e - b || seen[x]) { g_err = "set_order of read " + std::to_string(r) + " is not a permutation of the indices of its cells"; return FLORIA_E_INVALID; }
                seen[x] = 1;
            }
        }
    }
    return 0;
}

// Frag.positions of a fragment that combine_frags merged from several alignments (file_reader.rs:491-659) and / or that --ignore-monomorphic cut down
// (utils_frags.rs:745-755), as the reference's containers build it:
//   * every alignment's own set: seq_dict (an FxHashMap growing as the CIGAR walk inserts ascending SNP positions, :702-727) -> keys().collect(), i.e. an
//     empty set extended by the map's keys in the map's bucket order (`HashSet::from_iter` = extend on an empty set: reserve(size_hint) = all keys at once;
//     an alignment without SNPs leaves the unallocated empty set);
//   * `first.positions.extend(other.positions)` (:541, :639) in merge order: hashbrown's Extend reserves the whole hint when the receiving set is EMPTY and
//     (hint + 1) / 2 otherwise, then `insert`s the other set's keys in ITS bucket order — insert reserves room for one key before it looks the key up
//     (find_or_find_insert_slot), so a key both mates cover can still grow a full table;
//   * `positions.remove(pos)` for every removed position: the freed bucket becomes EMPTY or a tombstone, no other key moves.
// -> the positions in the final set's bucket order.
std::vector<uint32_t> positions_order(const uint32_t* seg_keys, const uint32_t* seg_off, uint32_t n_seg, const uint32_t* removed, uint32_t n_removed) {
    FxSet acc;
    for (uint32_t sgi = 0; sgi < n_seg; ++sgi) {
        const uint32_t b = seg_off[sgi], e = seg_off[sgi + 1];
        FxSet seq_dict, positions;
        for (uint32_t c = b; c < e; ++c) seq_dict.insert(seg_keys[c]);
        if (e > b) { positions.reserve(e - b); seq_dict.for_each([&](uint64_t pos) { positions.insert(pos); }); }
        if (sgi == 0) { acc = std::move(positions); continue; }
        const size_t hint = positions.items;
        acc.reserve(acc.items == 0 ? hint : (hint + 1) / 2);
        positions.for_each([&](uint64_t pos) { acc.insert(pos); });
    }
    for (uint32_t i = 0; i < n_removed; ++i) acc.remove(removed[i]);
    return acc.order();
}

}  // namespace

// ==================================================================================================
extern "C" {

const char* floria_oracle_last_error(void) { return g_err.c_str(); }
void floria_oracle_set_order_mode(int m) { g_order_mode.store(m); }
void floria_oracle_set_a14_tie_mode(int m) { g_a14_tie_mode.store(m); }
uint64_t floria_oracle_a14_dropped(int reset) { const uint64_t v = g_a14_dropped; if (reset) g_a14_dropped = 0; return v; }
void floria_oracle_set_arith_mode(int m) { g_arith_mode.store(m); }
int floria_oracle_get_arith_mode(void) { return g_arith_mode.load(); }

// The FxHashSet<&Frag> emulator on its own (tests; building the later sets of the chain from Python): ops[i] > 0 inserts key
// ops[i] - 1, ops[i] < 0 removes key -ops[i] - 1; out receives the iteration order of the final set.
int floria_oracle_fxset_order(const int64_t* ops, uint32_t n_ops, uint32_t* out, uint32_t* n_out) {
    FxSet s;
    for (uint32_t i = 0; i < n_ops; ++i) { if (ops[i] > 0) s.insert((uint64_t)(ops[i] - 1)); else if (ops[i] < 0) s.remove((uint64_t)(-ops[i] - 1)); }
    const auto o = s.order();
    for (size_t i = 0; i < o.size(); ++i) out[i] = o[i];
    *n_out = (uint32_t)o.size();
    return 0;
}
// Frag.positions of a merged / cut-down fragment (positions_order above): segment s = the SNP positions of alignment s, ascending, in merge order
int floria_oracle_positions_order(const uint32_t* seg_keys, const uint32_t* seg_off, uint32_t n_seg, const uint32_t* removed, uint32_t n_removed, uint32_t* out, uint32_t* n_out) {
    const auto o = positions_order(seg_keys, seg_off, n_seg, removed, n_removed);
    for (size_t i = 0; i < o.size(); ++i) out[i] = o[i];
    *n_out = (uint32_t)o.size();
    return 0;
}
// ... and filled the way set_to_seq_dict fills its position map, `entry(key).or_insert(..)` per key (lookup first, room reserved only for a new key)
int floria_oracle_fxset_entry_order(const uint64_t* keys, uint32_t n_keys, uint32_t* out, uint32_t* n_out, uint32_t* buckets) {
    FxSet s;
    for (uint32_t i = 0; i < n_keys; ++i) s.entry_insert(keys[i]);
    const auto o = s.order();
    for (size_t i = 0; i < o.size(); ++i) out[i] = o[i];
    *n_out = (uint32_t)o.size();
    if (buckets) *buckets = (uint32_t)s.buckets;
    return 0;
}
// Mode 2 only: for the blocks of the last floria_oracle_phase_blocks call, the reads of every block partition by partition
// (partition 0 first), each partition in the iteration order of its emulated set; same offsets as the result's read_off.
std::vector<uint32_t> g_last_set_order;
int floria_oracle_last_set_order(uint32_t* out, uint64_t n) {
    if (n != g_last_set_order.size()) { g_err = "no mode-2 phase_blocks result of that size"; return FLORIA_E_INVALID; }
    memcpy(out, g_last_set_order.data(), 4 * n);
    return 0;
}

int floria_oracle_weight_q24(uint32_t* out256) { memcpy(out256, g_w.q24, sizeof(g_w.q24)); return 0; }

int floria_oracle_block_ranges(const uint64_t* snp_to_genome_pos, uint32_t n_snps, uint64_t block_length,
                               uint64_t overlap_len, double minimal_density, floria_ranges** out) {
    std::vector<std::pair<uint32_t, uint32_t>> v;
    int rc = range_with_lengths(snp_to_genome_pos, n_snps, block_length, overlap_len, minimal_density, v);
    if (rc) return rc;
    floria_ranges* r = (floria_ranges*)calloc(1, sizeof(floria_ranges));
    r->n = (uint32_t)v.size();
    r->start = (uint32_t*)malloc(sizeof(uint32_t) * (v.size() + 1));
    r->end   = (uint32_t*)malloc(sizeof(uint32_t) * (v.size() + 1));
    for (size_t i = 0; i < v.size(); ++i) { r->start[i] = v[i].first; r->end[i] = v[i].second; }
    *out = r;
    return 0;
}
void floria_oracle_ranges_free(floria_ranges* r) { if (r) { free(r->start); free(r->end); free(r); } }

// S1 for a batch of blocks, `threads` worker threads pulling blocks from a shared counter
// (graph_processing.rs:345-362: one rayon task per block).
int floria_oracle_phase_blocks(const floria_pileup* pileup, const uint32_t* blk_start, const uint32_t* blk_end,
                               uint32_t n_blocks, const floria_params* params, uint32_t threads,
                               floria_block_result** out) {
    int rc = validate(pileup);
    if (rc) return rc;
    if (!params || params->max_ploidy < 1 || params->max_ploidy > FLORIA_MAX_PLOIDY || params->beam < 1) { g_err = "bad params"; return FLORIA_E_INVALID; }
    Pile P{pileup};
    std::vector<uint32_t> cell_order_;
    if (g_arith_mode.load() >= 1) { cell_order_ = build_cell_order(pileup); P.order = cell_order_.data(); }
    std::vector<BlockOut> outs(n_blocks);
    std::atomic<uint32_t> next{0};
    auto worker = [&]() {
        for (;;) {
            uint32_t b = next.fetch_add(1);
            if (b >= n_blocks) break;
            get_local_hap_blocks(P, blk_start[b], blk_end[b], *params, outs[b]);
        }
    };
    if (threads <= 1) worker();
    else { std::vector<std::thread> th; for (uint32_t t = 0; t < threads; ++t) th.emplace_back(worker); for (auto& t : th) t.join(); }

    floria_block_result* R = (floria_block_result*)calloc(1, sizeof(floria_block_result));
    R->n_blocks = n_blocks; R->max_ploidy = params->max_ploidy;
    R->best_ploidy = (uint32_t*)calloc(n_blocks + 1, sizeof(uint32_t));
    R->ploidies_tried = (uint32_t*)calloc(n_blocks + 1, sizeof(uint32_t));
    R->read_off = (uint64_t*)calloc(n_blocks + 1, sizeof(uint64_t));
    R->mec = (double*)calloc((size_t)n_blocks * params->max_ploidy + 1, sizeof(double));
    uint64_t tot = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) { R->read_off[b] = tot; tot += outs[b].reads.size(); }
    R->read_off[n_blocks] = tot;
    R->read_id = (uint32_t*)malloc(sizeof(uint32_t) * (tot + 1));
    R->part = (uint8_t*)malloc(tot + 1);
    R->min_prune_margin = std::numeric_limits<double>::infinity();
    g_last_set_order.clear();
    if (g_order_mode.load() == 2) for (uint32_t b = 0; b < n_blocks; ++b) g_last_set_order.insert(g_last_set_order.end(), outs[b].set_order.begin(), outs[b].set_order.end());
    for (uint32_t b = 0; b < n_blocks; ++b) {
        R->best_ploidy[b] = outs[b].best_ploidy;
        R->ploidies_tried[b] = outs[b].tried;
        for (size_t i = 0; i < outs[b].reads.size(); ++i) {
            R->read_id[R->read_off[b] + i] = outs[b].reads[i];
            R->part[R->read_off[b] + i] = outs[b].part.empty() ? 0 : outs[b].part[i];
        }
        for (uint32_t p = 0; p < params->max_ploidy; ++p) R->mec[(size_t)b * params->max_ploidy + p] = outs[b].mec.empty() ? 0.0 : outs[b].mec[p];
        if (outs[b].best_ploidy && outs[b].min_margin < R->min_prune_margin) R->min_prune_margin = outs[b].min_margin;
    }
    *out = R;
    return 0;
}
void floria_oracle_block_result_free(floria_block_result* r) {
    if (!r) return;
    free(r->best_ploidy); free(r->ploidies_tried); free(r->read_off); free(r->read_id); free(r->part); free(r->mec); free(r);
}

// Beam search + optimise for ONE ploidy of one block (unit-test hook): partition after beam search
// (part_beam) and after optimize_clustering (part_opt), plus the two MEC figures.
int floria_oracle_one_ploidy(const floria_pileup* pileup, uint32_t start, uint32_t end, uint32_t ploidy, double epsilon,
                             uint32_t beam, uint32_t* n_out, uint32_t* read_id, uint8_t* part_beam, uint8_t* part_opt,
                             double* mec_bad, double* num_alleles, int* iters) {
    int rc = validate(pileup);
    if (rc) return rc;
    Pile P{pileup};
    std::vector<uint32_t> cell_order_;
    if (g_arith_mode.load() >= 1) { cell_order_ = build_cell_order(pileup); P.order = cell_order_.data(); }
    std::vector<uint32_t> reads = find_reads_in_interval(P, start, end);
    *n_out = (uint32_t)reads.size();
    if (reads.empty()) return 0;
    std::vector<std::vector<uint32_t>> part;
    beam_search_phasing(P, reads, (int)ploidy, epsilon, DIV_FACTOR, std::log(PROB_CUTOFF), beam, part, nullptr);
    auto fill = [&](const std::vector<std::vector<uint32_t>>& pp, uint8_t* dst) {
        for (int k = 0; k < (int)pp.size(); ++k)
            for (uint32_t r : pp[k]) dst[std::lower_bound(reads.begin(), reads.end(), r) - reads.begin()] = (uint8_t)k;
    };
    for (size_t i = 0; i < reads.size(); ++i) read_id[i] = reads[i];
    fill(part, part_beam);
    auto opt = optimize_clustering(P, part, epsilon, NUM_ITER_OPTIMIZE, iters);
    fill(opt, part_opt);
    double mec = 0, na = 0;
    HapBlock np = hap_block_from_partition(P, opt, false);
    for (const QM& s : mec_stats_of_block(np, 1)) { double good = (double)s.bases, bad = (double)s.errors + (double)s.m * epsilon; mec += bad; na += good; na += bad; }
    *mec_bad = mec; *num_alleles = na;
    return 0;
}

// optimize_clustering (local_clustering.rs:71-130) on a GIVEN partition of the block's reads (test hook: the hand-traced opt_iterate KAT)
int floria_oracle_optimize_given(const floria_pileup* pileup, const uint32_t* read_id, const uint8_t* part_in, uint32_t n, uint32_t ploidy, double epsilon,
                                 uint8_t* part_out, int* iters) {
    int rc = validate(pileup);
    if (rc) return rc;
    Pile P{pileup};
    std::vector<uint32_t> cell_order_;
    if (g_arith_mode.load() >= 1) { cell_order_ = build_cell_order(pileup); P.order = cell_order_.data(); }
    std::vector<std::vector<uint32_t>> part(ploidy);
    for (uint32_t i = 0; i < n; ++i) { if (part_in[i] >= ploidy) return -1; part[part_in[i]].push_back(read_id[i]); }
    auto opt = optimize_clustering(P, part, epsilon, NUM_ITER_OPTIMIZE, iters);
    for (int k = 0; k < (int)opt.size(); ++k)
        for (uint32_t r : opt[k]) for (uint32_t i = 0; i < n; ++i) if (read_id[i] == r) part_out[i] = (uint8_t)k;
    return 0;
}

// S2
int floria_oracle_reassign_ordered(const floria_pileup* pileup, const uint64_t* grp_off, const uint32_t* grp_read,
                                   const uint32_t* grp_range, uint32_t n_groups, const uint32_t* read_order, uint32_t n_order,
                                   double epsilon, floria_groups** out);
int floria_oracle_reassign(const floria_pileup* pileup, const uint64_t* grp_off, const uint32_t* grp_read,
                           const uint32_t* grp_range, uint32_t n_groups, double epsilon, floria_groups** out) {
    return floria_oracle_reassign_ordered(pileup, grp_off, grp_read, grp_range, n_groups, nullptr, 0, epsilon, out);
}
int floria_oracle_reassign_ordered(const floria_pileup* pileup, const uint64_t* grp_off, const uint32_t* grp_read,
                                   const uint32_t* grp_range, uint32_t n_groups, const uint32_t* read_order, uint32_t n_order,
                                   double epsilon, floria_groups** out) {
    int rc = validate(pileup);
    if (rc) return rc;
    Pile P{pileup};
    std::vector<uint32_t> cell_order_;
    if (g_arith_mode.load() >= 1) { cell_order_ = build_cell_order(pileup); P.order = cell_order_.data(); }
    std::vector<std::vector<uint32_t>> parts(n_groups);
    std::vector<std::pair<uint32_t, uint32_t>> ranges(n_groups);
    for (uint32_t g = 0; g < n_groups; ++g) {
        for (uint64_t i = grp_off[g]; i < grp_off[g + 1]; ++i) {
            if (grp_read[i] >= P.n()) { g_err = "group read id out of range"; return FLORIA_E_INVALID; }
            parts[g].push_back(grp_read[i]);
        }
        std::sort(parts[g].begin(), parts[g].end());
        parts[g].erase(std::unique(parts[g].begin(), parts[g].end()), parts[g].end());   // FxHashSet semantics
        ranges[g] = {grp_range[2 * g], grp_range[2 * g + 1]};
    }
    for (uint32_t i = 0; i < n_order; ++i) if (read_order && read_order[i] >= P.n()) { g_err = "read_order id out of range"; return FLORIA_E_INVALID; }
    process_reads_for_final_parts(P, parts, ranges, epsilon, read_order, n_order);
    floria_groups* G = (floria_groups*)calloc(1, sizeof(floria_groups));
    G->n_groups = (uint32_t)parts.size();
    G->grp_off = (uint64_t*)calloc(parts.size() + 1, sizeof(uint64_t));
    G->range = (uint32_t*)calloc(2 * parts.size() + 2, sizeof(uint32_t));
    uint64_t tot = 0;
    for (size_t g = 0; g < parts.size(); ++g) { G->grp_off[g] = tot; tot += parts[g].size(); G->range[2 * g] = ranges[g].first; G->range[2 * g + 1] = ranges[g].second; }
    G->grp_off[parts.size()] = tot;
    G->grp_read = (uint32_t*)malloc(sizeof(uint32_t) * (tot + 1));
    for (size_t g = 0; g < parts.size(); ++g) std::copy(parts[g].begin(), parts[g].end(), G->grp_read + G->grp_off[g]);
    *out = G;
    return 0;
}
void floria_oracle_groups_free(floria_groups* g) { if (g) { free(g->grp_off); free(g->grp_read); free(g->range); free(g); } }

// Isolated BinaryHeap emulation for unit tests: ops[i] >= 0 pushes score[ops[i]] ... encoded as:
// scores[n] pushed in order with capacity `limit` (pop when len > limit, global_clustering.rs:130-134);
// returns the heap array (payload ids) and the into_sorted_vec order.
int floria_oracle_heap_trace(const double* scores, uint32_t n, uint32_t limit, int32_t* heap_ids, uint32_t* heap_len,
                             int32_t* sorted_ids) {
    BinaryHeap h;
    for (uint32_t i = 0; i < n; ++i) {
        HeapItem it; it.score = scores[i]; it.node = (int)i;
        h.push(std::move(it));
        if (h.len() > limit) h.pop();
    }
    *heap_len = (uint32_t)h.len();
    for (size_t i = 0; i < h.len(); ++i) heap_ids[i] = h.data[i].node;
    h.into_sorted_vec();
    for (size_t i = 0; i < h.len(); ++i) sorted_ids[i] = h.data[i].node;
    return 0;
}

// HapNode::new + update_hap_graph for the non-empty blocks of one contig, given S1's result for exactly these blocks.
// node_cov: one f64 per (non-empty block, partition) in block order; edge_w: for each consecutive pair of non-empty blocks
// the row-major best[b] x best[b'] matrix of out_weights (graph_processing.rs:28-47) before the >= 2 filter (:51).
int floria_oracle_hap_graph(const floria_pileup* pileup, const uint32_t* blk_start, const uint32_t* blk_end, uint32_t n_blocks,
                            const uint32_t* best_ploidy, const uint64_t* read_off, const uint32_t* read_id, const uint8_t* part,
                            double* node_cov, uint32_t* edge_w) {
    int rc = validate(pileup);
    if (rc) return rc;
    Pile P{pileup};
    std::vector<uint32_t> cell_order_;
    if (g_arith_mode.load() >= 1) { cell_order_ = build_cell_order(pileup); P.order = cell_order_.data(); }
    std::vector<std::vector<NodeO>> cols;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        if (best_ploidy[b] == 0) continue;                               // None -> no column (graph_processing.rs:355-361)
        std::vector<std::vector<uint32_t>> sets(best_ploidy[b]);
        for (uint64_t i = read_off[b]; i < read_off[b + 1]; ++i) sets[part[i]].push_back(read_id[i]);
        std::vector<NodeO> col;
        for (auto& fs : sets) col.push_back(hap_node_new(P, fs, blk_start[b], blk_end[b]));
        cols.push_back(std::move(col));
    }
    size_t nc = 0, ne = 0;
    for (auto& col : cols) for (auto& nd : col) node_cov[nc++] = nd.cov;
    for (size_t i = 0; i + 1 < cols.size(); ++i) {
        auto& b1 = cols[i]; auto& b2 = cols[i + 1];
        for (auto& n1 : b1) {
            std::vector<uint32_t> out_weights(b2.size(), 0);
            for (uint32_t read : n1.frags) {
                std::vector<std::pair<uint64_t, size_t>> sim;
                size_t hap_id_in = SIZE_MAX;
                for (size_t l = 0; l < b2.size(); ++l) {
                    if (std::binary_search(b2[l].frags.begin(), b2[l].frags.end(), read)) hap_id_in = l;
                    sim.push_back({distance_read_haplo_diff(P, read, b2[l].hap_map), l});
                }
                std::sort(sim.begin(), sim.end());
                if (sim.size() > 1) { if (sim[0].first != sim[1].first && hap_id_in != SIZE_MAX) out_weights[hap_id_in] += 1; }
                else if (hap_id_in != SIZE_MAX) out_weights[hap_id_in] += 1;
            }
            for (uint32_t w : out_weights) edge_w[ne++] = w;
        }
    }
    return 0;
}

// Iteration order of the inner allele map `FxHashMap<Genotype, GenotypeCount>` (types_structs.rs:15), derived — NOT observed, there is
// no Rust toolchain here — from fxhash 0.2.1 (hash(u8 k) = k * 0x517cc1b727220a95) and hashbrown's layout (slot = hash & bucket_mask,
// buckets scanned in ascending index; 4 buckets for <= 3 keys, 8 buckets from the 4th key on): keys {0,1,2,3} land in slots
// {0,1,2,3} of a 4-bucket table and in slots {0,5,2,7} of an 8-bucket one, so the order is ascending for <= 3 present alleles and
// 0,2,1,3 when all four are present.
static inline void allele_order(uint8_t present, int order[FLORIA_MAX_ALLELES], int* n) {
    static const int asc[4] = {0, 1, 2, 3}, four[4] = {0, 2, 1, 3};
    const int* o = (present & 15) == 15 ? four : asc;
    *n = 0;
    for (int x = 0; x < 4; ++x) if ((present >> o[x]) & 1) order[(*n)++] = o[x];
}
// consensus allele of a site: `.iter().max_by_key(|entry| entry.1)` (utils_frags.rs:674-688) — the LAST maximal element in
// iteration order
static inline int consensus_allele(const Site& s) {
    int order[FLORIA_MAX_ALLELES], n;
    allele_order(s.present, order, &n);
    int best = order[0];
    for (int x = 1; x < n; ++x) if (s.q[order[x]] >= s.q[best]) best = order[x];
    return best;
}

// get_errors_cov_from_frags (utils_frags.rs:596-655) for one haploset: unit-count histogram over [lo, hi]; alleles visited in the
// inner map's iteration order (allele_order below: ascending, except 0,2,1,3 at a site with all four alleles).
int floria_oracle_haploset_stats(const floria_pileup* pileup, const uint32_t* reads, uint32_t n, uint32_t lo, uint32_t hi, double* out4) {
    int rc = validate(pileup);
    if (rc) return rc;
    Pile P{pileup};
    std::vector<uint32_t> cell_order_;
    if (g_arith_mode.load() >= 1) { cell_order_ = build_cell_order(pileup); P.order = cell_order_.data(); }
    std::vector<uint32_t> fs(reads, reads + n);
    Hap hap_map = set_to_seq_dict(P, fs, false);                                   // :606
    double errors = 0., total_support = 0., sum_support = 0.;
    uint64_t nonzero = 0;
    for (uint64_t pos = lo; pos <= hi && hi >= lo; ++pos) {                          // :612
        double snp_support = 0., max_count_pos = 0.;
        const Site* s = hap_map.find((uint32_t)pos);
        if (s && s->present) {
            nonzero++;
            int order[FLORIA_MAX_ALLELES], na;
            allele_order(s->present, order, &na);                                     // the inner map's iteration order (see allele_order)
            for (int x = 0; x < na; ++x) {
                const int a = order[x];
                const double count = (double)s->q[a];
                if (count > snp_support) max_count_pos = count;                      // :622-624 (compares with the running sum)
                snp_support += count;
            }
        }
        total_support += snp_support;
        errors += snp_support - max_count_pos;
        sum_support += snp_support;
    }
    out4[0] = nonzero ? sum_support / (double)nonzero : 0.;                           // mean = true branch :641-647
    out4[1] = errors / total_support;
    out4[2] = errors;
    out4[3] = total_support;
    return 0;
}

// part_block_manip::get_hapq (part_block_manip.rs:517-616) for the haplosets of one contig: HAPQ and REL_ERR per haploset and the
// contig's avg_err.  groups = read-id lists (counter_ids), grp_range = their inclusive SNP ranges (1-based), snp_to_genome_pos is
// 0-based by SNP index - 1 as in the reference.  Order-independent: `find_overlapping_blocks` (rust-lapper: half-open overlap
// start < other.stop && stop > other.start, :484) only feeds a running maximum.
int floria_oracle_hapq(const floria_pileup* pileup, const uint64_t* grp_off, const uint32_t* grp_read, const uint32_t* grp_range,
                       uint32_t n_groups, const uint64_t* snp_to_genome_pos, uint32_t n_snps, uint64_t block_length,
                       uint8_t* hapq_out, double* rel_err_out, double* avg_err_out) {
    int rc = validate(pileup);
    if (rc) return rc;
    Pile P{pileup};
    std::vector<uint32_t> cell_order_;
    if (g_arith_mode.load() >= 1) { cell_order_ = build_cell_order(pileup); P.order = cell_order_.data(); }
    std::vector<std::vector<uint32_t>> parts(n_groups);
    for (uint32_t g = 0; g < n_groups; ++g) parts[g].assign(grp_read + grp_off[g], grp_read + grp_off[g + 1]);
    double weight = 0., error = 0.;
    std::vector<double> errs(n_groups);
    for (uint32_t g = 0; g < n_groups; ++g) {                                       // :529-539
        double o4[4];
        rc = floria_oracle_haploset_stats(pileup, parts[g].data(), (uint32_t)parts[g].size(), grp_range[2 * g], grp_range[2 * g + 1], o4);
        if (rc) return rc;
        weight += o4[3]; error += o4[2]; errs[g] = o4[1];
    }
    const double avg_err = error / weight;                                          // :540
    HapBlock all = hap_block_from_partition(P, parts, true);                        // :541
    for (uint32_t i = 0; i < n_groups; ++i) {
        const uint32_t x1 = grp_range[2 * i], x2 = grp_range[2 * i + 1];
        double max_penalty = 0.;
        for (uint32_t j = 0; j < n_groups; ++j) {                                   // find_overlapping_blocks :453-513
            if (j == i) continue;
            const uint32_t y1 = grp_range[2 * j], y2 = grp_range[2 * j + 1];
            if (!(y1 < x2 && y2 > x1)) continue;                                    // Lapper::find(x1, x2)
            const uint32_t a = x2 - y1 + 1, b = y2 - x1 + 1;                        // overlap_percent :13-24
            const uint32_t intersect = a < b ? a : b;
            double ol = (double)intersect / (double)(x2 - x1 + 1);
            if (ol > 1.) ol = 1.;
            if (!(ol > 0.05)) continue;                                             // :505
            double same = 0., diff = 0.;                                            // distance_between_haplotypes :659-700, range = (MIN, MAX)
            all.blocks[i].for_each([&](uint32_t pos, const Site& s1) {
                const Site* s2 = all.blocks[j].find(pos);
                if (!s2) return;
                if (consensus_allele(s1) == consensus_allele(*s2)) same += 1.; else diff += 1.;
            });
            const double dist = (same + diff) == 0. ? 1. : diff / (same + diff);    // :563-567
            if (ol * (1. - dist) > max_penalty) max_penalty = ol * (1. - dist);     // :568-572
        }
        uint32_t r0 = 0xffffffffu, r1 = 0;                                          // :584-592
        for (uint32_t r : parts[i]) {
            if (pileup->first[r] < r0) r0 = pileup->first[r];
            if (pileup->last[r] >= r1) r1 = pileup->last[r];
        }
        uint64_t base_range = 0;
        if (!(r0 > r1)) {
            if (x1 == 0 || x2 == 0 || x1 > n_snps || x2 > n_snps) { g_err = "haploset range outside snp_to_genome_pos"; return FLORIA_E_INVALID; }
            base_range = snp_to_genome_pos[x2 - 1] - snp_to_genome_pos[x1 - 1];     // :598-599
        }
        const double t1 = 40. * (1. - max_penalty);                                 // HAPQ_CONSTANT
        const double len = (double)parts[i].size();
        const double t2 = std::min(1., len / 3.);
        const double t3 = std::max(0.0, std::log(((double)base_range / (double)block_length) + 1.));
        const double prod = t1 * t2 * t3;
        uint64_t hq = prod > 0. ? (prod >= 18446744073709551615. ? ~0ull : (uint64_t)prod) : 0;    // `as usize` saturates, NaN -> 0
        if (parts[i].size() == 1) hq = 0;
        hapq_out[i] = (uint8_t)std::min<uint64_t>(hq, 60);
        rel_err_out[i] = errs[i] / avg_err;                                         // :614
    }
    *avg_err_out = avg_err;
    return 0;
}

double floria_oracle_binom(uint64_t n, uint64_t k, double p, double div) { return stable_binom_cdf_p_rev(n, k, p, div); }

}  // extern "C"
