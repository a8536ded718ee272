// arith_kernel.h — what the opt-in "reference arithmetic" mode (floria_hip_set_option("arith", 1)) needs beyond the kernels'
// ARITH template flag: the ORDERS in which the reference adds its running f64 sums.
//
// The reference walks a read's cells as `for pos in r.positions.iter()` (utils_frags.rs:35), an FxHashSet<SnpPosition> collected from
// the keys of an FxHashMap that the CIGAR walk filled in ascending order (file_reader.rs:661-733), and a haplotype's positions as
// `for seq_dict in hap.values()` (local_clustering.rs:227), an FxHashMap filled read by read.  Both orders are the bucket orders of
// std's hash table (hashbrown: 16-byte control groups, triangular probing, capacity 7/8 of the buckets, growth by re-insertion in bucket
// order) under FxHash (key * 0x517cc1b727220a95, fxhash 0.2.1).  FxTable below restates that table for keys that are inserted once and never
// removed — all this mode needs — in memory the caller provides (global scratch or LDS, reached through flat pointers).
// DESIGN.md §6 "Order of the f64 additions" says what this mode is and is not.
#pragma once
#include "common.h"

namespace fl {

constexpr uint32_t FX_W = 16;                        // control bytes per probe group (the SSE2 group of the x86-64 reference builds)
constexpr uint8_t  FX_EMPTY = 0xFF;

__host__ __device__ inline uint32_t fx_cap_of(uint32_t nb) { return nb == 0 ? 0 : (nb - 1 < 8 ? nb - 1 : nb / 8 * 7); }
__host__ __device__ inline uint32_t fx_buckets_for(uint32_t cap) {
    if (cap < 8) return cap < 4 ? 4 : 8;
    const uint64_t adj = (uint64_t)cap * 8 / 7;
    uint32_t b = 1;
    while (b < adj) b <<= 1;
    return b;
}
// bytes of ONE table able to hold `cap` keys: control bytes (buckets + one mirrored group) and keys — two arrays, so that the control bytes,
// which every probe reads, can sit in LDS while the keys stay in HBM scratch
// (+16: FxWave::probe16 reads the group at an unaligned position as five aligned words)
__host__ __device__ inline size_t fx_ctrl_bytes(uint32_t cap) { return (((size_t)fx_buckets_for(cap < 1 ? 1 : cap) + FX_W + 15) & ~(size_t)15) + 16; }
// K^-1 mod 2^32 for the low word of FxHash's multiplier (Newton: x <- x * (2 - K x) doubles the correct bits): bucket b of a C-bucket table is the home of the keys = b * K^-1 mod C
constexpr uint32_t fx_inv32(uint32_t k) { uint32_t x = k; for (int i = 0; i < 5; ++i) x *= 2u - k * x; return x; }
constexpr uint32_t FX_KINV32 = fx_inv32(0x27220a95u);
static_assert(FX_KINV32 * 0x27220a95u == 1u, "inverse of the hash multiplier");
constexpr uint32_t FX_TAGS_MIN = 128, FX_TAGS_MAX = 1024;      // FxWave::insert_batch: words of the conflict-detection table (the host takes the largest that costs no workgroup per CU)
__host__ __device__ inline size_t fx_slot_bytes(uint32_t cap) { return 4ull * fx_buckets_for(cap < 1 ? 1 : cap); }

struct FxTable {
    uint8_t*  ctrl = nullptr;        // [buckets + FX_W]
    uint32_t* slot = nullptr;        // [buckets]
    uint32_t  buckets = 0, items = 0, growth_left = 0;

    __device__ static uint64_t hash_of(uint32_t k) { return (uint64_t)k * 0x517cc1b727220a95ull; }
    __device__ void bind(uint8_t* c, uint32_t* s, uint32_t nb) {
        ctrl = c; slot = s;
        buckets = nb; items = 0; growth_left = fx_cap_of(nb);
        for (uint32_t i = 0; i < (nb + FX_W) / 4; ++i) ((uint32_t*)c)[i] = 0xffffffffu;        // (nb + FX_W is a multiple of 4, c 16-byte aligned)
    }
    __device__ void set_ctrl(uint32_t i, uint8_t c) { ctrl[i] = c; ctrl[((i - FX_W) & (buckets - 1)) + FX_W] = c; }
    // first EMPTY control byte along the probe sequence of h (there are no DELETED bytes: nothing is ever removed)
    __device__ uint32_t find_insert_slot(uint64_t h) const {
        const uint32_t mask = buckets - 1;
        uint32_t pos = (uint32_t)h & mask, stride = 0;
        for (;;) {
            uint32_t free_bits = 0;                          // the group's 16 control bytes are requested together (one memory round trip), then examined
#pragma unroll
            for (uint32_t x = 0; x < FX_W; ++x) free_bits |= (uint32_t)(ctrl[pos + x] >> 7) << x;
            if (free_bits) {
                uint32_t idx = (pos + (uint32_t)__builtin_ctz(free_bits)) & mask;
                if (!(ctrl[idx] & 0x80)) {                   // table smaller than a group: the hit was in the mirrored tail; the real slot is in group 0
                    for (uint32_t x = 0; x < FX_W; ++x) if (ctrl[x] & 0x80) { idx = x; break; }
                }
                return idx;
            }
            stride += FX_W; pos = (pos + stride) & mask;
        }
    }
    __device__ void put(uint32_t key) {
        const uint64_t h = hash_of(key);
        const uint32_t idx = find_insert_slot(h);
        --growth_left;
        set_ctrl(idx, (uint8_t)(h >> 57));
        slot[idx] = key;
        ++items;
    }
    // RawTable::resize: a new table of buckets_for(capacity), the old one's keys re-inserted in bucket order; `spare` is the other half of the pair
    __device__ void resize(uint32_t capacity, uint8_t*& spare_c, uint32_t*& spare_s) {
        FxTable n;
        n.bind(spare_c, spare_s, fx_buckets_for(capacity));
        for (uint32_t i0 = 0; i0 < buckets; i0 += 4) {       // (buckets is a multiple of 4; the four keys are requested together)
            bool full[4]; uint32_t key[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { full[u] = !(ctrl[i0 + u] & 0x80); key[u] = slot[i0 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (full[u]) n.put(key[u]);
        }
        spare_c = ctrl; spare_s = slot;
        *this = n;
    }
    // reserve(additional) on a table without tombstones (reserve_rehash never rehashes in place then)
    __device__ void reserve(uint32_t additional, uint8_t*& spare_c, uint32_t*& spare_s) {
        if (additional <= growth_left) return;
        const uint32_t new_items = items + additional, full = fx_cap_of(buckets);
        resize(new_items > full + 1 ? new_items : full + 1, spare_c, spare_s);
    }
    // insert of a key that is not in the table
    __device__ void insert_new(uint32_t key, uint8_t*& spare_c, uint32_t*& spare_s) {
        if (growth_left == 0) reserve(1, spare_c, spare_s);
        put(key);
    }
};

// The same table driven by a whole WAVEFRONT (every argument and member is wave-uniform; all 64 lanes must call): a probe group's 16 control bytes are
// one byte per lane and a ballot — what the reference's SSE2 group load and movemask do — instead of sixteen loads and a bit-gathering loop in one lane,
// and a resize reads 64 buckets at a time.  Used where one table is replayed by a workgroup with idle lanes (optimize_kernel's position maps).
struct FxWave {
    uint8_t*  ctrl = nullptr;
    uint32_t* slot = nullptr;
    uint32_t  buckets = 0, items = 0, growth_left = 0;
    uint32_t  tag_mask = FX_TAGS_MIN - 1;      // insert_batch: the claim of bucket b is word b & tag_mask (two buckets on one word are a false conflict: a shorter round, not a wrong one)
    bool      hbm = false;           // the control bytes are in HBM scratch, not LDS: one lane's stores are fenced before other lanes' loads of the next probe

    __device__ void bind(uint8_t* c, uint32_t* s, uint32_t nb, uint32_t lane) {
        ctrl = c; slot = s;
        buckets = nb; items = 0; growth_left = fx_cap_of(nb);
        for (uint32_t i = lane; i < (nb + FX_W) / 4; i += 64) ((uint32_t*)c)[i] = 0xffffffffu;
        if (hbm) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    __device__ uint32_t find_insert_slot(uint64_t h, uint32_t lane) const {
        const uint32_t mask = buckets - 1;
        uint32_t pos = (uint32_t)h & mask, stride = 0;
        for (;;) {
            const uint32_t c = lane < FX_W ? ctrl[pos + lane] : 0u;
            const uint32_t free_bits = (uint32_t)__ballot((c & 0x80u) != 0u) & 0xffffu;
            if (free_bits) {
                uint32_t idx = (pos + (uint32_t)__builtin_ctz(free_bits)) & mask;
                if (buckets < FX_W) {                        // table smaller than a group: a hit in the mirrored tail means the real slot is in group 0
                    const uint32_t at = ctrl[idx];
                    if (!(at & 0x80u)) {
                        const uint32_t c0 = lane < FX_W ? ctrl[lane] : 0u;
                        idx = (uint32_t)__builtin_ctz((uint32_t)__ballot((c0 & 0x80u) != 0u) & 0xffffu);
                    }
                }
                return idx;
            }
            stride += FX_W; pos = (pos + stride) & mask;
        }
    }
    __device__ void put(uint32_t key, uint32_t lane) {
        const uint64_t h = FxTable::hash_of(key);
        const uint32_t idx = find_insert_slot(h, lane);
        if (lane == 0) {
            const uint8_t h2 = (uint8_t)(h >> 57);
            ctrl[idx] = h2; ctrl[((idx - FX_W) & (buckets - 1)) + FX_W] = h2;
            slot[idx] = key;
        }
        if (hbm) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        --growth_left; ++items;
    }
    __device__ void resize(uint32_t capacity, uint8_t*& spare_c, uint32_t*& spare_s, uint32_t lane) {
        FxWave n;
        n.hbm = hbm;
        n.bind(spare_c, spare_s, fx_buckets_for(capacity), lane);
        for (uint32_t i0 = 0; i0 < buckets; i0 += 64) {
            const bool in = i0 + lane < buckets;
            const uint32_t c = in ? ctrl[i0 + lane] : 0xffu;
            const uint32_t key = in ? slot[i0 + lane] : 0u;
            uint64_t full = __ballot(!(c & 0x80u));
            while (full) {                                   // the old table's keys in bucket order
                const int l = __builtin_ctzll(full);
                full &= full - 1;
                n.put((uint32_t)__shfl((int)key, l), lane);
            }
        }
        spare_c = ctrl; spare_s = slot;
        *this = n;
    }
    __device__ void reserve(uint32_t additional, uint8_t*& spare_c, uint32_t*& spare_s, uint32_t lane) {
        if (additional <= growth_left) return;
        const uint32_t new_items = items + additional, full = fx_cap_of(buckets);
        resize(new_items > full + 1 ? new_items : full + 1, spare_c, spare_s, lane);
    }
    __device__ void insert_new(uint32_t key, uint8_t*& spare_c, uint32_t*& spare_s, uint32_t lane) {
        if (growth_left == 0) reserve(1, spare_c, spare_s, lane);
        put(key, lane);
    }

    // ---- many insertions per round (control bytes in LDS, tables of at least one full group) -------------------------------------------------
    // Sequential insertion puts key e into the first EMPTY byte along ITS probe sequence given what keys 0..e-1 filled.  If every key of a batch
    // computes that byte from the state BEFORE the batch and the results are pairwise distinct, they are the sequential result: a byte that an earlier
    // key of the batch fills cannot lie before key e's own choice on e's probe sequence (it was EMPTY, e would have chosen it), so it can only
    // matter by BEING e's choice.  Hence: every lane probes for its key at once (its group = five aligned LDS words, funnel-shifted), the lanes
    // claim their byte in a small tag table with an LDS atomic min on their sequence rank, and the longest prefix of ranks without a lost claim
    // is committed; the rest goes again.  A round places ~10 keys in the time the one-by-one form places one or two.
    __device__ void bind_lds(uint8_t* c, uint32_t* s, uint32_t nb, uint32_t lane) {        // bind with the control bytes in LDS, written with DS stores like every later access
        ctrl = c; slot = s; hbm = false;
        buckets = nb; items = 0; growth_left = fx_cap_of(nb);
        __attribute__((address_space(3))) uint32_t* const lc = (__attribute__((address_space(3))) uint32_t*)c;
        for (uint32_t i = lane; i < (nb + FX_W) / 4; i += 64) lc[i] = 0xffffffffu;
    }
    // put() for a table smaller than a group, control bytes in LDS (the group at any position reaches the always-EMPTY padding, so the first probe decides)
    __device__ void put_small_lds(uint32_t key, uint32_t lane) {
        __attribute__((address_space(3))) uint8_t* const lc = (__attribute__((address_space(3))) uint8_t*)ctrl;
        const uint64_t h = FxTable::hash_of(key);
        const uint32_t mask = buckets - 1, pos = (uint32_t)h & mask;
        const uint32_t c = lane < FX_W ? lc[pos + lane] : 0u;
        uint32_t idx = (pos + (uint32_t)__builtin_ctz((uint32_t)__ballot((c & 0x80u) != 0u) & 0xffffu)) & mask;
        if (!(lc[idx] & 0x80u)) {                                 // the hit was in the padding: the real slot is the first EMPTY one from 0
            const uint32_t c0 = lane < FX_W ? lc[lane] : 0u;
            idx = (uint32_t)__builtin_ctz((uint32_t)__ballot((c0 & 0x80u) != 0u) & 0xffffu);
        }
        if (lane == 0) { const uint8_t h2 = (uint8_t)(h >> 57); lc[idx] = h2; lc[((idx - FX_W) & mask) + FX_W] = h2; slot[idx] = key; }
        --growth_left; ++items;
    }
    // this lane's key: first EMPTY byte along the probe sequence of h (per-lane loop; buckets >= FX_W)
    __device__ uint32_t probe16(uint64_t h) const {
        const uint32_t mask = buckets - 1;
        uint32_t pos = (uint32_t)h & mask, stride = 0;
        for (;;) {
            const __attribute__((address_space(3))) uint32_t* w = (const __attribute__((address_space(3))) uint32_t*)(ctrl + (pos & ~3u));      // (LDS, not flat)
            const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
            const uint32_t sh = (pos & 3u) * 8u;
            const uint64_t lo = (uint64_t)(__builtin_amdgcn_alignbit(d1, d0, sh) & 0x80808080u) | ((uint64_t)(__builtin_amdgcn_alignbit(d2, d1, sh) & 0x80808080u) << 32);
            const uint64_t hi = (uint64_t)(__builtin_amdgcn_alignbit(d3, d2, sh) & 0x80808080u) | ((uint64_t)(__builtin_amdgcn_alignbit(d4, d3, sh) & 0x80808080u) << 32);
            if (lo | hi) {
                const uint32_t bit = lo ? (uint32_t)__builtin_ctzll(lo) >> 3 : 8u + ((uint32_t)__builtin_ctzll(hi) >> 3);
                return (pos + bit) & mask;
            }
            stride += FX_W; pos = (pos + stride) & mask;
        }
    }
    // keys of the lanes with `valid`, in the order of their ranks r = 0..cnt-1 (ascending with the lane); grow = may this table grow (false inside a resize).
    // tag: tag_mask + 1 LDS words, all ones between calls.
    // (GROW is a template parameter so that the re-insertion inside a resize is not a recursive call: the whole thing inlines, no stack frame)
    template <bool GROW>
    __device__ __forceinline__ void insert_batch(uint32_t key, bool valid, uint32_t r, uint32_t cnt, uint32_t* tag, uint8_t*& spare_c, uint32_t*& spare_s, uint32_t lane) {
        uint32_t done = 0;
        const uint64_t h = FxTable::hash_of(key);
        while (done < cnt) {
            if constexpr (GROW) { if (growth_left == 0) grow_batched(tag, spare_c, spare_s, lane); }
            if (buckets < FX_W) {                           // tables smaller than a group (the first seven keys): one by one
                const uint64_t at = __ballot(valid && r == done);
                put_small_lds((uint32_t)__shfl((int)key, (int)__builtin_ctzll(at)), lane);
                ++done;
                continue;
            }
            const uint32_t lim = cnt < done + growth_left ? cnt : done + growth_left;
            const bool act = valid && r >= done && r < lim;
            const uint32_t target = act ? probe16(h) : 0u;
            const uint32_t tg = target & tag_mask;
            __attribute__((address_space(3))) uint32_t* const tp = (__attribute__((address_space(3))) uint32_t*)tag + tg;
            if (act) (void)__hip_atomic_fetch_min(tp, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t won = act ? __hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : r;
            const uint64_t lm = __ballot(act && won != r);
            const uint32_t lstar = lm ? (uint32_t)__shfl((int)r, (int)__builtin_ctzll(lm)) : lim;
            if (act && r < lstar) {
                const uint8_t h2 = (uint8_t)(h >> 57);
                __attribute__((address_space(3))) uint8_t* const lc = (__attribute__((address_space(3))) uint8_t*)ctrl;
                lc[target] = h2; lc[((target - FX_W) & (buckets - 1)) + FX_W] = h2;
                slot[target] = key;
            }
            if (act) __hip_atomic_store(tp, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t nn = lstar - done;
            growth_left -= nn; items += nn; done = lstar;
        }
    }
    // reserve(1) on a full table, the old table's keys re-inserted in bucket order, 64 buckets at a time
    __device__ __forceinline__ void grow_batched(uint32_t* tag, uint8_t*& spare_c, uint32_t*& spare_s, uint32_t lane) {
        const uint32_t full_cap = fx_cap_of(buckets), want = items + 1 > full_cap + 1 ? items + 1 : full_cap + 1;
        FxWave n;
        n.tag_mask = tag_mask;
        n.bind_lds(spare_c, spare_s, fx_buckets_for(want), lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");           // the keys (HBM scratch) were stored by other lanes of this wave: same CU, same L1 — waiting for the stores is all it takes (an agent-scope fence writes the XCD's L2 back, for everybody)
        for (uint32_t i0 = 0; i0 < buckets; i0 += 64) {
            const bool in = i0 + lane < buckets;
            const uint32_t c = in ? ((const __attribute__((address_space(3))) uint8_t*)ctrl)[i0 + lane] : 0xffu;
            const uint32_t key = in ? slot[i0 + lane] : 0u;
            const bool fullb = !(c & 0x80u);
            const uint64_t fm = __ballot(fullb);
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            n.template insert_batch<false>(key, fullb, r, (uint32_t)__popcll(fm), tag, spare_c, spare_s, lane);
        }
        spare_c = ctrl; spare_s = slot;
        *this = n;
    }
};

// ---- Frag.positions: for every read its cells, permuted into the set's iteration order --------------------------------------------
// `positions = seq_dict.keys().collect()`: a set with room for all L keys at once (C = fx_buckets_for(L) buckets, never resized), filled in the iteration order of
// the growing map seq_dict.  THE HOME-BUCKET RULE (optimize_kernel.h, step (0)): a key's first probe is bucket (key * K) mod C, K odd — so when the read's SNP indices
// span fewer than C positions every key finds its home bucket empty, whatever the order of arrival: the set's iteration order is a function of the keys alone and
// bucket b holds the position congruent to b * K^-1 mod C.  cell_order_direct_kernel: a WAVEFRONT per read, a position -> cell table of C entries in LDS, the buckets
// in order 64 at a time; reads that span C positions or more (scattered cells: linked reads) are listed for cell_order_kernel, one thread per read, which emulates
// the three tables insertion by insertion (the growing seq_dict — a pair — and the set) in global scratch.
struct CellOrderArgs {
    const ContigDev* contigs;
    const uint64_t*  read_prefix;    // [n_contigs+1] reads before contig c in this launch
    const uint64_t*  cell_prefix;    // [n_contigs]   cells before contig c: where its part of `ord` starts
    uint32_t n_contigs;
    uint64_t n_reads;
    uint2*    ord;                   // [cells] {SNP, allele << 28 | weight} of the x-th cell of the read in set order (a permuted copy: one load per cell in the kernels)
    uint8_t*  scratch;               // [threads][3 * (ctrl_bytes + slot_bytes)]
    uint64_t  ctrl_bytes, slot_bytes;
    uint64_t* todo;                  // [1 + n_reads] number of reads left to cell_order_kernel, then their global indices
    uint32_t  replay_all;            // (tests) != 0: every read goes the long way
    uint64_t* bad;                   // ~(the smallest global index of a read whose host-given set_order is not a permutation of its cells), 0 = none
    uint64_t  read_base;             // this launch covers the reads [read_base, read_base + n_reads) of the batch (a pipelined upload computes the orders chunk by chunk)
    uint64_t  scratch_bytes;         // bytes behind `scratch`
    uint32_t  len_max;               // longest read (sizes the emulated tables); 0 = take it from `status` (the flatten kernel's per-contig maximum: known on the device only)
    const uint32_t* status_max_len;  // [n_contigs] with the stride below
    uint32_t  status_stride;         // in 32-bit words
};
constexpr uint32_t CO_IDX_MAX = 2048;      // largest set (buckets) the direct kernel takes: u16 entries, 4 KB of LDS per wavefront
__global__ __launch_bounds__(256) void cell_order_direct_kernel(CellOrderArgs g) {
    __shared__ uint16_t s_idx[4][CO_IDX_MAX];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wid, nw = (uint64_t)gridDim.x * 4;
    __attribute__((address_space(3))) uint16_t* const idx = (__attribute__((address_space(3))) uint16_t*)s_idx[wid];
    for (uint64_t gl = gw; gl < g.n_reads; gl += nw) {
        const uint64_t gr = g.read_base + gl;
        uint32_t lo = 0, hi = g.n_contigs;                                    // contig of global read gr
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g.read_prefix[mid] <= gr) lo = mid; else hi = mid; }
        const ContigDev cd = g.contigs[lo];
        const uint32_t r = (uint32_t)(gr - g.read_prefix[lo]);
        const uint32_t cb = cd.read_off[r], L = cd.read_off[r + 1] - cb;
        if (L == 0) continue;
        if (cd.set_order) {
            // The host wrote the set's iteration order down (include/floria_hip.h: a Rust host has the FxHashSet itself; floria-hip emulates merges and removals on
            // the CPU): a gather.  Checked: every entry indexes one of the read's cells (memory safety, any length) and no cell is named twice (a bitmap in
            // the wavefront's LDS table, reads of up to 16 * CO_IDX_MAX cells; longer reads are bounds-checked only).
            const uint32_t* so = cd.set_order + cb;
            uint2* out = g.ord + g.cell_prefix[lo] + cb;
            const bool bitmap = L <= 16u * CO_IDX_MAX;
            uint32_t* const bm = (uint32_t*)s_idx[wid];                          // (the wavefront's position table, as 32-bit words: a bit per cell)
            if (bitmap) { for (uint32_t x = lane; x < (L + 31u) / 32u; x += 64) bm[x] = 0; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
            bool ok = true;
            for (uint32_t j0 = 0; j0 < L; j0 += 64) {
                const uint32_t j = j0 + lane;
                uint32_t e = j < L ? so[j] : 0u;
                if (e >= L) { ok = false; e = 0; }
                if (j < L) out[j] = make_uint2(cd.cell_snp[cb + e], cd.cell_aw[cb + e]);
                if (bitmap && j < L) { const uint32_t bit = 1u << (e & 31u); if (atomicOr(&bm[e >> 5], bit) & bit) ok = false; }
            }
            if (__ballot(!ok) && lane == 0) atomicMax((unsigned long long*)g.bad, (unsigned long long)~gr);      // (max of ~index = the smallest index)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            continue;
        }
        const uint32_t first = cd.cell_snp[cb], range = cd.cell_snp[cb + L - 1] - first;
        const uint32_t C = fx_buckets_for(L);
        if (range >= C || C > CO_IDX_MAX || L > 65535u || g.replay_all) {       // (wave-uniform)
            if (lane == 0) g.todo[1 + atomicAdd((unsigned long long*)g.todo, 1ull)] = gr;
            continue;
        }
        // (a wavefront's LDS operations execute in order; the fences keep the compiler from moving them across each other)
        for (uint32_t x = lane; x <= range; x += 64) idx[x] = 0;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        for (uint32_t c = lane; c < L; c += 64) idx[cd.cell_snp[cb + c] - first] = (uint16_t)(c + 1);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        uint2* out = g.ord + g.cell_prefix[lo] + cb;
        uint32_t k = 0;
        for (uint32_t b0 = 0; b0 < C; b0 += 64) {
            const uint32_t b = b0 + lane;
            const uint32_t off = (b * FX_KINV32 - first) & (C - 1u);          // the one position of [first, first + C) whose home is bucket b
            const uint32_t e = (b < C && off <= range) ? idx[off] : 0u;
            const uint64_t m = __ballot(e != 0u);
            if (e) out[k + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = make_uint2(first + off, cd.cell_aw[cb + e - 1]);
            k += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}
__global__ void cell_order_kernel(CellOrderArgs g) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_todo = g.todo[0];
    if (tid >= n_todo) return;                                                // (nothing left over on BASELINE's configs: every read took the direct kernel)
    if (g.len_max == 0) {
        // a pipelined upload: the longest read of the chunk is known on the device only (flatten_kernel's status words); every thread sizes its three tables by it
        // and as many threads as the scratch holds share the list
        uint32_t lm = 1;
        for (uint32_t c = 0; c < g.n_contigs; ++c) { const uint32_t v = g.status_max_len[(uint64_t)c * g.status_stride]; lm = v > lm ? v : lm; }
        g.ctrl_bytes = fx_ctrl_bytes(lm); g.slot_bytes = fx_slot_bytes(lm);
        const uint64_t fit = g.scratch_bytes / (3 * (g.ctrl_bytes + g.slot_bytes));
        if (fit == 0) { atomicMax((unsigned long long*)g.bad, (unsigned long long)~(g.todo[1])); return; }      // (a read of > 10^7 cells: reported as a bad read rather than written out of bounds)
        nth = nth < fit ? nth : fit;
        if (tid >= nth) return;
    }
    const uint64_t tb = g.ctrl_bytes + g.slot_bytes;
    uint8_t* mine = g.scratch + tid * 3 * tb;
    for (uint64_t ti = tid; ti < n_todo; ti += nth) {
        const uint64_t gr = g.todo[1 + ti];
        uint32_t lo = 0, hi = g.n_contigs;                                    // contig of global read gr
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (g.read_prefix[mid] <= gr) lo = mid; else hi = mid; }
        const ContigDev cd = g.contigs[lo];
        const uint32_t r = (uint32_t)(gr - g.read_prefix[lo]);
        const uint32_t cb = cd.read_off[r], L = cd.read_off[r + 1] - cb;
        // seq_dict: an empty map, keys inserted ascending, growing as it goes
        FxTable seq;
        uint8_t* spare_c = mine + tb; uint32_t* spare_s = (uint32_t*)(mine + tb + g.ctrl_bytes);
        seq.bind(mine, (uint32_t*)(mine + g.ctrl_bytes), fx_buckets_for(1));          // (a read has at least one cell)
        for (uint32_t c = 0; c < L; ++c) seq.insert_new(cd.cell_snp[cb + c], spare_c, spare_s);
        // positions = seq_dict.keys().collect(): room for all keys at once, then the keys in the map's bucket order
        FxTable set;
        set.bind(mine + 2 * tb, (uint32_t*)(mine + 2 * tb + g.ctrl_bytes), fx_buckets_for(L));
        for (uint32_t i = 0; i < seq.buckets; ++i) if (!(seq.ctrl[i] & 0x80)) set.put(seq.slot[i]);
        uint2* out = g.ord + g.cell_prefix[lo] + cb;
        uint32_t k = 0;
        for (uint32_t i = 0; i < set.buckets; ++i) if (!(set.ctrl[i] & 0x80)) {
            const uint32_t pos = set.slot[i];
            uint32_t a = 0, b = L;                                            // the cell that holds pos (cells are strictly ascending)
            while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (cd.cell_snp[cb + mid] <= pos) a = mid; else b = mid; }
            out[k++] = make_uint2(pos, cd.cell_aw[cb + a]);
        }
    }
}

}  // namespace fl
