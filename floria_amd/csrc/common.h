// common.h — device-side data model shared by the gfx950 kernels of libfloria_hip.so.
//
// Histogram cell ("allele count" of types_structs.rs:15 `Haplotype`): one u64 per
// (partition, SNP position, allele).  Two encodings:
//   beam kernel     : bit 63 = allele key present, bits 0..62 = Q24 weighted sum
//   optimise kernel : bits 44..63 = number of reads (unit count), bits 0..43 = Q24 weighted sum
// Q24: w(q) = 1f32 - 10f32^(-q/10) is always k * 2^-24 (utils_frags.rs:702-711), so all weighted
// sums are exact integers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fl {

constexpr uint64_t PRESENT_BIT = 1ull << 63;
constexpr uint64_t QMASK63     = ~PRESENT_BIT;
constexpr int      CNT_SHIFT   = 44;
constexpr uint64_t QMASK44     = (1ull << CNT_SHIFT) - 1;
constexpr uint64_t ONE_Q24     = 1ull << 24;
constexpr int      MAX_PLOIDY  = 16;   // == FLORIA_MAX_PLOIDY

struct ContigDev {
    const uint32_t* read_off;   // [n_reads+1]
    const uint32_t* first;      // [n_reads]
    const uint32_t* last;       // [n_reads]
    const uint32_t* cell_snp;   // [n_cells] 1-based SNP index
    const uint32_t* cell_aw;    // [n_cells] allele << 28 | Q24 quality weight (w(q) <= 2^24), precomputed at upload
    const uint64_t* tw;         // [2*n_reads] per-read hash constants: sum over cells of Rq{1,2}[hash_idx(snp, allele)] * w
    const uint32_t* meta;       // [8*n_reads] packed per-read record {cell offset, cell count, first, last, tw1 lo/hi, tw2 lo/hi}: one 32-B load per beam step
    uint32_t        n_reads;
    uint32_t        pad;
    const uint32_t* set_order;  // [n_cells] or null: host-given iteration order of every read's position set (include/floria_hip.h), used by the reference-arithmetic mode
};

// A ContigDev is loaded from memory, so hipcc cannot tell that its pointers are global: loads through them would be FLAT loads, which
// count in lgkmcnt as well as vmcnt (every LDS wait then also waits for them) and take the slower flat path.  G(p) asserts "global".
template <class T> __device__ __forceinline__ const __attribute__((address_space(1))) T* G(const T* p) {
    return (const __attribute__((address_space(1))) T*)p;
}

// Index of (absolute SNP position, allele) in the linear-hash multiplier tables.  Any window of <= HASH_M consecutive
// positions maps injectively, and the index does not depend on the block, so a read's hash constant is precomputed once.
constexpr uint32_t HASH_M = 65536;
__host__ __device__ __forceinline__ uint32_t hash_idx(uint32_t snp, uint32_t allele) { return ((snp & (HASH_M - 1)) << 2) | allele; }

// One batch of SNP blocks (the work units of graph_processing.rs:345-362).
struct BlockSet {
    const ContigDev* contigs;
    const uint32_t*  blk_contig;    // [n_blocks]
    const uint32_t*  blk_start;     // [n_blocks] SNP range (1-based inclusive)
    const uint32_t*  blk_end;
    const uint32_t*  blk_pos0;      // [n_blocks] min first_position over the block's reads
    const uint32_t*  blk_span;      // [n_blocks] max last_position - pos0 + 1
    const uint64_t*  blk_read_off;  // [n_blocks+1] into blk_read
    const uint32_t*  blk_read;      // read ids (ascending) of every block, find_reads_in_interval
    uint32_t         n_blocks;
    uint32_t         pad;
};

// per-partition multipliers of the linear state hash (a by-value kernel-argument array indexed by lane would be
// copied to scratch)
__constant__ uint64_t c_rk1[MAX_PLOIDY], c_rk2[MAX_PLOIDY];

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
    return __longlong_as_double((long long)shfl_u64((uint64_t)__double_as_longlong(v), src));
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o), hi = __shfl_xor((uint32_t)(v >> 32), o);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double w = __longlong_as_double((long long)(((uint64_t)__shfl_xor((uint32_t)((uint64_t)__double_as_longlong(v) >> 32), o) << 32) |
                                                   __shfl_xor((uint32_t)(uint64_t)__double_as_longlong(v), o)));
        v = w < v ? w : v;
    }
    return v;
}

// value = Q*2^-24 + m*eps — the canonical conversion (DESIGN.md "Arithmetic"); compiled with
// -ffp-contract=off so the multiply and add round separately exactly like the host oracle.
__device__ __forceinline__ double qm_to_f64(uint64_t q, uint64_t m, double eps) {
    return (double)q * 0x1p-24 + (double)m * eps;
}
// same value for a 32-bit count (one v_cvt_f64_u32 instead of the u64 -> f64 sequence)
__device__ __forceinline__ double qm_to_f64(uint64_t q, uint32_t m, double eps) {
    return (double)q * 0x1p-24 + (double)m * eps;
}
__device__ __forceinline__ double qm_to_f64(uint64_t q, int m, double eps) { return qm_to_f64(q, (uint32_t)m, eps); }

// stable_binom_cdf_p_rev (utils_frags.rs:211-248) with device libm — only used beyond the host-built
// table (n > binom_nmax); see beam kernel.
// (out of line: only reached for n beyond the host-built table, and its libm temporaries would otherwise set the callers' register peak)
__device__ __attribute__((noinline)) double binom_device(uint64_t n, uint64_t k, double p, double div_factor) {
    if (n == 0) return 0.0;
    double n64 = (double)n, k64 = (double)k;
    double a = k64 / n64;
    if (a == 1.0) a = 0.9999999;
    if (a == 0.0) a = 0.0000001;
    double rel_ent = a * log(a / p) + (1.0 - a) * log((1.0 - a) / (1.0 - p));
    if (a < p) rel_ent = -rel_ent;
    return -1.0 * n64 / div_factor * rel_ent;
}

}  // namespace fl
