// hapq_kernel.h — the device half of part_block_manip::get_hapq (part_block_manip.rs:517-616): consensus haplotypes of the
// haplosets and distance_between_haplotypes (utils_frags.rs:659-700) of every overlapping pair.
//
// get_hapq itself is three pieces: (1) get_errors_cov_from_frags per haploset -> stats_kernel.h; (2) for every pair of haplosets
// whose SNP ranges overlap by more than 5 % (find_overlapping_blocks, :453-513 — host, integer ranges) the number of positions
// where the two phred-weighted consensus alleles agree / disagree, over ALL positions present in both (range = (MIN, MAX));
// (3) scalar f64 arithmetic with ln on the host.  Kernels here do (2):
//   consensus_kernel  one workgroup per haploset: count<<44 | Q24 histogram over the haploset's read span (u64 atomics), then one
//                     byte per position = the consensus allele, or 0xff where no read of the haploset has a cell;
//   pair_kernel       one wavefront per pair: compare the two byte arrays over the intersection of the spans.
// Consensus = `.iter().max_by_key(|e| e.1)`: the LAST maximal entry in the inner FxHashMap<u8, _>'s iteration order, which is
// ascending allele for <= 3 present alleles and 0,2,1,3 with all four (derivation: DESIGN.md §6, "inner allele maps").
#pragma once
#include "common.h"

namespace fl {

struct ConsensusArgs {
    const ContigDev* contigs;
    const uint32_t* grp_contig;     // [n_groups]
    const uint64_t* grp_off;        // [n_groups+1] into grp_read
    const uint32_t* grp_read;
    const uint32_t* span_lo;        // [n_groups] smallest first_position of the haploset's reads (1-based SNP index)
    const uint32_t* span_len;       // [n_groups] positions in the span (0 for an empty haploset)
    const uint64_t* cons_off;       // [n_groups+1] byte offset of the haploset's consensus array (= position offset of its histogram)
    unsigned long long* hist;       // zero-initialised, [cons_off[n] * A]
    uint8_t* cons;                  // out
    uint32_t n_groups;
};

constexpr int      HQ_CNT_SHIFT = 44;
constexpr uint64_t HQ_QMASK = (1ull << HQ_CNT_SHIFT) - 1;

template <int A>
__global__ __launch_bounds__(256) void consensus_kernel(ConsensusArgs g) {
    const uint32_t gi = blockIdx.x, tid = threadIdx.x;
    if (gi >= g.n_groups) return;
    const ContigDev cd = g.contigs[g.grp_contig[gi]];
    const uint32_t lo = g.span_lo[gi], len = g.span_len[gi];
    unsigned long long* hist = g.hist + g.cons_off[gi] * A;
    const uint32_t grp = tid >> 4, sub = tid & 15;
    const uint64_t r0 = g.grp_off[gi], r1 = g.grp_off[gi + 1];
    for (uint64_t i = r0 + grp; i < r1; i += 16) {                    // 16 lanes per read (set_to_seq_dict, utils_frags.rs:160-175)
        const uint32_t r = g.grp_read[i];
        const uint32_t cb = G(cd.read_off)[r], ce = G(cd.read_off)[r + 1];
        for (uint32_t c = cb + sub; c < ce; c += 16) {
            const uint32_t aw = G(cd.cell_aw)[c];
            atomicAdd(&hist[(uint64_t)(G(cd.cell_snp)[c] - lo) * A + (aw >> 28)], (1ull << HQ_CNT_SHIFT) | (aw & 0x0fffffffu));
        }
    }
    __syncthreads();
    uint8_t* cons = g.cons + g.cons_off[gi];
    for (uint32_t pr = tid; pr < len; pr += 256) {
        uint64_t q[A];
        uint32_t present = 0;
#pragma unroll
        for (int a = 0; a < A; ++a) { const uint64_t v = hist[(uint64_t)pr * A + a]; q[a] = v & HQ_QMASK; present |= (v >> HQ_CNT_SHIFT) ? (1u << a) : 0u; }
        uint32_t best = 0xff;
        if (present) {
            const bool all4 = A == 4 && present == 15u;
#pragma unroll
            for (int x = 0; x < A; ++x) {
                const int a = all4 ? (x == 1 ? 2 : x == 2 ? 1 : x) : x;       // iteration order of the inner map
                if ((present >> a) & 1u) { if (best == 0xff || q[a] >= q[best]) best = (uint32_t)a; }
            }
        }
        cons[pr] = (uint8_t)best;
    }
}

struct PairArgs {
    const uint32_t* pair_i;         // [n_pairs]
    const uint32_t* pair_j;
    const uint32_t* span_lo;
    const uint32_t* span_len;
    const uint64_t* cons_off;
    const uint8_t* cons;
    uint32_t* same_diff;            // out [2*n_pairs]
    uint32_t n_pairs;
};

__global__ __launch_bounds__(64) void pair_kernel(PairArgs g) {
    const uint32_t pi = blockIdx.x, lane = threadIdx.x;
    if (pi >= g.n_pairs) return;
    const uint32_t i = g.pair_i[pi], j = g.pair_j[pi];
    const uint32_t lo_i = g.span_lo[i], lo_j = g.span_lo[j];
    const uint64_t hi_i = (uint64_t)lo_i + g.span_len[i], hi_j = (uint64_t)lo_j + g.span_len[j];      // exclusive
    const uint32_t lo = lo_i > lo_j ? lo_i : lo_j;
    const uint64_t hi = hi_i < hi_j ? hi_i : hi_j;
    const uint8_t* ci = g.cons + g.cons_off[i];
    const uint8_t* cj = g.cons + g.cons_off[j];
    uint32_t same = 0, diff = 0;
    for (uint64_t p = (uint64_t)lo + lane; p < hi; p += 64) {
        const uint32_t a = ci[p - lo_i], b = cj[p - lo_j];
        if (a != 0xff && b != 0xff) { if (a == b) same++; else diff++; }
    }
    same = wave_sum_u32(same); diff = wave_sum_u32(diff);
    if (lane == 0) { g.same_diff[2 * (uint64_t)pi] = same; g.same_diff[2 * (uint64_t)pi + 1] = diff; }
}

}  // namespace fl
