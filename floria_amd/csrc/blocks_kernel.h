// blocks_kernel.h — find_reads_in_interval (local_clustering.rs:12-59) for every SNP block, on the device.
//
// Reads are sorted by first_position, so the candidates of a block are one binary-searched range
// [lower_bound(first >= start-10000), upper_bound(first <= end)) — reads that pass `last - first <= 10000`
// and `last >= start` cannot start earlier — filtered by the two predicates of :36-46.  One wavefront per block:
// pass 1 counts (and reduces min first / max last and the SURVEY.md §8(d) algorithmic bytes), the host turns the
// counts into offsets, pass 2 writes the ascending read-id lists with a ballot-ordered compaction.
#pragma once
#include "common.h"

namespace fl {

struct ScanArgs {
    const ContigDev* contigs;
    const uint32_t *blk_contig, *blk_start, *blk_end;
    uint32_t n_blocks, max_ploidy;
    uint32_t *cnt, *pos0, *span;       // out (pass 1)
    uint64_t* bytes;                   // out (pass 1)
    uint32_t *h_cnt, *h_span;          // out (pass 1): the same counts written straight into pinned host memory (null = not wanted); the host
    uint64_t* h_bytes;                 //   waits for the kernel instead of for three small copies that would queue behind the pileup's upload
    const uint64_t* roff;              // in  (pass 2)
    uint32_t* rids;                    // out (pass 2)
};

__device__ inline uint32_t lower_bound_u32(const uint32_t* a, uint32_t n, uint32_t key) {     // first i with a[i] >= key
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ inline uint32_t upper_bound_u32(const uint32_t* a, uint32_t n, uint32_t key) {     // first i with a[i] > key
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
    return lo;
}

template <bool FILL>
__global__ __launch_bounds__(64) void block_reads_kernel(ScanArgs g) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (b >= g.n_blocks) return;
    const ContigDev cd = g.contigs[g.blk_contig[b]];
    const uint32_t start = g.blk_start[b], end = g.blk_end[b];
    const uint32_t lo = lower_bound_u32(cd.first, cd.n_reads, start > 10000 ? start - 10000 : 0);
    const uint32_t hi = upper_bound_u32(cd.first, cd.n_reads, end);
    uint32_t cnt = 0, mn = 0xffffffffu, mx = 0;
    uint64_t bytes = 0;
    uint64_t base = FILL ? g.roff[b] : 0;
    const uint64_t lane_lt = (1ull << lane) - 1;
    for (uint32_t r0 = lo; r0 < hi; r0 += 64) {
        const uint32_t r = r0 + lane;
        bool pass = false;
        if (r < hi) {
            const uint32_t F = G(cd.first)[r], La = G(cd.last)[r];
            pass = La >= start && La - F <= 10000;                 // :36-38, :44-46  (first <= end by construction, :39-41)
            if (pass && !FILL) {
                cnt++; mn = F < mn ? F : mn; mx = La > mx ? La : mx;
                const uint64_t L = G(cd.read_off)[r + 1] - G(cd.read_off)[r];
                bytes += 8 + (L + 3) / 4 + (L + 7) / 8 + L;
            }
        }
        if (FILL) {
            const uint64_t m = __ballot(pass);
            if (pass) g.rids[base + __popcll(m & lane_lt)] = r;
            base += __popcll(m);
        }
    }
    if (!FILL) {
        cnt = wave_sum_u32(cnt);
        bytes = wave_sum_u64(bytes);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t a = __shfl_xor(mn, o), c = __shfl_xor(mx, o); mn = a < mn ? a : mn; mx = c > mx ? c : mx; }
        if (lane == 0) {
            g.cnt[b] = cnt;
            g.pos0[b] = cnt ? mn : 0;
            g.span[b] = cnt ? mx - mn + 1 : 0;
            g.bytes[b] = cnt ? 16 + bytes + cnt + 8ull * g.max_ploidy + 4 : 0;
            if (g.h_cnt) { g.h_cnt[b] = cnt; g.h_span[b] = cnt ? mx - mn + 1 : 0; g.h_bytes[b] = cnt ? 16 + bytes + cnt + 8ull * g.max_ploidy + 4 : 0; }
        }
    }
}

}  // namespace fl
