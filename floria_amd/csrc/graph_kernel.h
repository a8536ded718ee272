// graph_kernel.h — hap-graph nodes and edges right after S1, on the data that is still resident.
//
// HapNode::new (types_structs.rs:168-209): hap_map = phred histogram of the node's reads restricted to the block's SNP
// endpoints; cov = allele counts sorted ascending, element [len*2/3].
// update_hap_graph (graph_processing.rs:22-100): for consecutive non-empty blocks (b, b') of a contig and every read of
// node k1 of b: diff_l = round(distance_read_haplo(read, node l of b').1) (utils_frags.rs:77-108: positions absent from the
// node's map are skipped; a tie with the consensus never counts as diff); the read is unambiguous iff the two smallest
// diff_l differ (or b' has one node); an unambiguous read that also sits in node k2 of b' adds 1 to weight[k1][k2].
// Only reads present in BOTH blocks can contribute (hap_id_in, :33-35), so the job walks the intersection of the two
// ascending read lists.
//
// One 256-thread workgroup per target block b': histogram [pos][node][allele] (count << 44 | Q24) in LDS, 16 lanes per read
// with a DPP row reduction (as optimize_kernel), LDS counters for the weight matrix.
#pragma once
#include "optimize_kernel.h"

namespace fl {

constexpr int GRAPH_THREADS = 256;
constexpr int GRAPH_SORT_CAP = 2048;     // allele counts of one node sorted in LDS (range * alleles); larger blocks sort in HBM scratch

struct GraphArgs {
    BlockSet bs;
    const uint32_t* best_ploidy;    // [n_blocks]
    const uint8_t*  part;           // final partition of every read of every block (S1 output plane)
    const int32_t*  pred;           // [n_blocks] previous non-empty block of the same contig, or -1
    const uint64_t* node_off;       // [n_blocks+1]
    const uint64_t* edge_off;       // [n_blocks+1]
    double*   node_cov;
    uint32_t* edge_w;
    uint64_t* hist_pool;            // [grid][range_max * pmax * A] when the histogram does not fit LDS
    uint64_t* sort_pool;            // [grid][sort_cap]
    uint64_t  sort_cap;             // pow2 >= range_max * A
    uint32_t  range_max, hist_in_lds;
    uint64_t  hist_stride;          // u64 cells per block in hist_pool
};

template <int A>
__global__ __launch_bounds__(GRAPH_THREADS) void graph_kernel(GraphArgs g) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint64_t s_sort[GRAPH_SORT_CAP];
    __shared__ uint32_t s_w[MAX_PLOIDY * MAX_PLOIDY];
    __shared__ uint32_t s_cnt;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    if (b >= g.bs.n_blocks) return;
    const uint32_t p2 = g.best_ploidy[b];
    if (p2 == 0) return;
    const ContigDev cd = g.bs.contigs[g.bs.blk_contig[b]];
    const uint32_t lo = g.bs.blk_start[b], hi = g.bs.blk_end[b], range = hi - lo + 1;
    const uint64_t roff = g.bs.blk_read_off[b];
    const uint32_t n = (uint32_t)(g.bs.blk_read_off[b + 1] - roff);
    const uint32_t* reads = g.bs.blk_read + roff;
    const uint8_t* part = g.part + roff;
    const uint32_t PA = p2 * A;
    uint64_t* hist = g.hist_in_lds ? (uint64_t*)smem : g.hist_pool + (uint64_t)b * g.hist_stride;
    const uint32_t grp = tid >> 4, sub = tid & 15;

    // ---- hap_map of every node of b' (restricted to [lo, hi]) -------------------------------------------------------
    for (uint32_t x = tid; x < range * PA; x += GRAPH_THREADS) hist[x] = 0;
    for (uint32_t x = tid; x < MAX_PLOIDY * MAX_PLOIDY; x += GRAPH_THREADS) s_w[x] = 0;
    __syncthreads();
    const uint32_t n16 = (n + 15) & ~15u;
    for (uint32_t i = grp; i < n16; i += GRAPH_THREADS / 16) {
        if (i < n) {
            const uint32_t r = reads[i], k = part[i];
            const uint32_t cb = G(cd.read_off)[r], ce = G(cd.read_off)[r + 1];
            for (uint32_t c = cb + sub; c < ce; c += 16) {
                const uint32_t sn = G(cd.cell_snp)[c], aq = G(cd.cell_aw)[c];
                if (sn >= lo && sn <= hi)
                    atomicAdd((unsigned long long*)&hist[(uint64_t)(sn - lo) * PA + k * A + (aq >> 28)], (unsigned long long)((1ull << CNT_SHIFT) | (aq & 0x0fffffffu)));
            }
        }
    }
    __syncthreads();

    // ---- cov of every node: sorted allele counts [len*2/3] (types_structs.rs:179-193) -----------------------------------
    const bool sort_lds = (uint64_t)range * A <= GRAPH_SORT_CAP;
    uint64_t* sortbuf = g.sort_pool + (uint64_t)b * g.sort_cap;
    for (uint32_t k = 0; k < p2; ++k) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for (uint32_t x = tid; x < range * A; x += GRAPH_THREADS) {
            const uint64_t v = hist[(uint64_t)(x / A) * PA + k * A + (x % A)];
            if (v >> CNT_SHIFT) {                                   // allele key exists
                const uint32_t idx = atomicAdd(&s_cnt, 1u);
                if (sort_lds) s_sort[idx] = v & QMASK44; else sortbuf[idx] = v & QMASK44;
            }
        }
        __syncthreads();
        const uint32_t len = s_cnt;
        double cov = 0.0;
        if (len) {
            uint32_t m2 = 1;
            while (m2 < len) m2 <<= 1;
            if (sort_lds) { for (uint32_t x = len + tid; x < m2; x += GRAPH_THREADS) s_sort[x] = ~0ull; }
            else { for (uint32_t x = len + tid; x < m2; x += GRAPH_THREADS) sortbuf[x] = ~0ull; }
            __syncthreads();
            uint64_t* a = sort_lds ? s_sort : sortbuf;
            for (uint32_t kk = 2; kk <= m2; kk <<= 1)
                for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = tid; i < m2; i += GRAPH_THREADS) {
                        const uint32_t ixj = i ^ j;
                        if (ixj > i) {
                            const uint64_t x = a[i], y = a[ixj];
                            const bool up = (i & kk) == 0;
                            if (up ? x > y : x < y) { a[i] = y; a[ixj] = x; }
                        }
                    }
                    __syncthreads();
                }
            cov = (double)a[len * 2 / 3] * 0x1p-24;
        }
        if (tid == 0) g.node_cov[g.node_off[b] + k] = cov;
        __syncthreads();
    }

    // ---- edges pred(b') -> b' -----------------------------------------------------------------------------------------------
    const int32_t pb = g.pred[b];
    if (pb < 0) return;
    const uint32_t p1 = g.best_ploidy[pb];
    const uint64_t roff1 = g.bs.blk_read_off[pb];
    const uint32_t n1 = (uint32_t)(g.bs.blk_read_off[pb + 1] - roff1);
    const uint32_t* reads1 = g.bs.blk_read + roff1;
    const uint8_t* part1 = g.part + roff1;
    const uint32_t n1_16 = (n1 + 15) & ~15u;
    for (uint32_t i = grp; i < n1_16; i += GRAPH_THREADS / 16) {
        bool common = false;
        uint32_t r = 0, k1 = 0, k2 = 0, cb = 0, ce = 0;
        if (i < n1) {
            r = reads1[i]; k1 = part1[i];
            uint32_t l = 0, h = n;                                  // read lists ascend: binary search r in b'
            while (l < h) { const uint32_t mid = (l + h) >> 1; if (reads[mid] < r) l = mid + 1; else h = mid; }
            if (l < n && reads[l] == r) { common = true; k2 = part[l]; cb = G(cd.read_off)[r]; ce = G(cd.read_off)[r + 1]; }
        }
        uint64_t acc[MAX_PLOIDY];
#pragma unroll
        for (int l = 0; l < MAX_PLOIDY; ++l) acc[l] = 0;
        if (common) {
            for (uint32_t c = cb + sub; c < ce; c += 16) {
                const uint32_t sn = G(cd.cell_snp)[c], aq = G(cd.cell_aw)[c], al = aq >> 28;
                if (sn < lo || sn > hi) continue;                    // outside the block: not a key of any node's hap_map
                const uint64_t w = (aq & 0x0fffffffu);
                const uint64_t* row = hist + (uint64_t)(sn - lo) * PA;
#pragma unroll
                for (int l = 0; l < MAX_PLOIDY; ++l) {
                    if ((uint32_t)l < p2) {
                        uint64_t mx = 0, cn = 0, va = ~0ull;
#pragma unroll
                        for (int x = 0; x < A; ++x) {
                            const uint64_t v = row[l * A + x];
                            if (v >> CNT_SHIFT) { const uint64_t q = v & QMASK44; mx = q > mx ? q : mx; cn = 1; if (x == (int)al) va = q; }
                        }
                        if (cn && va != mx) acc[l] += w;             // absent allele (va = ~0) or strictly below the consensus
                    }
                }
            }
        }
        uint32_t nmin = 0;
        uint64_t dmin = ~0ull;
#pragma unroll
        for (int l = 0; l < MAX_PLOIDY; ++l) {
            if ((uint32_t)l < p2) {
                const uint64_t d = (row16_sum_u64(acc[l]) + (1ull << 23)) >> 24;      // .round() as usize (:107)
                if (d < dmin) { dmin = d; nmin = 1; } else if (d == dmin) nmin++;
            }
        }
        if (common && sub == 0 && (p2 == 1 || nmin == 1)) atomicAdd(&s_w[k1 * p2 + k2], 1u);          // :41-54
    }
    __syncthreads();
    for (uint32_t x = tid; x < p1 * p2; x += GRAPH_THREADS) g.edge_w[g.edge_off[b] + x] = s_w[x];
}

}  // namespace fl
