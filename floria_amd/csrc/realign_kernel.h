// realign_kernel.h — alignment::realign (alignment.rs:7-64) for the calls the host's exact shortcut cannot decide: around a SNP call the
// read's 32 bases are globally aligned to the reference's 32 bases with every candidate allele in the SNP column (match +1, mismatch -1,
// gap open -2, gap extend -1: NW1 / Gaps { open: -2, extend: -1 }, alignment.rs:15-18) and the FIRST best-scoring allele replaces the call.
//
// One wavefront per window, one half-wavefront per allele (lanes 0-31 allele a, lanes 32-63 allele a+1; two passes for 3-4 alleles).
// The 32 x 32 affine-gap DP (three matrices: M ends in a pair, I consumes a read base against a gap, D a reference base) runs as a
// systolic array over the anti-diagonals: lane c owns column c+1, at step t it computes row t-c+1.  What a cell needs from the column on its
// left arrives with ONE DPP wave shift per matrix (the left lane finished that row in the previous step), the diagonal is what arrived one
// step earlier, the cell above is the lane's own previous value; the read base of the row travels down the lanes the same way.  63 steps of
// ~25 integer instructions, no LDS, no memory traffic inside the loop.  Integer arithmetic: results are exact and equal the host DP
// (floria_amd/host/ingest.cpp: nw_affine_score) bit for bit; block-aligner, which the reference calls, is an adaptive-band
// approximation of exactly this score.
#pragma once
#include "common.h"

namespace fl {

struct RealignArgs {
    const uint8_t* q;         // [n][32] read window (A C G T)
    const uint8_t* r;         // [n][32] reference window; column 16 is replaced by the candidate allele
    const uint8_t* alleles;   // [n][FLORIA_MAX_ALLELES] candidate bases
    const uint8_t* n_alleles; // [n] 1..FLORIA_MAX_ALLELES
    uint8_t* best;            // [n] index of the first allele with the maximal score
    int32_t* score;           // [n] that score (tests), may be null
    uint64_t n;
};

template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

__global__ __launch_bounds__(256) void realign_kernel(RealignArgs g) {
    constexpr int W = 32, FLANK = 16, MATCH = 1, MISMATCH = -1, OPEN = -2, EXTEND = -1, NEG = -(1 << 28);
    const uint32_t lane = threadIdx.x & 63;
    const int c = (int)(lane & 31);
    const uint32_t half = lane >> 5;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), stride = (uint64_t)gridDim.x * (blockDim.x >> 6);
    for (uint64_t w = wave; w < g.n; w += stride) {
        const int qb = g.q[w * W + c];                                  // lane k of either half holds q[k]
        const int rb0 = g.r[w * W + c];
        const uint32_t na = g.n_alleles[w];
        int best_score = INT32_MIN;
        uint32_t best = 0;
        for (uint32_t a0 = 0; a0 < na; a0 += 2) {
            const uint32_t a = a0 + half;
            const int ab = g.alleles[w * FLORIA_MAX_ALLELES + (a < na ? a : na - 1)];
            const int rb = c == FLANK ? ab : rb0;
            // row 0 of column c+1: M = I = -inf, D = open + c * extend; the diagonal of the first row is row 0 of column c
            int M = NEG, I = NEG, D = OPEN + c * EXTEND;
            int diag_best = c == 0 ? 0 : OPEN + (c - 1) * EXTEND;
            int qsh = 0;
#pragma unroll 1
            for (int t = 0; t < 2 * W - 1; ++t) {
                const int i = t - c + 1;                                  // the row this lane computes now
                const bool active = i >= 1 && i <= W;
                // from the column on the left (wave_shr:1): its row i, finished in the previous step; column 0 for lane 0 of either half
                int lM = dpp_i32<0x138>(M), lI = dpp_i32<0x138>(I), lD = dpp_i32<0x138>(D);
                const int qin = __builtin_amdgcn_readlane(qb, t & (W - 1));   // q[t], injected at column 1 and handed down one lane per step
                qsh = dpp_i32<0x138>(qsh);
                if (c == 0) { lM = NEG; lI = OPEN + (i - 1) * EXTEND; lD = NEG; qsh = qin; }
                const int sub = qsh == rb ? MATCH : MISMATCH;               // q[i-1] against r[c]
                const int nM = diag_best + sub;
                const int mu = M > D ? M : D, nI0 = mu + OPEN, nI1 = I + EXTEND;
                const int nI = nI0 > nI1 ? nI0 : nI1;
                const int ml = lM > lI ? lM : lI, nD0 = ml + OPEN, nD1 = lD + EXTEND;
                const int nD = nD0 > nD1 ? nD0 : nD1;
                const int lb = ml > lD ? ml : lD;
                if (active) { M = nM; I = nI; D = nD; diag_best = lb; }
            }
            const int mx = M > I ? (M > D ? M : D) : (I > D ? I : D);     // lane 31 / 63: cell (32, 32)
            const int s0 = __builtin_amdgcn_readlane(mx, 31), s1 = __builtin_amdgcn_readlane(mx, 63);
            if (s0 > best_score) { best_score = s0; best = a0; }
            if (a0 + 1 < na && s1 > best_score) { best_score = s1; best = a0 + 1; }
        }
        if (lane == 0) { g.best[w] = (uint8_t)best; if (g.score) g.score[w] = best_score; }
    }
}

}  // namespace fl
