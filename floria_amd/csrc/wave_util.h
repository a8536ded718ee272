// wave_util.h — wavefront-level helpers shared by the beam, reassign and upload kernels: lane reads/writes with wave-uniform indices
// and std::collections::BinaryHeap held in registers, one heap slot per lane.
#pragma once
#include "beam_kernel.h"

namespace fl {

#ifndef FLORIA_FAST_WAVES
#define FLORIA_FAST_WAVES 3      // waves per SIMD of the beam_slab_kernel instances that are LDS-limited (ploidy >= 4, runtime-parameter instances)
#endif

__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t l) { return ((uint64_t)rl32((uint32_t)(v >> 32), l) << 32) | rl32((uint32_t)v, l); }
// "writelane": x and l are wave-uniform; a compare + select per dword (clang has no writelane builtin for HIP, and the
// select form lets the compiler handle the VALU->SGPR lane-select hazards itself)
__device__ __forceinline__ void wl32(uint32_t& v, uint32_t x, uint32_t l) { v = (threadIdx.x == l) ? x : v; }
__device__ __forceinline__ void wl64(uint64_t& v, uint64_t x, uint32_t l) { v = (threadIdx.x == l) ? x : v; }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// scalar (SMEM) load of read-only data at a wave-uniform address: the value lands in SGPRs, costs no VGPR and is
// tracked by lgkmcnt, so it can be requested a whole beam step before it is used
template <class T> __device__ __forceinline__ T sload(const T* p) {
    return *(const __attribute__((address_space(4))) T*)(uintptr_t)p;
}

// Scores are non-negative f64 (sums of non-negative terms), so their IEEE bit patterns order like the values:
// the heap compares u64 bit patterns.  Heap slot j lives in lane j: (hp_s, hp_id).
struct RegHeap {
    uint64_t hp_s; uint32_t hp_id;     // per-lane
    uint32_t len;                       // uniform
    __device__ __forceinline__ void sift_up(uint32_t pos, uint64_t xs, uint32_t xid) {
        while (pos > 0) {
            const uint32_t par = (pos - 1) >> 1;
            const uint64_t ps = rl64(hp_s, par);
            if (xs <= ps) break;
            const uint32_t pid = rl32(hp_id, par);
            wl64(hp_s, ps, pos); wl32(hp_id, pid, pos);
            pos = par;
        }
        wl64(hp_s, xs, pos); wl32(hp_id, xid, pos);
    }
#ifndef FLORIA_HEAP_PARALLEL_PUSH
    __device__ __forceinline__ void push(uint64_t xs, uint32_t xid) { const uint32_t pos = len++; sift_up(pos, xs, xid); }
#else
    // push = sift_up(0, old_len) of libstd's BinaryHeap, all levels at once: the ancestors of the new slot hold non-increasing values
    // towards the leaf, the new element stops below the DEEPEST ancestor whose value is >= its own (`if x <= parent break`), and the
    // ancestors below that one move down a level.  One ballot finds the stopping ancestor, one permute (each slot reads its parent)
    // moves the chain: no loop, no scalar round trips.  Same final array as the level-by-level loop.  MEASURED SLOWER (beam 137.5 vs 132.5 ms per
    // config-4 step, scripts/ab.sh): most pushes stop at the first comparison, which the scalar loop does in ~6 instructions.  Kept as a variant.
    __device__ __forceinline__ void push(uint64_t xs, uint32_t xid) {
        const uint32_t pos = len++;
        const uint32_t lane = threadIdx.x;
        const uint32_t pj = lane + 1, pp = pos + 1;
        const int dl = (int)__clz(pj) - (int)__clz(pp);                       // level(pos) - level(lane)
        const bool anc = dl >= 1 && (pp >> dl) == pj;
        const uint64_t m = __ballot(anc && hp_s >= xs);
        const int stop = m ? 63 - (int)__clzll((long long)m) : -1;
        const uint32_t par = (lane - 1) >> 1;                                  // (lane 0: unused)
        const uint64_t ps = shfl_u64(hp_s, (int)par);
        const uint32_t pid = __shfl(hp_id, (int)par);
        const bool on_chain = anc || lane == pos;
        const bool take = on_chain && lane > 0 && (int)par > stop;
        const bool land = on_chain && (lane == 0 ? stop < 0 : (int)par == stop);
        hp_s = land ? xs : (take ? ps : hp_s);
        hp_id = land ? xid : (take ? pid : hp_id);
    }
#endif
    __device__ __forceinline__ uint32_t pop() {                    // returns the evicted (max) entry id
        --len;
        const uint64_t xs = rl64(hp_s, len);
        const uint32_t xid = rl32(hp_id, len);
        if (len == 0) return xid;
        const uint32_t root = rl32(hp_id, 0);
        const uint32_t end = len, lim = end >= 2 ? end - 2 : 0;
        uint32_t pos = 0, child = 1;
        while (child <= lim) {                                     // sift_down_to_bottom(0)
            if (rl64(hp_s, child) <= rl64(hp_s, child + 1)) child++;
            wl64(hp_s, rl64(hp_s, child), pos); wl32(hp_id, rl32(hp_id, child), pos);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { wl64(hp_s, rl64(hp_s, child), pos); wl32(hp_id, rl32(hp_id, child), pos); pos = child; }
        sift_up(pos, xs, xid);
        return root;
    }
    // into_sorted_vec()[0]: heap-sort in place, return the id at array position 0
    __device__ __forceinline__ uint32_t sorted_first() {
        uint32_t end = len;
        while (end > 1) {
            --end;
            const uint64_t s0 = rl64(hp_s, 0), se = rl64(hp_s, end);
            const uint32_t i0 = rl32(hp_id, 0), ie = rl32(hp_id, end);
            wl64(hp_s, s0, end); wl32(hp_id, i0, end);              // swap(0, end)
            const uint64_t hs = se; const uint32_t hid = ie;        // hole element = old data[end], now at 0
            uint32_t pos = 0, child = 1;
            const uint32_t lim = end >= 2 ? end - 2 : 0;
            bool placed = false;
            while (child <= lim) {                                  // sift_down_range(0, end)
                if (rl64(hp_s, child) <= rl64(hp_s, child + 1)) child++;
                const uint64_t cs = rl64(hp_s, child);
                if (hs >= cs) { placed = true; break; }
                wl64(hp_s, cs, pos); wl32(hp_id, rl32(hp_id, child), pos);
                pos = child;
                child = 2 * pos + 1;
            }
            if (!placed && child == end - 1) {
                const uint64_t cs = rl64(hp_s, child);
                if (hs < cs) { wl64(hp_s, cs, pos); wl32(hp_id, rl32(hp_id, child), pos); pos = child; }
            }
            wl64(hp_s, hs, pos); wl32(hp_id, hid, pos);
        }
        return rl32(hp_id, 0);
    }
};

}  // namespace fl
