// wave_util.h — wavefront-level helpers shared by the beam, reassign and upload kernels: lane reads/writes with wave-uniform indices
// and std::collections::BinaryHeap held in registers, one heap slot per lane.
#pragma once
#include "beam_kernel.h"

namespace fl {

constexpr int SLAB_WAVES = 3;    // waves per SIMD of the beam_slab_kernel instances that are LDS-limited (ploidy >= 4, runtime-parameter instances)

__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t l) { return ((uint64_t)rl32((uint32_t)(v >> 32), l) << 32) | rl32((uint32_t)v, l); }
// "writelane": x and l are wave-uniform; a compare + select per dword (clang has no writelane builtin for HIP, and the
// select form lets the compiler handle the VALU->SGPR lane-select hazards itself)
__device__ __forceinline__ void wl32(uint32_t& v, uint32_t x, uint32_t l) { v = (threadIdx.x == l) ? x : v; }
__device__ __forceinline__ void wl64(uint64_t& v, uint64_t x, uint32_t l) { v = (threadIdx.x == l) ? x : v; }
// number of set bits of a wave-uniform mask below this lane (v_mbcnt_lo + v_mbcnt_hi): the rank of the lane among the set lanes
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// scalar (SMEM) load of read-only data at a wave-uniform address: the value lands in SGPRs, costs no VGPR and is
// tracked by lgkmcnt, so it can be requested a whole beam step before it is used
template <class T> __device__ __forceinline__ T sload(const T* p) {
    return *(const __attribute__((address_space(4))) T*)(uintptr_t)p;
}

// v_writelane_b32: lane l of v := x (x and l wave-uniform, in SGPRs): one instruction per dword instead of move + compare + select.
// gfx9 allows one SGPR operand per VALU instruction, so the lane select travels in M0.  M0 is on the clobber list: hipcc re-materialises its own M0 values (the
// LDS base of an LDS-DMA) behind the asm (checked in the ISA; round 3 saved and restored M0 by hand, two more scalar instructions per call, ten calls per beam
// step).  The LANE operand must come from scalar arithmetic, not straight from a v_readlane / v_cmp: the s_mov in between is the only separation the hardware
// gets ("VALU writes SGPR -> lane select" wants wait states that the assembler cannot see here).
__device__ __forceinline__ void wlane(uint32_t& v, uint32_t x, uint32_t l) {
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(x), "s"(l) : "m0");
}
__device__ __forceinline__ void wlane3(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t xa, uint32_t xb, uint32_t xc, uint32_t l) {
    asm("s_mov_b32 m0, %6\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
        : "+v"(a), "+v"(b), "+v"(c) : "s"(xa), "s"(xb), "s"(xc), "s"(l) : "m0");
}

// Scores are non-negative f64 (sums of non-negative terms), so their IEEE bit patterns order like the values:
// the heap compares u64 bit patterns.  Heap slot j lives in lane j: (hp_hi, hp_lo, hp_id) — the two halves of the score in separate
// registers, so that a comparison of two readlane results is three scalar instructions (gfx9 has no ordered 64-bit scalar compare and
// hipcc would bounce the operands through the vector unit).
struct RegHeap {
    uint32_t hp_hi, hp_lo, hp_id;      // per-lane
    uint32_t len;                       // uniform
    static __device__ __forceinline__ bool le(uint32_t ah, uint32_t al, uint32_t bh, uint32_t bl) { return ah < bh || (ah == bh && al <= bl); }   // a <= b
    __device__ __forceinline__ void sift_up(uint32_t pos, uint32_t xh, uint32_t xl, uint32_t xid) {
        while (pos > 0) {
            const uint32_t par = (pos - 1) >> 1;
            const uint32_t ph = rl32(hp_hi, par), pl = rl32(hp_lo, par);
            if (le(xh, xl, ph, pl)) break;
            const uint32_t pid = rl32(hp_id, par);
            wlane3(hp_hi, hp_lo, hp_id, ph, pl, pid, pos);
            pos = par;
        }
        wlane3(hp_hi, hp_lo, hp_id, xh, xl, xid, pos);
    }
    __device__ __forceinline__ void push(uint64_t xs, uint32_t xid) { const uint32_t pos = len++; sift_up(pos, (uint32_t)(xs >> 32), (uint32_t)xs, xid); }
    __device__ __forceinline__ void move(uint32_t from, uint32_t to) { wlane3(hp_hi, hp_lo, hp_id, rl32(hp_hi, from), rl32(hp_lo, from), rl32(hp_id, from), to); }
    __device__ __forceinline__ bool slot_le(uint32_t a, uint32_t b) const { return le(rl32(hp_hi, a), rl32(hp_lo, a), rl32(hp_hi, b), rl32(hp_lo, b)); }   // data[a] <= data[b]
    __device__ __forceinline__ uint32_t pop() {                    // returns the evicted (max) entry id
        --len;
        const uint32_t xh = rl32(hp_hi, len), xl = rl32(hp_lo, len), xid = rl32(hp_id, len);
        if (len == 0) return xid;
        const uint32_t root = rl32(hp_id, 0);
        const uint32_t end = len, lim = end >= 2 ? end - 2 : 0;
        uint32_t pos = 0, child = 1;
        while (child <= lim) {                                     // sift_down_to_bottom(0)
            if (slot_le(child, child + 1)) child++;
            move(child, pos);
            pos = child;
            child = 2 * pos + 1;
        }
        if (child == end - 1) { move(child, pos); pos = child; }
        sift_up(pos, xh, xl, xid);
        return root;
    }
    // into_sorted_vec()[0]: heap-sort in place, return the id at array position 0
    __device__ __forceinline__ uint32_t sorted_first() {
        uint32_t end = len;
        while (end > 1) {
            --end;
            const uint32_t hh = rl32(hp_hi, end), hl = rl32(hp_lo, end), hid = rl32(hp_id, end);      // hole element = old data[end]
            move(0, end);                                                                                // swap(0, end)
            uint32_t pos = 0, child = 1;
            const uint32_t lim = end >= 2 ? end - 2 : 0;
            bool placed = false;
            while (child <= lim) {                                  // sift_down_range(0, end)
                if (slot_le(child, child + 1)) child++;
                const uint32_t ch = rl32(hp_hi, child), cl = rl32(hp_lo, child);
                if (le(ch, cl, hh, hl)) { placed = true; break; }   // hole >= data[child]
                move(child, pos);
                pos = child;
                child = 2 * pos + 1;
            }
            if (!placed && child == end - 1) {
                const uint32_t ch = rl32(hp_hi, child), cl = rl32(hp_lo, child);
                if (!le(ch, cl, hh, hl)) { move(child, pos); pos = child; }      // hole < data[child]
            }
            wlane3(hp_hi, hp_lo, hp_id, hh, hl, hid, pos);
        }
        return rl32(hp_id, 0);
    }
};

}  // namespace fl
