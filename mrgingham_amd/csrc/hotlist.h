// Device-side helpers that append "hot" pixels (response > kRespMin) to a frame's hot list and
// maintain the pixel -> list-index map (CompTables::gidx).  Shared by the ChESS kernels (fused
// epilogue) and by the kernel that builds the list from a caller-supplied response.
#pragma once
#include "common.h"

namespace mrg {

// Exclusive prefix and total, over the wave, of a per-lane count in 0..8.
__device__ __forceinline__ void wave_prefix_0to8(int cnt, int& prefix, int& total) {
    prefix = 0;
    total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const unsigned long long m = __ballot((cnt >> b) & 1);
        prefix += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0)) << b;
        total += __popcll(m) << b;
    }
}

// A lane's aligned 8-pixel group (pixels xy0 + i for the set bits i of `bits`, xy0 = (y << 16) | x0,
// x0 a multiple of 8) goes to list entries base, base+1, ... in ascending x.
__device__ __forceinline__ void write_group_direct(const CompTables& t, int frame, int base, uint32_t bits,
                                                   uint32_t xy0) {
    const int x0 = (int)(xy0 & 0xffffu), y = (int)(xy0 >> 16);
    t.gidx[(long long)frame * t.gidx_pitch + (long long)y * t.gw + (x0 >> 3)] = make_uint2((uint32_t)base, bits);
    uint32_t* hot = t.hot_xy + (long long)frame * t.cap;
    int k = base;
    while (bits) {
        const int i = __ffs(bits) - 1;
        bits &= bits - 1;
        if (k < t.cap) hot[k] = xy0 + (uint32_t)i;
        ++k;
    }
}

// Appends the groups of a whole wave with ONE returning atomic on the frame counter.  All lanes of
// the wave must call it (bits may be 0).
__device__ __forceinline__ void append_groups_wave(const CompTables& t, int frame, uint32_t bits, uint32_t xy0) {
    int prefix, total;
    wave_prefix_0to8(__popc(bits), prefix, total);
    if (total == 0) return;
    int base = 0;
    if (__lane_id() == 0) base = atomicAdd(t.hot_cnt + frame, total);
    base = __builtin_amdgcn_readfirstlane(base);
    if (bits) write_group_direct(t, frame, base + prefix, bits, xy0);
}

// pixel (x, y), known to be hot -> its hot-list index
__device__ __forceinline__ int hot_index_of(const uint2* gidx_frame, int gw, int x, int y) {
    const uint2 e = gidx_frame[(long long)y * gw + (x >> 3)];
    return (int)e.x + __popc(e.y & ((1u << (x & 7)) - 1u));
}

}  // namespace mrg
