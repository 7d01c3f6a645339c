// Hot-pixel collection of the ChESS response kernels (chess.hip: 8 pixels per lane, chess16.hip: 16): records per aligned
// 8-pixel group in wave-private LDS segments, ONE global atomic per workgroup at the flush.  Device code only.
#pragma once
#include "common.h"
#include "hotlist.h"

namespace mrg {

// Hot pixels of a workgroup are collected in LDS and appended to the frame's list with ONE global atomic per
// workgroup: per-pixel (even per-wave) returning atomics on the frame counter serialise and were costing more than
// the response itself on the small pyramid levels.  What a lane collects is a RECORD per aligned 8-pixel group with
// a hot pixel -- (y << 16 | x0, 8-bit mask) -- in a segment of the buffer that belongs to its wave, at a position
// that is the wave's own running count (a scalar register) plus the lane's rank among the lanes that have one
// (one ballot): no atomic and no round trip in the loop.  (Round 2 expanded the pixels at once, with four ballots
// for the prefix of the per-lane counts and a returning LDS atomic per wave and iteration: +20 % on the level-0
// launch for a textured frame, where nearly every wave-iteration has a hot pixel.)  The flush expands the
// records into list entries (y << 16) | x -- a group's pixels consecutive, in ascending x -- and writes the
// pixel -> index record of every group.
constexpr int V1_HOTBUF = 768;            // words = records: mask | group column << 8 | row within the segment << 13
constexpr int V1_HOTSEG = V1_HOTBUF / 4;  // records per wave (192)

struct HotSink {          // where a workgroup's records go
    uint32_t* seg;        // this wave's segment of the buffer (LDS)
    int ys, strip_x;      // origin of the record coordinates: first row of the segment, first column of the strip
    int count;            // records in the segment (wave-uniform: a scalar register)
};

// Expands records [0, nrec) of a wave's segment into list entries starting at list index `base`, and writes the
// pixel -> index entry of every group.  All lanes of the wave call it.
__device__ __forceinline__ void expand_records(const HotSink& hs, int nrec, int base, const CompTables& t, int frame) {
    const int lane = threadIdx.x & 63;
    uint32_t* hot = t.hot_xy + (long long)frame * t.cap;
    for (int i0 = 0; i0 < nrec; i0 += 64) {
        const int i = i0 + lane;
        const uint32_t r = i < nrec ? hs.seg[i] : 0u;
        const uint32_t mask = r & 0xffu;
        const int c = __popc(mask);
        int incl = c;  // inclusive prefix over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const int chunk = __shfl(incl, 63);
        if (c) {
            int k = base + incl - c;
            const int y = hs.ys + (int)(r >> 13), x0 = hs.strip_x + 8 * (int)((r >> 8) & 31u);
            t.gidx[(long long)frame * t.gidx_pitch + (long long)y * t.gw + (x0 >> 3)] = make_uint2((uint32_t)k, mask);
            const uint32_t xy0 = ((uint32_t)y << 16) | (uint32_t)x0;
            uint32_t b = mask;
            while (b) {
                const int j = __ffs(b) - 1;
                b &= b - 1;
                if (k < t.cap) hot[k] = xy0 + (uint32_t)j;
                ++k;
            }
        }
        base += chunk;
    }
}

// hot pixels in records [0, nrec) of the wave's segment (wave-uniform result)
__device__ __forceinline__ int count_record_pixels(const HotSink& hs, int nrec) {
    const int lane = threadIdx.x & 63;
    int mine = 0;
    for (int i = lane; i < nrec; i += 64) mine += __popc(hs.seg[i] & 0xffu);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    return mine;
}

// All lanes of the wave call this; `bits` has bit i set when pixel i of the lane's group (row y, columns
// x0 .. x0 + 7, x0 a multiple of 8 inside the strip) is hot.
__device__ __forceinline__ void collect_hot(uint32_t bits, int y, int x0, HotSink& hs, const CompTables& t, int frame) {
    const unsigned long long m = __ballot(bits != 0);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    const int k = hs.count + rank;
    hs.count += __popcll(m);
    if (bits == 0) return;
    if (k < V1_HOTSEG) {
        hs.seg[k] = bits | ((uint32_t)((x0 - hs.strip_x) >> 3) << 8) | ((uint32_t)(y - hs.ys) << 13);
    } else {
        // rare (a wave's strip rows hold more than 192 groups with a hot pixel: dense texture): the group goes
        // straight to the list, one returning global atomic per lane
        write_group_direct(t, frame, atomicAdd(t.hot_cnt + frame, __popc(bits)), bits, ((uint32_t)y << 16) | (uint32_t)x0);
    }
}

// Flush at the end of the workgroup: one global atomic for the four waves' records.  Called by all 256 threads.
__device__ __forceinline__ void flush_hot(const HotSink& hs, int* hotcnt, int wave, const CompTables& t, int frame) {
    const int nrec = min(hs.count, V1_HOTSEG);  // (what did not fit went to the list directly)
    const int px = count_record_pixels(hs, nrec);
    __syncthreads();  // hotcnt is free
    if ((threadIdx.x & 63) == 0) hotcnt[wave] = px;
    __syncthreads();
    const int total = hotcnt[0] + hotcnt[1] + hotcnt[2] + hotcnt[3];
    if (total == 0) return;  // uniform
    if (threadIdx.x == 0) hotcnt[4] = atomicAdd(t.hot_cnt + frame, total);
    __syncthreads();
    int base = hotcnt[4];
    for (int w = 0; w < wave; ++w) base += hotcnt[w];
    expand_records(hs, nrec, base, t, frame);
}

}  // namespace mrg
