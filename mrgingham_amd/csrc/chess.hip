// ChESS radius-5 response kernels for gfx950 (wave64).
//
// What is computed: exactly mrgingham_ChESS_response_5 (ChESS.c:56-106) -- 16
// ring samples at radius 5 and a 3-pixel horizontal local mean per interior
// pixel, pure integer -- optionally fused with what the reference does right
// after it (find_chessboard_corners.cc:506-529): the zeroed 7-pixel frame, the
// clamp of negative responses to 0, and (new here) the compaction of "hot"
// pixels (response > 15, the only pixels the component search can ever seed
// from or extend through) into a per-frame list, so that the component search
// never has to scan the dense response.
//
// Algebra used by every kernel below (exact in integers):
//   with t1 = a+c, t2 = b+d per quadruple (a,b,c,d) = (s[i],s[i+4],s[i+8],s[i+12])
//     |a-c|     = 2*max(a,c) - t1
//     |t1 - t2| = 2*max(t1,t2) - (t1+t2)
//   so  sum_response - diff_response = 2*(Y - X),
//     Y = sum_i max(t1_i, t2_i),  X = sum_i max(a_i,c_i) + max(b_i,d_i)
//   and response = 2*(Y-X) - |M - local_mean|,  M = sum of the 16 samples.
#include "common.h"
#include "hotlist.h"
#include "chess_hot.h"
#include "kernels.h"

namespace mrg {

#ifdef MRG_EXPERIMENT  // the reference-shaped kernel: on-device cross-check of the tuned one (option "chess_v0"), experiment builds only
// ---------------------------------------------------------------------------
// v0 epilogue: a wave holds 64 consecutive pixels of one row (x0 a multiple of 64), one per lane.
// The first lane of every aligned group of 8 appends the group.  Uniform control flow required.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void append_hot_row64(bool hot, int x, int y, const CompTables& t, int frame) {
    const unsigned long long m = __ballot(hot);
    if (m == 0) return;
    const int lane = __lane_id();
    int base = 0;
    if (lane == 0) base = atomicAdd(t.hot_cnt + frame, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    const uint32_t bits = (uint32_t)(m >> (lane & 56)) & 0xffu;
    if ((lane & 7) == 0 && bits)
        write_group_direct(t, frame, base + __popcll(m & ((1ull << lane) - 1ull)), bits,
                           ((uint32_t)y << 16) | (uint32_t)x);
}

// ---------------------------------------------------------------------------
// v0: reference-shaped kernel.  One thread per pixel column of a 64x16 tile,
// bytes staged in LDS with a 5-pixel halo.  Kept as the on-device cross-check
// of the tuned kernel (tests compare the two) -- not the production path.
// ---------------------------------------------------------------------------
constexpr int V0_TW = 64, V0_TH = 16, V0_HALO = 5;
constexpr int V0_LW = V0_TW + 2 * V0_HALO;  // 74
constexpr int V0_LH = V0_TH + 2 * V0_HALO;  // 26
constexpr int V0_LS = 76;                   // LDS row stride

template <bool CLAMP, bool HOT>
__global__ __launch_bounds__(256) void chess_v0_kernel(LevelBatch lb, CompTables t, int frame0) {
    __shared__ uint8_t tile[V0_LH * V0_LS];
    const int frame = frame0 + blockIdx.z;
    const int w = lb.w, h = lb.h;
    const uint8_t* img = lb.img + (long long)frame * lb.img_pitch;
    int16_t* resp = lb.resp + (long long)frame * lb.resp_pitch;
    const int x0 = blockIdx.x * V0_TW, y0 = blockIdx.y * V0_TH;

    for (int i = threadIdx.x; i < V0_LH * V0_LW; i += 256) {
        const int ly = i / V0_LW, lx = i - ly * V0_LW;
        int gx = x0 + lx - V0_HALO, gy = y0 + ly - V0_HALO;
        gx = min(max(gx, 0), w - 1);
        gy = min(max(gy, 0), h - 1);
        tile[ly * V0_LS + lx] = img[(long long)gy * lb.img_stride + gx];
    }
    __syncthreads();

    const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ty = ty0 + 4 * k;
        const int x = x0 + tx, y = y0 + ty;
        const uint8_t* c = tile + (ty + V0_HALO) * V0_LS + tx + V0_HALO;
        auto S = [&](int dx, int dy) -> int { return c[dy * V0_LS + dx]; };
        // ring, ChESS.c:68-83
        const int s0 = S(2, -5), s1 = S(0, -5), s2 = S(-2, -5), s3 = S(-4, -4);
        const int s4 = S(-5, -2), s5 = S(-5, 0), s6 = S(-5, 2), s7 = S(-4, 4);
        const int s8 = S(-2, 5), s9 = S(0, 5), s10 = S(2, 5), s11 = S(4, 4);
        const int s12 = S(5, 2), s13 = S(5, 0), s14 = S(5, -2), s15 = S(4, -4);
        const int local_mean = (S(-1, 0) + S(0, 0) + S(1, 0)) * 16 / 3;  // ChESS.c:86
        int sum = 0, diff = 0, mean = 0;
#define MRG_QUAD(a, b, c_, d)                  \
    sum += abs((a) - (b) + (c_) - (d));        \
    diff += abs((a) - (c_)) + abs((b) - (d));  \
    mean += (a) + (b) + (c_) + (d);
        MRG_QUAD(s0, s4, s8, s12) MRG_QUAD(s1, s5, s9, s13) MRG_QUAD(s2, s6, s10, s14) MRG_QUAD(s3, s7, s11, s15)
#undef MRG_QUAD
        int r = sum - diff - abs(mean - local_mean);  // ChESS.c:104
        const bool inimg = x < w && y < h;
        const bool interior = x >= kMargin && x < w - kMargin && y >= kMargin && y < h - kMargin;
        if (!interior) r = 0;
        if (CLAMP) r = max(r, 0);
        if (inimg) resp[(long long)y * w + x] = (int16_t)r;
        if (HOT) append_hot_row64(interior && r > kRespMin, x, y, t, frame);
    }
}

void launch_chess_v0(const LevelBatch& lb, const CompTables& t, int frame0, int nframes, bool clamp, bool hot,
                     hipStream_t s) {
    dim3 grid((lb.w + V0_TW - 1) / V0_TW, (lb.h + V0_TH - 1) / V0_TH, nframes);
    if (hot)
        hipLaunchKernelGGL((chess_v0_kernel<true, true>), grid, dim3(256), 0, s, lb, t, frame0);
    else if (clamp)
        hipLaunchKernelGGL((chess_v0_kernel<true, false>), grid, dim3(256), 0, s, lb, t, frame0);
    else
        hipLaunchKernelGGL((chess_v0_kernel<false, false>), grid, dim3(256), 0, s, lb, t, frame0);
}

#endif  // MRG_EXPERIMENT

// ---------------------------------------------------------------------------
// v1: the production kernel.  HBM-bound design for gfx950 (wave64):
//
//  * a 256-thread workgroup owns a vertical strip SW = 256 pixels wide and SEG
//    rows tall and ROLLS down it 8 rows at a time, so every input row is
//    fetched from HBM once (plus a 32-px horizontal and 10-row per-segment halo)
//    and every output row is written once: ~3.3 B/px of traffic for 3 B/px of
//    algorithmic bytes;
//  * rows are staged through registers into an LDS ring of 32 rows as TWO
//    planes of packed u16 pixel pairs: P0 holds (I[2m], I[2m+1]), P1 holds
//    (I[2m+1], I[2m+2]).  Every ring sample at an even dx is then a whole-dword
//    offset in P0 and every odd dx (+-5, +-1) a whole-dword offset in P1: the
//    per-lane operands are picked by register renaming out of three aligned,
//    conflict-free ds_read_b128 per (plane,row), never by byte shuffles;
//  * each lane produces 8 adjacent pixels as 4 packed pairs with 16-bit packed
//    VALU ops (v_pk_max_u16, 32-bit adds of non-overflowing halves) using the
//    max-identities in the file header: 42 VALU ops per pixel PAIR;
//  * the group of 8 rows for the iteration after next is prefetched into
//    registers before the math and written to the ring after it, one barrier per
//    iteration; results leave as one 16-byte store per lane.
// ---------------------------------------------------------------------------
constexpr int V1_SW = 256;                    // strip width, output pixels
constexpr int V1_HL = 16;                     // left halo in the LDS window (16 keeps global loads 16-B aligned)
constexpr int V1_WIN = V1_SW + 2 * V1_HL;     // 288 window pixels per row
constexpr int V1_NCH = V1_WIN / 16;           // 18 staging chunks of 16 pixels per row
constexpr int V1_ROWB = V1_WIN * 2;           // 576 bytes per row per plane
constexpr int V1_NR = 32;                     // ring rows
constexpr int V1_RB = 8;                      // rows per iteration (4 waves x 2 rows)
constexpr int V1_PLANE = V1_NR * V1_ROWB;     // 18432 bytes

using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
using i16x2 = short __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a),
                                                                  __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a),
                                                                  __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (i16x2)(__builtin_bit_cast(i16x2, a) - __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (i16x2)(__builtin_bit_cast(i16x2, a) + __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ uint32_t dot2_u32_u16(uint32_t a, uint32_t b, uint32_t c) {  // a.lo * b.lo + a.hi * b.hi + c
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}
__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) {  // v_pk_sub_u16 ... clamp
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, a),
                                                                  __builtin_bit_cast(i16x2, b)));
}

struct StageRegs {
    uint4 g;        // 16 pixels
    uint32_t next;  // byte 0 = the pixel after them
};

// Global -> registers for one 16-pixel chunk of window row `r` (clamped: pixels
// outside the image read as 0; they only ever feed non-interior outputs).
__device__ __forceinline__ StageRegs stage_load(const uint8_t* img, int stride, int w, int h, int r, int gx) {
    StageRegs s;
    s.g = make_uint4(0, 0, 0, 0);
    s.next = 0;
    if (r < 0 || r >= h) return s;
    const uint8_t* row = img + (long long)r * stride;
    if (gx >= 0 && gx + 16 <= w) {
        __builtin_memcpy(&s.g, row + gx, 16);
        if (gx + 16 < w) s.next = row[gx + 16];
    } else if (gx + 16 > 0 && gx < w) {
        uint32_t v[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; ++i) {
            const int x = gx + i;
            if (x >= 0 && x < w) v[i >> 2] |= (uint32_t)row[x] << (8 * (i & 3));
        }
        s.g = make_uint4(v[0], v[1], v[2], v[3]);
        if (gx + 16 >= 0 && gx + 16 < w) s.next = row[gx + 16];
    }
    return s;
}

// Branch-free variant for frames whose width is a multiple of 16: every chunk is
// either wholly inside or wholly outside the frame, and what an outside chunk (or
// an outside row) holds never reaches an interior output, so addresses are simply
// clamped into the frame.  No divergent control flow means hipcc does not park an
// s_waitcnt vmcnt(0) behind the prefetch: the load stays in flight under the math.
__device__ __forceinline__ StageRegs stage_load_fast(const uint8_t* img, int stride, int h, int r, int gx_c, int nx_c) {
    StageRegs s;
    const int rc = min(max(r, 0), h - 1);
    const uint8_t* row = img + (long long)rc * stride;
    __builtin_memcpy(&s.g, row + gx_c, 16);
    s.next = row[nx_c];
    return s;
}

// Registers -> both LDS planes (packed u16 pairs).
template <int ROWB = V1_ROWB, int PLANE = V1_PLANE>
__device__ __forceinline__ void stage_store(char* lds, int slot, int ch, const StageRegs& s) {
    const uint32_t g0 = s.g.x, g1 = s.g.y, g2 = s.g.z, g3 = s.g.w;
    constexpr uint32_t S01 = 0x0c010c00u, S23 = 0x0c030c02u;  // (b0,b1) / (b2,b3) of the low operand
    constexpr uint32_t S12 = 0x0c020c01u, S34 = 0x0c040c03u;  // (b1,b2) / (b3, b0 of the high operand)
    uint4 a, b, c, d;
    a.x = __builtin_amdgcn_perm(g0, g0, S01); a.y = __builtin_amdgcn_perm(g0, g0, S23);
    a.z = __builtin_amdgcn_perm(g1, g1, S01); a.w = __builtin_amdgcn_perm(g1, g1, S23);
    b.x = __builtin_amdgcn_perm(g2, g2, S01); b.y = __builtin_amdgcn_perm(g2, g2, S23);
    b.z = __builtin_amdgcn_perm(g3, g3, S01); b.w = __builtin_amdgcn_perm(g3, g3, S23);
    c.x = __builtin_amdgcn_perm(g0, g0, S12); c.y = __builtin_amdgcn_perm(g1, g0, S34);
    c.z = __builtin_amdgcn_perm(g1, g1, S12); c.w = __builtin_amdgcn_perm(g2, g1, S34);
    d.x = __builtin_amdgcn_perm(g2, g2, S12); d.y = __builtin_amdgcn_perm(g3, g2, S34);
    d.z = __builtin_amdgcn_perm(g3, g3, S12); d.w = __builtin_amdgcn_perm(s.next, g3, S34);
    char* p = lds + slot * ROWB + ch * 32;
    *reinterpret_cast<uint4*>(p) = a;
    *reinterpret_cast<uint4*>(p + 16) = b;
    *reinterpret_cast<uint4*>(p + PLANE) = c;
    *reinterpret_cast<uint4*>(p + PLANE + 16) = d;
}

// ---------------------------------------------------------------------------
// Typed staging: `buffer_load_format_d16_xyzw` through an 8_8_8_8 UINT buffer descriptor turns 4 bytes
// into two packed-u16 pair registers in the texture unit, i.e. the P0 plane entries with no VALU work
// at all (the v_perm path above spends 16 v_perm_b32 per 16 pixels on it).  Probed on the chip
// (tools/ubench/typed_load.hip): element addresses need no alignment, and an element that is out of
// range of the descriptor (negative offsets included) reads as 0 instead of faulting.  A staging task
// is one aligned group of 8 window pixels ("oct") of one row: 36 octs x 8 rows = 288 tasks per
// iteration for 256 threads, every wave takes part.
//   STAGE_TYPED2: P1 (odd pairs) comes from two more typed loads at byte offset +1 / +5;
//   STAGE_TYPED1: P1 is built from the P0 registers with four v_alignbit_b32 plus one byte load.
// Works for any width: what a row holds beyond the frame (the next row's pixels) only ever reaches
// outputs the interior mask zeroes.
// ---------------------------------------------------------------------------
enum : int { STAGE_GENERIC = 0, STAGE_PERM16 = 1, STAGE_TYPED2 = 2, STAGE_TYPED1 = 3 };
using i32x4 = int __attribute__((ext_vector_type(4)));
using u32x4v = uint32_t __attribute__((ext_vector_type(4)));
using u16x4 = unsigned short __attribute__((ext_vector_type(4)));
__device__ u16x4 buffer_load_u8x4_as_u16x4(i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.format.v4i16");
__device__ unsigned char buffer_load_u8(i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.i8");

// Untyped buffer accesses: address = resource base (scalar registers) + VGPR offset (32 bits, bounds-checked
// against the resource's size: an access beyond it reads 0 / is dropped) + SGPR offset (not bounds-checked).
// Used where the lane's part of an address is a constant and the rest is wave-uniform: no 64-bit vector
// arithmetic per access (hipcc otherwise spends a v_mad_i64_i32 + v_lshl_add_u64 per lane and iteration).
using u32x2 = uint32_t __attribute__((ext_vector_type(2)));
__device__ u32x4v raw_buffer_load_b128(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ void raw_buffer_store_b128(u32x4v data, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.v4i32");
__device__ void raw_buffer_store_b64(u32x2 data, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.v2i32");
__device__ void raw_buffer_store_b32(uint32_t data, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.i32");
__device__ void raw_buffer_store_b8(unsigned char data, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.i8");
constexpr int kAuxNT = 2;  // non-temporal (gfx94x/gfx950 cache-policy bit "nt")
__device__ __forceinline__ i32x4 raw_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};  // data_format 32, stride 0
}

__device__ __forceinline__ i32x4 typed_rsrc(const uint8_t* base, uint32_t bytes) {
    // word 3: dst_sel x,y,z,w = R,G,B,A (4,5,6,7); num_format UINT (4) << 12; data_format 8_8_8_8 (10) << 15
    constexpr uint32_t w3 = (4u | (5u << 3) | (6u << 6) | (7u << 9)) | (4u << 12) | (10u << 15);
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, (int)w3};
}

struct OctRegs {
    uint2 p0a, p0b;  // pairs (I0,I1) (I2,I3) | (I4,I5) (I6,I7)
    uint2 p1a, p1b;  // TYPED2: pairs (I1,I2) (I3,I4) | (I5,I6) (I7,I8)
    uint32_t next;   // TYPED1: I8
};

template <int STAGE>
__device__ __forceinline__ OctRegs oct_load(const i32x4& rsrc, int voff) {
    OctRegs o;
    o.p0a = __builtin_bit_cast(uint2, buffer_load_u8x4_as_u16x4(rsrc, voff, 0, 0));
    o.p0b = __builtin_bit_cast(uint2, buffer_load_u8x4_as_u16x4(rsrc, voff + 4, 0, 0));
    if (STAGE == STAGE_TYPED2) {
        o.p1a = __builtin_bit_cast(uint2, buffer_load_u8x4_as_u16x4(rsrc, voff + 1, 0, 0));
        o.p1b = __builtin_bit_cast(uint2, buffer_load_u8x4_as_u16x4(rsrc, voff + 5, 0, 0));
        o.next = 0;
    } else {
        o.next = buffer_load_u8(rsrc, voff + 8, 0, 0);
        o.p1a = o.p1b = make_uint2(0, 0);
    }
    return o;
}

template <int STAGE>
__device__ __forceinline__ void oct_store(char* lds, int slot, int oct, const OctRegs& o) {
    char* p = lds + slot * V1_ROWB + oct * 16;
    *reinterpret_cast<uint4*>(p) = make_uint4(o.p0a.x, o.p0a.y, o.p0b.x, o.p0b.y);
    if (STAGE == STAGE_TYPED2) {
        *reinterpret_cast<uint4*>(p + V1_PLANE) = make_uint4(o.p1a.x, o.p1a.y, o.p1b.x, o.p1b.y);
    } else {
        // (hi half of one pair, lo half of the next) = v_alignbit_b32(next, this, 16)
        *reinterpret_cast<uint4*>(p + V1_PLANE) =
            make_uint4(__builtin_amdgcn_alignbit(o.p0a.y, o.p0a.x, 16), __builtin_amdgcn_alignbit(o.p0b.x, o.p0a.y, 16),
                       __builtin_amdgcn_alignbit(o.p0b.y, o.p0b.x, 16), __builtin_amdgcn_alignbit(o.next, o.p0b.y, 16));
    }
}

// 12 dwords D[-4..7] around the lane's 4 dwords of one (plane,row).
// The empty asm statements make each 16-byte value opaque, so hipcc keeps the
// aligned, bank-conflict-free ds_read_b128 instead of narrowing it to the few
// dwords that are used (at a 16-byte lane stride those narrow reads are 2- to
// 4-way bank conflicted).
using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 lds_read_b128(const char* p) {
    u32x4 v = *reinterpret_cast<const u32x4*>(p);
    asm("" : "+v"(v));
    return v;
}
__device__ __forceinline__ void load12(uint32_t (&R)[12], const char* p) {
    const u32x4 a = lds_read_b128(p);
    const u32x4 b = lds_read_b128(p + 16);
    const u32x4 c = lds_read_b128(p + 32);
    R[0] = a.x; R[1] = a.y; R[2] = a.z; R[3] = a.w;
    R[4] = b.x; R[5] = b.y; R[6] = b.z; R[7] = b.w;
    R[8] = c.x; R[9] = c.y; R[10] = c.z; R[11] = c.w;
}

// ---------------------------------------------------------------------------
// Pyramid levels 1..3 out of the ring (level-0 launches of the chain; frames whose width is a multiple
// of 16 and height a multiple of 8, i.e. whole cells only, where every level is (a+b+c+d+2)>>2 of four
// FULL-RESOLUTION pixels -- see decimate.hip).  The 8 frame rows y..y+7 an iteration produces responses
// for sit complete in the ring, already split into the two pair planes, and a 2x2 cell is exactly one
// pair of two consecutive rows:
//   level 1, pixel X of row Y: rows 2Y, 2Y+1, columns 2X, 2X+1     -> plane P0, pair  X      (4 rows x 128)
//   level 2                  : rows 4Y+1, 4Y+2, columns 4X+1, 4X+2 -> plane P1, pair 2X      (2 rows x 64)
//   level 3                  : rows 8Y+3, 8Y+4, columns 8X+3, 8X+4 -> plane P1, pair 4X+1    (1 row  x 32)
// (pairs counted from the strip's first pixel).  That saves the separate pyramid kernel its read of the
// whole batch.  Wave 3 does all of it: level 1 with eight pixels per lane, then levels 2 (lanes 0..31, four
// pixels each) and 3 (lanes 32..63, one pixel each) in one pass: 8 ds_read_b128 and ~42 VALU per iteration.
// It has to be that wave: it is the one that stages nothing, so (a) it has some issue slots to spare and
// (b) it never waits on vmcnt inside the loop.  On a staging wave ANY store is poison: vmcnt retires in
// order, the wave waits for its prefetched rows twice per iteration, and hipcc's wait-count insertion turns
// a store in divergent control flow into vmcnt(0) at those waits -- the workgroup then sits at the barrier
// for the latency of a store.  Measured per launch (64 frames of 4096x3072, pipeline, plain kernel 645 us):
//   all of it on wave 3 via LDS (this code)                       743 us
//   levels 2, 3 on wave 2 instead                                 +55 us
//   cells built in registers on every wave with v_permlane32_swap (no LDS reads, 10-19 VALU per wave),
//   stores from the waves that hold them                          778 us; stores moved behind the ring
//   store 875 us; the cell arithmetic alone (nothing stored) already +72 us: the kernel is VALU-bound,
//   so 8 pixels per lane on one wave beats 2 pixels per lane on four
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cells4(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    // v = (left column sum, right column sum) of a cell as u16 halves; four cells -> four bytes
    const uint32_t lo01 = __builtin_amdgcn_perm(v1, v0, 0x05040100u), hi01 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const uint32_t lo23 = __builtin_amdgcn_perm(v3, v2, 0x05040100u), hi23 = __builtin_amdgcn_perm(v3, v2, 0x07060302u);
    const uint32_t s01 = (lo01 + hi01 + 0x00020002u) >> 2, s23 = (lo23 + hi23 + 0x00020002u) >> 2;
    return __builtin_amdgcn_perm(s23, s01, 0x06040200u);
}
__device__ __forceinline__ void emit_pyramid_rows(const char* lds, int y, int strip_x, int w, int frame, int wvu, int lane,
                                                  const PyramidOut& po) {
    if (wvu != 3) return;
    // wave-uniform parts (scalar registers): ring slot of row y, first output pixel of this strip and row group
    const uint32_t slot0 = (uint32_t)((y + 64) & (V1_NR - 1));  // y is a multiple of 8: rows y..y+7 do not wrap
    const int j = lane & 15, R = (lane >> 4) & 3;
    // all eight reads first, then the arithmetic: one LDS latency per iteration instead of two
    const bool l3 = lane >= 32;
    const char* r1 = lds + slot0 * V1_ROWB + 2 * V1_HL + (uint32_t)(R * (2 * V1_ROWB) + 32 * j);
    const uint32_t loff = l3 ? 3 * V1_ROWB + 16 * (lane - 32) : (1 + 4 * (R & 1)) * V1_ROWB + 32 * j;
    const char* r2 = lds + slot0 * V1_ROWB + V1_PLANE + 2 * V1_HL + loff;
    const u32x4 a0 = lds_read_b128(r1), a1 = lds_read_b128(r1 + 16);
    const u32x4 b0 = lds_read_b128(r1 + V1_ROWB), b1 = lds_read_b128(r1 + V1_ROWB + 16);
    const u32x4 c0 = lds_read_b128(r2), c1 = lds_read_b128(r2 + 16);
    const u32x4 d0 = lds_read_b128(r2 + V1_ROWB), d1 = lds_read_b128(r2 + V1_ROWB + 16);
    // stores: the level image of the frame as a buffer resource, the lane's offset within the iteration's rows a
    // constant, the rows' offset uniform (SGPR)
    if (po.out[0]) {
        const i32x4 rs = raw_rsrc(po.out[0] + (long long)frame * po.h[0] * po.w[0], (uint32_t)(po.h[0] * po.w[0]));
        u32x2 o;
        o.x = cells4(a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w);
        o.y = cells4(a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w);
        if (strip_x + 16 * j < w)
            raw_buffer_store_b64(o, rs, R * po.w[0] + 8 * j + (strip_x >> 1), (y >> 1) * po.w[0], 0);
    }
    if (po.out[1]) {
        const uint32_t v0 = l3 ? c0.y + d0.y : c0.x + d0.x;
        const uint32_t o = cells4(v0, c0.z + d0.z, c1.x + d1.x, c1.z + d1.z);
        if (!l3) {
            if (strip_x + 16 * j < w) {
                const i32x4 rs = raw_rsrc(po.out[1] + (long long)frame * po.h[1] * po.w[1], (uint32_t)(po.h[1] * po.w[1]));
                raw_buffer_store_b32(o, rs, (R & 1) * po.w[1] + 4 * j + (strip_x >> 2), (y >> 2) * po.w[1], 0);
            }
        } else if (po.out[2] && strip_x + 8 * (lane - 32) < w) {
            const i32x4 rs = raw_rsrc(po.out[2] + (long long)frame * po.h[2] * po.w[2], (uint32_t)(po.h[2] * po.w[2]));
            raw_buffer_store_b8((unsigned char)o, rs, lane - 32 + (strip_x >> 3), (y >> 3) * po.w[2], 0);
        }
    }
}

// The body of the kernel for workgroup `bid` of `nwg` of one level (the multi-level launch below runs
// several levels in one grid).
// The responses of pixel pair k (0..3) of a lane's aligned 8-pixel group out of its window registers: rows -5, +5,
// -4, +4 of plane P0 (pairs that start at an even pixel), rows -2, +2 and the centre row of plane P1 (pairs that
// start at an odd pixel), 12 registers each = pixels x0 - 8 .. x0 + 15, and the centre row's own pairs z0.
// Returns response + 8192 in each half.
__device__ __forceinline__ uint32_t response_pair_biased(const uint32_t (&m5)[12], const uint32_t (&p5)[12],
                                                         const uint32_t (&m4)[12], const uint32_t (&p4)[12],
                                                         const uint32_t (&m2)[12], const uint32_t (&p2)[12],
                                                         const uint32_t (&z1)[12], const uint32_t (&z0)[4], int k) {
    const int c = 4 + k;
    // quadruples (a,b,c,d) = (s[i], s[i+4], s[i+8], s[i+12]); ring offsets from ChESS.c:68-83
    const uint32_t a0 = m5[c + 1], c0 = p5[c - 1], b0 = m2[c - 3], d0 = p2[c + 2];
    const uint32_t a1 = m5[c], c1 = p5[c], b1 = z1[c - 3], d1 = z1[c + 2];
    const uint32_t a2 = m5[c - 1], c2 = p5[c + 1], b2 = p2[c - 3], d2 = m2[c + 2];
    const uint32_t a3 = m4[c - 2], c3 = p4[c + 2], b3 = p4[c - 2], d3 = m4[c + 2];
    // Both pixels of a pair sit in one register and no half ever overflows or borrows, so
    // every add / subtract below is a plain 32-bit op (2-cycle issue class on gfx950; the
    // packed 16-bit forms are 4).  Only the twelve maxima need v_pk_max_u16.
    const uint32_t t10 = a0 + c0, t20 = b0 + d0, t11 = a1 + c1, t21 = b1 + d1;
    const uint32_t t12 = a2 + c2, t22 = b2 + d2, t13 = a3 + c3, t23 = b3 + d3;
    const uint32_t M = ((t10 + t20) + (t11 + t21)) + ((t12 + t22) + (t13 + t23));
    // Y carries a +4096 bias per half so that Y - X (>= -2040) stays positive
    const uint32_t Yb = ((pk_max_u16(t10, t20) + pk_max_u16(t11, t21)) +
                         (pk_max_u16(t12, t22) + pk_max_u16(t13, t23))) + 0x10001000u;
    const uint32_t X = ((pk_max_u16(a0, c0) + pk_max_u16(b0, d0)) + (pk_max_u16(a1, c1) + pk_max_u16(b1, d1))) +
                       ((pk_max_u16(a2, c2) + pk_max_u16(b2, d2)) + (pk_max_u16(a3, c3) + pk_max_u16(b3, d3)));
    // local_mean = (I[x-1]+I[x]+I[x+1])*16/3, truncating (ChESS.c:86): floor(16n/3) = (n*349536)>>16, n <= 765
    const uint32_t n = z1[c - 1] + z0[k] + z1[c];
    const uint32_t lm_lo = __umul24(n & 0xffffu, 349536u);
    const uint32_t lm_hi = __umul24(n >> 16, 349536u);
    const uint32_t LM = __builtin_amdgcn_perm(lm_hi, lm_lo, 0x07060302u);
    const uint32_t dev = pk_max_u16(M, LM) - pk_min_u16(M, LM);  // |M - LM| per half
    const uint32_t d1x = Yb - X;
    return (d1x + d1x) - dev;  // halves = response + 8192, in [2072, 10232]  (ChESS.c:104)
}

template <bool CLAMP, bool HOT, int STAGE, bool PYR = false, bool FILTER = false>
__device__ __forceinline__ void chess_v1_body(const LevelBatch& lb, const CompTables& t, int frame0, int nsegs, unsigned bid,
                                              unsigned nwg_level, char* lds, const PyramidOut* po = nullptr) {
    // XCD-aware work order: workgroup b is dispatched to XCD b % 8 (observed, used
    // for speed only).  Give each XCD a contiguous run of work items, strips
    // fastest, so the 32-pixel column halo and the 10-row segment halo a
    // workgroup shares with its neighbours is found in that XCD's L2 instead of
    // being fetched from HBM once per neighbour.
    const int nstrips = (lb.w + V1_SW - 1) / V1_SW;
    int work;
    {
        const unsigned b = bid, nwg = nwg_level, xcd = b & 7u, j = b >> 3;
        const unsigned q = nwg >> 3, r = nwg & 7u;
        work = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j);
    }
    const bool probe = lb.clk != nullptr && bid == 0;  // (mrgingham_amd_sclk_mhz)
    const ClockProbe clkp = clock_probe_begin(probe);
    const int strip = work % nstrips, rest = work / nstrips;
    // FILTER, the dense repeat of a sparse chain (CompTables::only): the grid is laid out for kOnlySlots frames and
    // the workgroup takes frames slot, slot + kOnlySlots, ... of the list (usually none: it leaves at once)
    const int nlisted = FILTER ? t.only[0] : 0;
  for (int li = rest / nsegs; FILTER ? li < nlisted : li == rest / nsegs; li += kOnlySlots) {
    const int frame = FILTER ? t.only[1 + li] : frame0 + li;
    if (FILTER && li != rest / nsegs) __syncthreads();  // the ring and the record buffer of the frame before are free
    const int w = lb.w, h = lb.h, stride = lb.img_stride;
    const uint8_t* img = lb.img + (long long)frame * lb.img_pitch;
    int16_t* resp = lb.resp + (long long)frame * lb.resp_pitch;
    const int strip_x = strip * V1_SW;
    int ys, ye;  // the workgroup's rows: segment rest % nsegs of the frame's nsegs balanced segments (common.h)
    segment_rows(h, nsegs, rest % nsegs, V1_RB, ys, ye);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, half = lane >> 5, lx = lane & 31;
    uint32_t* hotbuf = reinterpret_cast<uint32_t*>(lds + 2 * V1_PLANE);
    int* hotcnt = reinterpret_cast<int*>(hotbuf + V1_HOTBUF);
    HotSink hsink{hotbuf + (tid >> 6) * V1_HOTSEG, ys, strip_x, 0};
    constexpr bool W16 = STAGE == STAGE_PERM16;
    constexpr bool TYPED = STAGE == STAGE_TYPED2 || STAGE == STAGE_TYPED1;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    // staging role (v_perm paths): 8 rows x 18 chunks = 144 threads
    const bool stager = tid < V1_RB * V1_NCH;
    const int st_row = tid / V1_NCH, st_ch = tid - st_row * V1_NCH;
    const int st_gx = strip_x - V1_HL + 16 * st_ch;
    const int st_gx_c = min(max(st_gx, 0), max(w - 16, 0));  // W16: clamped chunk start / following pixel
    const int st_nx_c = min(max(st_gx + 16, 0), w - 1);
    // W16, inside the loop: the frame as a buffer resource; a lane's chunk of row group g is at byte offset
    // (st_row * stride + st_gx_c) + g's first row * stride -- one 32-bit add per load, and a row below the frame is
    // out of range and reads 0 (it only feeds outputs the interior mask zeroes), so no clamp either
    const i32x4 img_rsrc = raw_rsrc(img, (uint32_t)min((long long)(h - 1) * stride + w, 0xffffffffLL));
    const int st_voff_g = st_row * stride + st_gx_c, st_voff_n = st_row * stride + st_nx_c;
    // staging role (typed paths): 8 rows x 36 octs = 288 tasks; thread tid takes task tid, and wave 0
    // also takes tasks 256..287 (both of its half-waves do the same 32, so the branch is wave-uniform)
    constexpr int NOCT = V1_WIN / 8;  // 36
    const int oc_row0 = tid / NOCT, oc_c0 = tid - oc_row0 * NOCT;
    const int oc_t1 = 256 + (lane & 31);
    const int oc_row1 = oc_t1 / NOCT, oc_c1 = oc_t1 - oc_row1 * NOCT;
    const int oc_v0 = oc_row0 * stride + strip_x - V1_HL + 8 * oc_c0;  // + (group's first row) * stride
    const int oc_v1 = oc_row1 * stride + strip_x - V1_HL + 8 * oc_c1;
    i32x4 rsrc = {0, 0, 0, 0};
    if (TYPED) {
        // this frame and whatever of the batch follows it, up to 2 GB: rows above frame 0 and below
        // the last frame are out of range and read as 0; between frames the neighbour's pixels are
        // read, which is as good (they only reach masked outputs)
        const long long rest = (long long)(lb.nframes - 1 - frame) * lb.img_pitch + (long long)(h - 1) * stride + w;
        rsrc = typed_rsrc(img, (uint32_t)(rest < 0x7fffffffLL ? rest : 0x7fffffffLL));
    }

    // prologue: row groups G0..G2 = rows ys-5 .. ys+18
    if (TYPED) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int r0 = ys - 5 + V1_RB * g;
            const OctRegs a = oct_load<STAGE>(rsrc, oc_v0 + r0 * stride);
            oct_store<STAGE>(lds, (r0 + oc_row0 + 64) & (V1_NR - 1), oc_c0, a);
            if (wvu == 0) {
                const OctRegs b = oct_load<STAGE>(rsrc, oc_v1 + r0 * stride);
                oct_store<STAGE>(lds, (r0 + oc_row1 + 64) & (V1_NR - 1), oc_c1, b);
            }
        }
    } else if (stager) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int r = ys - 5 + V1_RB * g + st_row;
            const StageRegs s = W16 ? stage_load_fast(img, stride, h, r, st_gx_c, st_nx_c)
                                    : stage_load(img, stride, w, h, r, st_gx);
            stage_store(lds, (r + 64) & (V1_NR - 1), st_ch, s);
        }
    }
    __syncthreads();

    // per-lane constants
    const int x0 = strip_x + 8 * lx;
    // Results are carried with a +8192 bias per half (see below).  CLAMP: ONE saturating subtraction strips the
    // bias, clamps at 0 and zeroes the 7-pixel frame columns -- its per-lane constant is 8192 for a pixel inside
    // the frame columns and 0xffff (more than any biased response: saturates to 0) for one outside.
    uint32_t xmask[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xa = x0 + 2 * k, xb = xa + 1;
        const bool ina = xa >= kMargin && xa < w - kMargin, inb = xb >= kMargin && xb < w - kMargin;
        if (CLAMP) xmask[k] = (ina ? 0x2000u : 0xffffu) | (inb ? 0x20000000u : 0xffff0000u);
        else xmask[k] = (ina ? 0xffffu : 0u) | (inb ? 0xffff0000u : 0u);
    }
    // Row addressing: wave index and loop row are wave-uniform (SALU); the lane-dependent part is a
    // constant.  A wave covers rows yy = y + 2*wv + half; for even dy the slot of half 1 is the
    // slot of half 0 plus one (never wraps, the base slot is even), for odd dy (+-5) the two
    // half-waves get separately wrapped uniform slots, selected with a mask.
    const uint32_t lane_off = 16u + 16u * lx;                      // D[-4] of the lane within a row
    const uint32_t lane_off_h = lane_off + (half ? V1_ROWB : 0u);  // + the half-wave's row
    const uint32_t halfmask = half ? 0xffffffffu : 0u;
    const bool seg_interior = ys >= kMargin && ye <= h - kMargin;  // workgroup-uniform
    // response rows as a buffer resource: the lane's byte offset within the wave's two rows is a constant, the
    // rows' offset is wave-uniform (SGPR offset of the store)
    const i32x4 resp_rsrc = raw_rsrc(resp, (uint32_t)min((long long)w * h * 2, 0xffffffffLL));
    const int st_resp_voff = (half * w + x0) * 2;

    int grp = 3;
    for (int y = ys; y < ye; y += V1_RB, ++grp) {
        // prefetch the group needed two iterations from now (rows y+19 .. y+26)
        StageRegs pre;
        OctRegs opre0, opre1;
        const int pr0 = ys - 5 + V1_RB * grp;  // first row of the group (uniform)
        const int pr = pr0 + st_row;
        if (TYPED) {
            opre0 = oct_load<STAGE>(rsrc, oc_v0 + pr0 * stride);
            if (wvu == 0) opre1 = oct_load<STAGE>(rsrc, oc_v1 + pr0 * stride);
        } else if (stager) {
            if (W16) {
                const int rowoff = pr0 * stride;  // uniform; pr0 >= 0 inside the loop
                pre.g = __builtin_bit_cast(uint4, raw_buffer_load_b128(img_rsrc, st_voff_g + rowoff, 0, 0));
                pre.next = buffer_load_u8(img_rsrc, st_voff_n + rowoff, 0, 0);
            } else {
                pre = stage_load(img, stride, w, h, pr, st_gx);
            }
        }

        const int s0 = y + 2 * wvu + 64;  // uniform, even
        const int yy = s0 - 64 + half;
        auto row_even = [&](int dy) -> const char* {
            return lds + (lane_off_h + (uint32_t)(((s0 + dy) & (V1_NR - 1)) * V1_ROWB));
        };
        auto row_odd = [&](int dy) -> const char* {
            const uint32_t o0 = (uint32_t)(((s0 + dy) & (V1_NR - 1)) * V1_ROWB);
            const uint32_t o1 = (uint32_t)(((s0 + dy + 1) & (V1_NR - 1)) * V1_ROWB);
            return lds + (lane_off + o0 + (halfmask & (o1 - o0)));
        };
        uint32_t m5[12], p5[12], m4[12], p4[12], m2[12], p2[12], z1[12];
        load12(m5, row_odd(-5));
        load12(p5, row_odd(+5));
        load12(m4, row_even(-4));
        load12(p4, row_even(+4));
        load12(m2, row_even(-2) + V1_PLANE);
        load12(p2, row_even(+2) + V1_PLANE);
        const char* r0 = row_even(0);
        load12(z1, r0 + V1_PLANE);
        const u32x4 z0v = lds_read_b128(r0 + 16);
        const uint32_t z0[4] = {z0v.x, z0v.y, z0v.z, z0v.w};

        uint32_t out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t P = response_pair_biased(m5, p5, m4, p4, m2, p2, z1, z0, k);
            if (CLAMP) out[k] = pk_sub_sat_u16(P, xmask[k]);  // max(r, 0), 0 in the frame columns
            else out[k] = pk_sub_i16(P, 0x20002000u) & xmask[k];
        }
        if (!seg_interior) {
            const uint32_t rm = (yy >= kMargin && yy < h - kMargin) ? 0xffffffffu : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] &= rm;
        }

        if (HOT) {
            // responses are clamped here, so "> 15" is "any bit above bit 3"
            const bool live = yy < ye;
            const uint32_t any = (out[0] | out[1] | out[2] | out[3]) & 0xfff0fff0u;
            if (__ballot(any != 0 && live) != 0ull) {
                // bit i = pixel i of the lane is hot: per register, both halves to 0 / 1 (v_pk_min_u16), then the
                // two flags go to their bit positions and into the mask in one v_dot2_u32_u16 (lo * 2^2k + hi * 2^(2k+1))
                uint32_t bits = 0;
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    uint32_t m;  // (as an instruction: hipcc turns min(x, 1) on the halves into two compares, two selects and a v_perm)
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(out[k] & 0xfff0fff0u), "v"(0x00010001u));
                    bits = dot2_u32_u16(m, (1u << (2 * k)) | (2u << (2 * k + 16)), bits);
                }
                if (!live) bits = 0;
                collect_hot(bits, yy, x0, hsink, t, frame);
            }
        }

        // ring first, results second: the wait for the prefetched rows must not also
        // wait for this iteration's output stores (loads and stores share vmcnt)
        // The waves that stage are the last to reach the barrier (16 v_perm + four 13-cycle LDS stores
        // more than the others), and the rest of the workgroup waits for them: issue priority for
        // exactly that stretch (-1.2 ... -2.6 % on the launch, A/B on three boxes).
        if (TYPED) {
            oct_store<STAGE>(lds, (pr0 + oc_row0 + 64) & (V1_NR - 1), oc_c0, opre0);
            if (wvu == 0) oct_store<STAGE>(lds, (pr0 + oc_row1 + 64) & (V1_NR - 1), oc_c1, opre1);
        } else if (stager) {
            __builtin_amdgcn_s_setprio(2);
            stage_store(lds, (pr + 64) & (V1_NR - 1), st_ch, pre);
            __builtin_amdgcn_s_setprio(0);
        }
        if (yy < ye && x0 < w) {
            if (x0 + 8 <= w) {
                // streamed out: the dense response is only ever re-read around the few hot pixels
                const u32x4 v = {out[0], out[1], out[2], out[3]};
                raw_buffer_store_b128(v, resp_rsrc, st_resp_voff, (int)((uint32_t)(s0 - 64) * (uint32_t)w * 2u), kAuxNT);
            } else {
                int16_t* dst = resp + (long long)yy * w + x0;
                for (int i = 0; i < w - x0; ++i) dst[i] = (int16_t)(out[i >> 1] >> (16 * (i & 1)));
            }
        }
        if (PYR) {
            // after the math, when the window registers are dead: the max-ILP scheduler would otherwise
            // hoist these LDS reads into the math block and push the kernel over 128 VGPRs (3 waves/SIMD)
            __builtin_amdgcn_sched_barrier(0);
            emit_pyramid_rows(lds, y, strip_x, w, frame, wvu, lane, *po);
            __builtin_amdgcn_sched_barrier(0);
        }
        // LDS-only workgroup barrier: the output stores stay in flight across it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    if (HOT) flush_hot(hsink, hotcnt, wvu, t, frame);
  }
    clock_probe_end(probe, clkp, lb.clk);
}

template <bool CLAMP, bool HOT, int STAGE, bool FILTER = false>
__global__ __launch_bounds__(256) void chess_v1_kernel(LevelBatch lb, CompTables t, int frame0, int nsegs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    chess_v1_body<CLAMP, HOT, STAGE, false, FILTER>(lb, t, frame0, nsegs, blockIdx.x, gridDim.x, lds);
}

// Level 0 of the chain: response + clamp + hot list + the level images 1..3 (see emit_pyramid_rows).
__global__ __launch_bounds__(256, 4) void chess_v1_pyr_kernel(LevelBatch lb, CompTables t, int nsegs, PyramidOut po) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    chess_v1_body<true, true, STAGE_PERM16, true>(lb, t, 0, nsegs, blockIdx.x, gridDim.x, lds, &po);
}

// ---------------------------------------------------------------------------
// Sparse refinement, step 2: the clamped response -- and the hot list -- in a LIST OF CELLS of a level instead of
// the whole level (api.hip, option "sparse_refine"): the refinement of ~100 known points only ever looks at the
// response around them, and computing it for the other 99 % of the frame is what the dense schedule spends its
// time on.  Cells are squares of 2^cs pixels (cs >= 4) on the grid that starts at pixel (0, 0), listed per frame
// by sparse_cells_kernel (cc.hip).  The unit of work is a 16 x 16 MICRO-TILE -- a cell is 4^(cs-4) of them --
// computed by a half-wave with the window layout and the arithmetic of the production kernel above (two planes of
// packed pixel pairs; a lane owns an aligned group of 8 pixels of one row: 2 lanes per row, 16 rows).  A
// workgroup is ONE wave (two micro-tiles per pass, nothing to synchronise with) and takes micro-tile pairs
// blockIdx.x, + gridDim.x, ... of its frame.  Everything outside the listed cells of the response buffer is
// STALE: the refinement kernel knows (its SPARSE instantiation, WinSel::marked<true>).
// ---------------------------------------------------------------------------
constexpr int VC_T = 16;                     // micro-tile edge
constexpr int VC_ROWS = VC_T + 10;           // window rows: 5 above, 5 below
constexpr int VC_ROWB = 96;                  // bytes per window row and plane: 32 pixels (x - 8 .. x + 23) as pairs (64) + 32, so
                                             // that the rows of 8 consecutive lanes (4 rows) sit 8 banks apart
constexpr int VC_PLANE = VC_ROWS * VC_ROWB;  // 2496
constexpr int VC_TILEB = 2 * VC_PLANE;       // 4992 per micro-tile
__global__ __launch_bounds__(64) void chess_cells_kernel(LevelBatch lb, CompTables t, const uint32_t* cell_list,
                                                         const int32_t* cell_cnt, int list_pitch, int frame0) {
    __shared__ __attribute__((aligned(16))) char lds_all[2 * VC_TILEB];
    const int frame = frame0 + blockIdx.y;
    const int w = lb.w, h = lb.h, stride = lb.img_stride;
    const uint8_t* img = lb.img + (long long)frame * lb.img_pitch;
    int16_t* resp = lb.resp + (long long)frame * lb.resp_pitch;
    const uint32_t* list = cell_list + (long long)frame * list_pitch;
    const int lane = threadIdx.x, half = lane >> 5, hl = lane & 31, lx = hl & 1, trow = hl >> 1;
    const int ncell = min(cell_cnt[kCellHdr * frame], list_pitch), cs = cell_cnt[kCellHdr * frame + 1];
    if (ncell <= 0 || cs < 4) return;
    const int sub = cs - 4, nitems = ncell << (2 * sub);  // micro-tiles of the frame (<= sparse_mask_items(t): sparse_cells_kernel)
    uint8_t* masks = reinterpret_cast<uint8_t*>(t.gidx + (long long)frame * t.gidx_pitch);
    char* lds = lds_all + half * VC_TILEB;
    const char* lane_base = lds + 16 * lx + (trow + 5) * VC_ROWB;  // D[-4] of the lane's centre row
    for (int it0 = 2 * blockIdx.x; it0 < nitems; it0 += 2 * gridDim.x) {
        const int it = it0 + half;
        const bool have = it < nitems;
        const uint32_t c = list[min(it, nitems - 1) >> (2 * sub)];
        const int si = it & ((1 << (2 * sub)) - 1);  // micro-tile within the cell, row-major
        const int xt = ((int)(c & 0xfffu) << cs) + VC_T * (si & ((1 << sub) - 1));  // (bits 12..15: the cell's subset, cc.hip)
        const int yt = ((int)(c >> 16) << cs) + VC_T * (si >> sub);
        __syncthreads();  // (one wave: orders the LDS reads of the pass before against these writes)
        for (int task = hl; task < VC_ROWS * 2; task += 32) {
            const int row = task >> 1, ch = task & 1;
            const StageRegs sr = stage_load(img, stride, w, h, yt - 5 + row, xt - 8 + 16 * ch);
            stage_store<VC_ROWB, VC_PLANE>(lds, row, ch, sr);
        }
        __syncthreads();
        uint32_t m5[12], p5[12], m4[12], p4[12], m2[12], p2[12], z1[12];
        load12(m5, lane_base - 5 * VC_ROWB);
        load12(p5, lane_base + 5 * VC_ROWB);
        load12(m4, lane_base - 4 * VC_ROWB);
        load12(p4, lane_base + 4 * VC_ROWB);
        load12(m2, lane_base - 2 * VC_ROWB + VC_PLANE);
        load12(p2, lane_base + 2 * VC_ROWB + VC_PLANE);
        load12(z1, lane_base + VC_PLANE);
        const u32x4 z0v = lds_read_b128(lane_base + 16);
        const uint32_t z0[4] = {z0v.x, z0v.y, z0v.z, z0v.w};
        const int x0 = xt + 8 * lx, yy = yt + trow;
        const bool rowin = yy >= kMargin && yy < h - kMargin;
        uint32_t out[4], bits = 0;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const int xa = x0 + 2 * k, xb = xa + 1;
            const bool ina = rowin && xa >= kMargin && xa < w - kMargin, inb = rowin && xb >= kMargin && xb < w - kMargin;
            const uint32_t xmask = (ina ? 0x2000u : 0xffffu) | (inb ? 0x20000000u : 0xffff0000u);
            // one saturating subtraction strips the bias, clamps at 0 and zeroes the 7-pixel frame
            out[k] = pk_sub_sat_u16(response_pair_biased(m5, p5, m4, p4, m2, p2, z1, z0, k), xmask);
            uint32_t m;  // both halves to 0 / 1, then to their bit positions (as in the production kernel)
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(out[k] & 0xfff0fff0u), "v"(0x00010001u));
            bits = dot2_u32_u16(m, (1u << (2 * k)) | (2u << (2 * k + 16)), bits);
        }
        if (have && yy < h && x0 < w) {
            int16_t* dst = resp + (long long)yy * w + x0;
            if (x0 + 8 <= w) {
                const u32x4 v = {out[0], out[1], out[2], out[3]};
                __builtin_memcpy(dst, &v, 16);
            } else {
                for (int i = 0; i < w - x0; ++i) dst[i] = (int16_t)(out[i >> 1] >> (16 * (i & 1)));
            }
        } else {
            bits = 0;
        }
        // which pixels are hot: 32 bytes per micro-tile (byte = a lane's 8 pixels, 2 bytes per row), in the frame's part
        // of the pixel -> list-index map (used by the global-memory kernels only, which never see a sparse level).
        // NOT appended to the hot list here: an atomic per wave on the frame's counter -- one address for workgroups on
        // all eight XCDs -- costs 0.3 us and they serialise (80 us per launch, whatever else the kernel did); the
        // refinement kernel, one workgroup per frame, expands the masks itself (cc.hip, hot_list_from_masks).
        if (have) masks[(long long)it * 32 + hl] = (uint8_t)bits;
    }
}

void launch_chess_cells(const LevelBatch& lb, const CompTables& t, const uint32_t* cell_list, const int32_t* cell_cnt,
                        int list_pitch, int frame0, int nframes, hipStream_t s) {
    if (nframes <= 0 || lb.w <= 0 || lb.h <= 0) return;
    // a frame has a few hundred micro-tiles and each is a chain of dependent round trips (list -> pixels -> counter):
    // one pair per workgroup where the grid allows it (an idle workgroup costs two loads)
    int per_frame = nframes >= 32768 ? 1 : (32768 + nframes - 1) / nframes;
    if (per_frame > 512) per_frame = 512;
    hipLaunchKernelGGL(chess_cells_kernel, dim3(per_frame, nframes), dim3(64), 0, s, lb, t, cell_list, cell_cnt, list_pitch, frame0);
}

// Several pyramid levels of the same batch in ONE grid (clamp + hot list, widths that are multiples
// of 16): the small levels have too few workgroups to fill the chip on their own, and every kernel
// boundary on the pixel stream costs 7-12 us (end-of-kernel cache write-back + dispatch).  The
// levels are laid out largest first, so the small ones fill the tail of the large one.
constexpr int kMultiMax = 4;
struct ChessMulti {
    LevelBatch lb[kMultiMax];
    CompTables t[kMultiMax];
    int first_wg[kMultiMax];  // first workgroup of level slot k (a multiple of 8: workgroup b runs on XCD b % 8,
    int nwg[kMultiMax];       // and the XCD-aware work order of the body counts from the slot's first workgroup)
    int nsegs[kMultiMax];     // balanced row segments per frame (common.h, segment_rows)
    int n;
};
template <bool FILTER>  // FILTER: only the frames a sparse chain reported (CompTables::only, see chess_v1_body)
__global__ __launch_bounds__(256, 4) void chess_v1_multi_kernel(ChessMulti a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int b = blockIdx.x;
    int k = 0;  // uniform
#pragma unroll
    for (int j = 1; j < kMultiMax; ++j)
        if (j < a.n && b >= a.first_wg[j]) k = j;
    const int rel = b - a.first_wg[k];
    if (rel >= a.nwg[k]) return;  // padding between slots
    chess_v1_body<true, true, STAGE_PERM16, false, FILTER>(a.lb[k], a.t[k], 0, a.nsegs[k], (unsigned)rel, (unsigned)a.nwg[k], lds);
}


int chess_multi_min_blocks = 768;  // tuning hook "chess_multi_min_blocks": per-level block target inside a merged launch
int chess_stage_override = 0;  // tuning hook "chess_stage": 0 = automatic, 2 / 3 = typed staging, -1 = generic

// Row segments per frame (balanced: common.h, segment_rows).  Tall segments amortise the three-group prologue (24
// rows staged before the first response row), short ones fill the chip when the batch is small and leave a shorter
// drain; the model and its fit are in common.h (pick_balanced_segments).  Before round 6 the choice was among 256 / 128 /
// 64 / 32 rows, cut from the top of the frame: 1080 rows became eight segments of 128 and one of 56, and the short one
// every ninth workgroup put the tall ones on the same CUs (64 x 1920x1080, hot kernel: 132 -> 123 us with six of 180;
// 2048x1536: 198 -> 166 us).
// `seg_rows` > 0: the caller's segment height (option "chess_seg").
// `min_blocks` > 0: the older rule for the levels inside a merged launch (tallest segment that still gives that
// many workgroups), where the largest level fills the chip and the others only pack its tail.
static const SegModel kV1Model = {6.8, 0.57, 13.0, 32, 1024};      // plain response
static const SegModel kV1PyrModel = {12.0, 0.8, 13.0, 32, 1024};   // with the hot list and the level images (level 0 of a chain)
// with the hot list alone: 10 rows' worth per workgroup, not 12 -- at 64 x 1920x1080 the two models differ by one segment
// (six of 184 rows against five of 216) and the sweep has six to eight at 122 us, five at 125 (profiles/r06_seg_rounds_sweep.txt)
static const SegModel kV1HotModel = {10.0, 0.8, 13.0, 32, 1024};
static int pick_nsegs(int w, int h, int nframes, int seg_rows, bool hot, int min_blocks = 0, bool pyr = false) {
    if (seg_rows > 0) return segments_for_rows(h, seg_rows, V1_RB);
    const int strips = (w + V1_SW - 1) / V1_SW;
    if (min_blocks > 0) {
        for (int seg : {256, 128, 64, 32}) {
            const long long blocks = (long long)strips * ((h + seg - 1) / seg) * nframes;
            if (blocks >= min_blocks || seg == 32) return segments_for_rows(h, seg, V1_RB);
        }
    }
    return pick_balanced_segments(strips, h, nframes, V1_RB, pyr ? kV1PyrModel : (hot ? kV1HotModel : kV1Model));
}

// Production entry point.
void launch_chess(const LevelBatch& lb, const CompTables& t, int frame0, int nframes, bool clamp, bool hot,
                  hipStream_t s, int seg_rows) {
    if (hot && t.only) nframes = kOnlySlots;  // a frame list: the grid is laid out for that many frames (chess_v1_body)
    const int nsegs = pick_nsegs(lb.w, lb.h, nframes, (hot && t.only) ? 256 : seg_rows, hot);
    dim3 grid(((lb.w + V1_SW - 1) / V1_SW) * nsegs * nframes);
    const size_t lds = 2 * V1_PLANE + (hot ? (V1_HOTBUF + 12) * sizeof(int) : 0);
    const bool w16 = lb.w >= 16 && lb.w % 16 == 0 && (long long)(lb.h + V1_RB) * lb.img_stride < 0x7fffffffLL;
    // staging path: chess_stage_override (tuning hook "chess_stage") 0 = automatic
    // widths that are a multiple of 16 take the v_perm staging (fastest there: 624 vs 637-644 us per 64 frames of
    // 4096x3072); every other width takes the typed staging, which needs no edge handling at all (4090x3070:
    // 0.377 ms per 32 frames against 0.552 ms with the generic staging)
    const bool typed_ok = (long long)lb.h * lb.img_stride < 0x7fffffffLL && lb.img_pitch >= 0;
    int stage = w16 ? STAGE_PERM16 : (typed_ok ? STAGE_TYPED1 : STAGE_GENERIC);
    if (chess_stage_override == STAGE_TYPED2 || chess_stage_override == STAGE_TYPED1) {
        if (typed_ok) stage = chess_stage_override;
    } else if (chess_stage_override == -1) {
        stage = STAGE_GENERIC;
    }
    // (the P0 + P1 typed staging lost to the others everywhere it was measured: instantiated in experiment builds only)
#ifdef MRG_EXPERIMENT
#define MRG_CASE_TYPED2(C, H) case STAGE_TYPED2: MRG_LAUNCH(C, H, STAGE_TYPED2); break;
#else
#define MRG_CASE_TYPED2(C, H)
#endif
#define MRG_LAUNCH(C, H, A) hipLaunchKernelGGL((chess_v1_kernel<C, H, A>), grid, dim3(256), lds, s, lb, t, frame0, nsegs)
#define MRG_LAUNCH_ST(C, H)                                             \
    switch (stage) {                                                    \
        case STAGE_PERM16: MRG_LAUNCH(C, H, STAGE_PERM16); break;       \
        MRG_CASE_TYPED2(C, H)                                           \
        case STAGE_TYPED1: MRG_LAUNCH(C, H, STAGE_TYPED1); break;       \
        default: MRG_LAUNCH(C, H, STAGE_GENERIC); break;                \
    }
    if (hot && t.only) {  // only the frames a sparse chain reported (see chess_v1_body)
        switch (stage) {
            case STAGE_PERM16: hipLaunchKernelGGL((chess_v1_kernel<true, true, STAGE_PERM16, true>), grid, dim3(256), lds, s, lb, t, frame0, nsegs); break;
            case STAGE_TYPED1: hipLaunchKernelGGL((chess_v1_kernel<true, true, STAGE_TYPED1, true>), grid, dim3(256), lds, s, lb, t, frame0, nsegs); break;
            default: hipLaunchKernelGGL((chess_v1_kernel<true, true, STAGE_GENERIC, true>), grid, dim3(256), lds, s, lb, t, frame0, nsegs); break;
        }
    }
    else if (hot) { MRG_LAUNCH_ST(true, true) }
    else if (clamp) { MRG_LAUNCH_ST(true, false) }
    else { MRG_LAUNCH_ST(false, false) }
#undef MRG_LAUNCH_ST
#undef MRG_LAUNCH
#undef MRG_CASE_TYPED2
}

// Level 0 with the pyramid fused in; false when the shape does not qualify (whole cells only, 16-byte rows).
bool chess_pyramid_ok(const LevelBatch& lb, int nframes) {
    return nframes > 0 && lb.w >= 16 && lb.w % 16 == 0 && lb.h >= 8 && lb.h % 8 == 0 && lb.img_stride % 16 == 0 &&
           lb.img_pitch % 16 == 0 && ((uintptr_t)lb.img & 15) == 0 &&
           (long long)(lb.h + V1_RB) * lb.img_stride < 0x7fffffffLL;  // buffer offsets of the staging loads fit 31 bits
}
bool launch_chess_pyramid(const LevelBatch& lb, const CompTables& t, const PyramidOut& po, int nframes, hipStream_t s,
                          int seg_rows) {
    if (!chess_pyramid_ok(lb, nframes)) return false;
    const int nsegs = pick_nsegs(lb.w, lb.h, nframes, seg_rows, true, 0, true);  // segments of whole 8-row granules: an iteration's rows are whole cells
    dim3 grid(((lb.w + V1_SW - 1) / V1_SW) * nsegs * nframes);
    const size_t lds = 2 * V1_PLANE + (V1_HOTBUF + 12) * sizeof(int);
    PyramidOut p2 = po;
#ifdef MRG_EXPERIMENT
    // MRGINGHAM_AMD_PYR_SKIP (timing ablation, tools/interference_ab.py "dbg fused"): bit k = level k + 1 is not written
    static const int skip = [] { const char* e = getenv("MRGINGHAM_AMD_PYR_SKIP"); return e ? atoi(e) : 0; }();
    for (int k = 0; k < 3; ++k)
        if (skip >> k & 1) p2.out[k] = nullptr;
#endif
    hipLaunchKernelGGL(chess_v1_pyr_kernel, grid, dim3(256), lds, s, lb, t, nsegs, p2);
    return true;
}

// Levels lbs[0..n) (n <= 3, largest first) of one batch in one launch; returns false when the shapes do
// not qualify (then the caller launches them one by one).
bool chess_multi_ok(const LevelBatch* lbs, int n, int nframes) {
    if (n < 2 || n > kMultiMax || nframes <= 0) return false;
    for (int k = 0; k < n; ++k)
        if (lbs[k].w < 16 || lbs[k].w % 16 != 0 || lbs[k].h <= 0 ||
            (long long)(lbs[k].h + V1_RB) * lbs[k].img_stride >= 0x7fffffffLL) return false;
    return true;
}

bool launch_chess_multi(const LevelBatch* lbs, const CompTables* ts, int n, int nframes, hipStream_t s, int seg_rows) {
    if (!chess_multi_ok(lbs, n, nframes)) return false;
    if (ts[0].only) nframes = kOnlySlots;  // a frame list: the grid is laid out for that many frames (chess_v1_body)
    ChessMulti a;
    a.n = n;
    int total = 0;
    for (int k = 0; k < kMultiMax; ++k) {
        const int j = k < n ? k : 0;
        a.lb[k] = lbs[j];
        a.t[k] = ts[j];
        // inside a merged launch the largest level fills the chip; the smaller ones only need enough
        // workgroups to pack its tail, so they can afford taller segments than on their own
        // (a frame list: tall segments, i.e. as few workgroups as possible -- nearly always all of them leave at once)
        a.nsegs[k] = pick_nsegs(lbs[j].w, lbs[j].h, nframes, ts[0].only ? 256 : seg_rows, true, k == 0 ? 2048 : chess_multi_min_blocks);
        a.first_wg[k] = total;
        a.nwg[k] = k < n ? ((lbs[j].w + V1_SW - 1) / V1_SW) * a.nsegs[k] * nframes : 0;
        total += (a.nwg[k] + 7) / 8 * 8;
    }
    const size_t lds = 2 * V1_PLANE + (V1_HOTBUF + 12) * sizeof(int);
    if (ts[0].only) hipLaunchKernelGGL(chess_v1_multi_kernel<true>, dim3(total), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(chess_v1_multi_kernel<false>, dim3(total), dim3(256), lds, s, a);
    return true;
}

}  // namespace mrg
