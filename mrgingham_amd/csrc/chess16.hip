// ChESS response, second decomposition (round 5): SIXTEEN pixels per lane.  The library's kernel for the response WITHOUT a
// hot list (mrgingham_amd_chess_response_batch, the literal output of ChESS.c:56-106) on widths that are multiples of 16:
// 3.5-4.5 % faster than chess_v1_kernel at 4096x3072, 1-8 % at the smaller sizes (tools/chess16_sweep.py).  The variants with
// the hot list, the level images and several levels per launch exist in experiment builds only: measured in the chain they
// are NOT faster than chess_v1's.  Same algebra, same LDS format (two planes of packed u16 pixel pairs) and the same
// rolling strip as chess_v1 (chess.hip); what changes is the shape of a lane's work:
//
//   * a lane owns 16 adjacent pixels (8 pairs) of one row; 16 lanes cover the 256-pixel strip, the four quarter-waves
//     of a wave take four consecutive rows, the four waves 16 rows per iteration (v1: 8 pixels, 2 rows, 8 rows);
//   * the window of a (plane, row) is 16 dwords = 4 ds_read_b128 for 16 pixels (v1: 3 for 8): 30 reads and 120
//     returned dwords per 16 pixels instead of 44 and 176; window addresses, masks and the barrier are paid per
//     lane-iteration, i.e. half as often per pixel;
//   * the window is 272 pixels (8 of halo on either side: v1 has 16) and rows are 560 bytes apart (544 + 16), so that consecutive rows sit an ODD number of 16-byte slots apart: a
//     ds_read_b128 is serviced in groups of 16 lanes that here span two rows of 8 lanes at a 32-byte stride, and the
//     odd pitch puts the second row on the slots the first leaves free (conflict-free; with an even pitch every read is 2-way);
//   * every thread stages the same: one 16-pixel chunk of the strip (16-byte load, v_perm into the two planes) and one pixel
//     pair of the halo (a typed two-byte load that arrives as two u16): 256 + 256 tasks for 256 threads, no wave does more;
//   * the ring has 44 rows (16 computed + 10 halo + 16 arriving, rounded to a multiple of 4): 49 KB, three
//     workgroups per CU, and the kernel may use 168 VGPRs.  Slots are NOT a power of two: a wave's four rows are an
//     aligned group of four for dy = 0, +-4 and straddle two groups otherwise -- the group offsets are wave-uniform
//     (SALU), the lane's choice between them a precomputed mask.
#include "common.h"
#include "hotlist.h"
#include "chess_hot.h"
#include "kernels.h"


namespace mrg {

namespace v16 {

constexpr int SW = 256;                 // strip width, output pixels
constexpr int HL = 8;                   // halo of the LDS window on either side
constexpr int WIN = SW + 2 * HL;        // 272
constexpr int ROWB = WIN * 2 + 16;      // 560 bytes per row and plane: 35 sixteen-byte slots, an ODD number
constexpr int NR = 44;                  // ring rows
constexpr int RB = 16;                  // rows per iteration
constexpr int PLANE = NR * ROWB;        // 24640
constexpr int RING = NR * ROWB;         // bytes of one plane's ring (wrap length)

using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
using i16x2 = short __attribute__((ext_vector_type(2)));
using i32x4 = int __attribute__((ext_vector_type(4)));
using u32x4 = uint32_t __attribute__((ext_vector_type(4)));

__device__ u32x4 raw_buffer_load_b128(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ unsigned char buffer_load_u8(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i8");
// two bytes -> one register of two u16 (the texture unit unpacks: 8_8 UINT elements, any byte address, 0 out of range)
__device__ u16x2 buffer_load_u8x2_as_u16x2(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v2i16");
__device__ void raw_buffer_store_b128(u32x4 data, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.v4i32");
constexpr int kAuxNT = 2;
__device__ __forceinline__ i32x4 raw_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};
}
__device__ __forceinline__ i32x4 typed88_rsrc(const void* base, uint32_t bytes) {
    // word 3: dst_sel x, y = R, G (4, 5), z, w = 0; num_format UINT (4) << 12; data_format 8_8 (3) << 15
    constexpr uint32_t w3 = (4u | (5u << 3)) | (4u << 12) | (3u << 15);
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, (int)w3};
}

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (i16x2)(__builtin_bit_cast(i16x2, a) - __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

__device__ __forceinline__ uint32_t dot2_u32_u16(uint32_t a, uint32_t b, uint32_t c) {  // a.lo * b.lo + a.hi * b.hi + c
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

struct StageRegs {
    uint4 g;        // 16 pixels of the strip
    uint32_t next;  // byte 0 = the pixel after them
    uint32_t halo;  // one pair of the 8-pixel halo, already as two u16
};

// 16 bytes -> the P0 and P1 entries of the chunk (as chess_v1's stage_store), at byte offset `off` of plane P0
__device__ __forceinline__ void stage_store(char* lds, uint32_t off, uint32_t hoff, const StageRegs& s) {
    const uint32_t g0 = s.g.x, g1 = s.g.y, g2 = s.g.z, g3 = s.g.w;
    constexpr uint32_t S01 = 0x0c010c00u, S23 = 0x0c030c02u, S12 = 0x0c020c01u, S34 = 0x0c040c03u;
    uint4 a, b, c, d;
    a.x = __builtin_amdgcn_perm(g0, g0, S01); a.y = __builtin_amdgcn_perm(g0, g0, S23);
    a.z = __builtin_amdgcn_perm(g1, g1, S01); a.w = __builtin_amdgcn_perm(g1, g1, S23);
    b.x = __builtin_amdgcn_perm(g2, g2, S01); b.y = __builtin_amdgcn_perm(g2, g2, S23);
    b.z = __builtin_amdgcn_perm(g3, g3, S01); b.w = __builtin_amdgcn_perm(g3, g3, S23);
    c.x = __builtin_amdgcn_perm(g0, g0, S12); c.y = __builtin_amdgcn_perm(g1, g0, S34);
    c.z = __builtin_amdgcn_perm(g1, g1, S12); c.w = __builtin_amdgcn_perm(g2, g1, S34);
    d.x = __builtin_amdgcn_perm(g2, g2, S12); d.y = __builtin_amdgcn_perm(g3, g2, S34);
    d.z = __builtin_amdgcn_perm(g3, g3, S12); d.w = __builtin_amdgcn_perm(s.next, g3, S34);
    char* p = lds + off;
    *reinterpret_cast<uint4*>(p) = a;
    *reinterpret_cast<uint4*>(p + 16) = b;
    *reinterpret_cast<uint4*>(p + PLANE) = c;
    *reinterpret_cast<uint4*>(p + PLANE + 16) = d;
    *reinterpret_cast<uint32_t*>(lds + hoff) = s.halo;
}

__device__ __forceinline__ u32x4 lds_read_b128(const char* p) {
    u32x4 v = *reinterpret_cast<const u32x4*>(p);
    asm("" : "+v"(v));
    return v;
}
__device__ __forceinline__ void load16(uint32_t (&R)[16], const char* p) {
    const u32x4 a = lds_read_b128(p), b = lds_read_b128(p + 16), c = lds_read_b128(p + 32), d = lds_read_b128(p + 48);
    R[0] = a.x; R[1] = a.y; R[2] = a.z; R[3] = a.w;
    R[4] = b.x; R[5] = b.y; R[6] = b.z; R[7] = b.w;
    R[8] = c.x; R[9] = c.y; R[10] = c.z; R[11] = c.w;
    R[12] = d.x; R[13] = d.y; R[14] = d.z; R[15] = d.w;
}

// response + 8192 per half of pixel pair k (0..7) of the lane (chess.hip, response_pair_biased; ChESS.c:68-104)
__device__ __forceinline__ uint32_t response_pair_biased(const uint32_t (&m5)[16], const uint32_t (&p5)[16],
                                                         const uint32_t (&m4)[16], const uint32_t (&p4)[16],
                                                         const uint32_t (&m2)[16], const uint32_t (&p2)[16],
                                                         const uint32_t (&z1)[16], const uint32_t (&z0)[8], int k) {
    const int c = 4 + k;
    const uint32_t a0 = m5[c + 1], c0 = p5[c - 1], b0 = m2[c - 3], d0 = p2[c + 2];
    const uint32_t a1 = m5[c], c1 = p5[c], b1 = z1[c - 3], d1 = z1[c + 2];
    const uint32_t a2 = m5[c - 1], c2 = p5[c + 1], b2 = p2[c - 3], d2 = m2[c + 2];
    const uint32_t a3 = m4[c - 2], c3 = p4[c + 2], b3 = p4[c - 2], d3 = m4[c + 2];
    const uint32_t t10 = a0 + c0, t20 = b0 + d0, t11 = a1 + c1, t21 = b1 + d1;
    const uint32_t t12 = a2 + c2, t22 = b2 + d2, t13 = a3 + c3, t23 = b3 + d3;
    const uint32_t M = ((t10 + t20) + (t11 + t21)) + ((t12 + t22) + (t13 + t23));
    const uint32_t Yb = ((pk_max_u16(t10, t20) + pk_max_u16(t11, t21)) + (pk_max_u16(t12, t22) + pk_max_u16(t13, t23))) + 0x10001000u;
    const uint32_t X = ((pk_max_u16(a0, c0) + pk_max_u16(b0, d0)) + (pk_max_u16(a1, c1) + pk_max_u16(b1, d1))) +
                       ((pk_max_u16(a2, c2) + pk_max_u16(b2, d2)) + (pk_max_u16(a3, c3) + pk_max_u16(b3, d3)));
    const uint32_t n = z1[c - 1] + z0[k] + z1[c];
    const uint32_t lm_lo = __umul24(n & 0xffffu, 349536u);
    const uint32_t lm_hi = __umul24(n >> 16, 349536u);
    const uint32_t LM = __builtin_amdgcn_perm(lm_hi, lm_lo, 0x07060302u);
    const uint32_t dev = pk_max_u16(M, LM) - pk_min_u16(M, LM);
    const uint32_t d1x = Yb - X;
    return (d1x + d1x) - dev;
}

#ifdef MRG_EXPERIMENT  // (the fused variants: measured, not faster than chess_v1's -- DESIGN.md section 9; experiment builds only)
// ---------------------------------------------------------------------------
// Level images 1..3 out of the ring (chess.hip, emit_pyramid_rows: frames of whole 16 x 8 blocks; every level pixel is
// (a+b+c+d+2)>>2 of four FULL-RESOLUTION pixels, and a 2x2 cell is one packed pair of two consecutive rows).  Here an
// iteration holds 16 frame rows = two blocks of 8: wave 2 emits the block y .. y+7, wave 3 the block y+8 .. y+15, each
// exactly like chess_v1's wave 3 (level 1: eight pixels per lane, levels 2 and 3 on the two half-waves).  A block is two
// aligned groups of four ring rows, whose offsets (gA, gB) are wave-uniform; the ring does not wrap inside a group.
// ---------------------------------------------------------------------------
__device__ void raw_buffer_store_b64(uint32_t __attribute__((ext_vector_type(2))) data, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.v2i32");
__device__ void raw_buffer_store_b32(uint32_t data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");
__device__ void raw_buffer_store_b8(unsigned char data, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i8");

__device__ __forceinline__ uint32_t cells4(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    const uint32_t lo01 = __builtin_amdgcn_perm(v1, v0, 0x05040100u), hi01 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const uint32_t lo23 = __builtin_amdgcn_perm(v3, v2, 0x05040100u), hi23 = __builtin_amdgcn_perm(v3, v2, 0x07060302u);
    const uint32_t s01 = (lo01 + hi01 + 0x00020002u) >> 2, s23 = (lo23 + hi23 + 0x00020002u) >> 2;
    return __builtin_amdgcn_perm(s23, s01, 0x06040200u);
}
// y8: first frame row of the block (a multiple of 8); gA / gB: ring byte offsets of rows y8 .. y8+3 / y8+4 .. y8+7
__device__ __forceinline__ void emit_pyramid_block(const char* lds, uint32_t gA, uint32_t gB, int y8, int strip_x, int w, int frame,
                                                   int lane, const PyramidOut& po) {
    using u32x2 = uint32_t __attribute__((ext_vector_type(2)));
    const int j = lane & 15, R = (lane >> 4) & 3;
    const bool l3 = lane >= 32;
    // level 1, output row R of the block's four: frame rows 2R, 2R+1 -- group A for R < 2, B otherwise
    const uint32_t g1 = R < 2 ? gA : gB;
    const char* r1 = lds + g1 + (uint32_t)(2 * HL + (R & 1) * (2 * ROWB) + 32 * j);
    // levels 2 (lanes 0..31: output row R & 1, frame rows 4(R&1)+1, +2) and 3 (lanes 32..63: frame rows 3 and 4)
    const uint32_t g2 = (R & 1) ? gB : gA;
    const char* r2a = lds + PLANE + 2 * HL + (l3 ? gA + 3 * ROWB + 16 * (lane - 32) : g2 + ROWB + 32 * j);
    const char* r2b = lds + PLANE + 2 * HL + (l3 ? gB + 16 * (lane - 32) : g2 + 2 * ROWB + 32 * j);
    const u32x4 a0 = lds_read_b128(r1), a1 = lds_read_b128(r1 + 16);
    const u32x4 b0 = lds_read_b128(r1 + ROWB), b1 = lds_read_b128(r1 + ROWB + 16);
    const u32x4 c0 = lds_read_b128(r2a), c1 = lds_read_b128(r2a + 16);
    const u32x4 d0 = lds_read_b128(r2b), d1 = lds_read_b128(r2b + 16);
    if (po.out[0]) {
        const i32x4 rs = raw_rsrc(po.out[0] + (long long)frame * po.h[0] * po.w[0], (uint32_t)(po.h[0] * po.w[0]));
        u32x2 o;
        o.x = cells4(a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w);
        o.y = cells4(a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w);
        if (strip_x + 16 * j < w)
            raw_buffer_store_b64(o, rs, R * po.w[0] + 8 * j + (strip_x >> 1), (y8 >> 1) * po.w[0], 0);
    }
    if (po.out[1]) {
        const uint32_t v0 = l3 ? c0.y + d0.y : c0.x + d0.x;
        const uint32_t o = cells4(v0, c0.z + d0.z, c1.x + d1.x, c1.z + d1.z);
        if (!l3) {
            if (strip_x + 16 * j < w) {
                const i32x4 rs = raw_rsrc(po.out[1] + (long long)frame * po.h[1] * po.w[1], (uint32_t)(po.h[1] * po.w[1]));
                raw_buffer_store_b32(o, rs, (R & 1) * po.w[1] + 4 * j + (strip_x >> 2), (y8 >> 2) * po.w[1], 0);
            }
        } else if (po.out[2] && strip_x + 8 * (lane - 32) < w) {
            const i32x4 rs = raw_rsrc(po.out[2] + (long long)frame * po.h[2] * po.w[2], (uint32_t)(po.h[2] * po.w[2]));
            raw_buffer_store_b8((unsigned char)o, rs, lane - 32 + (strip_x >> 3), (y8 >> 3) * po.w[2], 0);
        }
    }
}

#endif  // MRG_EXPERIMENT

}  // namespace v16

// Frames whose width is a multiple of 16 only (the staging loads are whole 16-byte chunks inside or outside the frame).
// The body for workgroup `bid` of `nwg` of one level (the multi-level launch runs several levels in one grid).
// HOT (implies CLAMP): the hot-pixel records of chess_hot.h, two aligned 8-pixel groups per lane.
// PAIR (experiment builds; widths with w % 256 == 128, an even number of segments, no hot list): the half-empty last
// strip's workgroups take TWO consecutive row segments at once -- lanes 0-7 of a quarter-wave on segment A, lanes 8-15 on
// segment B, the window laid out [A's left halo 8 | A 128 | B's left halo 8 | B 128] = the 272 pixels it has (A's right
// halo lies outside the frame and only feeds masked outputs).  The grid is nwg_level ordinary workgroups over the full
// strips followed by (nsegs / 2) * frames paired ones; everything that differs is a per-lane constant set up front.
template <bool CLAMP, bool HOT, bool PYR = false, bool PAIR = false>
__device__ __forceinline__ void chess_v16_body(const LevelBatch& lb, const CompTables& t, int frame0, int nsegs, unsigned bid,
                                               unsigned nwg_level, char* lds, const PyramidOut* po = nullptr) {
    using namespace v16;
    const int nstrips = PAIR ? lb.w / SW : (lb.w + SW - 1) / SW;
    const bool paired = PAIR && bid >= nwg_level;  // (workgroup-uniform)
    int work;
    {   // XCD-aware work order (chess.hip, chess_v1_body)
        const unsigned b = bid, nwg = nwg_level, xcd = b & 7u, j = b >> 3;
        const unsigned q = nwg >> 3, r = nwg & 7u;
        work = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j);
    }
    const bool probe = lb.clk != nullptr && bid == 0;  // (mrgingham_amd_sclk_mhz)
    const ClockProbe clkp = clock_probe_begin(probe);
    int strip = work % nstrips, rest = work / nstrips;
    if (paired) {  // the half strip, segments 2 i and 2 i + 1 of a frame
        const int pw = (int)(bid - nwg_level), half_n = nsegs >> 1;
        strip = nstrips;
        rest = (pw / half_n) * nsegs + 2 * (pw % half_n);
    }
    const int frame = frame0 + rest / nsegs;
    const int w = lb.w, h = lb.h, stride = lb.img_stride;
    const uint8_t* img = lb.img + (long long)frame * lb.img_pitch;
    int16_t* resp = lb.resp + (long long)frame * lb.resp_pitch;
    const int strip_x = strip * SW;
    int ys, ye;  // segment rest % nsegs of the frame's nsegs balanced segments (common.h)
    segment_rows(h, nsegs, rest % nsegs, RB, ys, ye);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, jx = lane & 15;
    // paired: this lane's segment (B for jx >= 8) starts dB rows below A's and ends at ye_lane; the loop runs to the longer one
    int dB = 0, ye_lane = ye, ye_loop = ye;
    const bool laneB = paired && jx >= 8;
    if (paired) {
        int ysB, yeB;
        segment_rows(h, nsegs, rest % nsegs + 1, RB, ysB, yeB);
        ye_loop = ys + max(ye - ys, yeB - ysB);
        if (laneB) { dB = ysB - ys; ye_lane = yeB; }
    }
    uint32_t* hotbuf = reinterpret_cast<uint32_t*>(lds + 2 * PLANE);
    int* hotcnt = reinterpret_cast<int*>(hotbuf + V1_HOTBUF);
    HotSink hsink{hotbuf + (tid >> 6) * V1_HOTSEG, ys, strip_x, 0};

    // Staging, the same for every thread (16 rows per group):
    //   one 16-pixel chunk of the strip: row tid / 16, chunk tid % 16 (a 16-byte load + the byte after it, v_perm into
    //   the two planes, four ds_write_b128);
    //   one pixel pair of the 8-pixel halo on either side: row tid / 16, then side | plane | pair in tid % 16 (a typed
    //   two-byte load that arrives as two u16 = the plane's entry as it is, one ds_write_b32).  Out-of-range offsets
    //   (columns left of the frame, rows above and below it) read 0: such pixels only reach outputs that are masked.
    const int trow = tid >> 4, tch = tid & 15;
    const int hside = (tid >> 3) & 1, hplane = (tid >> 2) & 1, hpair = tid & 3;
    const uint32_t fbytes = (uint32_t)min((long long)(h - 1) * stride + w, 0x7fffffffLL);
    const i32x4 img_rsrc = raw_rsrc(img, fbytes);
    const i32x4 img88 = typed88_rsrc(img, fbytes);
    int vg = trow * stride + strip_x + 16 * tch;                             // + first row of the group * stride
    int vn = trow * stride + min(strip_x + 16 * tch + 16, w - 1);
    int vh = trow * stride + strip_x + (hside ? SW : -HL) + 2 * hpair + hplane;
    uint32_t chunk_col = (uint32_t)(2 * HL + 32 * tch);                      // byte offset of the chunk within a row of P0
    uint32_t halo_col = (uint32_t)((hside ? 2 * (SW + HL) : 0) + 4 * hpair + hplane * PLANE);
    if (paired) {
        // chunks 8..15 are B's chunks 0..7 (B's rows, 8 window pixels further right than the ordinary layout puts them);
        // "side 1" of the halo is B's LEFT halo, in front of B's pixels
        int ysB, yeB;
        segment_rows(h, nsegs, rest % nsegs + 1, RB, ysB, yeB);
        const int rowsB = (ysB - ys) * stride;
        if (tch >= 8) {
            vg = trow * stride + strip_x + 16 * (tch - 8) + rowsB;
            vn = trow * stride + min(strip_x + 16 * (tch - 8) + 16, w - 1) + rowsB;
            chunk_col += 2 * HL;
        }
        if (hside) {
            vh = trow * stride + strip_x - HL + 2 * hpair + hplane + rowsB;
            halo_col = (uint32_t)(2 * (HL + SW / 2) + 4 * hpair + hplane * PLANE);
        }
    }
    auto stage_load = [&](int row0) {
        StageRegs s;
        const int rowoff = row0 * stride;
        s.g = __builtin_bit_cast(uint4, raw_buffer_load_b128(img_rsrc, vg + rowoff, 0, 0));
        s.next = buffer_load_u8(img_rsrc, vn + rowoff, 0, 0);
        s.halo = __builtin_bit_cast(uint32_t, buffer_load_u8x2_as_u16x2(img88, vh + rowoff, 0, 0));
        return s;
    };
    // ring byte offset (plane P0) of frame row ys + rel: slot (rel mod 44)
    auto slot_off = [&](int rel) { return (uint32_t)(((rel % NR) + NR) % NR) * ROWB; };

    // prologue: rows ys - 11 .. ys + 20 (two groups of 16; the first six are never read)
    {
        const StageRegs a = stage_load(ys - 11), b = stage_load(ys + 5);
        const uint32_t oa = slot_off(-11 + trow), ob = slot_off(5 + trow);
        stage_store(lds, oa + chunk_col, oa + halo_col, a);
        stage_store(lds, ob + chunk_col, ob + halo_col, b);
    }
    __syncthreads();

    // per-lane constants
    const int x0 = strip_x + 16 * jx - (laneB ? SW / 2 : 0);
    uint32_t xmask[8], xadd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int xa = x0 + 2 * k, xb = xa + 1;
        const bool ina = xa >= kMargin && xa < w - kMargin, inb = xb >= kMargin && xb < w - kMargin;
        if (CLAMP) xmask[k] = (ina ? 0x2000u : 0xffffu) | (inb ? 0x20000000u : 0xffff0000u);
        else {
            xmask[k] = (ina ? 1u : 0u) | (inb ? 0x10000u : 0u);              // the multiplier m of the raw epilogue
            xadd[k] = (ina ? 0xe000u : 0u) | (inb ? 0xe0000000u : 0u);      // -8192 * m per half
        }
    }
    const uint32_t lane_col = 32u * jx + (laneB ? 2u * HL : 0u);  // D[-4] of the lane within a row: pixel x0 - 8 = window pixel 16 jx (paired, B: + 8)
    // dy classes c = dy mod 4: the lane's row is row (c + q) & 3 of group B (c + q < 4) or of the group after it
    uint32_t LC[4], MK[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        LC[c] = lane_col + (uint32_t)((c + q) & 3) * ROWB;
        MK[c] = (c + q >= 4) ? 0xffffffffu : 0u;
    }
    const bool seg_interior = !paired && ys >= kMargin && ye <= h - kMargin;
    const i32x4 resp_rsrc = raw_rsrc(resp, (uint32_t)min((long long)w * h * 2, 0xffffffffLL));
    const int st_resp_voff = ((q + dB) * w + x0) * 2;
    // running ring offset of the staging row (row y + 21 + trow of iteration y)
    uint32_t so = slot_off(21 + trow);

    int am = 4 * wvu;  // (y - ys + 4 * wave) mod 44: the ring slot of the wave's first row, a multiple of 4
    for (int y = ys; y < ye_loop; y += RB) {
        // prefetch rows y + 21 .. y + 36 (needed by the next iteration)
        const StageRegs pre = stage_load(y + 21);

        // window rows: wave-uniform group offsets (the aligned groups of four rows that start -8, -4, 0, +4, +8 rows from the
        // wave's own group), per-lane choice between a group and the next by mask
        uint32_t goff[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int b = am + 4 * (i - 2);
            b = b < 0 ? b + NR : b;
            b = b >= NR ? b - NR : b;
            goff[i] = (uint32_t)b * ROWB;
        }
        auto row_ptr = [&](int dy, int plane) -> const char* {
            const int c = ((dy % 4) + 4) % 4;
            const int gi = (dy - c) / 4 + 2;  // group of the lanes with c + q < 4
            if (c == 0) return lds + (LC[0] + goff[gi] + (uint32_t)plane);
            return lds + (LC[c] + goff[gi] + (uint32_t)plane + (MK[c] & (goff[gi + 1] - goff[gi])));
        };
        uint32_t m5[16], p5[16], m4[16], p4[16], m2[16], p2[16], z1[16], z0[8];
        load16(m5, row_ptr(-5, 0));
        load16(p5, row_ptr(+5, 0));
        load16(m4, row_ptr(-4, 0));
        load16(p4, row_ptr(+4, 0));
        load16(m2, row_ptr(-2, PLANE));
        load16(p2, row_ptr(+2, PLANE));
        load16(z1, row_ptr(0, PLANE));
        {
            const char* rz = row_ptr(0, 0);
            const u32x4 z0a = lds_read_b128(rz + 16), z0b = lds_read_b128(rz + 32);
            z0[0] = z0a.x; z0[1] = z0a.y; z0[2] = z0a.z; z0[3] = z0a.w;
            z0[4] = z0b.x; z0[5] = z0b.y; z0[6] = z0b.z; z0[7] = z0b.w;
        }

        const int yy = y + 4 * wvu + q + dB;
        uint32_t out[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t P = response_pair_biased(m5, p5, m4, p4, m2, p2, z1, z0, k);
            if (CLAMP) out[k] = pk_sub_sat_u16(P, xmask[k]);  // max(r, 0), 0 in the frame columns (chess.hip)
            // raw: un-bias and frame-column mask in ONE v_pk_mad_i16: P * m - 8192 * m per half, m = 1 inside the frame columns and
            // 0 outside (-6 us of 595 against a subtraction and an AND; per-lane constants, there are registers to spare here)
            else out[k] = __builtin_bit_cast(uint32_t, (i16x2)(__builtin_bit_cast(i16x2, P) * __builtin_bit_cast(i16x2, xmask[k]) + __builtin_bit_cast(i16x2, xadd[k])));
        }
        if (!seg_interior) {
            const uint32_t rm = (yy >= kMargin && yy < h - kMargin) ? 0xffffffffu : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) out[k] &= rm;
        }

        if (HOT) {
            // responses are clamped here, so "> 15" is "any bit above bit 3" (chess.hip); a lane has two aligned 8-pixel groups
            const bool live = yy < ye_lane;
            const uint32_t any = (((out[0] | out[1]) | (out[2] | out[3])) | ((out[4] | out[5]) | (out[6] | out[7]))) & 0xfff0fff0u;
            if (__ballot(any != 0 && live) != 0ull) {
                uint32_t bits = 0;
#pragma unroll
                for (int k = 7; k >= 0; --k) {
                    uint32_t m;
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(out[k] & 0xfff0fff0u), "v"(0x00010001u));
                    bits = dot2_u32_u16(m, (1u << (2 * (k & 3))) | (2u << (2 * (k & 3) + 16)), k == 3 ? bits << 8 : bits);
                }
                // (pairs 7..4 were collected first and moved up by 8 in front of pair 3: bits 8..15 = the second group)
                if (!live) bits = 0;
                collect_hot(bits & 0xffu, yy, x0, hsink, t, frame);
                collect_hot(bits >> 8, yy, x0 + 8, hsink, t, frame);
            }
        }

        // ring first, results second (chess.hip)
        __builtin_amdgcn_s_setprio(2);  // (-10 us of 636 on the launch, A/B)
        stage_store(lds, so + chunk_col, so + halo_col, pre);
        __builtin_amdgcn_s_setprio(0);
        so += RB * ROWB;
        if (so >= (uint32_t)RING) so -= RING;
        if (yy < ye_lane && x0 < w) {
            const u32x4 va = {out[0], out[1], out[2], out[3]}, vb = {out[4], out[5], out[6], out[7]};
            const int soff = (int)((uint32_t)(y + 4 * wvu) * (uint32_t)w * 2u);
            raw_buffer_store_b128(va, resp_rsrc, st_resp_voff, soff, kAuxNT);
            raw_buffer_store_b128(vb, resp_rsrc, st_resp_voff + 16, soff, kAuxNT);
        }
#ifdef MRG_EXPERIMENT
        if (PYR && wvu >= 2) {
            // behind the math, when the window registers are dead (chess.hip); rows y .. y+15 sit complete in the ring
            __builtin_amdgcn_sched_barrier(0);
            const int a0 = am - 4 * wvu;  // ring slot of row y (the wave's own group is 4 * wave further; may be negative)
            auto goff_of = [&](int g) {   // aligned group g (0..3) of the iteration's sixteen rows
                int b = a0 + 4 * g;
                b = b < 0 ? b + NR : b;
                b = b >= NR ? b - NR : b;
                return (uint32_t)b * ROWB;
            };
            const int blk = wvu - 2;      // wave 2: rows y .. y+7, wave 3: rows y+8 .. y+15
            emit_pyramid_block(lds, goff_of(2 * blk), goff_of(2 * blk + 1), y + 8 * blk, strip_x, w, frame, lane, *po);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        am += RB;
        am = am >= NR ? am - NR : am;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    if (HOT) flush_hot(hsink, hotcnt, wvu, t, frame);
    clock_probe_end(probe, clkp, lb.clk);
}

// Frames whose width is a multiple of 16 only (the staging loads are whole 16-byte chunks inside or outside the frame).
template <bool CLAMP>
__global__ __launch_bounds__(256, 3) void chess_v16_kernel(LevelBatch lb, int frame0, int nsegs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    chess_v16_body<CLAMP, false>(lb, CompTables{}, frame0, nsegs, blockIdx.x, gridDim.x, lds);
}

#ifdef MRG_EXPERIMENT
// the plain response with the half strip's segments paired (chess_v16_body, PAIR): `nfull` ordinary workgroups first
template <bool CLAMP>
__global__ __launch_bounds__(256, 3) void chess_v16_pair_kernel(LevelBatch lb, int frame0, int nsegs, unsigned nfull) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    chess_v16_body<CLAMP, false, false, true>(lb, CompTables{}, frame0, nsegs, blockIdx.x, nfull, lds);
}

// clamp + hot list (the levels of a chain)
__global__ __launch_bounds__(256, 3) void chess_v16_hot_kernel(LevelBatch lb, CompTables t, int frame0, int nsegs) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    chess_v16_body<true, true>(lb, t, frame0, nsegs, blockIdx.x, gridDim.x, lds);
}

// level 0 of a chain: clamp + hot list + the level images 1..3
__global__ __launch_bounds__(256, 3) void chess_v16_pyr_kernel(LevelBatch lb, CompTables t, int nsegs, PyramidOut po) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    chess_v16_body<true, true, true>(lb, t, 0, nsegs, blockIdx.x, gridDim.x, lds, &po);
}

// Several pyramid levels of the same batch in ONE grid (chess.hip, chess_v1_multi_kernel): clamp + hot list.
constexpr int kMulti16Max = 4;
struct ChessMulti16 {
    LevelBatch lb[kMulti16Max];
    CompTables t[kMulti16Max];
    int first_wg[kMulti16Max];  // first workgroup of level slot k (a multiple of 8: the XCD-aware order counts from it)
    int nwg[kMulti16Max];
    int nsegs[kMulti16Max];
    int n;
};
__global__ __launch_bounds__(256, 3) void chess_v16_multi_kernel(ChessMulti16 a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int b = blockIdx.x;
    int k = 0;  // uniform
#pragma unroll
    for (int j = 1; j < kMulti16Max; ++j)
        if (j < a.n && b >= a.first_wg[j]) k = j;
    const int rel = b - a.first_wg[k];
    if (rel >= a.nwg[k]) return;  // padding between slots
    chess_v16_body<true, true>(a.lb[k], a.t[k], 0, a.nsegs[k], (unsigned)rel, (unsigned)a.nwg[k], lds);
}

#endif  // MRG_EXPERIMENT

bool chess16_ok(const LevelBatch& lb) {
    return lb.w >= 16 && lb.w % 16 == 0 && lb.h > 0 && ((uintptr_t)lb.img & 3) == 0 && lb.img_stride % 4 == 0 && lb.img_pitch % 4 == 0 && (long long)(lb.h + 64) * lb.img_stride < 0x7fffffffLL &&
           (long long)lb.w * lb.h * 2 < 0x7fffffffLL;
}

// Row segments per frame (balanced, whole 16-row granules: common.h, where the model and its fit are).  Before round 6:
// 1024 / 512 / 256 / 128 / 64 rows cut from the top (1080 rows = 8 x 128 + 56: 119.8 us per 64 frames; five balanced
// segments: 107.2 us).
// `seg_rows` > 0: the caller's segment height (option "chess16_seg").
static const SegModel kV16Model = {10.8, 0.34, 22.0, 32, 1024};
static const SegModel kV16HotModel = {10.8, 0.34, 22.0, 32, 512};  // (hot kernels: at most 512 rows -- records per wave)
static int pick_nsegs16(int w, int h, int nframes, int seg_rows, bool hot = false) {
    if (seg_rows > 0) return segments_for_rows(h, seg_rows, v16::RB);
    return pick_balanced_segments((w + v16::SW - 1) / v16::SW, h, nframes, v16::RB, hot ? kV16HotModel : kV16Model);
}

// enough workgroups for the three-per-CU kernel to be worth it (below that chess_v1's shorter segments fill the chip better)
bool chess16_pays(const LevelBatch& lb, int nframes) {
    const long long strips = (lb.w + v16::SW - 1) / v16::SW;
    return chess16_ok(lb) && strips * nframes * ((lb.h + 63) / 64) >= 256;
}

#ifdef MRG_EXPERIMENT
int chess16_pair = 0;  // tuning hook "chess16_pair": 1 = pair the half strip's segments where the shape allows it
#endif
void launch_chess16(const LevelBatch& lb, int frame0, int nframes, bool clamp, hipStream_t s, int seg_rows) {
    int nsegs = pick_nsegs16(lb.w, lb.h, nframes, seg_rows);
#ifdef MRG_EXPERIMENT
    if (chess16_pair && lb.w % v16::SW == v16::SW / 2 && lb.w > v16::SW) {
        nsegs += nsegs & 1;  // (an even number of segments)
        if (nsegs <= (lb.h + v16::RB - 1) / v16::RB) {
            const unsigned nfull = (unsigned)((lb.w / v16::SW) * nsegs * nframes);
            dim3 grid(nfull + (unsigned)((nsegs / 2) * nframes));
            const size_t lds = 2 * v16::PLANE;
            if (clamp) hipLaunchKernelGGL(chess_v16_pair_kernel<true>, grid, dim3(256), lds, s, lb, frame0, nsegs, nfull);
            else hipLaunchKernelGGL(chess_v16_pair_kernel<false>, grid, dim3(256), lds, s, lb, frame0, nsegs, nfull);
            return;
        }
    }
#endif
    dim3 grid(((lb.w + v16::SW - 1) / v16::SW) * nsegs * nframes);
    const size_t lds = 2 * v16::PLANE;
    if (clamp) hipLaunchKernelGGL(chess_v16_kernel<true>, grid, dim3(256), lds, s, lb, frame0, nsegs);
    else hipLaunchKernelGGL(chess_v16_kernel<false>, grid, dim3(256), lds, s, lb, frame0, nsegs);
}

#ifdef MRG_EXPERIMENT
// clamp + hot list of one level through chess_v16_hot_kernel
void launch_chess16_hot(const LevelBatch& lb, const CompTables& t, int frame0, int nframes, hipStream_t s) {
    const int nsegs = pick_nsegs16(lb.w, lb.h, nframes, 0, true);
    dim3 grid(((lb.w + v16::SW - 1) / v16::SW) * nsegs * nframes);
    const size_t lds = 2 * v16::PLANE + (V1_HOTBUF + 12) * sizeof(int);
    hipLaunchKernelGGL(chess_v16_hot_kernel, grid, dim3(256), lds, s, lb, t, frame0, nsegs);
}

// level 0 with the level images fused in (shapes: chess_pyramid_ok of chess.hip, whole 16 x 8 blocks)
bool launch_chess16_pyramid(const LevelBatch& lb, const CompTables& t, const PyramidOut& po, int nframes, hipStream_t s) {
    if (!chess16_ok(lb) || !chess_pyramid_ok(lb, nframes) || lb.h % 16 != 0) return false;
    const int nsegs = pick_nsegs16(lb.w, lb.h, nframes, 0, true);
    dim3 grid(((lb.w + v16::SW - 1) / v16::SW) * nsegs * nframes);
    const size_t lds = 2 * v16::PLANE + (V1_HOTBUF + 12) * sizeof(int);
    hipLaunchKernelGGL(chess_v16_pyr_kernel, grid, dim3(256), lds, s, lb, t, nsegs, po);
    return true;
}

// levels lbs[0 .. n) (largest first) of one batch in one launch; false when a shape does not qualify
bool chess16_multi_ok(const LevelBatch* lbs, int n, int nframes) {
    if (n < 2 || n > kMulti16Max || nframes <= 0) return false;
    for (int k = 0; k < n; ++k)
        if (!chess16_ok(lbs[k])) return false;
    return true;
}
bool launch_chess16_multi(const LevelBatch* lbs, const CompTables* ts, int n, int nframes, hipStream_t s) {
    if (!chess16_multi_ok(lbs, n, nframes) || ts[0].only) return false;  // (frame lists: the dense repeat stays on chess_v1)
    ChessMulti16 a;
    a.n = n;
    int total = 0;
    for (int k = 0; k < kMulti16Max; ++k) {
        const int j = k < n ? k : 0;
        a.lb[k] = lbs[j];
        a.t[k] = ts[j];
        a.nsegs[k] = pick_nsegs16(lbs[j].w, lbs[j].h, nframes, 0, true);
        a.first_wg[k] = total;
        a.nwg[k] = k < n ? ((lbs[j].w + v16::SW - 1) / v16::SW) * a.nsegs[k] * nframes : 0;
        total += (a.nwg[k] + 7) / 8 * 8;
    }
    const size_t lds = 2 * v16::PLANE + (V1_HOTBUF + 12) * sizeof(int);
    hipLaunchKernelGGL(chess_v16_multi_kernel, dim3(total), dim3(256), lds, s, a);
    return true;
}

#endif  // MRG_EXPERIMENT

}  // namespace mrg
