// ChESS response, second decomposition (round 5 experiment; option "chess_variant" 16 of experiment builds):
// SIXTEEN pixels per lane.  Same algebra, same LDS format (two planes of packed u16 pixel pairs) and the same
// rolling strip as chess_v1 (chess.hip); what changes is the shape of a lane's work:
//
//   * a lane owns 16 adjacent pixels (8 pairs) of one row; 16 lanes cover the 256-pixel strip, the four quarter-waves
//     of a wave take four consecutive rows, the four waves 16 rows per iteration (v1: 8 pixels, 2 rows, 8 rows);
//   * the window of a (plane, row) is 16 dwords = 4 ds_read_b128 for 16 pixels (v1: 3 for 8): 30 reads and 120
//     returned dwords per 16 pixels instead of 44 and 176; window addresses, masks and the barrier are paid per
//     lane-iteration, i.e. half as often per pixel;
//   * rows are 592 bytes apart (576 + 16), so that consecutive rows sit an ODD number of 16-byte slots apart: a
//     ds_read_b128 is serviced in groups of 16 lanes that here span two rows of 8 lanes at a 32-byte stride, and the
//     odd pitch puts the second row on the slots the first leaves free (conflict-free; with 576 every read is 2-way);
//   * the ring has 44 rows (16 computed + 10 halo + 16 arriving, rounded to a multiple of 4): 52 KB, three
//     workgroups per CU, and the kernel may use 168 VGPRs.  Slots are NOT a power of two: a wave's four rows are an
//     aligned group of four for dy = 0, +-4 and straddle two groups otherwise -- the group offsets are wave-uniform
//     (SALU), the lane's choice between them a precomputed mask.
#include "common.h"
#include "kernels.h"

namespace mrg {

namespace v16 {

constexpr int SW = 256;                 // strip width, output pixels
constexpr int HL = 16;                  // left halo of the LDS window
constexpr int WIN = SW + 2 * HL;        // 288
constexpr int NCH = WIN / 16;           // 18 staging chunks per row
constexpr int ROWB = WIN * 2 + 16;      // 592 bytes per row and plane
constexpr int NR = 44;                  // ring rows
constexpr int RB = 16;                  // rows per iteration
constexpr int PLANE = NR * ROWB;        // 26048
constexpr int RING = NR * ROWB;         // bytes of one plane's ring (wrap length)

using u16x2 = unsigned short __attribute__((ext_vector_type(2)));
using i16x2 = short __attribute__((ext_vector_type(2)));
using i32x4 = int __attribute__((ext_vector_type(4)));
using u32x4 = uint32_t __attribute__((ext_vector_type(4)));

__device__ u32x4 raw_buffer_load_b128(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ unsigned char buffer_load_u8(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i8");
__device__ void raw_buffer_store_b128(u32x4 data, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.v4i32");
constexpr int kAuxNT = 2;
__device__ __forceinline__ i32x4 raw_rsrc(const void* base, uint32_t bytes) {
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, 0x00020000};
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

struct StageRegs {
    uint4 g;
    uint32_t next;
};

__device__ __forceinline__ StageRegs stage_load_clamped(const uint8_t* img, int stride, int h, int r, int gx_c, int nx_c) {
    StageRegs s;
    const int rc = min(max(r, 0), h - 1);
    const uint8_t* row = img + (long long)rc * stride;
    __builtin_memcpy(&s.g, row + gx_c, 16);
    s.next = row[nx_c];
    return s;
}

// 16 bytes -> the P0 and P1 entries of the chunk (as chess_v1's stage_store), at byte offset `off` of plane P0
__device__ __forceinline__ void stage_store(char* lds, uint32_t off, const StageRegs& s) {
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

}  // namespace v16

// Frames whose width is a multiple of 16 only (the staging loads are whole 16-byte chunks inside or outside the frame).
template <bool CLAMP>
__global__ __launch_bounds__(256, 3) void chess_v16_kernel(LevelBatch lb, int frame0, int seg) {
    using namespace v16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nstrips = (lb.w + SW - 1) / SW, nsegs = (lb.h + seg - 1) / seg;
    int work;
    {   // XCD-aware work order (chess.hip, chess_v1_body)
        const unsigned b = blockIdx.x, nwg = gridDim.x, xcd = b & 7u, j = b >> 3;
        const unsigned q = nwg >> 3, r = nwg & 7u;
        work = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j);
    }
    const int strip = work % nstrips, rest = work / nstrips;
    const int frame = frame0 + rest / nsegs;
    const int w = lb.w, h = lb.h, stride = lb.img_stride;
    const uint8_t* img = lb.img + (long long)frame * lb.img_pitch;
    int16_t* resp = lb.resp + (long long)frame * lb.resp_pitch;
    const int strip_x = strip * SW;
    const int ys = (rest % nsegs) * seg;
    const int ye = min(ys + seg, h);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, jx = lane & 15;

    // staging tasks: task t = row t / 18 of the group, chunk t % 18; thread tid takes task tid, wave 0 also tasks 256..287
    // (both of its half-waves the same 32: the branch is wave-uniform)
    const int t0 = tid, t1 = 256 + (lane & 31);
    const int r0 = t0 / NCH, c0 = t0 - r0 * NCH, r1 = t1 / NCH, c1 = t1 - r1 * NCH;
    auto gxc = [&](int ch) { return min(max(strip_x - HL + 16 * ch, 0), max(w - 16, 0)); };
    auto nxc = [&](int ch) { return min(max(strip_x - HL + 16 * ch + 16, 0), w - 1); };
    const int g0x = gxc(c0), n0x = nxc(c0), g1x = gxc(c1), n1x = nxc(c1);
    const i32x4 img_rsrc = raw_rsrc(img, (uint32_t)min((long long)(h - 1) * stride + w, 0xffffffffLL));
    const int v0g = r0 * stride + g0x, v0n = r0 * stride + n0x, v1g = r1 * stride + g1x, v1n = r1 * stride + n1x;
    // ring byte offset (plane P0) of the thread's task rows: row r of the frame lives in slot (r - ys) mod 44
    auto slot_off = [&](int rel) { return (uint32_t)(((rel % NR) + NR) % NR) * ROWB; };

    // prologue: rows ys - 11 .. ys + 20 (two groups of 16; the first six are never read)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int rel = -11 + RB * g;
        const StageRegs a = stage_load_clamped(img, stride, h, ys + rel + r0, g0x, n0x);
        stage_store(lds, slot_off(rel + r0) + c0 * 32, a);
        if (wvu == 0) {
            const StageRegs b = stage_load_clamped(img, stride, h, ys + rel + r1, g1x, n1x);
            stage_store(lds, slot_off(rel + r1) + c1 * 32, b);
        }
    }
    __syncthreads();

    // per-lane constants
    const int x0 = strip_x + 16 * jx;
    uint32_t xmask[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int xa = x0 + 2 * k, xb = xa + 1;
        const bool ina = xa >= kMargin && xa < w - kMargin, inb = xb >= kMargin && xb < w - kMargin;
        if (CLAMP) xmask[k] = (ina ? 0x2000u : 0xffffu) | (inb ? 0x20000000u : 0xffff0000u);
        else xmask[k] = (ina ? 0xffffu : 0u) | (inb ? 0xffff0000u : 0u);
    }
    const uint32_t lane_col = 16u + 32u * jx;  // D[-4] of the lane within a row
    // dy classes c = dy mod 4: the lane's row is row (c + q) & 3 of group B (c + q < 4) or of the group after it
    uint32_t LC[4], MK[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        LC[c] = lane_col + (uint32_t)((c + q) & 3) * ROWB;
        MK[c] = (c + q >= 4) ? 0xffffffffu : 0u;
    }
    const bool seg_interior = ys >= kMargin && ye <= h - kMargin;
    const i32x4 resp_rsrc = raw_rsrc(resp, (uint32_t)min((long long)w * h * 2, 0xffffffffLL));
    const int st_resp_voff = (q * w + x0) * 2;
    // running ring offsets of the staging tasks' rows (rows y + 21 + r of iteration y)
    uint32_t so0 = slot_off(21 + r0) + c0 * 32, so1 = slot_off(21 + r1) + c1 * 32;

    int am = 4 * wvu;  // (y - ys + 4 * wave) mod 44: the ring slot of the wave's first row, a multiple of 4
    for (int y = ys; y < ye; y += RB) {
        // prefetch rows y + 21 .. y + 36 (needed by the next iteration)
        StageRegs pre0, pre1;
        {
            const int rowoff = (y + 21) * stride;
            pre0.g = __builtin_bit_cast(uint4, raw_buffer_load_b128(img_rsrc, v0g + rowoff, 0, 0));
            pre0.next = buffer_load_u8(img_rsrc, v0n + rowoff, 0, 0);
            if (wvu == 0) {
                pre1.g = __builtin_bit_cast(uint4, raw_buffer_load_b128(img_rsrc, v1g + rowoff, 0, 0));
                pre1.next = buffer_load_u8(img_rsrc, v1n + rowoff, 0, 0);
            }
        }

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
        uint32_t m5[16], p5[16], m4[16], p4[16], m2[16], p2[16], z1[16];
        load16(m5, row_ptr(-5, 0));
        load16(p5, row_ptr(+5, 0));
        load16(m4, row_ptr(-4, 0));
        load16(p4, row_ptr(+4, 0));
        load16(m2, row_ptr(-2, PLANE));
        load16(p2, row_ptr(+2, PLANE));
        load16(z1, row_ptr(0, PLANE));
        const char* rz = row_ptr(0, 0);
        const u32x4 z0a = lds_read_b128(rz + 16), z0b = lds_read_b128(rz + 32);
        const uint32_t z0[8] = {z0a.x, z0a.y, z0a.z, z0a.w, z0b.x, z0b.y, z0b.z, z0b.w};

        const int yy = y + 4 * wvu + q;
        uint32_t out[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t P = response_pair_biased(m5, p5, m4, p4, m2, p2, z1, z0, k);
            if (CLAMP) out[k] = pk_sub_sat_u16(P, xmask[k]);
            else out[k] = pk_sub_i16(P, 0x20002000u) & xmask[k];
        }
        if (!seg_interior) {
            const uint32_t rm = (yy >= kMargin && yy < h - kMargin) ? 0xffffffffu : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) out[k] &= rm;
        }

        // ring first, results second (chess.hip)
        stage_store(lds, so0, pre0);
        so0 += RB * ROWB;
        if (so0 >= (uint32_t)RING) so0 -= RING;
        if (wvu == 0) {
            stage_store(lds, so1, pre1);
            so1 += RB * ROWB;
            if (so1 >= (uint32_t)RING) so1 -= RING;
        }
        if (yy < ye && x0 < w) {
            const u32x4 va = {out[0], out[1], out[2], out[3]}, vb = {out[4], out[5], out[6], out[7]};
            const int soff = (int)((uint32_t)(y + 4 * wvu) * (uint32_t)w * 2u);
            raw_buffer_store_b128(va, resp_rsrc, st_resp_voff, soff, kAuxNT);
            raw_buffer_store_b128(vb, resp_rsrc, st_resp_voff + 16, soff, kAuxNT);
        }
        am += RB;
        am = am >= NR ? am - NR : am;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
}

bool chess16_ok(const LevelBatch& lb) {
    return lb.w >= 16 && lb.w % 16 == 0 && lb.h > 0 && (long long)(lb.h + 64) * lb.img_stride < 0x7fffffffLL &&
           (long long)lb.w * lb.h * 2 < 0x7fffffffLL;
}

void launch_chess16(const LevelBatch& lb, int frame0, int nframes, bool clamp, hipStream_t s) {
    const int seg = 256;
    dim3 grid(((lb.w + v16::SW - 1) / v16::SW) * ((lb.h + seg - 1) / seg) * nframes);
    const size_t lds = 2 * v16::PLANE;
    if (clamp) hipLaunchKernelGGL(chess_v16_kernel<true>, grid, dim3(256), lds, s, lb, frame0, seg);
    else hipLaunchKernelGGL(chess_v16_kernel<false>, grid, dim3(256), lds, s, lb, frame0, seg);
}

}  // namespace mrg
