// Memory-pattern microbenchmark: how fast can 64 x 4096x3072 int16 be written (and u8 read) with
// the strip-rolling pattern of the ChESS kernel vs a linear sweep?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int W = 4096, H = 3072, NF = 64;

// linear: every workgroup writes contiguous 4 KB chunks
__global__ __launch_bounds__(256) void k_lin_w(int16_t* out, long long n16) {
    uint4 v = make_uint4(blockIdx.x, 2, 3, 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256)
        reinterpret_cast<uint4*>(out)[i] = v;
}
// strip pattern: workgroup = strip SWx(seg rows); per iteration RB rows; lane writes 16 B (8 px)
template <int SW, int RB>
__global__ __launch_bounds__(256) void k_strip_w(int16_t* out, int seg) {
    constexpr int LPR = SW / 8;             // lanes per row
    static_assert(LPR * RB == 256, "");
    const int nstrips = W / SW, nsegs = H / seg;
    const int work = blockIdx.x;
    const int strip = work % nstrips, rest = work / nstrips, frame = rest / nsegs, ys = (rest % nsegs) * seg;
    const int row = threadIdx.x / LPR, lx = threadIdx.x % LPR;
    int16_t* base = out + (long long)frame * W * H + strip * SW + lx * 8;
    uint4 v = make_uint4(blockIdx.x, 2, 3, 4);
    for (int y = ys; y < ys + seg; y += RB) {
        *reinterpret_cast<uint4*>(base + (long long)(y + row) * W) = v;
        v.x += 1;
    }
}
// same with the XCD remap of the real kernel
template <int SW, int RB>
__global__ __launch_bounds__(256) void k_strip_w_xcd(int16_t* out, int seg) {
    constexpr int LPR = SW / 8;
    const int nstrips = W / SW, nsegs = H / seg;
    const unsigned b = blockIdx.x, nwg = gridDim.x, xcd = b & 7u, j = b >> 3, q = nwg >> 3;
    const int work = (int)(xcd * q + j);
    const int strip = work % nstrips, rest = work / nstrips, frame = rest / nsegs, ys = (rest % nsegs) * seg;
    const int row = threadIdx.x / LPR, lx = threadIdx.x % LPR;
    int16_t* base = out + (long long)frame * W * H + strip * SW + lx * 8;
    uint4 v = make_uint4(blockIdx.x, 2, 3, 4);
    for (int y = ys; y < ys + seg; y += RB) {
        *reinterpret_cast<uint4*>(base + (long long)(y + row) * W) = v;
        v.x += 1;
    }
}
// read side: strip rows of u8, 16 B per lane, SW/16 lanes per row, 8 rows per iteration
template <int SW>
__global__ __launch_bounds__(256) void k_strip_r(const uint8_t* in, uint32_t* sink, int seg) {
    constexpr int LPR = SW / 16;
    const int nstrips = W / SW, nsegs = H / seg;
    const unsigned b = blockIdx.x, nwg = gridDim.x, xcd = b & 7u, j = b >> 3, q = nwg >> 3;
    const int work = (int)(xcd * q + j);
    const int strip = work % nstrips, rest = work / nstrips, frame = rest / nsegs, ys = (rest % nsegs) * seg;
    const int row = threadIdx.x / LPR, lx = threadIdx.x % LPR;
    uint32_t acc = 0;
    if (row < 8) {
        const uint8_t* base = in + (long long)frame * W * H + strip * SW + lx * 16;
        for (int y = ys; y < ys + seg; y += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(base + (long long)(y + row) * W);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}
// read + write together (the real kernel's traffic without LDS/math)
template <int SW, int RB>
__global__ __launch_bounds__(256) void k_strip_rw(const uint8_t* in, int16_t* out, int seg) {
    constexpr int LPR = SW / 8;
    const int nstrips = W / SW, nsegs = H / seg;
    const unsigned b = blockIdx.x, nwg = gridDim.x, xcd = b & 7u, j = b >> 3, q = nwg >> 3;
    const int work = (int)(xcd * q + j);
    const int strip = work % nstrips, rest = work / nstrips, frame = rest / nsegs, ys = (rest % nsegs) * seg;
    const int row = threadIdx.x / LPR, lx = threadIdx.x % LPR;
    const uint8_t* ibase = in + (long long)frame * W * H + strip * SW + lx * 8;
    int16_t* base = out + (long long)frame * W * H + strip * SW + lx * 8;
    for (int y = ys; y < ys + seg; y += RB) {
        const uint2 g = *reinterpret_cast<const uint2*>(ibase + (long long)(y + row) * W);
        uint4 v;
        v.x = __builtin_amdgcn_perm(g.x, g.x, 0x0c010c00u); v.y = __builtin_amdgcn_perm(g.x, g.x, 0x0c030c02u);
        v.z = __builtin_amdgcn_perm(g.y, g.y, 0x0c010c00u); v.w = __builtin_amdgcn_perm(g.y, g.y, 0x0c030c02u);
        *reinterpret_cast<uint4*>(base + (long long)(y + row) * W) = v;
    }
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    int16_t* out; uint8_t* in; uint32_t* sink;
    const long long npx = (long long)W * H * NF;
    (void)hipMalloc(&out, npx * 2); (void)hipMalloc(&in, npx); (void)hipMalloc(&sink, 4096);
    (void)hipMemset(in, 7, npx);
    const double wb = npx * 2.0, rb = npx * 1.0;
    float ms;
    for (int g : {2048, 8192, 32768}) {
        ms = timeit([&] { hipLaunchKernelGGL(k_lin_w, dim3(g), dim3(256), 0, 0, out, npx / 8); });
        printf("linear write grid %5d          %7.1f us  %6.0f GB/s\n", g, ms * 1e3, wb / ms / 1e6);
    }
    for (int seg : {128, 256, 512}) {
#define RUNW(K, SW, RB, name) ms = timeit([&] { hipLaunchKernelGGL((K<SW, RB>), dim3((W / SW) * (H / seg) * NF), dim3(256), 0, 0, out, seg); }); \
        printf("%-22s SW=%4d RB=%2d seg=%3d  %7.1f us  %6.0f GB/s\n", name, SW, RB, seg, ms * 1e3, wb / ms / 1e6);
        RUNW(k_strip_w, 256, 8, "strip write");
        RUNW(k_strip_w_xcd, 256, 8, "strip write xcd");
        RUNW(k_strip_w_xcd, 512, 4, "strip write xcd");
        RUNW(k_strip_w_xcd, 1024, 2, "strip write xcd");
        RUNW(k_strip_w_xcd, 2048, 1, "strip write xcd");
        ms = timeit([&] { hipLaunchKernelGGL((k_strip_r<256>), dim3((W / 256) * (H / seg) * NF), dim3(256), 0, 0, in, sink, seg); });
        printf("strip read  SW=256 seg=%3d                 %7.1f us  %6.0f GB/s\n", seg, ms * 1e3, rb / ms / 1e6);
#define RUNRW(SW, RB) ms = timeit([&] { hipLaunchKernelGGL((k_strip_rw<SW, RB>), dim3((W / SW) * (H / seg) * NF), dim3(256), 0, 0, in, out, seg); }); \
        printf("strip read+write       SW=%4d RB=%2d seg=%3d  %7.1f us  %6.0f GB/s (3 B/px)\n", SW, RB, seg, ms * 1e3, (wb + rb) / ms / 1e6);
        RUNRW(256, 8); RUNRW(512, 4); RUNRW(1024, 2); RUNRW(2048, 1);
    }
    return 0;
}
