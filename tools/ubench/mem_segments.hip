// Write-pattern study 2: segment height and skew vs achieved write bandwidth (strip 256 px, 8 rows / iteration)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
constexpr int W = 4096, H = 3072, NF = 64;
template <int MODE>
__global__ __launch_bounds__(256) void k_w(int16_t* out, const uint8_t* in, int seg, int nsegs) {
    constexpr int SW = 256, LPR = 32;
    const int nstrips = W / SW;
    const unsigned b = blockIdx.x, nwg = gridDim.x, xcd = b & 7u, j = b >> 3, q = nwg >> 3, r = nwg & 7u;
    const int work = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j);
    const int strip = work % nstrips, rest = work / nstrips, frame = rest / nsegs, sg = rest % nsegs;
    const int ys = sg * seg, ye = min(ys + seg, H);
    const int row = threadIdx.x / LPR, lx = threadIdx.x % LPR;
    int16_t* base = out + (long long)frame * W * H + strip * SW + lx * 8;
    const uint8_t* ibase = in + (long long)frame * W * H + strip * SW + lx * 8;
    uint4 v = make_uint4(blockIdx.x, 2, 3, 4);
    for (int y = ys; y < ye; y += 8) {
        if (MODE == 1) {   // paced by a read of the same rows
            const uint2 g = *reinterpret_cast<const uint2*>(ibase + (long long)(y + row) * W);
            v.x = g.x; v.y = g.y;
        }
        if (y + row < ye) *reinterpret_cast<uint4*>(base + (long long)(y + row) * W) = v;
        v.z += 1;
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    int16_t* out; uint8_t* in;
    const long long npx = (long long)W * H * NF;
    (void)hipMalloc(&out, npx * 2); (void)hipMalloc(&in, npx);
    (void)hipMemset(in, 7, npx);
    for (int seg : {32, 64, 96, 128, 136, 160, 192, 200, 248, 256, 264, 328, 384, 512, 520, 1024, 3072}) {
        const int nsegs = (H + seg - 1) / seg;
        const int grid = (W / 256) * nsegs * NF;
        float a = timeit([&] { hipLaunchKernelGGL(k_w<0>, dim3(grid), dim3(256), 0, 0, out, in, seg, nsegs); });
        float b = timeit([&] { hipLaunchKernelGGL(k_w<1>, dim3(grid), dim3(256), 0, 0, out, in, seg, nsegs); });
        printf("seg %4d (%3d segs, grid %6d)  write-only %7.1f us %5.0f GB/s   read+write %7.1f us %5.0f GB/s(3B/px)\n", seg, nsegs, grid,
               a * 1e3, npx * 2.0 / a / 1e6, b * 1e3, npx * 3.0 / b / 1e6);
    }
    return 0;
}
