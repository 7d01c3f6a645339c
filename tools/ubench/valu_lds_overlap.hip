// Does LDS read traffic (data returning into the VGPR file) slow a VALU-bound wave down?
// Every wave runs ITER iterations of: NLDS wide LDS reads (independent, conflict-free, results never used),
// then a ~200-op VALU block in the ChESS mix (half packed 16-bit max, half 32-bit adds), then s_waitcnt lgkmcnt(0).
// 4 workgroups of 256 threads per CU (40 KB of LDS each), every CU busy: the occupancy of the production kernel.
// Output: shader cycles per wave-iteration per SIMD (s_memtime), for NLDS = 0 / 11 / 22 and read widths 128 / 64 / 32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
using u32x2 = uint32_t __attribute__((ext_vector_type(2)));

#define A(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define P(i) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define BLOCK24 A(0) P(1) A(2) P(3) A(4) P(5) A(6) P(7) P(0) A(1) P(2) A(3) P(4) A(5) P(6) A(7) A(0) P(1) A(2) P(3) A(4) P(5) A(6) P(7)

template <int NLDS, int WIDTH, bool WAIT_FIRST>
__global__ __launch_bounds__(256, 4) void k(uint32_t* out, unsigned long long* cyc, uint32_t seed, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 9216; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = i * seed;
    __syncthreads();
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17,
             a7 = a0 * 19, b = seed ^ 0x00030005u;
    const uint32_t addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    u32x4 r[NLDS > 0 ? NLDS : 1];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NLDS; ++j) {
            if (WIDTH == 128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[j]) : "v"(addr), "n"(j * 1536));
            if (WIDTH == 64) {
                u32x2 t;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(j * 1536));
                r[j].x = t.x; r[j].y = t.y;
            }
            if (WIDTH == 32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[j].x) : "v"(addr), "n"(j * 1536));
        }
        if (WAIT_FIRST) asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int q = 0; q < 8; ++q) { BLOCK24 }
        BLOCK24  // 9 x 24 = 216 ops
        if (!WAIT_FIRST) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
#pragma unroll
    for (int j = 0; j < NLDS; ++j) x ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <typename K> void run(const char* name, K kern) {
    const int blocks = 1024 * 2, iters = 400;  // two rounds of resident workgroups
    uint32_t* d; unsigned long long* c;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipMalloc(&c, (size_t)blocks * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 40000, 0, d, c, 12345u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 40000, 0, d, c, 12345u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    // a wave shares its SIMD with 3 others: cycles per wave-iteration per SIMD = wave cycles / iters / 4
    printf("%-28s %8.3f ms   %7.1f cyc per wave-iteration per SIMD (s_memtime)   %7.1f (events @2.3 GHz)\n", name, ms,
           s / h.size() / iters / 4, ms * 1e-3 * 2.3e9 / (2.0 * 4 * iters));
    hipFree(d); hipFree(c);
}
int main() {
#define R(...) run(#__VA_ARGS__, k<__VA_ARGS__>)
    R(0, 128, false);
    R(11, 128, false); R(22, 128, false); R(22, 128, true);
    R(11, 64, false);  R(22, 64, false);
    R(11, 32, false);  R(22, 32, false);
    R(0, 128, false);
    return 0;
}
