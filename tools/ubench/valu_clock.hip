// Effective shader clock and per-wave issue cost under chip-wide VALU load.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define BODY8(ASMSTR)                                   \
    asm volatile(ASMSTR : "+v"(a0) : "v"(b), "v"(c));   \
    asm volatile(ASMSTR : "+v"(a1) : "v"(b), "v"(c));   \
    asm volatile(ASMSTR : "+v"(a2) : "v"(b), "v"(c));   \
    asm volatile(ASMSTR : "+v"(a3) : "v"(b), "v"(c));   \
    asm volatile(ASMSTR : "+v"(a4) : "v"(b), "v"(c));   \
    asm volatile(ASMSTR : "+v"(a5) : "v"(b), "v"(c));   \
    asm volatile(ASMSTR : "+v"(a6) : "v"(b), "v"(c));   \
    asm volatile(ASMSTR : "+v"(a7) : "v"(b), "v"(c));
#define DEF(NAME, A1, A2)                                                                           \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, long long* tm, uint32_t seed, int iters) { \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,     \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed ^ 0x00030005u, c = seed | 1;   \
        const long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();                                \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) { BODY8(A1) BODY8(A2) }                  \
        }                                                                                          \
        const long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();                                \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
        if ((threadIdx.x & 63) == 0) {                                                             \
            const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);                                    \
            tm[2 * wv] = t1 - t0; tm[2 * wv + 1] = w1 - w0;                                        \
        }                                                                                          \
    }
DEF(k_add, "v_add_u32 %0, %0, %1", "v_add_u32 %0, %0, %1")
DEF(k_pkmax, "v_pk_max_u16 %0, %0, %1", "v_pk_max_u16 %0, %0, %1")
DEF(k_mix, "v_add_u32 %0, %0, %1", "v_pk_max_u16 %0, %0, %1")
DEF(k_fma, "v_fma_f32 %0, %0, %1, %2", "v_fma_f32 %0, %0, %1, %2")
DEF(k_mov, "v_mov_b32 %0, %1", "v_mov_b32 %0, %1")

template <typename K> void run(const char* name, K kern, int wgs_per_cu, int ncu) {
    const int blocks = ncu * wgs_per_cu, iters = 2000;
    uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    long long* tm; hipMalloc(&tm, (size_t)blocks * 4 * 2 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, tm, 12345u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, tm, 12345u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)blocks * 8);
    hipMemcpy(h.data(), tm, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int i = 0; i < blocks * 4; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
    cyc /= blocks * 4; wall /= blocks * 4;
    const double ninstr = (double)iters * 128;
    printf("%-8s wg/cu=%d cus=%3d  %8.3f ms  cyc/instr/wave %6.2f  wall-ticks %9.0f (kernel %.0f ticks@100MHz)  clk %.0f MHz  chip %.1f T lane-ops/s\n",
           name, wgs_per_cu, ncu, ms, cyc / ninstr, wall, ms * 1e5, cyc / wall * 100.0, (double)blocks * 256 * ninstr / ms / 1e9);
    hipFree(d); hipFree(tm);
}
int main() {
    for (int ncu : {16, 256})
        for (int k : {1, 2, 4, 8}) {
            run("add", k_add, k, ncu);
            run("pkmax", k_pkmax, k, ncu);
            run("mix", k_mix, k, ncu);
            run("fma", k_fma, k, ncu);
            run("mov", k_mov, k, ncu);
        }
    return 0;
}
