// How the order of full-rate (v_add_u32) and half-rate (v_pk_max_u16) VALU ops changes throughput.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define A(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define P(i) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define A2(i, j) asm volatile("v_add_u32 %0, %1, %2" : "=v"(a##i) : "v"(a##j), "v"(a##i));
#define P2(i, j) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(a##i) : "v"(a##j), "v"(a##i));
#define DEF(NAME, N, SEQ)                                                                           \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed, int iters) {         \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,     \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed ^ 0x00030005u;                 \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) { SEQ }                                  \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
    }                                                                                              \
    static const int NAME##_n = N;
// 24 instructions per SEQ: 16 A + 8 P (2:1, like the ChESS inner loop)
DEF(k_grouped, 24, A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7))
DEF(k_aap, 24, A(0) A(1) P(2) A(3) A(4) P(5) A(6) A(7) P(0) A(1) A(2) P(3) A(4) A(5) P(6) A(7) A(0) P(1) A(2) A(3) P(4) A(5) A(6) P(7))
DEF(k_aaaapp, 24, A(0) A(1) A(2) A(3) P(4) P(5) A(6) A(7) A(0) A(1) P(2) P(3) A(4) A(5) A(6) A(7) P(0) P(1) A(2) A(3) A(4) A(5) P(6) P(7))
DEF(k_allA, 24, A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7) A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7))
DEF(k_allP, 24, P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7) P(0) P(1) P(2) P(3) P(4) P(5) P(6) P(7))
// dependent: each instruction reads the result of the previous one
DEF(k_depA, 24, A2(1,0) A2(2,1) A2(3,2) A2(4,3) A2(5,4) A2(6,5) A2(7,6) A2(0,7) A2(1,0) A2(2,1) A2(3,2) A2(4,3) A2(5,4) A2(6,5) A2(7,6) A2(0,7) A2(1,0) A2(2,1) A2(3,2) A2(4,3) A2(5,4) A2(6,5) A2(7,6) A2(0,7))
DEF(k_depP, 24, P2(1,0) P2(2,1) P2(3,2) P2(4,3) P2(5,4) P2(6,5) P2(7,6) P2(0,7) P2(1,0) P2(2,1) P2(3,2) P2(4,3) P2(5,4) P2(6,5) P2(7,6) P2(0,7) P2(1,0) P2(2,1) P2(3,2) P2(4,3) P2(5,4) P2(6,5) P2(7,6) P2(0,7))
DEF(k_depmix, 24, A2(1,0) A2(2,1) P2(3,2) A2(4,3) A2(5,4) P2(6,5) A2(7,6) A2(0,7) P2(1,0) A2(2,1) A2(3,2) P2(4,3) A2(5,4) A2(6,5) P2(7,6) A2(0,7) A2(1,0) P2(2,1) A2(3,2) A2(4,3) P2(5,4) A2(6,5) A2(7,6) P2(0,7))

template <typename K> void run(const char* name, K kern, int n, int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu * 4, iters = 1000;   // 4 rounds of resident workgroups
    uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 4 * n;
    // cycles per wave-instruction per SIMD at 2.3 GHz
    printf("%-10s %8.3f ms  %6.2f cyc/instr/SIMD@2.3GHz   ideal(2/4): %.2f\n", name, ms,
           ms * 1e-3 * 2.3e9 * 1024 / winstr, 0.0);
    hipFree(d);
}
int main() {
#define R(x) run(#x, x, x##_n, k)
    for (int k : {4, 8}) { printf("wg/cu=%d\n", k);
        R(k_allA); R(k_allP); R(k_grouped); R(k_aap); R(k_aaaapp); R(k_depA); R(k_depP); R(k_depmix);
    }
    return 0;
}
