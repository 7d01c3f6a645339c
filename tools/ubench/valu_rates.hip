// VALU issue-rate microbenchmark (inline asm so nothing folds). 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define DEF(NAME, ASMSTR)                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed, int iters) {         \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,     \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed ^ 0x00030005u, c = seed | 1;   \
        for (int it = 0; it < iters; ++it) {                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                       \
                asm volatile(ASMSTR : "+v"(a0) : "v"(b), "v"(c));                                  \
                asm volatile(ASMSTR : "+v"(a1) : "v"(b), "v"(c));                                  \
                asm volatile(ASMSTR : "+v"(a2) : "v"(b), "v"(c));                                  \
                asm volatile(ASMSTR : "+v"(a3) : "v"(b), "v"(c));                                  \
                asm volatile(ASMSTR : "+v"(a4) : "v"(b), "v"(c));                                  \
                asm volatile(ASMSTR : "+v"(a5) : "v"(b), "v"(c));                                  \
                asm volatile(ASMSTR : "+v"(a6) : "v"(b), "v"(c));                                  \
                asm volatile(ASMSTR : "+v"(a7) : "v"(b), "v"(c));                                  \
            }                                                                                      \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
    }
DEF(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEF(k_add, "v_add_u32 %0, %0, %1")
DEF(k_and, "v_and_b32 %0, %0, %1")
DEF(k_maxu32, "v_max_u32 %0, %0, %1")
DEF(k_pkmax, "v_pk_max_u16 %0, %0, %1")
DEF(k_pkadd, "v_pk_add_u16 %0, %0, %1")
DEF(k_pksub, "v_pk_sub_i16 %0, %0, %1")
DEF(k_pkmad, "v_pk_mad_u16 %0, %0, %1, %2")
DEF(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF(k_mul24, "v_mul_u32_u24 %0, %0, %1")
DEF(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEF(k_sadu8, "v_sad_u8 %0, %0, %1, %2")
DEF(k_sadu16, "v_sad_u16 %0, %0, %1, %2")
DEF(k_sadu32, "v_sad_u32 %0, %0, %1, %2")
DEF(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")
DEF(k_maxf32, "v_max_f32 %0, %0, %1")
DEF(k_addf32, "v_add_f32 %0, %0, %1")
DEF(k_fmaf32, "v_fma_f32 %0, %0, %1, %2")
DEF(k_pkmaxf16, "v_pk_max_f16 %0, %0, %1")
DEF(k_pkaddf16, "v_pk_add_f16 %0, %0, %1")
DEF(k_max3u32, "v_max3_u32 %0, %0, %1, %2")
DEF(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
DEF(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
DEF(k_lshlor, "v_lshl_or_b32 %0, %0, 8, %1")
DEF(k_mov, "v_mov_b32 %0, %1")
DEF(k_or, "v_or_b32 %0, %0, %1")
DEF(k_xor, "v_xor_b32 %0, %0, %1")
DEF(k_sub, "v_sub_u32 %0, %0, %1")
DEF(k_lshl, "v_lshlrev_b32 %0, 1, %0")
DEF(k_lshr, "v_lshrrev_b32 %0, 1, %0")
DEF(k_ashr, "v_ashrrev_i32 %0, 1, %0")
DEF(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF(k_andor, "v_and_or_b32 %0, %0, %1, %2")
DEF(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
DEF(k_or3, "v_or3_b32 %0, %0, %1, %2")
DEF(k_addlshl, "v_add_lshl_u32 %0, %0, %1, 1")
DEF(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1")
DEF(k_minu16pk, "v_pk_min_u16 %0, %0, %1")
DEF(k_mulf32, "v_mul_f32 %0, %0, %1")
DEF(k_subf32, "v_sub_f32 %0, %0, %1")
DEF(k_maxi16, "v_max_i16 %0, %0, %1")
DEF(k_addu16, "v_add_u16 %0, %0, %1")
DEF(k_movdpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF(k_adddpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF(k_movdppw, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
DEF(k_pksubclamp, "v_pk_sub_u16 %0, %0, %1 clamp")
DEF(k_pkmaxi16, "v_pk_max_i16 %0, %0, %1")
DEF(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_2")
DEF(k_maxi16_sdwa, "v_max_i16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1")
DEF(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
DEF(k_lshl2, "v_lshlrev_b32 %0, %1, %0")
DEF(k_maxu16, "v_max_u16 %0, %0, %1")

template <typename K> void run(const char* name, K kern) {
    uint32_t* d; hipMalloc(&d, 256 * 4096 * 4);
    const int iters = 1000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 128;
    printf("%-16s %8.3f ms  %7.2f T lane-ops/s  %6.1f lanes/clk/CU@2.4GHz\n", name, ms, ops / ms / 1e9, ops / ms * 1e3 / 256 / 2.4e9);
    hipFree(d);
}
int main() {
#define R(x) run(#x, x)
    R(k_add3); R(k_add); R(k_and); R(k_maxu32); R(k_pkmax); R(k_pkadd); R(k_pksub); R(k_pkmad); R(k_perm); R(k_mul24); R(k_mad24);
    R(k_sadu8); R(k_sadu16); R(k_sadu32); R(k_dot4); R(k_maxf32); R(k_addf32); R(k_fmaf32); R(k_pkmaxf16); R(k_pkaddf16); R(k_max3u32);
    R(k_or); R(k_xor); R(k_sub); R(k_lshl); R(k_lshr); R(k_ashr); R(k_cndmask); R(k_andor); R(k_bfi); R(k_or3); R(k_addlshl); R(k_lshladd); R(k_minu16pk); R(k_mulf32); R(k_subf32); R(k_maxi16); R(k_addu16);
    R(k_movdpp); R(k_adddpp); R(k_movdppw); R(k_pksubclamp); R(k_pkmaxi16); R(k_sub_sdwa); R(k_maxi16_sdwa); R(k_bitop3); R(k_lshl2); R(k_maxu16);
    R(k_alignbit); R(k_bfe); R(k_lshlor); R(k_mov);
    return 0;
}
