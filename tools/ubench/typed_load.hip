// Probe: buffer_load_format_d16_xyzw with an 8_8_8_8 UINT descriptor = 4 bytes -> two packed-u16 pair
// registers in the texture unit (what the ChESS staging does with four v_perm_b32 today).
// Checks (1) values at byte offsets 0..3 mod 4 (unaligned element addresses), (2) out-of-range reads
// return 0, (3) streaming rate next to a dwordx4 byte stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
using i32x4 = int __attribute__((ext_vector_type(4)));
using u16x4 = unsigned short __attribute__((ext_vector_type(4)));
__device__ u16x4 ld_fmt(i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4i16");
__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint32_t bytes) {
    const uint32_t w3 = (4u | (5u << 3) | (6u << 6) | (7u << 9)) | (4u << 12) | (10u << 15);
    const uint64_t a = (uint64_t)p;
    return i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)bytes, (int)w3};
}
__global__ void probe(const uint8_t* img, uint32_t bytes, int shift, uint2* out) {
    const i32x4 r = make_rsrc(img, bytes);
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    const u16x4 v = ld_fmt(r, t * 4 + shift, 0, 0);
    out[t] = make_uint2(v.x | ((uint32_t)v.y << 16), v.z | ((uint32_t)v.w << 16));
}
__global__ void stream_fmt(const uint8_t* img, uint32_t bytes, uint32_t* sink, int iters) {
    const i32x4 r = make_rsrc(img, bytes);
    uint32_t acc = 0;
    long long off = (long long)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (int i = 0; i < iters; ++i, off += stride) {
        const u16x4 v = ld_fmt(r, (int)off, 0, 0);
        const u16x4 w = ld_fmt(r, (int)off + 1, 0, 0);
        acc += v.x + v.y + v.z + v.w + w.x + w.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void stream_x4(const uint8_t* img, uint32_t bytes, uint32_t* sink, int iters) {
    uint32_t acc = 0;
    long long off = (long long)(blockIdx.x * blockDim.x + threadIdx.x) * 16;
    const long long stride = (long long)gridDim.x * blockDim.x * 16;
    for (int i = 0; i < iters; ++i, off += stride) {
        const uint4 v = *reinterpret_cast<const uint4*>(img + off);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
int main() {
    const uint32_t N = 1u << 30;
    uint8_t* d; hipMalloc(&d, N);
    std::vector<uint8_t> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)((i * 37 + (i >> 8)) & 255);
    hipMemset(d, 7, N);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    uint2* out; hipMalloc(&out, 4096 * 8);
    std::vector<uint2> ho(4096);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(probe, dim3(16), dim3(256), 0, 0, d, N, shift, out);
        hipMemcpy(ho.data(), out, 4096 * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 4096; ++t) {
            const uint8_t* p = h.data() + t * 4 + shift;
            const uint32_t e0 = p[0] | (p[1] << 16), e1 = p[2] | (p[3] << 16);
            if (ho[t].x != e0 || ho[t].y != e1) { if (bad < 3) printf("  shift %d t %d got %08x %08x want %08x %08x\n", shift, t, ho[t].x, ho[t].y, e0, e1); ++bad; }
        }
        printf("typed load, byte shift %d: %s (%d bad of 4096)\n", shift, bad ? "MISMATCH" : "ok", bad);
    }
    // out of range: descriptor of 1000 bytes, read elements around the end and at "negative" offsets
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, 1000u, 0, out);
    hipMemcpy(ho.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    printf("range check (num_records 1000): t=249 %08x %08x  t=250 %08x %08x  t=251 %08x %08x\n", ho[249].x, ho[249].y, ho[250].x, ho[250].y, ho[251].x, ho[251].y);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d + 4096, 1000u, -8, out);
    hipMemcpy(ho.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    printf("negative offset (-8, -4, 0): %08x %08x | %08x %08x | %08x %08x\n", ho[0].x, ho[0].y, ho[1].x, ho[1].y, ho[2].x, ho[2].y);
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, 1001u, 2, out);   // element straddling the end
    hipMemcpy(ho.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    printf("straddling the end (records 1001, shift 2): t=249 %08x %08x (bytes 998..1001)\n", ho[249].x, ho[249].y);
    // streaming rates
    uint32_t* sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        const int blocks = 4096, thr = 256;
        int it = (int)((N / 2) / ((size_t)blocks * thr * 4));
        hipEventRecord(e0); hipLaunchKernelGGL(stream_fmt, dim3(blocks), dim3(thr), 0, 0, d, N, sink, it); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("typed d16 xyzw, two loads (+0,+1) per 4 px: %.1f GB/s of image bytes (%.3f ms)\n", (N / 2) / ms / 1e6, ms);
        it = (int)((N / 2) / ((size_t)blocks * thr * 16));
        hipEventRecord(e0); hipLaunchKernelGGL(stream_x4, dim3(blocks), dim3(thr), 0, 0, d, N, sink, it); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("global_load_dwordx4: %.1f GB/s (%.3f ms)\n", (N / 2) / ms / 1e6, ms);
    }
    return 0;
}
