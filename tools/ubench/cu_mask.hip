// Probe: which CUs (XCC id, SE, CU) a stream created with hipExtStreamCreateWithCUMask runs on, for a
// few mask shapes, and what a masked latency-bound kernel costs a chip-filling kernel next to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <set>
#include <vector>
__global__ void whereami(uint32_t* out, int spin) {
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int nb = 2048;
    uint32_t* d; hipMalloc(&d, nb * 8);
    hipLaunchKernelGGL(whereami, dim3(nb), dim3(64), 0, s, d, 20000);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(nb * 2);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;
    for (int b = 0; b < nb; ++b) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    int total = 0;
    printf("%s:", name);
    for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  -> %d distinct CUs\n", total);
    hipFree(d);
    hipStreamDestroy(s);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    run("all 256", std::vector<uint32_t>(8, 0xffffffffu));
    run("bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0..15", {0xffffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("word 7", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu});
    run("bit 0 of each word", std::vector<uint32_t>(8, 1u));
    run("every 32nd.. bits 0,32,..", {1u, 1u, 1u, 1u, 1u, 1u, 1u, 1u});
    run("every 8th bit", std::vector<uint32_t>(8, 0x01010101u));
    return 0;
}
