# generates dep.hip: kernels with a repeating op pattern and a fixed dependency distance
pats = {"A": "A", "P": "P", "AAP": "AAP", "AP": "AP", "AAAAPP": "AAAAPP", "AAAAAAAAPPPP": "AAAAAAAAPPPP"}
dists = [1, 2, 3, 4, 6, 8]
NI = 48
out = ['#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <stdint.h>']
names = []
for pn, pat in pats.items():
    for d in dists:
        name = f"k_{pn}_d{d}"
        names.append((name, pn, d))
        body = []
        for i in range(NI):
            op = "v_add_u32" if pat[i % len(pat)] == "A" else "v_pk_max_u16"
            body.append(f'asm volatile("{op} %0, %1, %2" : "=v"(r{i % 16}) : "v"(r{(i - d) % 16}), "v"(b));')
        regs = ", ".join(f"r{i} = seed * {2*i+3} + threadIdx.x" for i in range(16))
        xor = " ^ ".join(f"r{i}" for i in range(16))
        out.append(f"""__global__ __launch_bounds__(256) void {name}(uint32_t* out, uint32_t seed, int iters) {{
    uint32_t {regs}, b = seed ^ 0x00030005u;
    for (int it = 0; it < iters; ++it) {{
        {' '.join(body)}
    }}
    out[blockIdx.x * 256 + threadIdx.x] = {xor};
}}""")
out.append(f"""
template <typename K> void run(const char* name, K kern, int k) {{
    const int blocks = 256 * k * 4, iters = 1000;
    uint32_t* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * {NI};
    printf("%-22s wg/cu=%d %8.3f ms  %6.2f cyc/instr/SIMD@2.3GHz\\n", name, k, ms, ms * 1e-3 * 2.3e9 * 1024 / winstr);
    (void)hipFree(d);
}}
int main() {{
    for (int k : {{4, 2, 1}}) {{
""")
for name, pn, d in names:
    out.append(f'        run("{name}", {name}, k);')
out.append("    }\n    return 0;\n}")
open("dep.hip", "w").write("\n".join(out))
