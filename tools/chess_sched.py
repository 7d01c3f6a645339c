#!/usr/bin/env python3
"""Instruction-order generator for the ChESS inner math (tools/, not shipped in the library).

The level-0 ChESS kernel is VALU-issue bound, and on gfx950 the cost of a VALU instruction in a
stream that mixes full-rate ops (v_add_u32 ...) with half-rate ops (v_pk_max_u16, v_mul_u32_u24,
v_perm_b32) depends on the order of the stream (scratch/ub/dep.hip: 3.05 .. 4.0 cycles per
instruction for the same 2:1 mix).  hipcc's scheduler does not model this, so the order is chosen
here: the math of the four pixel pairs a lane produces is a DAG of ~200 ops; this script list-
schedules it under a policy and emits the ops as one `asm volatile` statement each (hipcc keeps
volatile asms in program order and still does the register allocation and the s_waitcnt
insertion).

  chess_sched.py bench  out.hip     -> a standalone microbenchmark with many candidate orders
  chess_sched.py emit POLICY out.inc -> the include file used by chess.hip
"""
import itertools
import sys

FAST, SLOW = "A", "P"


class Op:
    __slots__ = ("dst", "kind", "asm", "srcs", "pair", "idx")

    def __init__(self, dst, kind, asm, srcs, pair):
        self.dst, self.kind, self.asm, self.srcs, self.pair = dst, kind, asm, srcs, pair


def build_dag(clamp=True, add3=False):
    """Ops of the 4 pixel pairs; inputs are the C arrays m5,p5,m4,p4,m2,p2,z1 (12 dwords) and z0 (4)."""
    ops = []

    def add(dst, kind, asm, srcs, k):
        ops.append(Op(dst, kind, asm, srcs, k))
        return dst

    for k in range(4):
        c = 4 + k
        s = lambda arr, i: f"{arr}[{i}]"
        a = [s("m5", c + 1), s("m5", c), s("m5", c - 1), s("m4", c - 2)]
        cc = [s("p5", c - 1), s("p5", c), s("p5", c + 1), s("p4", c + 2)]
        b = [s("m2", c - 3), s("z1", c - 3), s("p2", c - 3), s("p4", c - 2)]
        d = [s("p2", c + 2), s("z1", c + 2), s("m2", c + 2), s("m4", c + 2)]
        ADD = "v_add_u32 %0, %1, %2"
        SUB = "v_sub_u32 %0, %1, %2"
        PMAX = "v_pk_max_u16 %0, %1, %2"
        PMIN = "v_pk_min_u16 %0, %1, %2"
        t1 = [add(f"t1_{i}_{k}", FAST, ADD, [a[i], cc[i]], k) for i in range(4)]
        t2 = [add(f"t2_{i}_{k}", FAST, ADD, [b[i], d[i]], k) for i in range(4)]
        mx = [add(f"mx_{i}_{k}", SLOW, PMAX, [a[i], cc[i]], k) for i in range(4)]
        my = [add(f"my_{i}_{k}", SLOW, PMAX, [b[i], d[i]], k) for i in range(4)]
        yy = [add(f"yy_{i}_{k}", SLOW, PMAX, [t1[i], t2[i]], k) for i in range(4)]

        def tree(prefix, leaves):
            lvl, n = list(leaves), 0
            while add3 and len(lvl) > 2:
                nxt = []
                j = 0
                while j + 2 < len(lvl):
                    nxt.append(add(f"{prefix}{n}_{k}", SLOW, "v_add3_u32 %0, %1, %2, %3", [lvl[j], lvl[j + 1], lvl[j + 2]], k))
                    n += 1
                    j += 3
                nxt.extend(lvl[j:])
                lvl = nxt
            while len(lvl) > 1:
                nxt = []
                for j in range(0, len(lvl) - 1, 2):
                    nxt.append(add(f"{prefix}{n}_{k}", FAST, ADD, [lvl[j], lvl[j + 1]], k))
                    n += 1
                if len(lvl) % 2:
                    nxt.append(lvl[-1])
                lvl = nxt
            return lvl[0]

        M = tree("M", [t1[0], t2[0], t1[1], t2[1], t1[2], t2[2], t1[3], t2[3]])
        X = tree("X", [mx[0], my[0], mx[1], my[1], mx[2], my[2], mx[3], my[3]])
        Y = tree("Y", yy)
        Yb = add(f"Yb_{k}", FAST, "v_add_u32 %0, 0x10001000, %1", [Y], k)
        n0 = add(f"n0_{k}", FAST, ADD, [s("z1", c - 1), s("z0", k)], k)
        n = add(f"n_{k}", FAST, ADD, [n0, s("z1", c)], k)
        lo = add(f"lmlo_{k}", SLOW, "v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD", [n, "kmul"], k)
        hi = add(f"lmhi_{k}", SLOW, "v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", [n, "kmul"], k)
        LM = add(f"LM_{k}", SLOW, "v_perm_b32 %0, %1, %2, %3", [hi, lo, "kperm"], k)
        dmax = add(f"dmax_{k}", SLOW, PMAX, [M, LM], k)
        dmin = add(f"dmin_{k}", SLOW, PMIN, [M, LM], k)
        dev = add(f"dev_{k}", FAST, SUB, [dmax, dmin], k)
        d1 = add(f"d1_{k}", FAST, SUB, [Yb, X], k)
        d2 = add(f"d2_{k}", FAST, ADD, [d1, d1], k)
        Pf = add(f"Pf_{k}", FAST, SUB, [d2, dev], k)
        if clamp:
            cl = add(f"cl_{k}", SLOW, PMAX, [Pf, "kbias"], k)
        else:
            cl = add(f"cl_{k}", SLOW, "v_pk_sub_i16 %0, %1, %2", [Pf, "kbias"], k)
        add(f"out[{k}]", FAST, "v_and_b32 %0, %1, %2", [cl, f"xmask[{k}]"], k)
    for i, o in enumerate(ops):
        o.idx = i
    return ops


def schedule(ops, window=1, D=1, run_a=0, run_p=0, crit=True):
    """List scheduling.  window: pairs in flight; D: wanted distance between an op and its producers;
    run_a/run_p: preferred run length of fast / slow ops (0 = no type preference)."""
    by_dst = {o.dst: o for o in ops}
    deps = {o.dst: [by_dst[s] for s in o.srcs if s in by_dst] for o in ops}
    users = {o.dst: [] for o in ops}
    for o in ops:
        for p in deps[o.dst]:
            users[p.dst].append(o)
    height = {}
    for o in reversed(ops):
        height[o.dst] = 1 + max((height[u.dst] for u in users[o.dst]), default=0)
    pos, order, done = {}, [], set()
    last_kind, run = None, 0
    pairs_open = list(range(window))
    remaining = {k: sum(1 for o in ops if o.pair == k) for k in range(4)}
    nxt_pair = window
    while len(order) < len(ops):
        ready = [o for o in ops if o.dst not in done and o.pair in pairs_open and all(p.dst in done for p in deps[o.dst])]
        t = len(order)

        def score(o):
            dist = min((t - pos[p.dst] for p in deps[o.dst]), default=99)
            ok = dist >= D
            want = None
            if run_a or run_p:
                if last_kind == FAST:
                    want = FAST if run < run_a else SLOW
                elif last_kind == SLOW:
                    want = SLOW if run < run_p else FAST
            typ = (o.kind == want) if want else True
            return (ok, typ, min(dist, D), height[o.dst] if crit else 0, -o.idx)

        best = max(ready, key=score)
        order.append(best)
        pos[best.dst] = t
        done.add(best.dst)
        if best.kind == last_kind:
            run += 1
        else:
            last_kind, run = best.kind, 1
        remaining[best.pair] -= 1
        if remaining[best.pair] == 0:
            pairs_open.remove(best.pair)
            if nxt_pair < 4:
                pairs_open.append(nxt_pair)
                nxt_pair += 1
    return order


def emit(order, indent="        "):
    lines, declared = [], set()
    for o in order:
        if not o.dst.startswith("out["):
            lines.append(f"{indent}uint32_t {o.dst};")
    for o in order:
        ins = ", ".join(f'"v"({s})' for s in o.srcs)
        lines.append(f'{indent}asm volatile("{o.asm}" : "=v"({o.dst}) : {ins});')
    return "\n".join(lines)


POLICIES = {
    "seq_d1": dict(window=1, D=1),
    "seq_d2": dict(window=1, D=2),
    "seq_d3": dict(window=1, D=3),
    "seq_d4": dict(window=1, D=4),
    "w2_d2": dict(window=2, D=2),
    "w2_d4": dict(window=2, D=4),
    "w2_d6": dict(window=2, D=6),
    "w4_d4": dict(window=4, D=4),
    "seq_d2_r84": dict(window=1, D=2, run_a=8, run_p=4),
    "seq_d2_r21": dict(window=1, D=2, run_a=2, run_p=1),
    "seq_d3_r42": dict(window=1, D=3, run_a=4, run_p=2),
    "seq_d3_r63": dict(window=1, D=3, run_a=6, run_p=3),
    "seq_d4_r21": dict(window=1, D=4, run_a=2, run_p=1),
    "w2_d4_r21": dict(window=2, D=4, run_a=2, run_p=1),
    "w2_d4_r42": dict(window=2, D=4, run_a=4, run_p=2),
    "w2_d4_r84": dict(window=2, D=4, run_a=8, run_p=4),
    "w2_d4_r168": dict(window=2, D=4, run_a=16, run_p=8),
    "w2_d3_r11": dict(window=2, D=3, run_a=1, run_p=1),
    "w2_d2_r2416": dict(window=2, D=2, run_a=24, run_p=16),
    "w4_d4_r84": dict(window=4, D=4, run_a=8, run_p=4),
    "w4_d6_r3216": dict(window=4, D=6, run_a=32, run_p=16),
    "seq_d1_nocrit": dict(window=1, D=1, crit=False),
}


def bench_source(policies=None, add3=False):
    global POLICIES
    if policies:
        POLICIES = policies
    ops = build_dag(add3=add3)
    out = ['#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <stdint.h>']
    for name, pol in POLICIES.items():
        order = schedule(ops, **pol)
        out.append(f"""
__global__ __launch_bounds__(256) void k_{name}(uint32_t* outp, const uint32_t* inp, int iters) {{
    uint32_t m5[12], p5[12], m4[12], p4[12], m2[12], p2[12], z1[12], z0[4], xmask[4], out[4] = {{0, 0, 0, 0}};
    const uint32_t* q = inp + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 12; ++i) {{ m5[i] = q[i * 256]; p5[i] = q[(12 + i) * 256]; m4[i] = q[(24 + i) * 256]; p4[i] = q[(36 + i) * 256];
        m2[i] = q[(48 + i) * 256]; p2[i] = q[(60 + i) * 256]; z1[i] = q[(72 + i) * 256]; }}
#pragma unroll
    for (int i = 0; i < 4; ++i) {{ z0[i] = q[(84 + i) * 256]; xmask[i] = q[(88 + i) * 256] | 0x1fff1fffu; }}
    uint32_t kmul = 349536u, kperm = 0x07060302u, kbias = 0x20002000u, acc = 0;
    asm volatile("" : "+v"(kmul), "+v"(kperm), "+v"(kbias));
    for (int it = 0; it < iters; ++it) {{
{emit(order)}
        acc ^= out[0] ^ out[1] ^ out[2] ^ out[3];
    }}
    outp[blockIdx.x * 256 + threadIdx.x] = acc;
}}""")
    names = list(POLICIES)
    out.append("""
template <typename K> void run(const char* name, K kern, uint32_t* d, uint32_t* in) {
    const int blocks = 256 * 4 * 4, iters = 200;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, in, 10);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, in, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    uint32_t h; (void)hipMemcpy(&h, d + 777, 4, hipMemcpyDeviceToHost);
    // cycles per wave-iteration (the 4 pairs of one lane) per SIMD at 2.3 GHz
    printf("%-18s %8.3f ms  %7.1f cyc per wave-iteration per SIMD @2.3GHz   chk %08x\\n", name, best,
           best * 1e-3 * 2.3e9 * 1024 / ((double)blocks * 4 * iters), h);
}
int main() {
    uint32_t *d, *in; (void)hipMalloc(&d, 256 * 16 * 256 * 4); (void)hipMalloc(&in, 92 * 256 * 4);
    uint32_t* h = (uint32_t*)malloc(92 * 256 * 4);
    uint32_t s = 12345;
    for (int i = 0; i < 92 * 256; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xff) | (((s >> 16) & 0xff) << 16); }
    (void)hipMemcpy(in, h, 92 * 256 * 4, hipMemcpyHostToDevice);""")
    for n in names:
        out.append(f'    run("{n}", k_{n}, d, in);')
    out.append("    return 0;\n}")
    return "\n".join(out)


if __name__ == "__main__":
    if sys.argv[1] == "bench":
        open(sys.argv[2], "w").write(bench_source())
    elif sys.argv[1] == "bench3":
        pols = {k: POLICIES[k] for k in ("seq_d1", "seq_d3_r42", "seq_d3_r63", "w2_d4_r21", "seq_d2", "seq_d4", "w2_d4", "seq_d1_nocrit")}
        open(sys.argv[2], "w").write(bench_source(pols, add3=True))
    elif sys.argv[1] == "bench2":
        pols = {}
        for w, D, ra, rp in itertools.product([1, 2], [3, 4, 5], [2, 3, 4, 5, 6], [1, 2, 3]):
            pols[f"w{w}_d{D}_r{ra}{rp}"] = dict(window=w, D=D, run_a=ra, run_p=rp)
        open(sys.argv[2], "w").write(bench_source(pols))
    elif sys.argv[1] == "emit":
        pol = POLICIES[sys.argv[2]]
        clamp = (len(sys.argv) < 5 or sys.argv[4] != "noclamp")
        open(sys.argv[3], "w").write(emit(schedule(build_dag(clamp), **pol)) + "\n")
