#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the committed
summaries under profiles/:  python tools/make_profiles.py r02a [r02]   (second argument: file prefix = round)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
RND = sys.argv[2] if len(sys.argv) > 2 else tag[:3]
R = os.path.join(ROOT, "gpurun_out", tag)
P = os.path.join(ROOT, "profiles")
rd = lambda n: open(os.path.join(R, n)).read()
strip = lambda t: re.sub(r"/tmp/[^ ]*?/gpurun_out/", "gpurun_out/", t)

# 1. kernel trace of the bench command
tb, b = json.loads(rd("trace_bench.json")), json.loads(rd("bench.json"))
hdr = (f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-find-boards   (round {RND[1:]}, final kernels; tools/collect_profiles.sh {tag})\n"
       f"# bench.py under the tracer: {tb['value']:.0f} frames/s, {tb['ms_per_step']:.3f} ms/step, level-0 ChESS launch {tb['roofline']['avg_launch_ms']*1e3:.1f} us by hipEvents (trace below: every launch of the process incl. set-up passes and warm-up; the quartile line is closer to the steady state)\n"
       f"# bench.py with its defaults (200 steps) without the tracer, same box, same build: {b['value']:.0f} frames/s, {b['ms_per_step']:.3f} ms/step, level-0 ChESS launch {b['roofline']['avg_launch_ms']*1e3:.1f} us -> {b['roofline']['achieved']:.0f} GB/s = {b['roofline']['frac']*100:.1f} % of 8 TB/s\n")
open(os.path.join(P, f"{RND}_bench_kernel_trace.txt"), "w").write(hdr + strip(rd("bench_kernel_trace.txt")))
open(os.path.join(P, f"{RND}_bench.json"), "w").write(rd("bench.json"))

# 2. EA traffic of the bench command, separate passes
parts = []
for n, title in [("pmc_rd", "pass 1: --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B"),
                 ("pmc_wr", "pass 2: --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"), ("pmc_fetch", "pass 3: --pmc FETCH_SIZE"),
                 ("pmc_write", "pass 4: --pmc WRITE_SIZE")]:
    parts.append(f"## {title}\n" + rd(n + ".txt"))
open(os.path.join(P, f"{RND}_bench_pmc_ea_traffic.txt"), "w").write(
    "# rocprofv3 --pmc ... --kernel-trace --output-format csv -- python bench.py --distinct 4 --steps 3 --warmup 1 --no-cpu-baseline (+ the legs beside the timed steps off: tools/collect_pmc_bench.sh)\n"
    f"# separate passes, counters only; means over the dispatches of each kernel (tools/pmc_summary.py); final round-{RND[1:]} kernels\n" + "\n".join(parts))


def grab(fn, kernel_grid, ctr):
    s = rd(fn + ".txt")
    blk = s[s.index(kernel_grid):]
    return float(re.search(re.escape(ctr) + r"\s+mean\s+([0-9.]+)", blk).group(1))


def kernel_line(fn, name):
    """'<name>  grid=N' of the kernel's block in a pmc summary (the grid follows the segment count the launcher picked)"""
    m = re.search(re.escape(name) + r"  grid=\d+", rd(fn + ".txt"))
    assert m, (fn, name)
    return m.group(0)


fused = "chess_v1_pyr_kernel" in rd("pmc_rd.txt")
kg = kernel_line("pmc_rd", "chess_v1_pyr_kernel" if fused else "chess_v1_kernel<true, true, 1>")
rdb = 128 * grab("pmc_rd", kg, "TCC_EA0_RDREQ_128B") + 64 * grab("pmc_rd", kg, "TCC_EA0_RDREQ_64B") + 32 * grab("pmc_rd", kg, "TCC_EA0_RDREQ_32B")
wr64, wrall = grab("pmc_wr", kg, "TCC_EA0_WRREQ_64B"), grab("pmc_wr", kg, "TCC_EA0_WRREQ ")
wrb = 64 * wr64 + 32 * (wrall - wr64)
px = 64 * 4096 * 3072
j = json.load(open(os.path.join(P, "chess_l0_traffic.json")))
j["kernel"] = ("mrg::chess_v1_pyr_kernel (CLAMP, HOT, STAGE_PERM16, + level images 1..3)" if fused else
               "mrg::chess_v1_kernel<true, true, 1> (CLAMP, HOT, STAGE_PERM16)")
j["algorithmic_bytes_per_pixel"] = 3.328125 if fused else 3.0
j["source"] = f"profiles/{RND}_bench_pmc_ea_traffic.txt"
# (written from scratch every time: the round and the file it names are the ones in `source`)
j["_comment"] = (f"HBM-side (L2 <-> EA fabric) traffic of ONE launch of the dominant kernel, {j['kernel'].split(' ')[0]} at level 0 over 64 "
                 f"frames of 4096x3072, from rocprofv3 --pmc passes of `python bench.py --distinct 4 --steps 3 --warmup 1 --no-cpu-baseline` "
                 f"(round {RND[1:]}, {j['source']}). Separate passes: {{TCC_EA0_RDREQ, _32B, _64B, _128B}}, {{TCC_EA0_WRREQ, _64B}}, "
                 "{FETCH_SIZE}, {WRITE_SIZE}. Reads = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B. FETCH_SIZE (KiB) reads exactly half of "
                 "that on gfx950 (MI355X_MICROARCH.md, HBM section), so it is doubled; WRITE_SIZE (KiB) matches 64*WRREQ_64B to 0.1 %.")
j["kernel_id"] = b.get("kernel_id")   # the library the passes ran on (bench.py prints it; mrgingham_amd_kernel_id)
j.update(read_bytes=int(rdb), write_bytes=int(wrb), fetch_size_kib_raw=grab("pmc_fetch", kg, "FETCH_SIZE"),
         write_size_kib_raw=grab("pmc_write", kg, "WRITE_SIZE"), bytes_per_pixel=round((rdb + wrb) / px, 4))
json.dump(j, open(os.path.join(P, "chess_l0_traffic.json"), "w"), indent=1)
print(f"level-0 ChESS: read {rdb/1e6:.1f} MB + written {wrb/1e6:.1f} MB per launch = {(rdb+wrb)/px:.4f} B/px")

# 3. issue / wait / LDS counters of the level-0 response kernel alone
open(os.path.join(P, f"{RND}_chess_l0_sq_counters.txt"), "w").write(
    "# rocprofv3 --pmc <SQ counters> --kernel-trace --output-format csv -- python tools/chess_l0_alone.py   (32 frames 4096x3072, clamp kernel without the hot list; two passes)\n"
    "# per wave-iteration (512 px): divide by the number of wave-iterations (pixels / 512); SQ_* cycle counters are in quad-cycles (DESIGN.md section 7)\n"
    + rd("pmc_sq1.txt") + rd("pmc_sq2.txt"))

# 4. preprocessing kernels
open(os.path.join(P, f"{RND}_preprocess_kernel_trace.txt"), "w").write(
    "# rocprofv3 --kernel-trace --stats -- python tools/preprocess_bench.py   (64 frames 4096x3072; clahe+blur, clahe only, blur only; row (f)-2)\n"
    + strip(rd("preprocess_kernel_trace.txt")) + "\n" + rd("prebench.txt"))

# 5. the same SQ counters on the kernels as bench.py launches them (production level-0 kernel), and the VALU issue figure
#    bench.py quotes next to the HBM roofline
if os.path.exists(os.path.join(R, "pmc_sqp1.txt")):
    open(os.path.join(P, f"{RND}_bench_sq_counters.txt"), "w").write(
        "# rocprofv3 --pmc <SQ counters> --kernel-trace --output-format csv -- python bench.py --distinct 4 --steps 3 --warmup 1 --prime 2 --no-cpu-baseline --no-end-to-end (+ the legs beside the timed steps off: tools/collect_pmc_bench.sh)   (two passes)\n"
        "# the kernels exactly as the bench launches them: chess_v1_pyr_kernel = 64 frames of 4096x3072, clamp + hot list + level images 1..3\n"
        "# per wave-iteration (512 px): divide by pixels / 512 = 1 572 864 for the level-0 launch; SQ_* cycle counters are in quad-cycles\n"
        + rd("pmc_sqp1.txt") + rd("pmc_sqp2.txt"))
    kgq = kernel_line("pmc_sqp1", "chess_v1_pyr_kernel")
    wi = 64 * 4096 * 3072 / 512
    insts, wavecyc = grab("pmc_sqp1", kgq, "SQ_INSTS_VALU"), grab("pmc_sqp1", kgq, "SQ_WAVE_CYCLES")
    waves = grab("pmc_sqp1", kgq, "SQ_WAVES")
    json.dump({
        "_comment": "VALU issue of ONE launch of the dominant kernel (mrg::chess_v1_pyr_kernel, 64 frames of 4096x3072) from the rocprofv3 --pmc "
                    f"passes of the bench command (profiles/{RND}_bench_sq_counters.txt).  A SIMD issues at most one VALU instruction per quad-cycle "
                    "from one wave; with 4 waves resident per SIMD the wall quad-cycles per wave-iteration are SQ_WAVE_CYCLES / 4 / wave-iterations, "
                    "and valu_issue_frac = SQ_INSTS_VALU / (SQ_WAVE_CYCLES / 4): the fraction of the SIMDs' quad-cycle issue slots that carry a VALU instruction.",
        "kernel": "mrg::chess_v1_pyr_kernel", "kernel_id": b.get("kernel_id"), "frames": 64, "width": 4096, "height": 3072,
        "valu_insts_per_wave_iteration": insts / wi, "wave_quad_cycles_per_wave_iteration": wavecyc / wi,
        "waves_per_simd": 4, "valu_issue_frac": insts / (wavecyc / 4.0),
        "wait_any_frac": grab("pmc_sqp1", kgq, "SQ_WAIT_ANY") / wavecyc, "wait_inst_any_frac": grab("pmc_sqp1", kgq, "SQ_WAIT_INST_ANY") / wavecyc,
        "lds_insts_per_wave_iteration": grab("pmc_sqp2", kgq, "SQ_INSTS_LDS") / wi, "waves": waves,
        "source": f"profiles/{RND}_bench_sq_counters.txt"}, open(os.path.join(P, "chess_l0_valu.json"), "w"), indent=1)
for n, out in [("cluttered_kernel_trace.txt", f"{RND}_cluttered_kernel_trace.txt")]:
    if os.path.exists(os.path.join(R, n)):
        open(os.path.join(P, out), "w").write("# rocprofv3 --kernel-trace --stats -- python bench.py --workload c3_cluttered --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end\n" + strip(rd(n)))
if os.path.exists(os.path.join(R, "sparse_kernel_trace.txt")):
    open(os.path.join(P, f"{RND}_sparse_kernel_trace.txt"), "w").write("# rocprofv3 --kernel-trace --stats -- python bench.py --sparse-refine --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end\n" + strip(rd("sparse_kernel_trace.txt")))
for n in ("bench_cluttered.json", "bench_c4.json", "bench_14x14.json", "bench_sparse.json", "bench_sparse_c4.json"):
    if os.path.exists(os.path.join(R, n)) and rd(n).strip():
        open(os.path.join(P, f"{RND}_{n}"), "w").write(rd(n))

# 6. round 5: the plain ChESS pass alone, config 2, the rehearsal, chess_v16 against chess_v1
if os.path.exists(os.path.join(R, "chess_alone_kernel_trace.txt")):
    ca = json.loads(rd("chess_alone.json").strip().splitlines()[-1])
    cu = json.loads(rd("chess_alone_untraced.json").strip().splitlines()[-1])
    open(os.path.join(P, f"{RND}_chess_alone_kernel_trace.txt"), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python tools/chess_pass_alone.py   (mrgingham_amd_chess_response_batch(level 0, clamp 0) = the output of ChESS.c:56-106,\n"
        "# 64 frames of 4096x3072, 130 launches back to back, nothing else on the device; 3 B/px = 2 415 919 104 bytes per launch)\n"
        f"# the script's own hipEvents under the tracer: avg {ca['avg_launch_ms']*1e3:.1f} us (first event to last / 120, dispatch gaps included) = {ca['frac']*100:.1f} % of 8 TB/s, median launch {ca['median_launch_ms']*1e3:.1f} us = {ca['frac_median_launch']*100:.1f} %\n"
        f"# the same without the tracer, same box: avg {cu['avg_launch_ms']*1e3:.1f} us = {cu['frac']*100:.1f} %, median {cu['median_launch_ms']*1e3:.1f} us = {cu['frac_median_launch']*100:.1f} %\n"
        + strip(rd("chess_alone_kernel_trace.txt")))
if os.path.exists(os.path.join(R, "c2_kernel_trace.txt")):
    c2 = json.loads(rd("bench_c2.json").strip().splitlines()[-1])
    open(os.path.join(P, f"{RND}_c2_kernel_trace.txt"), "w").write(
        "# rocprofv3 --kernel-trace --stats -- python bench.py --workload c2_1920x1080_level0 --steps 40 --warmup 5 --no-cpu-baseline --no-find-boards --no-end-to-end\n"
        f"# BASELINE config 2 (64 x 1920x1080, level-0 detect).  bench.py with its defaults, untraced, same box: {c2['value']:.0f} frames/s, {c2['ms_per_step']:.4f} ms/step, level-0 launch "
        f"{c2['roofline']['avg_launch_ms']*1e3:.1f} us = {c2['roofline']['frac']*100:.1f} % of 8 TB/s on 3 B/px; chess_pass_alone {c2['chess_pass_alone']['avg_launch_ms']*1e3:.1f} us = {c2['chess_pass_alone']['frac']*100:.1f} %\n"
        + strip(rd("c2_kernel_trace.txt")))
    open(os.path.join(P, f"{RND}_bench_c2.json"), "w").write(rd("bench_c2.json"))
for n in ("bench_rehearsal.json", "chess16_sweep.txt", "sparse_subsets_ab.txt", "seg_rounds_sweep.txt"):
    if os.path.exists(os.path.join(R, n)) and rd(n).strip():
        open(os.path.join(P, f"{RND}_{n}"), "w").write(strip(rd(n)))
if os.path.exists(os.path.join(R, "pmc_a16.txt")):
    open(os.path.join(P, f"{RND}_chess16_sq_counters.txt"), "w").write(
        "# rocprofv3 --pmc <SQ counters> --kernel-trace --output-format csv -- python tools/chess16_pmc.py 16 | 1   (32 frames of 4096x3072, plain response, two passes each)\n"
        "# chess_v16_kernel (16 pixels per lane, 3 waves per SIMD) beside chess_v1_kernel (8 pixels per lane, 4 waves per SIMD); per 512 px: divide by 786 432\n"
        + rd("pmc_a16.txt") + rd("pmc_b16.txt") + rd("pmc_a1.txt") + rd("pmc_b1.txt"))

# 7. EA traffic of the plain ChESS pass (chess_v16_kernel alone): what bench.py replays as chess_pass_alone.traffic
if os.path.exists(os.path.join(R, "pmc_ard.txt")):
    kga = kernel_line("pmc_ard", "chess_v16_kernel<false>")
    ardb = 128 * grab("pmc_ard", kga, "TCC_EA0_RDREQ_128B") + 64 * grab("pmc_ard", kga, "TCC_EA0_RDREQ_64B") + 32 * grab("pmc_ard", kga, "TCC_EA0_RDREQ_32B")
    aw64, awall = grab("pmc_awr", kga, "TCC_EA0_WRREQ_64B"), grab("pmc_awr", kga, "TCC_EA0_WRREQ ")
    awrb = 64 * aw64 + 32 * (awall - aw64)
    apx = 32 * 4096 * 3072
    open(os.path.join(P, f"{RND}_chess_alone_pmc_ea_traffic.txt"), "w").write(
        "# rocprofv3 --pmc ... --kernel-trace --output-format csv -- python tools/chess16_pmc.py 16   (chess_v16_kernel alone, 32 frames of 4096x3072, plain response; two passes)\n"
        "## pass 1: --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B\n" + rd("pmc_ard.txt") +
        "## pass 2: --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B\n" + rd("pmc_awr.txt"))
    json.dump({"_comment": f"HBM-side (L2 <-> EA fabric) traffic of ONE launch of mrg::chess_v16_kernel<false> (the plain ChESS pass, 32 frames of 4096x3072) from the rocprofv3 --pmc passes "
                           f"of tools/chess16_pmc.py (round {RND[1:]}, profiles/{RND}_chess_alone_pmc_ea_traffic.txt).  Reads = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B, writes = 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B).",
               "kernel": "mrg::chess_v16_kernel<false>", "frames": 32, "width": 4096, "height": 3072, "read_bytes": int(ardb), "write_bytes": int(awrb),
               "bytes_per_pixel": round((ardb + awrb) / apx, 4), "algorithmic_bytes_per_pixel": 3.0,
               "source": f"profiles/{RND}_chess_alone_pmc_ea_traffic.txt", "kernel_id": b.get("kernel_id")},
              open(os.path.join(P, "chess_alone_traffic.json"), "w"), indent=1)
    print(f"plain ChESS pass: read {ardb/1e6:.1f} MB + written {awrb/1e6:.1f} MB per 32-frame launch = {(ardb+awrb)/apx:.4f} B/px")

# 8. round 6: EA traffic of the preprocessing kernels
if os.path.exists(os.path.join(R, "pmc_prd.txt")) and rd("pmc_prd.txt").strip():
    open(os.path.join(P, f"{RND}_preprocess_pmc_ea_traffic.txt"), "w").write(
        "# rocprofv3 --pmc ... --kernel-trace --output-format csv -- python tools/preprocess_bench.py 3   (64 frames of 4096x3072 = 805.3 Mpx per launch; two passes)\n"
        "# bytes per launch = 128 * RDREQ_128B + 64 * RDREQ_64B + 32 * RDREQ_32B (reads), 64 * WRREQ_64B + 32 * (WRREQ - WRREQ_64B) (writes); algorithmic: clahe_hist 1 B/px read,\n"
        "# clahe_blur3 1 B/px read + 1 B/px written (two-kernel path for comparison: clahe_apply_fast 1 + 1, box_blur3 1 + 1)\n"
        "## pass 1: --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B\n" + rd("pmc_prd.txt") +
        "## pass 2: --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B\n" + rd("pmc_pwr.txt"))
    import hashlib
    def g(fn, kern, ctr):
        blk = rd(fn + ".txt")
        blk = blk[blk.index(kern):]
        return float(re.search(re.escape(ctr) + r"\s+mean\s+([0-9.]+)", blk).group(1))
    def rdb(kern):
        return 128 * g("pmc_prd", kern, "TCC_EA0_RDREQ_128B") + 64 * g("pmc_prd", kern, "TCC_EA0_RDREQ_64B") + 32 * g("pmc_prd", kern, "TCC_EA0_RDREQ_32B")
    def wrb(kern):
        w64, wall = g("pmc_pwr", kern, "TCC_EA0_WRREQ_64B"), g("pmc_pwr", kern, "TCC_EA0_WRREQ ")
        return 64 * w64 + 32 * (wall - w64)
    kernels = ["clahe_hist_kernel<16>", "minmax_from_hist_kernel", "clahe_lut_kernel", "clahe_quad_kernel", "clahe_blur3_kernel<32>"]
    per = {k: {"read_bytes": int(rdb(k)), "write_bytes": int(wrb(k))} for k in kernels}
    total = sum(v["read_bytes"] + v["write_bytes"] for v in per.values())
    src = open(os.path.join(ROOT, "mrgingham_amd", "csrc", "preprocess.hip"), "rb").read()
    json.dump({"_comment": "HBM-side (L2 <-> EA fabric) traffic of ONE call of the tool's default preprocessing chain (normalize + CLAHE(8) + 3x3 blur, fused path) on 64 frames "
                           f"of 4096x3072, per kernel, from the rocprofv3 --pmc passes of tools/preprocess_bench.py (profiles/{RND}_preprocess_pmc_ea_traffic.txt); bench.py replays "
                           "the total as configs.preprocess.traffic when preprocess.hip is the file these were collected on",
               "frames": 64, "width": 4096, "height": 3072, "kernels": per, "bytes_per_call": int(total),
               "bytes_per_pixel": round(total / (64 * 4096 * 3072), 4), "algorithmic_bytes_per_pixel": 3.0,
               "preprocess_hip_sha16": hashlib.sha256(src).hexdigest()[:16], "source": f"profiles/{RND}_preprocess_pmc_ea_traffic.txt"},
              open(os.path.join(P, "preprocess_traffic.json"), "w"), indent=1)
    print(f"preprocessing chain: {total/1e6:.1f} MB per 64-frame call = {total/(64*4096*3072):.4f} B/px")
