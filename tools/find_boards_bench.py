"""Throughput of the full detector over device-resident batches (mrgingham_amd_find_boards_submit / _collect:
per-frame adaptive pyramid depth, device candidates + refinement, host grid-finder threads), pipelined and
synchronous.  usage: python tools/find_boards_bench.py [out.json] [--quick] [--one W H B depth nthreads]
(MRGINGHAM_AMD_LIB=.../libmrgingham_amd_experiment.so MRG_DBG_FB=1: host milliseconds by phase on stderr)"""
import sys, os, time, json; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # a context uses four HIP streams that must overlap (as bench.py)
import numpy as np
import torch
import mrgingham_amd
from mrgingham_amd import synth


def run(W, H, B, depth, nthreads=0, n=300, pipeline=1, repeats=3):
    det = mrgingham_amd.Detector(0)
    det.set_option("find_boards_pipeline", pipeline)
    batches = [synth.board_batch(B, W, H, 10, 100 * i, device='cuda') for i in range(2)]
    jobs = []
    last = [None]
    def step(i):
        jobs.append(det.find_boards_submit(batches[i % 2], gridn=10, nthreads=nthreads))
        if len(jobs) >= depth:
            last[0] = det.find_boards_collect(jobs.pop(0))
    def drain():
        while jobs:
            last[0] = det.find_boards_collect(jobs.pop(0))
    for i in range(8):
        step(i)
    drain()
    torch.cuda.synchronize()
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        drain()
        times.append((time.perf_counter() - t0) / n)
    found = last[0][1]
    det.close()
    times.sort()
    return {"W": W, "H": H, "B": B, "depth": depth, "pipeline": pipeline, "nthreads": nthreads, "batches_timed": n,
            "ms_per_batch_best": times[0] * 1e3, "ms_per_batch_median": times[len(times) // 2] * 1e3,
            "frames_per_s": B / times[len(times) // 2], "frames_per_s_best": B / times[0],
            "found_levels": np.bincount(found[found >= 0], minlength=4).tolist(), "not_found": int((found < 0).sum())}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    quick = "--quick" in sys.argv
    res = []
    if "--one" in sys.argv:
        W, H, B, depth, nt = (int(a) for a in sys.argv[sys.argv.index("--one") + 1: sys.argv.index("--one") + 6])
        r = run(W, H, B, depth, nthreads=nt)
        print(json.dumps(r))
        sys.exit(0)
    for (W, H, B) in [(4096, 3072, 64)] if quick else [(4096, 3072, 64), (1920, 1080, 64), (640, 480, 64)]:
        for (depth, pipeline) in [(1, 0), (1, 1), (2, 1), (3, 1)]:
            r = run(W, H, B, depth, pipeline=pipeline, n=100 if not pipeline else 300)
            res.append(r)
            print(f"{W}x{H} x{B} {'pipelined' if pipeline else 'synchronous'} depth {depth}: {r['ms_per_batch_median']:7.3f} ms per batch "
                  f"(best {r['ms_per_batch_best']:.3f}) -> {r['frames_per_s']:8.0f} frames/s; found at levels {r['found_levels']}, "
                  f"not found {r['not_found']}", flush=True)
    for nt in (32, 16, 12, 8, 4):
        r = run(4096, 3072, 64, 3, nthreads=nt)
        res.append(r)
        print(f"4096x3072 x64 pipelined depth 3, {nt} grid-finder threads: {r['ms_per_batch_median']:7.3f} ms (best {r['ms_per_batch_best']:.3f}) "
              f"-> {r['frames_per_s']:8.0f} frames/s", flush=True)
    if args:
        json.dump(res, open(args[0], "w"), indent=1)
