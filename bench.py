#!/usr/bin/env python3
"""Benchmark of the MI355X corner-candidate path (contract: see the task brief).

  python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic frames that
already live in HBM: for each of B = 64 frames of 4096x3072 (10x10 board, the
size BASELINE.json's metric is quoted on) the reference's found-frame schedule
for image_pyramid_level < 0 -- detect at pyramid level 3, then refine through
levels 2, 1, 0 (mrgingham.cc:50, :81-99; 1.328*W*H ChESS pixels per frame) --
ending with the corner list on the device (and, for N > 1, ONE gather of the
corner lists to rank 0 over RCCL).  Steps are queued back to back like a
streaming pipeline would (no host sync between steps; the timed region is
fenced by a full device sync + barrier on both sides), so the component search
of step N overlaps the pixel kernels of step N+1.  Weak scaling: every rank owns its own B
frames.  Rank 0 prints ONE JSON line.

  roofline      dominant kernel = the level-0 ChESS response kernel; achieved =
                algorithmic bytes per launch (3 B/px: u8 read once + int16
                written once) / average launch duration measured
                with hipEvents on the stream each launch ran on.
  cpu_baseline  the C oracle (a port of the reference's algorithm) running the
                same schedule on a bounded sample of the same frames, one frame
                per host thread at a time like the reference CLI's --jobs: T = 1
                and T = all physical cores, both as numbers.
  end_to_end    (N = 1) the same step fed from PINNED HOST memory: H2D of the
                batch on two copy streams (double-buffered device frames) ->
                chain -> D2H of the packed corner lists; never `value`.
  configs       (N = 1) every other BASELINE config that fits one GPU as a short
                leg of its own, never `value`: c2_level0 (64 x 1920x1080, level-0
                detect), preprocess (the reference tool's default CLAHE + blur on
                the bench frames), c5_mixed_one_rank (1-12 MP stream through the
                pipelined detector), c1_tool (the built tool on 640x480 files).
  sclk_mhz      the engine clock the timed level-0 launches actually ran at
                (mrgingham_amd_sclk_mhz: s_memtime / s_memrealtime inside the
                kernel), and `frac_at_2400mhz` = frac * 2400 / sclk_mhz: the
                kernels are VALU-bound, so a slow box shows here, not in the code.

`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment)
starts the N ranks itself, one process per GPU, on a free local port.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A context uses three HIP streams (pixel kernels + two component chains) that must overlap; HIP maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share one serialise.
# With torch's and RCCL's own streams in the process, leave room (must be set before HIP initialises).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (W, H, gridn, start_level, batch, background)
    "c3_4096x3072_chain": (4096, 3072, 10, 3, 64, "flat"),          # the size and board BASELINE.json's metric names
    "c3_4096x3072_14x14_chain": (4096, 3072, 14, 3, 64, "flat"),    # configs[2] as stated: 14x14 board
    # configs[3]: 2048 frames sharded 256 per GPU at N = 8, one gather of the corner lists (weak scaling: every
    # rank owns 256 frames whatever N is; `--gpus 1` is one GPU's shard of the job)
    "c4_4096x3072_shard256": (4096, 3072, 10, 3, 256, "flat"),
    # the board over a textured background (synth.cluttered_board_frame): ~7e4 hot pixels per frame at level 0,
    # ~1e4 at level 1 -- the component search cannot run out of its LDS tables there
    "c3_cluttered": (4096, 3072, 10, 3, 64, "clutter"),
    "c2_1920x1080_level0": (1920, 1080, 10, 0, 64, "flat"),
    "c1_640x480_chain": (640, 480, 10, 3, 64, "flat"),
}


def gpu_numa_cpus(local_rank):
    """(numa node, cpu list) of the host CPUs next to HIP device `local_rank`, from sysfs; (None, None) if the
    topology cannot be read.  Must not initialise HIP in the parent of a launcher: uses rocm-smi's sysfs tree."""
    try:
        bus = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(bus, "pci_domain_id", 0), bus.pci_bus_id, bus.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        node = int(open(base + "/numa_node").read())
        cpus = open(base + "/local_cpulist").read().strip()
        out = []
        for part in cpus.split(","):
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
        return node, out
    except Exception:
        return None, None


def bind_rank_to_gpu_numa(local_rank):
    """One rank per GPU: keep the rank's host threads (H2D staging, launches, RCCL proxy) on the cores of the
    GPU's NUMA node -- on a two-socket 8-GPU node a rank that runs on the far socket feeds its GPU over the
    socket interconnect (the reference's worker model has no such notion: mrgingham-from-image.cc:374-379)."""
    node, cpus = gpu_numa_cpus(local_rank)
    if not cpus:
        return {"numa_node": node, "cpus": None}
    try:
        allowed = os.sched_getaffinity(0)
        want = set(cpus) & allowed
        if want:
            os.sched_setaffinity(0, want)
        return {"numa_node": node, "cpus": len(want)}
    except (AttributeError, OSError):
        return {"numa_node": node, "cpus": None}


def physical_cores():
    """(physical cores, logical cpus, model name) of this host from /proc/cpuinfo."""
    logical = os.cpu_count() or 1
    cores, model, phys, core = set(), "unknown", None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    try:
        logical = min(logical, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    n = len(cores) if cores else max(1, logical // 2)
    return min(n, logical), logical, model


def cpu_quota_cores():
    """CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited.
    The GPU boxes of this pool expose all 256 logical CPUs of the host but cap the container at 16 cores' worth
    of CPU time: more runnable threads than that only get throttled."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(frames_host, start_level, leg_seconds=5.0):
    """Oracle (kind "port") on the host cores through oracle/cpu_bench.c: a pthread harness with the reference
    CLI's worker model (mrgingham-from-image.cc:50, :374-379: T threads, one frame per thread at a time, the
    per-call allocations of the reference kept), every leg at least `leg_seconds` long, no Python in the timed
    region.  Legs: T = 1; T = all physical cores under glibc's default allocator policy (what the reference
    binary gets: every 25 MB level buffer is an mmap / munmap + page faults); the same with freed blocks kept on
    the heap.  `value` = the better of the two all-core legs.  The upstream ChESS.c built as shipped
    (oracle/_ref, level-0 response only) runs in the same harness for scale."""
    import numpy as np
    from oracle import oracle
    oracle.lib()
    nphys, nlogical, model = physical_cores()
    quota = cpu_quota_cores()
    # threads of the all-core legs: one per physical core the container can actually keep busy
    ncores = nphys if quota is None else max(1, min(nphys, int(quota + 0.5)))
    frames_host = np.ascontiguousarray(frames_host)
    n = len(frames_host)
    oracle.bench_chain(frames_host[:1], start_level, 1, 0.0)                       # warm
    p1, e1, _ = oracle.bench_chain(frames_host, start_level, 1, leg_seconds)
    pa, ea, pts_a = oracle.bench_chain(frames_host, start_level, ncores, leg_seconds)
    oracle.bench_heap_reuse(True)
    oracle.bench_chain(frames_host, start_level, ncores, 1.0)                      # warm the heap
    ph, eh, _ = oracle.bench_chain(frames_host, start_level, ncores, leg_seconds)
    p1h, e1h, _ = oracle.bench_chain(frames_host, start_level, 1, min(leg_seconds, 3.0))
    oracle.bench_heap_reuse(False)
    t1, tall, tall_h, t1h = p1 / e1, pa / ea, ph / eh, p1h / e1h
    best = max(tall, tall_h)
    out = {"value": best, "unit": "frames/s", "cores": ncores, "kind": "port",
           "physical_cores": nphys, "logical_cpus": nlogical, "cpu_model": model,
           "cpu_quota_cores": quota,
           "t1_frames_s": t1, "tall_frames_s": tall, "tall_frames_s_heap_reuse": tall_h, "t1_frames_s_heap_reuse": t1h,
           "threads_all": ncores,
           "parallel_efficiency": tall / (t1 * ncores), "parallel_efficiency_heap_reuse": tall_h / (t1h * ncores),
           "candidates_per_frame": pts_a / max(pa, 1),
           "sample": f"oracle/cpu_bench.c (pthreads, one frame per thread at a time, per-call allocations kept): "
                     f"T=1 {p1} frame passes in {e1:.1f} s; T={ncores} "
                     f"({nphys} physical cores, CPU quota {'none' if quota is None else '%.1f cores' % quota}) {pa} passes in {ea:.1f} s (glibc default "
                     f"allocator policy), {ph} passes in {eh:.1f} s (freed blocks kept on the heap); {n} distinct "
                     f"frames; every pass is the full detect(L{start_level})+refine chain of the C oracle (gcc -O3)",
           "what": "`value` = the better all-core leg of the oracle port (whole chain).  `cores` = threads of the "
                   "all-core legs = min(physical cores, the container's CPU quota).  parallel_efficiency = "
                   "tall / (t1 * cores).  The upstream ChESS.c built as shipped (oracle/_ref) is timed beside it "
                   "(level-0 response only) for scale."}
    if oracle.have_reference_build():                       # the upstream ChESS.c itself, level 0 only
        pr1, er1 = oracle.bench_ref_chess(frames_host, 1, 2.0)
        pra, era = oracle.bench_ref_chess(frames_host, ncores, 3.0)
        out["upstream_chess_level0_ms_per_frame_t1"] = er1 / max(pr1, 1) * 1e3
        out["upstream_chess_level0_frames_s_t1"] = pr1 / er1
        out["upstream_chess_level0_frames_s_tall"] = pra / era
    return out


def end_to_end(det, frames, start_level, P, steps=12, warmup=3):
    """The step fed from pinned host memory (SURVEY.md 8d's second rate): host batch -> H2D on one of
    two copy streams into one of two device buffers -> chain on the context's streams -> packed corner
    lists D2H on a third stream; uploads of step i+1 overlap the kernels of step i."""
    from mrgingham_amd import parallel
    B, H, W = frames.shape
    dev = frames.device
    host = torch.empty((B, H, W), dtype=torch.uint8, pin_memory=True)
    host.copy_(frames)
    dbuf = [torch.empty_like(frames) for _ in range(2)]
    packs = [parallel.packed_outputs(B, P, dev) for _ in range(2)]
    hout = [torch.empty(packs[0][0].numel(), dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    up = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    down = torch.cuda.Stream(dev)
    done = [None, None]
    torch.cuda.synchronize()

    def step(i):
        b = i & 1
        with torch.cuda.stream(up[b]):
            if done[b] is not None:
                up[b].wait_event(done[b])                    # the chain that read dbuf[b] / wrote packs[b] is complete
            dbuf[b].copy_(host, non_blocking=True)
        det.after_stream(up[b])                              # the chain starts behind the upload, on the device
        det.chain(dbuf[b], start_level=start_level, max_points=P, out=packs[b][1:], sync=False)
        det.stream_wait(down)
        with torch.cuda.stream(down):
            hout[b].copy_(packs[b][0], non_blocking=True)
            done[b] = torch.cuda.Event()
            done[b].record(down)

    for i in range(warmup):
        step(i)
    det.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        step(i)
    det.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    npts = parallel.unpack_outputs(hout[(warmup + steps - 1) & 1], B, P)[2]
    return {"value": B * steps / dt, "unit": "frames/s", "h2d_GBs": B * H * W * steps / dt / 1e9,
            "d2h_bytes_per_step": int(packs[0][0].numel()), "steps": steps, "ms_per_step": dt / steps * 1e3,
            "frames_with_points_last_step": int((npts > 0).sum()),
            "what": "pinned host batch -> H2D (2 copy streams, double-buffered) -> chain -> corner lists D2H to "
                    "pinned host; PCIe-bound (the link, not the kernels)"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU,
    rendezvous on a free local port) and wait for them; rank 0 prints the JSON line.  (`--rehearse`: the ranks
    all use HIP device 0 -- main() ignores LOCAL_RANK for the device then.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def find_boards_leg(device_index, frames, gridn, batches=150, depth=3):
    """The reference's product call -- find_chessboard_from_image_array: level search + host grid finder + refinement
    (mrgingham.cc:106-140) -- over the bench frames, pipelined (mrgingham_amd_find_boards_submit / _collect, `depth`
    batches in flight) and one batch at a time.  NOT `value`: it is host-bound (the grid finder on this box's cores);
    the boards of the first pipelined batch are compared with the synchronous dense schedule's, double for double."""
    import numpy as np
    import mrgingham_amd
    ref = mrgingham_amd.Detector(device_index)
    ref.set_option("find_boards_pipeline", 0)
    ref.set_option("sparse_refine", 0)
    want = ref.find_boards(frames, gridn=gridn)
    t0 = time.perf_counter()
    nsync = 10
    for _ in range(nsync):
        ref.find_boards(frames, gridn=gridn)
    sync_ms = (time.perf_counter() - t0) / nsync * 1e3
    ref.close()
    det = mrgingham_amd.Detector(device_index)
    try:
        jobs, first, last = [], None, None
        def step():
            nonlocal first, last
            jobs.append(det.find_boards_submit(frames, gridn=gridn))
            if len(jobs) >= depth:
                last = det.find_boards_collect(jobs.pop(0))
                if first is None:
                    first = (last[0].copy(), last[1].copy())
        for _ in range(10):
            step()
        while jobs:
            det.find_boards_collect(jobs.pop(0))
        det.find_boards_stats(reset=True)                    # the phase clock starts on an empty pipeline ...
        for _ in range(depth - 1):                           # ... which is filled again before the timed batches
            step()
        import resource
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        for _ in range(batches):
            step()
        while jobs:
            last = det.find_boards_collect(jobs.pop(0))
        dt = (time.perf_counter() - t0) / batches
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu_ms = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / batches * 1e3   # all threads of the process
        quota = cpu_quota_cores()
        st = det.find_boards_stats(reset=True)
        nb = max(st["batches"], 1.0)
        ok = bool(np.array_equal(first[1], want[1]))
        for f in range(len(want[1])):
            if want[1][f] >= 0:
                ok = ok and bool(np.array_equal(first[0][f], want[0][f]))
        B = frames.shape[0]
        return {"value": B / dt, "unit": "frames/s", "ms_per_batch": dt * 1e3, "batches": batches, "in_flight": depth,
                "one_batch_at_a_time_synchronous_ms": sync_ms, "one_batch_at_a_time_synchronous_frames_per_s": B / (sync_ms / 1e3),
                "boards_found": int((want[1] >= 0).sum()), "found_at_level": np.bincount(want[1][want[1] >= 0], minlength=4).tolist(),
                "identical_to_synchronous_dense": ok, "repeated_densely_by_the_library": det.sparse_fallbacks(),
                # what the line is made of (mrgingham_amd_find_boards_stats; per batch of B frames, averaged over the timed batches)
                "host_threads_used": int(st["host_threads"]),
                "host_threads_rule": "one per core the process may use (std::thread::hardware_concurrency), at most 32; "
                                     "cpu quota of this container: %s cores" % ("none" if cpu_quota_cores() is None else "%.0f" % cpu_quota_cores()),
                "grid_finder_calls_per_frame": st["grid_calls"] / (nb * B),
                "grid_finder_us_per_call": (st["grid_us_graph"] + st["grid_us_adjacency"] + st["grid_us_sequences"] + st["grid_us_cycles_rows"]) / max(st["grid_calls"], 1.0),
                "grid_finder_us_per_frame": (st["grid_us_graph"] + st["grid_us_adjacency"] + st["grid_us_sequences"] + st["grid_us_cycles_rows"]) / (nb * B),
                "grid_finder_us_per_call_by_phase": {"neighbour_graph": st["grid_us_graph"] / max(st["grid_calls"], 1.0),
                                                     "adjacency": st["grid_us_adjacency"] / max(st["grid_calls"], 1.0),
                                                     "sequences": st["grid_us_sequences"] / max(st["grid_calls"], 1.0),
                                                     "cycles_rows": st["grid_us_cycles_rows"] / max(st["grid_calls"], 1.0)},
                "host_part_ms_per_batch": (st["ms_submit_checks"] + st["ms_submit_prev_host_begin"] + st["ms_submit_device_queued"] +
                                           st["ms_grid_finder_joined"] + st["ms_refinement_queued"] + st["ms_collect_wait_refinement"] +
                                           st["ms_collect_boards_copied"]) / nb,
                "host_part_ms_per_batch_by_phase": {k[3:]: st[k] / nb for k in st if k.startswith("ms_")},
                "grid_finder_cpu_ms_per_batch_over_all_threads": (st["grid_us_graph"] + st["grid_us_adjacency"] + st["grid_us_sequences"] + st["grid_us_cycles_rows"]) / nb / 1e3,
                # the container's CPU quota as a floor: a batch of a process that is held to its quota cannot take less wall time than its CPU time /
                # the cores it may use (boxes that let a burst through come in a few per cent under it); `process_cpu_ms_per_batch` counts every thread (grid finder, submit / collect, the runtime's own)
                "process_cpu_ms_per_batch": cpu_ms,
                "cpu_quota_floor_ms_per_batch": (cpu_ms / quota) if quota else None,
                "device_part_ms_per_batch": {"first_pass": st["device_ms_first_pass"] / nb, "refinement": st["device_ms_refinement"] / nb,
                                             "note": "hipEvents; the two run on different streams and overlap each other and the host part"},
                "bound_by": "host" if (st["ms_grid_finder_joined"] / nb) > 0.5 * dt * 1e3 else "device",
                "what": "find_boards over the bench frames already in HBM, boards (gridn^2 refined corners per frame) on the host: "
                        "first pass (level images, responses + candidates of levels 3, 2, 1) on the device, grid finder on "
                        "the host threads under the next batch's first pass, refinement out of the cells around the corners"}
    finally:
        det.close()


def chess_pass_alone_leg(det, frames, launches=120, warm=10):
    import mrgingham_amd
    """north_star's sentence as a number: the plain ChESS pass -- mrgingham_amd_chess_response_batch(level 0, clamp 0),
    the literal output of mrgingham_ChESS_response_5 (ChESS.c:56-106) for every frame of the batch: u8 read once, int16
    written once, no clamp, no hot list, no level images -- alone on the device, every launch bracketed by hipEvents on
    the stream it runs on (torch's current stream: that is the stream the call is given).  NOT `value`."""
    B, H, W = frames.shape
    out = torch.empty((B, H, W), dtype=torch.int16, device=frames.device)
    det.sync()
    torch.cuda.synchronize()
    for _ in range(warm):
        det.chess_response(frames, 0, clamp=False, out=out)
    torch.cuda.synchronize()
    det.set_kernel_timing(2)                                 # the engine-clock probe alone: no events of the library on the stream
    det.sclk_mhz()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record()
    for i in range(launches):
        det.chess_response(frames, 0, clamp=False, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(launches))
    avg = ev[0].elapsed_time(ev[launches]) / launches        # back-to-back launches: includes the dispatch gaps
    med = per[launches // 2]
    alg = B * W * H * 3.0
    sclk = det.sclk_mhz()
    det.set_kernel_timing(False)
    del out
    # HBM-side bytes per launch: replayed from the committed rocprofv3 --pmc passes of this kernel alone
    # (profiles/chess_alone_traffic.json), like `roofline.traffic`: only for a library built from the same kernel sources
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "chess_alone_traffic.json")))
        kid = mrgingham_amd._lib.lib().mrgingham_amd_kernel_id().decode()
        if (tj["width"], tj["height"]) == (W, H):
            if tj.get("kernel_id") == kid:
                traffic, traffic_src = tj["bytes_per_pixel"] * B * W * H, tj["source"]
            else:
                traffic_src = f"none: {tj.get('source')} was collected on kernel sources {tj.get('kernel_id')}, this library is {kid}"
    except (OSError, KeyError, ValueError):
        pass
    return dict(clock_fields(alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, sclk), **{"traffic": traffic, "traffic_source": traffic_src,
            "kernel": "chess_v16_kernel<CLAMP 0> (plain ChESS response, the output of ChESS.c:56-106; sixteen pixels per lane, "
                      "mrgingham_amd/csrc/chess16.hip: the library's kernel for the response without a hot list)",
            "bytes_model": "3 B/px (u8 read once + int16 written once)", "bytes_per_launch": alg,
            "launches_timed": launches, "avg_launch_ms": avg, "median_launch_ms": med,
            "min_launch_ms": per[0], "p90_launch_ms": per[int(launches * 0.9)],
            "achieved": alg / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_median_launch": alg / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frames_per_s": B / (avg * 1e-3),
            "what": "the ChESS pass ALONE (nothing else on the device): `frac` from the average over back-to-back launches "
                    "(first event to last / launches, dispatch gaps included), `frac_median_launch` from the median of the "
                    "per-launch hipEvent intervals; `sclk_mhz` = the engine clock inside those launches, `frac_at_2400mhz` = "
                    "frac * 2400 / sclk_mhz (the kernel is bound by VALU issue: its time goes with 1 / clock)"})


def sparse_leg(det, frames, start_level, P, steps):
    """The same workload with option "sparse_refine" (include/mrgingham_amd.h): level images and the start level's
    response for whole frames, the response below it only in the cells around the points.  NOT `value`: the judged
    metric prices the dense per-level ChESS pass; this is what the same answer costs when that pass is not asked
    for.  The outputs are compared with the dense schedule's on every frame of the batch, inside this function."""
    B = frames.shape[0]
    want = det.chain(frames, start_level, P)
    det.set_option("sparse_refine", 1)
    try:
        outs = [tuple(torch.empty_like(t) for t in want) for _ in range(3)]
        det.chain(frames, start_level, P, out=outs[0])
        repeated = det.sparse_fallbacks()    # frames the sparse kernels handed to the dense ones (inside the call)
        same = bool(torch.equal(want[2], outs[0][2]))
        n = want[2].clamp(max=P).tolist()
        for f in range(B):
            same = same and bool(torch.equal(want[0][f, :n[f]], outs[0][0][f, :n[f]]) and
                                 torch.equal(want[1][f, :n[f]], outs[0][1][f, :n[f]]))
        for i in range(10):
            det.chain(frames, start_level, P, out=outs[i % 3], sync=False)
        det.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            det.chain(frames, start_level, P, out=outs[i % 3], sync=False)
        det.sync()
        dt = time.perf_counter() - t0
        return {"accepted": repeated == 0, "frames_repeated_densely": repeated, "identical_to_dense": same, "value": B * steps / dt, "unit": "frames/s",
                "ms_per_step": dt / steps * 1e3, "steps": steps, "scratch_GiB": det.scratch_bytes() / 2**30,
                "what": "option sparse_refine on the same frames: response below the start level only in the 16-px cells "
                        "around the points; outputs compared with the dense schedule's on every frame"}
    finally:
        det.set_option("sparse_refine", 0)


MAX_SCLK_MHZ = 2400.0  # MI355X peak engine clock (MI355X_MICROARCH.md)


def clock_fields(frac, sclk_mhz):
    """`sclk_mhz` + the fraction rescaled to the peak engine clock: the response kernels are bound by VALU issue, so
    their time goes with 1 / sclk -- a box that holds 2.1 GHz under this load reads 12 % lower than one that holds 2.4."""
    if not sclk_mhz or sclk_mhz <= 0:
        return {"sclk_mhz": None, "frac_at_2400mhz": None}
    return {"sclk_mhz": sclk_mhz, "frac_at_2400mhz": frac * MAX_SCLK_MHZ / sclk_mhz}


def power_state(local_rank):
    """Best effort, from sysfs (hwmon of the GPU's PCI device): power cap and the instantaneous average power in watts."""
    out = {"power_cap_w": None, "power_w": None}
    try:
        import glob
        bus = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(bus, "pci_domain_id", 0), bus.pci_bus_id, bus.pci_device_id)
        for hw in glob.glob("/sys/bus/pci/devices/" + bdf + "/hwmon/hwmon*"):
            for key, name in (("power_cap_w", "power1_cap"), ("power_w", "power1_average"), ("power_w", "power1_input")):
                try:
                    out[key] = out[key] if out[key] is not None else int(open(os.path.join(hw, name)).read()) / 1e6
                except (OSError, ValueError):
                    pass
    except Exception:
        pass
    return out


def c2_level0_leg(device_index, steps=150, warm=30):
    """BASELINE configs[1] as stated: 64 x 1920x1080, 10x10 board, image_pyramid_level = 0 -- one level-0 detect call per
    step (response + clamp + hot list -> connected components -> candidate list), pipelined like the timed steps; the
    level-0 launch on 3 B/px with its engine clock, and the plain ChESS pass alone at that size.  NOT `value`."""
    import mrgingham_amd
    from mrgingham_amd import synth
    W, H, B, gridn = 1920, 1080, 64, 10
    dev = torch.device("cuda", device_index)
    frames = synth.board_batch(B, W, H, gridn=gridn, seed0=0, device=dev)
    det = mrgingham_amd.Detector(device_index)
    try:
        outs = [det.detect(frames, 0, capacity=256, sync=True) for _ in range(3)]   # three output sets in rotation, like the main steps
        for i in range(warm):
            xy, counts = det.detect(frames, 0, sync=False, out=outs[i % 3])
        det.sync()
        # twice: without the library's kernel timing (what a caller gets: `value`), then with it (two events around every
        # response launch -- they cost the pixel stream of a 0.13-ms step several per cent: `level0_launch_ms`, `frac`, clock)
        t0 = time.perf_counter()
        for i in range(steps):
            xy, counts = det.detect(frames, 0, sync=False, out=outs[i % 3])
        det.sync()
        dt = time.perf_counter() - t0
        det.set_kernel_timing(True)
        det.chess_kernel_ms()
        det.sclk_mhz()
        t1 = time.perf_counter()
        for i in range(steps):
            xy, counts = det.detect(frames, 0, sync=False, out=outs[i % 3])
        det.sync()
        dt_timed = time.perf_counter() - t1
        kern_ms, nl = det.chess_kernel_ms()
        sclk = det.sclk_mhz()
        det.set_kernel_timing(False)
        frac = B * W * H * 3.0 / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kern_ms > 0 else 0.0
        alone = chess_pass_alone_leg(det, frames, launches=100, warm=10)
        return dict({"workload": f"{B} x {W}x{H} u8, {gridn}x{gridn} board, level-0 detect (BASELINE configs[1])",
                     "value": B * steps / dt, "unit": "frames/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
                     "ms_per_step_with_kernel_timing": dt_timed / steps * 1e3,
                     "frames_with_all_candidates_last_step": int((counts >= gridn * gridn).sum().item()),
                     "level0_launch_ms": kern_ms, "launches_timed": nl, "bytes_model": "3 B/px (u8 read once + int16 written once)",
                     "frac": frac,
                     "chess_pass_alone": {k: alone[k] for k in ("avg_launch_ms", "median_launch_ms", "frac", "frac_median_launch",
                                                                "sclk_mhz", "frac_at_2400mhz")}},
                    **clock_fields(frac, sclk))
    finally:
        det.close()


def preprocess_leg(det, frames, runs=60, prime=40):
    """The reference tool's DEFAULT chain in front of the detector (mrgingham-from-image.cc:71-111: normalize + CLAHE(8) +
    3x3 blur) on the bench frames, as the device runs it: tile histograms (1 B/px read), LUTs, then the blend and the blur
    in one pass (1 B/px read + 1 B/px written).  Events on the stream the call is given; compared with the two-kernel
    path (blend, then blur: what the oracle pins at test sizes) byte for byte on the whole batch.  NOT `value`."""
    B, H, W = frames.shape
    for _ in range(prime):                                   # (the engine clock ramps over the first dozens of passes of a leg)
        out = det.preprocess(frames, clahe=True, blur_radius=1)
    torch.cuda.synchronize()
    det.set_kernel_timing(2)                                 # the engine-clock probe (here: in the blend + blur kernel)
    det.sclk_mhz()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(runs + 1)]
    ev[0].record()
    for i in range(runs):
        out = det.preprocess(frames, clahe=True, blur_radius=1)
        ev[i + 1].record()
    torch.cuda.synchronize()
    sclk = det.sclk_mhz()
    det.set_kernel_timing(False)
    per = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(runs))
    ms = ev[0].elapsed_time(ev[runs]) / runs
    det.set_option("preprocess_fused", 0)
    try:
        two = det.preprocess(frames, clahe=True, blur_radius=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            two = det.preprocess(frames, clahe=True, blur_radius=1)
        e1.record()
        torch.cuda.synchronize()
        two_ms = e0.elapsed_time(e1) / 3
        same = bool(torch.equal(out, two))
    finally:
        det.set_option("preprocess_fused", 1)
    del two, out
    alg = B * W * H * 3.0
    # HBM-side bytes of the chain: replayed from the committed counter passes (profiles/preprocess_traffic.json), only for the
    # same frame shape and the same preprocess.hip
    traffic, traffic_src = None, None
    try:
        import hashlib
        tj = json.load(open(os.path.join(ROOT, "profiles", "preprocess_traffic.json")))
        sha = hashlib.sha256(open(os.path.join(ROOT, "mrgingham_amd", "csrc", "preprocess.hip"), "rb").read()).hexdigest()[:16]
        if (tj["frames"], tj["width"], tj["height"]) == (B, W, H):
            if tj.get("preprocess_hip_sha16") == sha:
                traffic, traffic_src = tj["bytes_per_call"], tj["source"]
            else:
                traffic_src = f"none: {tj.get('source')} was collected on another preprocess.hip"
    except (OSError, KeyError, ValueError):
        pass
    return {"traffic": traffic, "traffic_source": traffic_src,
            "workload": f"normalize + CLAHE(8) + 3x3 blur of {B} x {W}x{H} u8 (mrgingham-from-image.cc:71-111)",
            "ms_per_batch": ms, "median_ms_per_batch": per[runs // 2], "runs": runs, "frames_per_s": B / (ms * 1e-3),
            "bytes_model": "3 B/px (histograms: 1 read; blend + blur in one pass: 1 read + 1 written)", "bytes_per_batch": alg,
            "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "sclk_mhz": sclk or None,
            "ms_per_batch_at_2400mhz": ms * sclk / MAX_SCLK_MHZ if sclk else None,
            "kernels": "clahe_hist_kernel (LDS-atomic bound: one ds_add per pixel at the unit's 8 lanes per clock), clahe_lut_kernel, "
                       "clahe_quad_kernel, clahe_blur3_kernel (VALU-issue bound: ~15 instructions per pixel, scales with the engine clock)",
            "two_kernel_path_ms_per_batch": two_ms, "identical_to_two_kernel_path": same}


MIXED_RES = [(1280, 800), (1920, 1080), (2560, 1440), (4096, 2160), (4096, 3072)]   # 1 .. 12 MP


def c5_mixed_leg(device_index, nframes=400, gridn=10, unit=32, depth=3):
    """BASELINE configs[4] on ONE rank: a stream of `nframes` frames of mixed resolution (1-12 MP, seeded draw) through the
    full detector with per-frame adaptive pyramid depth (mrgingham.cc:106-140) -- per-resolution units of at most `unit`
    frames, `depth` units in flight (mrgingham_amd_find_boards_submit / _collect), heaviest first as the multi-rank
    work queue hands them out (mrgingham_amd.parallel.stream_units).  A sample of frames goes through the single-frame
    entry point (find_board) and must give the same board, double for double.  NOT `value`."""
    import random
    import numpy as np
    import mrgingham_amd
    from mrgingham_amd import parallel, synth
    dev = torch.device("cuda", device_index)
    rnd = random.Random(5)
    sizes = [MIXED_RES[rnd.randrange(len(MIXED_RES))] for _ in range(nframes)]
    units = parallel.stream_units(sizes, unit_frames=unit)
    frames_of = [torch.stack([synth.board_frame(wh[0], wh[1], gridn, seed=i, device=dev) for i in idx]) for wh, idx in units]
    det = mrgingham_amd.Detector(device_index)
    try:
        def run_once():
            recs, jobs = [], []
            for u, (wh, idx) in enumerate(units):
                jobs.append((u, det.find_boards_submit(frames_of[u], gridn=gridn)))
                if len(jobs) >= depth:
                    uu, job = jobs.pop(0)
                    recs.append((uu,) + det.find_boards_collect(job))
            while jobs:
                uu, job = jobs.pop(0)
                recs.append((uu,) + det.find_boards_collect(job))
            return recs
        run_once()                                           # allocations
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs = run_once()
        dt = time.perf_counter() - t0
        levels = np.concatenate([np.asarray(found, dtype=np.int64) for _, _, found in recs])
        # a sample through the single-frame entry point
        same, checked = True, 0
        for u, boards, found in recs[:: max(1, len(recs) // 6)]:
            img = frames_of[u][0].cpu().numpy()
            single = mrgingham_amd.find_board(img, gridn=gridn)
            checked += 1
            if found[0] < 0:
                same = same and single is None
            else:
                same = same and single is not None and bool(np.array_equal(single, boards[0]))
        mpx = sum(w * h for w, h in sizes) / 1e6
        return {"workload": f"{nframes} frames drawn from {MIXED_RES} (seed 5), {gridn}x{gridn} board, full detector "
                            f"(level search + grid finder + refinement), units of <= {unit} frames of one resolution, {depth} in flight",
                "value": nframes / dt, "unit": "frames/s", "seconds_per_pass": dt, "megapixels": mpx, "megapixels_per_s": mpx / dt,
                "units": len(units), "boards_found": int((levels >= 0).sum()), "found_at_level": np.bincount(levels[levels >= 0], minlength=4).tolist(),
                "single_frame_path_checked": checked, "identical_to_single_frame_path": same,
                "what": "one rank's view of BASELINE configs[4]; on N ranks the same units are pulled off one shared counter "
                        "(mrgingham_amd.parallel.WorkQueue, tools/mixed_stream_bench.py)"}
    finally:
        det.close()


def c1_tool_leg(device_index, nfiles=256, jobs=4):
    """BASELINE configs[0]: the command-line tool (mrgingham_amd/bin/mrgingham-amd-from-image, the reference's
    mrgingham-from-image.cc) on `nfiles` 640x480 PGM files in a RAM-backed directory with its DEFAULT options (CLAHE + blur,
    level search, refinement): wall time of the whole process (start + HIP initialisation + images), the steady-state rate
    from the difference between a long and a short run, and the vnlog rows of one file against the Python mirror's
    find_board on the preprocessed image.  NOT `value`."""
    import shutil
    import subprocess
    import tempfile
    import numpy as np
    import mrgingham_amd
    from mrgingham_amd import synth
    cli = os.path.join(ROOT, "mrgingham_amd", "bin", "mrgingham-amd-from-image")
    if not os.access(cli, os.X_OK):
        return {"skipped": "the tool is not built (make -C mrgingham_amd/csrc)"}
    W, H, gridn = 640, 480, 10
    d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        frames = synth.board_batch(nfiles, W, H, gridn, 0, device=torch.device("cuda", device_index)).cpu().numpy()
        names = []
        for i in range(nfiles):
            names.append(os.path.join(d, f"f{i:03d}.pgm"))
            with open(names[-1], "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (W, H))
                f.write(frames[i].tobytes())
        env = dict(os.environ, MRGINGHAM_AMD_DEVICE=str(device_index))

        def run(files):
            t0 = time.perf_counter()
            r = subprocess.run([cli, "--jobs", str(jobs)] + files, capture_output=True, text=True, env=env, timeout=120)
            return time.perf_counter() - t0, r
        # a short run (process start + HIP initialisation + a few images) and a long one (every file `reps` times over: the
        # difference of two sub-second wall times is the measurement, so it has to be a few tenths of a second)
        small, reps = max(nfiles // 8, 1), 4
        t_small, r_small = run(names[:small])
        t_all, r = run(names * reps)
        rows = {}
        for ln in r_small.stdout.splitlines():
            if ln.startswith("#"):
                continue
            name, x, y, lv = ln.split()
            rows.setdefault(name, []).append(None if x == "-" else (float(x), float(y)))
        found = sum(1 for ln in r.stdout.splitlines() if not ln.startswith("#") and ln.split()[1] != "-") // (gridn * gridn)
        want = mrgingham_amd.find_board(mrgingham_amd.preprocess(frames[0], clahe=True, blur_radius=1), gridn=gridn)
        got = rows.get(names[0])
        same = (want is None and got == [None]) or (want is not None and got is not None and len(got) == gridn * gridn and
                                                     None not in got and bool(np.abs(np.array(got) - want).max() < 1e-6))
        nlong = nfiles * reps
        return {"workload": f"{nfiles} PGM files of {W}x{H} in {os.path.dirname(names[0])}, default options, --jobs {jobs} (BASELINE configs[0])",
                "value": (nlong - small) / max(t_all - t_small, 1e-9), "unit": "images/s",
                "value_note": "steady state: (images of the long run - images of the short run) / (difference of their wall times); "
                              f"the long run names every file {reps} times",
                "images_long_run": nlong, "wall_s": t_all, "wall_s_short_run": t_small, "files_short_run": small,
                "images_per_s_with_process_start": nlong / t_all, "returncode": r.returncode,
                "boards_found": found, "vnlog_of_first_file_matches_find_board": bool(same)}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is ~1.2 ms: the first few dozen run below the steady-state rate (clocks / TLBs warming up),
    # so the defaults are long enough to measure the steady state and still finish in well under a minute
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3_4096x3072_chain", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-points", type=int, default=256)
    ap.add_argument("--distinct", type=int, default=0,
                    help="render only this many distinct frames and repeat them to fill the batch (used for the "
                         "rocprofv3 --pmc passes, where tracing the ~37k tiny kernels of the frame generator is "
                         "the bottleneck); default: every frame distinct")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--sparse-refine", action="store_true",
                    help="run the TIMED steps with option sparse_refine (not the judged configuration: `config.sparse_refine` "
                         "says so, and `roofline` then describes the kernel that reads the frames, pyramid_fast_kernel)")
    ap.add_argument("--no-find-boards", action="store_true",
                    help="skip the leg that times the full detector (level search + host grid finder + refinement; N = 1 only)")
    ap.add_argument("--no-sparse-leg", action="store_true",
                    help="skip the extra leg that times the same workload with option sparse_refine (N = 1 only)")
    ap.add_argument("--no-chess-alone", action="store_true",
                    help="skip the leg that times the plain ChESS pass alone (level 0, no clamp; N = 1 only)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` legs (BASELINE configs 1, 2, 5 and the preprocessing chain as short legs; N = 1 only)")
    ap.add_argument("--scratch-sets", type=int, default=0,
                    help="option scratch_sets of the library (0 = its default: chosen from the batch shape): calls' component "
                         "searches in flight")
    ap.add_argument("--force-gather", action="store_true",
                    help="with one rank: still create the (one-rank) RCCL group and issue the gather every step")
    ap.add_argument("--rehearse", action="store_true",
                    help="REHEARSAL of the N > 1 flow on ONE GPU: the N ranks share HIP device 0, the process group is "
                         "gloo and the packed corner lists go through pinned host memory for the gather.  Exercises the "
                         "launcher, per-rank shards, the cross-rank aggregation and the one-JSON-line rule; the line is "
                         "marked `rehearsal: true, backend: gloo` and is NOT a scaling measurement")
    ap.add_argument("--bind-numa", action="store_true",
                    help="bind this process to the cores of its GPU's NUMA node (always done for --gpus > 1)")
    ap.add_argument("--prime", type=int, default=30,
                    help="untimed set-up passes before the W warm-up steps (scratch allocation, clock ramp); reported")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if torch.cuda.device_count() < args.gpus and not (args.rehearse and torch.cuda.device_count() >= 1):
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) here")
        sys.exit(launch_ranks(args.gpus))                    # no launcher: start the ranks ourselves
    # stdout carries ONE line, the JSON: keep a private handle to it and point file descriptor 1 at stderr for
    # everything else in the process (RCCL prints a version banner to stdout, `make` of the oracle talks, ...)
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: mrgingham_amd has no CPU path")
    rehearse = bool(args.rehearse)
    if rehearse:
        local_rank = 0                                       # every rank of a rehearsal shares device 0
    torch.cuda.set_device(local_rank)
    binding = bind_rank_to_gpu_numa(local_rank) if (world > 1 or args.bind_numa) else None
    collective = world > 1 or args.force_gather
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import mrgingham_amd
    from mrgingham_amd import parallel, synth

    W, H, gridn, start_level, batch, background = WORKLOADS[args.workload]
    render = synth.cluttered_board_batch if background == "clutter" else synth.board_batch
    if args.batch > 0:
        batch = args.batch
    dev = torch.device("cuda", local_rank)
    # every rank renders its own shard of the global batch (seed = global frame index)
    lo, _ = parallel.shard_range(world * batch, rank, world)
    if args.distinct and args.distinct < batch:
        base = render(args.distinct, W, H, gridn=gridn, seed0=lo, device=dev)
        frames = base.repeat((batch + args.distinct - 1) // args.distinct, 1, 1)[:batch].contiguous()
        del base
    else:
        frames = render(batch, W, H, gridn=gridn, seed0=lo, device=dev)
    det = mrgingham_amd.Detector(local_rank)
    if args.scratch_sets:
        det.set_option("scratch_sets", args.scratch_sets)
    # `value` is the reference's schedule: the dense ChESS response of every level.  The library's default
    # (sparse_refine 1: below the start level only around the points, where that pays) is pinned OFF for it.
    det.set_option("sparse_refine", 1 if args.sparse_refine else 0)
    P = args.max_points
    # Output ring: consecutive steps overlap on the device (step N+1's pixel kernels run while step
    # N's component kernels and gather finish), so a step must not overwrite a predecessor whose
    # results may still be in use.  A buffer is reused three steps later, and only after the gather
    # that read it has completed (event recorded behind the gather, host-waited before reuse).
    NBUF = 3
    packs = [parallel.packed_outputs(batch, P, dev) for _ in range(NBUF)]   # (pack, points, levels, npoints)
    outs = [p[1:] for p in packs]
    # where the collectives' tensors live: on the device for RCCL; a rehearsal (gloo) moves them through pinned host memory
    cdev = torch.device("cpu") if rehearse else dev
    gathered = [torch.empty((world, packs[0][0].numel()), dtype=torch.uint8, device=cdev) if (collective and rank == 0)
                else None for _ in range(NBUF)]
    hpack = [torch.empty(packs[0][0].numel(), dtype=torch.uint8, pin_memory=True) for _ in range(NBUF)] if (collective and rehearse) else None
    consumed = [None] * NBUF
    torch.cuda.synchronize()
    nstep = [0]

    def step():
        k = nstep[0] % NBUF
        nstep[0] += 1
        if consumed[k] is not None:
            consumed[k].synchronize()                        # gather of three steps ago: long done
        pts, lv, npts = det.chain(frames, start_level=start_level, max_points=P, out=outs[k], sync=False)
        if collective and rehearse:
            det.stream_wait()
            hpack[k].copy_(packs[k][0], non_blocking=True)   # device -> pinned host behind the step ...
            torch.cuda.current_stream().synchronize()        # ... which gloo needs on the host before it can send
            parallel.gather_packed(hpack[k], dst=0, out=gathered[k], force=True)
        elif collective:
            det.stream_wait()                                # torch's stream waits for this step on the device
            parallel.gather_packed(packs[k][0], dst=0, out=gathered[k], force=True)   # the ONE collective of the path
            consumed[k] = torch.cuda.Event()
            consumed[k].record()
        return npts

    def fence():
        det.sync()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.prime):                              # set-up: first call allocates the scratch
        step()
    fence()
    ranks_seen = 1
    if collective:
        seen = torch.ones(1, dtype=torch.int32, device=cdev)
        dist.all_reduce(seen)
        ranks_seen = int(seen.item())
    for _ in range(args.warmup):
        step()
    det.set_kernel_timing(True)
    det.chess_kernel_ms()
    det.sclk_mhz()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        npts = step()
    fence()
    dt = time.perf_counter() - t0
    power = power_state(local_rank)                          # (right behind the timed steps)
    det.set_kernel_timing(False)
    kern_ms, nlaunch = det.chess_kernel_ms()
    sclk_mhz = det.sclk_mhz()                                # engine clock inside the timed level-0 launches

    if collective:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    found = int((npts >= gridn * gridn).sum().item())
    # Everything below is a LEG beside the timed region.  A leg that fails (a box without /dev/shm, a tool that was not
    # built, ...) reports {"error": ...} in its place: it must never cost the run its JSON line.
    def leg(fn, *a, **kw):
        try:
            if os.environ.get("MRG_BENCH_FAIL_LEG") == getattr(fn, "__name__", ""):   # (tests: a leg that breaks)
                raise RuntimeError("forced by MRG_BENCH_FAIL_LEG")
            return fn(*a, **kw)
        except Exception as e:                               # noqa: BLE001 -- reported, not swallowed
            import traceback
            sys.stderr.write("bench.py: leg %s failed:\n%s" % (getattr(fn, "__name__", "?"), traceback.format_exc()))
            try:
                torch.cuda.synchronize()
            except Exception:                                # noqa: BLE001
                pass
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    # the host-fed leg on EVERY rank (each rank feeds its own GPU from its own pinned buffer, all at once: what
    # the node's PCIe / memory topology gives when all GPUs are fed together); rank 0 reports min / max over ranks
    e2e = None
    if not args.no_end_to_end:
        if collective:
            dist.barrier()
        e2e = end_to_end(det, frames, start_level, P) if collective else leg(end_to_end, det, frames, start_level, P)
        if collective:
            mine = torch.tensor([e2e["h2d_GBs"], e2e["value"]], dtype=torch.float64, device=cdev)
            allr = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            h2d = [float(t[0]) for t in allr]
            e2e = dict(e2e, ranks=world, h2d_GBs_min=min(h2d), h2d_GBs_max=max(h2d), h2d_GBs_per_rank=h2d,
                       value=float(sum(float(t[1]) for t in allr)),
                       what=e2e["what"] + "; all ranks at once, `value` = sum over ranks")
    fused, merged = det.chain_info()                         # (of the timed steps: before the sparse leg makes its calls)
    scratch_gib = det.scratch_bytes() / 2**30                # (likewise: a context that runs sparse chains keeps a third set)
    alone = None
    if world == 1 and not args.no_chess_alone:
        alone = leg(chess_pass_alone_leg, det, frames)
    sparse = None
    if world == 1 and start_level >= 1 and not args.no_sparse_leg and not args.sparse_refine:
        sparse = leg(sparse_leg, det, frames, start_level, P, 100)   # (its own length: not the timed region of the contract)
    fboards = None
    if world == 1 and not args.no_find_boards:
        fboards = leg(find_boards_leg, local_rank, frames, gridn)
    configs = None
    if world == 1 and not args.no_configs:
        configs = {"preprocess": leg(preprocess_leg, det, frames)}
        torch.cuda.empty_cache()
        configs["c2_level0"] = leg(c2_level0_leg, local_rank)
        configs["c5_mixed_one_rank"] = leg(c5_mixed_leg, local_rank)
        torch.cuda.empty_cache()
        configs["c1_tool"] = leg(c1_tool_leg, local_rank)
        configs["what"] = ("the other BASELINE configs that fit one GPU, and the reference tool's preprocessing chain, as short "
                           "legs of the default command; none of them is `value`")
    bindings = None
    if collective and binding is not None:
        objs = [None] * world
        dist.all_gather_object(objs, binding)
        bindings = objs
    gather_ok = None
    if collective and rank == 0:                             # what rank 0 received equals what the ranks produced
        last = (nstep[0] - 1) % NBUF
        gp, gl, gn = parallel.unpack_outputs(gathered[last], batch, P)
        gather_ok = bool(torch.equal(gn[0].to(dev), packs[last][3]) and torch.equal(gp[0].to(dev), packs[last][1]))
        assert gather_ok, "gathered corner lists differ from rank 0's own"
    # which shard every rank worked on, as rank 0 received it: first global frame, corner count, a checksum of the corner
    # bytes -- and (rehearsal) rank 0 renders every other rank's shard itself, runs the chain on it and compares
    shards_seen, shards_differ, shards_verified = None, None, None
    if collective and rank == 0:
        last = (nstep[0] - 1) % NBUF
        gp, gl, gn = parallel.unpack_outputs(gathered[last], batch, P)
        shards_seen = []
        for r in range(world):
            rlo, _ = parallel.shard_range(world * batch, r, world)
            live = gp[r].reshape(batch, P * 2).to(torch.float64)
            shards_seen.append({"rank": r, "first_frame": rlo, "corners": int(gn[r].clamp(max=P).sum()),
                                "corner_checksum": float(live.nan_to_num().sum())})
        shards_differ = all(not torch.equal(gp[r], gp[0]) for r in range(1, world)) if world > 1 else None
        if rehearse and world > 1 and not args.distinct:
            shards_verified = True
            for r in range(1, world):
                rlo, _ = parallel.shard_range(world * batch, r, world)
                fr_r = render(batch, W, H, gridn=gridn, seed0=rlo, device=dev)
                wp, wl, wn = det.chain(fr_r, start_level=start_level, max_points=P)
                n = wn.clamp(max=P).tolist()
                ok = bool(torch.equal(gn[r].to(dev), wn))
                for f in range(batch):
                    ok = ok and bool(torch.equal(gp[r][f, :n[f]].to(dev), wp[f, :n[f]]) and
                                     torch.equal(gl[r][f, :n[f]].to(dev), wl[f, :n[f]]))
                shards_verified = shards_verified and ok
                del fr_r
            assert shards_verified, "a rank's gathered corner lists are not those of its shard of the global batch"
    if rank == 0:
        total_frames = world * batch * args.steps
        # level-0 ChESS launches per step = number of stream chunks; frames per launch follows
        launches_per_step = max(1, nlaunch // max(1, args.steps))
        frames_per_launch = batch / launches_per_step
        # 3 B/px: the frame read once, the int16 response written once.  When the launch also writes the
        # level images 1..3 (fused pyramid) those bytes are its algorithmic output too: 1/4 + 1/16 + 1/64 B/px.
        bpp = 3.0 + (sum(0.25 ** L for L in range(1, min(start_level, 3) + 1)) if fused else 0.0)
        sparse_step = merged < 0
        if sparse_step:      # the timed launch is pyramid_fast_kernel: the frame read once, the level images written once
            bpp = 1.0 + sum(0.25 ** L for L in range(1, min(start_level, 3) + 1))
        alg_bytes = frames_per_launch * W * H * bpp
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # HBM-side bytes per launch: PMC counters cannot be read from inside this process, so the
        # figure is the per-pixel traffic measured by the committed rocprofv3 --pmc passes of this
        # same command (profiles/chess_l0_traffic.json), scaled to this launch; null if absent.
        # Both replayed figures carry the id of the kernel sources they were collected on (tools/make_profiles.py); a
        # library built from other sources gets null, not somebody else's counters.
        kernel_id = mrgingham_amd._lib.lib().mrgingham_amd_kernel_id().decode()
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "chess_l0_traffic.json")))
            if (tj["width"], tj["height"]) == (W, H) and start_level >= 0 and not sparse_step:
                if tj.get("kernel_id") == kernel_id:
                    traffic = tj["bytes_per_pixel"] * frames_per_launch * W * H
                    traffic_src = tj["source"]
                else:
                    traffic_src = f"none: {tj.get('source')} was collected on kernel sources {tj.get('kernel_id')}, this library is {kernel_id}"
        except (OSError, KeyError, ValueError):
            pass
        # What binds the kernel (it moves its bytes once and is not waiting for them): the issue slots of the SIMDs.
        # From the committed SQ counter passes of this same command (profiles/chess_l0_valu.json); null if absent.
        valu = None
        try:
            vj = json.load(open(os.path.join(ROOT, "profiles", "chess_l0_valu.json")))
            if (vj["width"], vj["height"]) == (W, H) and not sparse_step and vj.get("kernel_id") == kernel_id:
                valu = {"bound": "valu_issue", "frac": vj["valu_issue_frac"], "unit": "fraction of SIMD quad-cycle issue slots "
                        "carrying a VALU instruction (4 waves per SIMD)", "valu_insts_per_512px": vj["valu_insts_per_wave_iteration"],
                        "wave_parked_frac": vj["wait_any_frac"], "source": vj["source"]}
        except (OSError, KeyError, ValueError):
            pass
        res = {
            "metric": "frames/sec, 4096x3072 10x10 board, corner-candidate path (ChESS + level decimation + "
                      "connected components), frames resident in HBM",
            "value": total_frames / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "kernel_id": kernel_id,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8->int16 (ChESS), int64/f64 (centroids)",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {batch} frames/GPU of {W}x{H} u8, {gridn}x{gridn} board, "
                                   f"detect at level {start_level} + refine to level 0, corner lists "
                                   f"{'gathered to rank 0' if collective else 'left on the device'}",
                       "frames_per_gpu": batch, "width": W, "height": H, "gridn": gridn, "background": background,
                       "start_level": start_level, "sparse_refine": bool(args.sparse_refine),
                       "sparse_refine_note": "the library's default (1: response below the start level only around the "
                                             "points, where that pays) is switched OFF for the timed steps: `value` prices "
                                             "the reference's dense per-level ChESS pass" if not args.sparse_refine else
                                             "timed WITH option sparse_refine: not the judged configuration",
                       "parallelism": f"frames sharded x{world}",
                       "frames_with_full_grid_last_step": found},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "pyramid_fast_kernel (level images 1..3 from the frames; option sparse_refine)" if sparse_step else
                                   "level-0 ChESS response (+clamp +hot-pixel compaction" +
                                   (" +level images 1..3)" if fused else ")"),
                         "bytes_model": ("%.4f B/px (u8 read once + u8 level images written once)" % bpp) if sparse_step else
                                        ("%.4f B/px (u8 read once + int16 written once + u8 level images 1..3 "
                                         "written once)" % bpp) if fused else
                                        "3 B/px (u8 read once + int16 written once)",
                         "bytes_per_launch": alg_bytes, "avg_launch_ms": kern_ms,
                         # the same launch on SURVEY.md 8d's two other ways of counting: its ChESS-pass model alone
                         # (3 B/px, as if the level images were free), and what the unfused schedule moves for the same
                         # outputs (ChESS pass 3 B/px + decimation pass 1 B/px read + the level images written)
                         "frac_3Bpx_chess_pass_model": None if sparse_step else
                                                       (frames_per_launch * W * H * 3.0 / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                                       if kern_ms > 0 else 0.0,
                         "frac_two_pass_equivalent": None if sparse_step else
                                                     (frames_per_launch * W * H * (bpp + (1.0 if fused else 0.0)) /
                                                      (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kern_ms > 0 else 0.0,
                         "launches_timed": nlaunch, "binding": valu},
        }
        res["roofline"].update(clock_fields(achieved / HBM_PEAK_GBS, sclk_mhz))
        res["roofline"].update(power)
        res["roofline"]["clock_note"] = ("sclk_mhz: engine clock inside the timed level-0 launches (workgroup 0 of every launch reads "
                                         "s_memtime and s_memrealtime: mrgingham_amd_sclk_mhz); frac_at_2400mhz = frac * 2400 / sclk_mhz: "
                                         "what the same instructions give at the part's peak clock -- the kernel is VALU-issue bound "
                                         "(`binding`), so a low sclk_mhz means a power- or thermally-limited box, not a slower kernel")
        res["gather_checked"] = gather_ok
        if shards_seen is not None:
            res["shards_seen"] = shards_seen
            res["shards_differ"] = shards_differ
            res["shards_verified_by_rerender"] = shards_verified
        if rehearse:
            res["rehearsal"] = True
            res["backend"] = "gloo"
            res["physical_gpus"] = 1
            res["metric"] = "REHEARSAL (%d ranks share ONE GPU, gloo through pinned host memory; not a scaling measurement): " % world + res["metric"]
        elif collective:
            res["backend"] = "nccl"
        res["scratch_sets"] = args.scratch_sets or "auto"
        res["setup_prime_steps"] = args.prime
        res["timed_region_s"] = dt
        res["notes"] = ("steps are queued back to back (streaming pipeline); the first few dozen passes of a process "
                        "run ~5-7 % below the steady state (clock ramp), which is why `setup_prime_steps` untimed "
                        "set-up passes precede the W warm-up steps.  The timed region is synchronised on both sides, so it "
                        "carries one pipeline fill and one drain (the last step's component chain runs with nothing "
                        "beside it): measured time = ~0.45 ms + K x 0.93 ms on the default workload (a single step is 1.28 ms end to end), "
                        "i.e. a 20-step region reads ~3 % below a 200-step one")
        res["scratch_GiB"] = scratch_gib
        if e2e is not None:
            res["end_to_end"] = e2e
        if alone is not None:
            res["chess_pass_alone"] = alone
        if sparse is not None:
            res["sparse_refine"] = sparse
        if fboards is not None:
            res["find_boards"] = fboards
        if configs is not None:
            res["configs"] = configs
        if bindings is not None or binding is not None:
            res["cpu_binding"] = bindings if bindings is not None else [binding]
        if world == 1 and not args.no_cpu_baseline:
            nhost = min(batch, 64)
            try:
                res["cpu_baseline"] = cpu_baseline(frames[:nhost].cpu().numpy(), start_level)
            except Exception as e:                           # noqa: BLE001 -- the line must come out; the leg says what broke
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": None, "kind": "port", "sample": None,
                                       "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        json_out.write(json.dumps(res) + "\n")
        json_out.flush()
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
