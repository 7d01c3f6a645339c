"""BASELINE config 5 end to end: a mixed-resolution stream (1 MP .. 12 MP) with per-frame adaptive pyramid
depth, balanced over the ranks with the LPT plan, boards gathered to rank 0.

  python tools/mixed_stream_bench.py [--frames 200]                      # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P tools/mixed_stream_bench.py --frames 1600       # N GPUs, one process each

Every rank computes the same plan without communicating (parallel.plan_mixed_stream), renders its own
frames, runs mrgingham_amd_find_boards_batch once per resolution it was assigned, and sends
(frame index, found level, board) records to rank 0 in one gather.  Prints ONE JSON line on rank 0."""
import argparse, json, os, random, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

RES = [(1280, 800), (1920, 1080), (2560, 1440), (4096, 2160), (4096, 3072)]   # SURVEY.md 8d, all divisible by 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--gridn", type=int, default=10)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--balance", choices=["queue", "lpt"], default="queue",
                    help="queue: ranks pull per-resolution sub-batches off a shared counter (parallel.WorkQueue); "
                         "lpt: the static plan on the cost model 1.328*W*H")
    ap.add_argument("--unit", type=int, default=32, help="frames per work unit (queue)")
    ap.add_argument("--depth", type=int, default=3, help="work units in flight per rank (find_boards_submit / _collect; 1 = one at a time)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import mrgingham_amd
    from mrgingham_amd import parallel, synth
    rnd = random.Random(5)
    sizes = [RES[rnd.randrange(len(RES))] for _ in range(args.frames)]
    det = mrgingham_amd.Detector(local)
    N = args.gridn * args.gridn
    passes = [0]
    if args.balance == "lpt":
        mine = parallel.plan_mixed_stream(sizes, world, rank)              # {(w, h): [frame indices]}
        batches = {}
        for (w, h), idx in mine.items():
            batches[(w, h)] = (idx, torch.stack([synth.board_frame(w, h, args.gridn, seed=i, device=dev) for i in idx]))

        def run_once():
            recs = []
            for (w, h), (idx, frames) in batches.items():
                for lo in range(0, len(idx), 64):                          # sub-batches of at most 64 frames
                    boards, found = det.find_boards(frames[lo:lo + 64], gridn=args.gridn)
                    recs.append((idx[lo:lo + 64], found, boards))
            return recs
    else:
        # Dynamic balance: every rank can reach every frame (here: has rendered it; in a deployment: reads it from
        # the host / the shared store when it pulls the unit) and pulls per-resolution sub-batches off ONE shared
        # counter, heaviest first.  The cost of a frame is only known once its grid has been found
        # (mrgingham.cc:127-138), so whoever is free takes the next unit.
        units = parallel.stream_units(sizes, unit_frames=args.unit)
        frames_of = {}
        for wh, idx in units:
            frames_of[idx[0]] = torch.stack([synth.board_frame(wh[0], wh[1], args.gridn, seed=i, device=dev) for i in idx])

        def run_once():
            recs = []
            q = parallel.WorkQueue(len(units), name=f"mixed_stream/{passes[0]}")
            passes[0] += 1
            jobs = []                                                      # up to three units in flight per rank

            def collect():
                idx, job = jobs.pop(0)
                boards, found = det.find_boards_collect(job)
                recs.append((idx, found, boards))                          # (per unit: frame indices, found levels, boards)
            for u in q:
                wh, idx = units[u]
                jobs.append((idx, det.find_boards_submit(frames_of[idx[0]], gridn=args.gridn)))
                if len(jobs) >= args.depth:
                    collect()
            while jobs:
                collect()
            return recs
    torch.cuda.synchronize()

    run_once()                                                             # warm-up (allocations)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.repeat):
        recs = run_once()
        # one gather of fixed-size records: [frame, level, 2N coordinates]
        mine_n = sum(len(idx) for idx, _, _ in recs)
        pack = torch.full((args.frames, 2 + 2 * N), -1.0, dtype=torch.float64, device=dev)
        if mine_n:
            arr = np.concatenate([np.concatenate([np.asarray(idx, dtype=np.float64)[:, None], np.asarray(found, dtype=np.float64)[:, None],
                                                  np.nan_to_num(np.asarray(boards, dtype=np.float64).reshape(len(idx), 2 * N), nan=-1.0)], axis=1)
                                  for idx, found, boards in recs])
            pack[:mine_n] = torch.from_numpy(arr).to(dev)
        if world > 1:
            bufs = [torch.empty_like(pack) for _ in range(world)] if rank == 0 else None
            dist.gather(pack, bufs, dst=0)
        else:
            bufs = [pack]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / args.repeat
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        allrec = torch.cat(bufs).cpu().numpy()
        allrec = allrec[allrec[:, 0] >= 0]
        levels = allrec[:, 1].astype(int)
        mpx = sum(w * h for w, h in sizes) / 1e6
        loads = [sum(parallel.frame_cost(*sizes[i]) for i in p) for p in parallel.lpt_assign([parallel.frame_cost(w, h) for w, h in sizes], world)]
        print(json.dumps({"metric": "frames/sec, mixed-resolution stream, full detector with adaptive pyramid depth", "value": args.frames / dt,
                          "unit": "frames/s", "n_gpus": world, "frames": args.frames, "megapixels": mpx, "seconds_per_pass": dt,
                          "records_on_rank0": int(len(allrec)), "found_at_level": np.bincount(levels[levels >= 0], minlength=4).tolist(),
                          "not_found": int((levels < 0).sum()), "balance": args.balance, "units_in_flight": args.depth if args.balance == "queue" else 1,
                          "lpt_model_imbalance": max(loads) / (sum(loads) / world)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
