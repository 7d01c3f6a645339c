"""world_size-2 gloo run of the multi-GPU host logic (shards + the one gather)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    from mrgingham_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, P = 3, 8
    lo, hi = parallel.shard_range(world * B, rank, world)
    assert hi - lo == B
    pts = torch.arange(B * P * 2, dtype=torch.float64).reshape(B, P, 2) + 1000 * rank
    lv = (torch.arange(B * P, dtype=torch.int64).reshape(B, P) % 4).to(torch.int8) - rank
    npts = torch.tensor([rank + 1, rank + 2, rank + 3], dtype=torch.int32)
    out = parallel.gather_corner_lists(pts, lv, npts, dst=0)
    if rank == 0:
        gp, gl, gn = out
        ok = gp.shape == (world * B, P, 2) and gl.dtype == torch.int8
        for r in range(world):
            ok &= bool(torch.equal(gp[r * B:(r + 1) * B], torch.arange(B * P * 2, dtype=torch.float64).reshape(B, P, 2) + 1000 * r))
            ok &= bool(torch.equal(gl[r * B:(r + 1) * B], (torch.arange(B * P, dtype=torch.int64).reshape(B, P) % 4).to(torch.int8) - r))
            ok &= gn[r * B:(r + 1) * B].tolist() == [r + 1, r + 2, r + 3]
        ret.put(ok)
    else:
        assert out is None
    # the packed form bench.py uses: every output of a step lives in one byte buffer, one gather
    pack, ppts, plv, pnp = parallel.packed_outputs(B, P, "cpu")
    ppts.copy_(pts)
    plv.copy_(lv)
    pnp.copy_(npts)
    pre = torch.empty((world, pack.numel()), dtype=torch.uint8) if rank == 0 else None
    got = parallel.gather_packed(pack, dst=0, out=pre)
    if rank == 0:
        assert got is pre
        gp, gl, gn = parallel.unpack_outputs(got, B, P)
        ok2 = gp.shape == (world, B, P, 2) and gl.shape == (world, B, P) and gn.shape == (world, B)
        for r in range(world):
            ok2 &= bool(torch.equal(gp[r], torch.arange(B * P * 2, dtype=torch.float64).reshape(B, P, 2) + 1000 * r))
            ok2 &= bool(torch.equal(gl[r], (torch.arange(B * P, dtype=torch.int64).reshape(B, P) % 4).to(torch.int8) - r))
            ok2 &= gn[r].tolist() == [r + 1, r + 2, r + 3]
        ret.put(ok2)
    else:
        assert got is None
    # exact-size form: counts first, then grouped send / recv of exactly the live records
    ex = parallel.gather_exact(pts, lv, npts, dst=0)
    if rank == 0:
        ok3 = len(ex) == world
        for r in range(world):
            f, xy, l = ex[r]
            n_r = [r + 1, r + 2, r + 3]
            want_f = sum(([k] * n for k, n in enumerate(n_r)), [])
            rp = torch.arange(B * P * 2, dtype=torch.float64).reshape(B, P, 2) + 1000 * r
            rl = (torch.arange(B * P, dtype=torch.int64).reshape(B, P) % 4).to(torch.int8) - r
            ok3 &= f.tolist() == want_f and xy.shape == (sum(n_r), 2)
            ok3 &= bool(torch.equal(xy, torch.cat([rp[k, :n] for k, n in enumerate(n_r)])))
            ok3 &= bool(torch.equal(l, torch.cat([rl[k, :n] for k, n in enumerate(n_r)])))
        ret.put(ok3)
    else:
        assert ex is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    sys.path.insert(0, ROOT)
    from mrgingham_amd import parallel
    for n in (0, 1, 7, 64, 2048):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_gather_corner_lists_world2_gloo():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    ok = ret.get(timeout=120) and ret.get(timeout=120) and ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_lpt_plan_for_mixed_resolution_stream():
    sys.path.insert(0, ROOT)
    from mrgingham_amd import parallel
    import random
    rnd = random.Random(5)
    res = [(1280, 800), (1920, 1080), (2560, 1440), (4096, 2160), (4096, 3072)]
    sizes = [res[rnd.randrange(5)] for _ in range(200)]
    costs = [parallel.frame_cost(w, h) for (w, h) in sizes]
    assert abs(parallel.frame_cost(4096, 3072) / (4096 * 3072) - 1.328125) < 1e-9
    for world in (1, 2, 8):
        plan = parallel.lpt_assign(costs, world)
        assert sorted(i for p in plan for i in p) == list(range(200))          # a partition
        loads = [sum(costs[i] for i in p) for p in plan]
        assert max(loads) - min(loads) <= max(costs) + 1e-6                      # LPT bound
        assert max(loads) <= (4 / 3) * sum(costs) / world + 1e-6                 # classic 4/3 guarantee
        seen = []
        for r in range(world):
            groups = parallel.plan_mixed_stream(sizes, world, r)
            for sz, idx in groups.items():
                assert all(tuple(sizes[i]) == sz for i in idx) and idx == sorted(idx)
                seen += idx
        assert sorted(seen) == list(range(200))


def _queue_worker(rank, world, port, ret):
    """A stream whose true cost is off the model by up to 4x: the static LPT plan (computed on the model) leaves
    one rank idle for a large part of the run, the shared work counter does not."""
    import time
    sys.path.insert(0, ROOT)
    from mrgingham_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import random
    rnd = random.Random(11)
    RES = [(1280, 800), (1920, 1080), (2560, 1440), (4096, 2160), (4096, 3072)]
    sizes = [RES[rnd.randrange(len(RES))] for _ in range(96)]
    units = parallel.stream_units(sizes, unit_frames=4)
    model = [parallel.frame_cost(*wh) * len(idx) for wh, idx in units]
    # the static plan on the model, and then a truth that is deliberately wrong by 4x where it hurts that plan
    # most: the frames rank 0 was given happen to be the ones whose board is found at the first level tried (a
    # quarter of the modelled cost), rank 1's run all levels.  (Which frames stop early is a property of the
    # images, mrgingham.cc:127-138: no plan made before looking at them can know.)
    plan = parallel.lpt_assign(model, world)
    cheap = set(plan[0])
    true = [m * (0.25 if u in cheap else 1.0) for u, m in enumerate(model)]
    scale = 1.2 / sum(true)                                   # the whole stream is 1.2 s of "work"

    def run(my_units):
        t0 = time.perf_counter()
        done = []
        for u in my_units:
            time.sleep(true[u] * scale)
            done.append(u)
        return time.perf_counter() - t0, done

    dist.barrier()
    t_lpt, _ = run(plan[rank])
    # shared counter
    dist.barrier()
    q = parallel.WorkQueue(len(units), name="test/wq1")
    t_q, mine = run(iter(q))
    tt = torch.tensor([t_lpt, t_q], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    got = [None] * world
    dist.all_gather_object(got, mine)
    # two queues with the DEFAULT name back to back (the next stream, the next pass): each counts from its own key, so the
    # second does not start from the exhausted counter of the first
    dist.barrier()
    second = []
    for n in (7, 5):
        q2 = parallel.WorkQueue(n)
        mine2 = list(q2)
        got2 = [None] * world
        dist.all_gather_object(got2, mine2)
        second.append(sorted(u for g in got2 for u in g) == list(range(n)))
        dist.barrier()
    if rank == 0:
        ideal = 1.2 / world
        every = sorted(u for g in got for u in g)
        ret.put({"lpt": float(tt[0]) / ideal, "queue": float(tt[1]) / ideal, "complete": every == list(range(len(units))),
                 "both_worked": all(len(g) > 0 for g in got), "default_named_queues_back_to_back": second})
    dist.barrier()
    dist.destroy_process_group()


def test_work_queue_balances_where_the_static_plan_cannot():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_queue_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    res = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["complete"] and res["both_worked"], res       # every unit done exactly once
    assert res["default_named_queues_back_to_back"] == [True, True], res
    assert res["queue"] <= 1.15, res                          # within 15 % of the ideal makespan
    assert res["lpt"] >= 1.3, res                             # the static plan on the wrong model is not
    print(res)


def test_stream_units_and_single_process_queue():
    sys.path.insert(0, ROOT)
    from mrgingham_amd import parallel
    sizes = [(640, 480)] * 5 + [(1920, 1080)] * 3 + [(640, 480)] * 2
    units = parallel.stream_units(sizes, unit_frames=4)
    assert sorted(i for _, idx in units for i in idx) == list(range(10))
    assert all(len(idx) <= 4 and len({sizes[i] for i in idx}) == 1 for _, idx in units)
    assert units[0][0] == (1920, 1080)                        # heaviest unit first
    q = parallel.WorkQueue(len(units))                        # no process group: a local counter
    assert list(q) == list(range(len(units))) and q.next() is None
