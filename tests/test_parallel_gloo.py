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
