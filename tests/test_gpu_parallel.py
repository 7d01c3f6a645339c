"""The multi-GPU path on whatever GPUs this box has: bench.py's own step (chain -> stream_wait -> the one
gather over RCCL) launched exactly the way the driver launches it -- `python bench.py --gpus N` with no
launcher in the environment -- with one rank (collective forced through a one-rank RCCL group) and,
when a second device is present, two.  Plus BASELINE config 5 at its stated resolutions on one rank:
mixed 1-12 MP stream, LPT plan for 8 ranks, per-frame adaptive pyramid depth (find_boards)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import mrgingham_amd
from mrgingham_amd import parallel, synth
from oracle import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]   # stdout is ONE json line (rank 0), nothing else
    return json.loads(lines[0])


SMALL = ["--workload", "c1_640x480_chain", "--steps", "6", "--warmup", "2", "--prime", "3", "--no-cpu-baseline",
         "--no-end-to-end"]


def test_bench_step_with_the_collective_on_one_rank():
    res = _bench("--gpus", "1", "--force-gather", *SMALL)
    assert res["n_gpus"] == 1 and res["ranks_seen"] == 1 and res["steps"] == 6
    assert res["gather_checked"] is True                   # rank 0 compared what it received with what it sent
    assert "gathered to rank 0" in res["config"]["workload"]
    assert res["roofline"]["launches_timed"] == 6 and res["value"] > 0


def test_bench_under_a_launcher_environment():
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment, as torch.distributed.run sets them."""
    res = _bench("--gpus", "1", *SMALL, env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert res["n_gpus"] == 1 and res["ranks_seen"] == 1


def test_bench_line_survives_a_leg_that_breaks():
    """The legs beside the timed region (chess_pass_alone, sparse_refine, find_boards, configs.*, end_to_end) run guarded: one that
    throws is reported in its place and the ONE json line still comes out with every key of the contract."""
    res = _bench("--gpus", "1", "--workload", "c1_640x480_chain", "--steps", "6", "--warmup", "2", "--prime", "3", "--no-cpu-baseline",
                 "--no-find-boards", "--no-sparse-leg", env_extra={"MRG_BENCH_FAIL_LEG": "c5_mixed_leg"})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in res, key
    assert res["config"]["workload"].startswith("c1_640x480_chain") and res["value"] > 0 and res["steps"] == 6
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "sclk_mhz", "frac_at_2400mhz"):
        assert key in res["roofline"], key
    cfg = res["configs"]
    assert cfg["c5_mixed_one_rank"] == {"error": "RuntimeError: forced by MRG_BENCH_FAIL_LEG"}
    assert cfg["c2_level0"]["value"] > 0 and cfg["c2_level0"]["frames_with_all_candidates_last_step"] == 64
    assert cfg["preprocess"]["identical_to_two_kernel_path"] is True
    assert cfg["c1_tool"].get("vnlog_of_first_file_matches_find_board", True) is True and "error" not in cfg["c1_tool"]
    assert res["end_to_end"]["value"] > 0 and res["chess_pass_alone"]["frac"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices")
def test_bench_launches_its_own_two_ranks():
    res = _bench("--gpus", "2", *SMALL)
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2
    assert res["scaling"] == "weak"


def test_bench_rehearsal_two_ranks_on_one_gpu():
    """The whole N > 1 flow of bench.py with world = 2 on the ONE GPU there is (`--rehearse`: both ranks on device 0,
    gloo, the packed lists through pinned host memory): the self-launcher, per-rank shards of the global batch (seed =
    global frame index), `ranks_seen`, the gather compared with rank 0's own lists AND with a re-render of rank 1's
    shard, the per-rank host-fed leg aggregated over ranks, one JSON line on stdout -- marked so that it can never be
    read as a scaling measurement."""
    res = _bench("--gpus", "2", "--rehearse", "--workload", "c1_640x480_chain", "--steps", "6", "--warmup", "2",
                 "--prime", "3", "--no-cpu-baseline")
    assert res["rehearsal"] is True and res["backend"] == "gloo" and res["physical_gpus"] == 1
    assert res["metric"].startswith("REHEARSAL")
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["scaling"] == "weak"
    assert res["gather_checked"] is True
    sh = res["shards_seen"]
    assert [x["rank"] for x in sh] == [0, 1] and [x["first_frame"] for x in sh] == [0, 64]
    assert res["shards_differ"] is True                    # rank 1 did not work on rank 0's frames
    assert sh[0]["corner_checksum"] != sh[1]["corner_checksum"] and min(x["corners"] for x in sh) >= 64 * 50   # (640x480: ~63 candidates per frame at level 3)
    assert res["shards_verified_by_rerender"] is True       # ... but on frames 64..127 of the global batch
    e = res["end_to_end"]
    assert e["ranks"] == 2 and len(e["h2d_GBs_per_rank"]) == 2 and min(e["h2d_GBs_per_rank"]) > 0
    assert res["config"]["frames_per_gpu"] == 64 and "gathered to rank 0" in res["config"]["workload"]
    assert len(res["cpu_binding"]) == 2
    assert "find_boards" not in res and "sparse_refine" not in res and "chess_pass_alone" not in res   # N = 1 legs


def _rccl():
    """The RCCL of this process (PyTorch's), through ctypes: (library, communicator of ONE rank)."""
    import ctypes
    import glob
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + ["librccl.so.1", "librccl.so"]
    lib = None
    for c in cands:
        try:
            lib = ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if lib is None:
        pytest.skip("no librccl in this environment")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    assert lib.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    return lib, comm


def test_gather_rccl_c_entry_on_a_one_rank_communicator():
    """mrgingham_amd_gather_rccl, the exchange of a C++ host that runs one process per GPU: chain_batch into ONE packed
    block (mrgingham_amd_packed_layout = the layout of parallel.packed_outputs), ncclGather of it on a stream, queued
    behind the chain on the device.  One rank here (a communicator made with RCCL's own C API, not torch.distributed);
    what arrives is what the chain wrote."""
    import ctypes
    torch.cuda.set_device(0)
    lib, comm = _rccl()
    det = mrgingham_amd.Detector(0)
    try:
        B, P = 5, 256
        L = det.L
        o_lv, o_np, nbytes = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        assert L.mrgingham_amd_packed_layout(B, P, ctypes.byref(o_lv), ctypes.byref(o_np), ctypes.byref(nbytes)) == 0
        pack, pts, lv, npts = parallel.packed_outputs(B, P, "cuda:0")
        assert nbytes.value == pack.numel() and o_lv.value == B * P * 16                      # one layout on both sides
        assert lv.data_ptr() - pack.data_ptr() == o_lv.value and npts.data_ptr() - pack.data_ptr() == o_np.value
        frames = synth.board_batch(B, 640, 480, 10, 40, device="cuda:0")
        gathered = torch.zeros((1, pack.numel()), dtype=torch.uint8, device="cuda:0")
        st = torch.cuda.Stream(torch.device("cuda", 0))
        torch.cuda.synchronize()
        for rep in range(3):                                       # asynchronous: chain -> gather, back to back
            det.chain(frames, 3, P, out=(pts, lv, npts), sync=False)
            rc = L.mrgingham_amd_gather_rccl(det.ctx, comm, 0, pack.data_ptr(), pack.numel(), gathered.data_ptr(), st.cuda_stream)
            assert rc == 0, L.mrgingham_amd_last_error(det.ctx)
        st.synchronize()
        det.sync()
        want = det.chain(frames, 3, P)
        gp, gl, gn = parallel.unpack_outputs(gathered, B, P)
        assert torch.equal(gn[0], want[2]) and int(gn[0].min()) >= 50
        for f in range(B):
            n = int(want[2][f])
            assert torch.equal(gp[0][f, :n], want[0][f, :n]) and torch.equal(gl[0][f, :n], want[1][f, :n])
        # argument errors come back as codes with a message, nothing is queued
        assert L.mrgingham_amd_gather_rccl(det.ctx, None, 0, pack.data_ptr(), pack.numel(), gathered.data_ptr(), st.cuda_stream) == -1
        assert b"gather_rccl" in L.mrgingham_amd_last_error(det.ctx)
    finally:
        det.close()
        lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        lib.ncclCommDestroy(comm)


def test_bench_refuses_more_ranks_than_devices():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *SMALL],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0 and "HIP device" in r.stderr


def test_bench_end_to_end_leg_small():
    res = _bench("--gpus", "1", "--workload", "c1_640x480_chain", "--steps", "4", "--warmup", "1", "--prime", "2",
                 "--no-cpu-baseline")
    e = res["end_to_end"]
    assert e["value"] > 0 and e["h2d_GBs"] > 0 and e["frames_with_points_last_step"] == 64


def test_exact_size_gather_matches_the_padded_one_on_device():
    det = mrgingham_amd.Detector(0)
    try:
        frames = synth.board_batch(3, 640, 480, gridn=10, seed0=0, device="cuda")
        pts, lv, npts = det.chain(frames, 3, 256)
        (f, xy, l), = parallel.gather_exact(pts, lv, npts)
        k = 0
        for b in range(3):
            n = int(npts[b])
            assert f[k:k + n].tolist() == [b] * n
            assert torch.equal(xy[k:k + n], pts[b, :n]) and torch.equal(l[k:k + n], lv[b, :n])
            k += n
        assert k == len(f)
    finally:
        det.close()


def test_c5_mixed_stream_at_baseline_resolutions_one_rank_of_eight():
    """BASELINE config 5 as stated (mixed 1-12 MP stream, per-frame adaptive pyramid depth, load balanced
    over 8 GPUs), on the one GPU there is: the 8-rank LPT plan is computed, rank 3's share is run as one
    find_boards batch per resolution, and every board equals what the single-frame path gives."""
    import random
    rnd = random.Random(11)
    res = [(1280, 800), (1920, 1080), (2560, 1440), (4096, 2160), (4096, 3072)]
    sizes = [res[rnd.randrange(5)] for _ in range(64)]
    costs = [parallel.frame_cost(w, h) for (w, h) in sizes]
    plan = parallel.lpt_assign(costs, 8)
    loads = [sum(costs[i] for i in p) for p in plan]
    assert max(loads) / (sum(loads) / 8) < 1.15                        # balanced to within 15 %
    det = mrgingham_amd.Detector(0)
    try:
        nfound = 0
        for (w, h), idx in parallel.plan_mixed_stream(sizes, 8, 3).items():
            frames = torch.stack([synth.board_frame(w, h, 10, seed=i, device="cuda") for i in idx])
            boards, found = det.find_boards(frames, gridn=10)
            for j, i in enumerate(idx):
                img = frames[j].cpu().numpy()
                want = mrgingham_amd.find_board(img, gridn=10)
                if want is None:
                    assert found[j] < 0, (w, h, i)
                    continue
                assert found[j] >= 0 and np.array_equal(boards[j], want), (w, h, i)
                nfound += 1
                # the level the batch stopped at is the first of 3, 2, 1, 0 with a grid (mrgingham.cc:127-138):
                # no grid among the oracle's candidates of any higher level
                for L in range(3, int(found[j]), -1):
                    cand = oracle.find_corners(img, L)
                    assert len(cand) < 100 or mrgingham_amd.find_grid_from_points(cand, 10) is None
        assert nfound >= 4
    finally:
        det.close()


def test_bench_c4_shard_of_256_frames_with_the_gather():
    """BASELINE config 4 as a bench workload: 256 frames per GPU (2048 at N = 8), the step ends in the ONE gather
    of the corner lists (a one-rank RCCL group here), the host-fed leg runs on the rank with the collective up,
    and the rank binds itself to its GPU's NUMA node."""
    res = _bench("--gpus", "1", "--workload", "c4_4096x3072_shard256", "--force-gather", "--bind-numa", "--steps", "3",
                 "--warmup", "1", "--prime", "2", "--no-cpu-baseline")
    assert res["config"]["frames_per_gpu"] == 256 and res["scaling"] == "weak"
    assert res["gather_checked"] is True and res["config"]["frames_with_full_grid_last_step"] == 256
    e = res["end_to_end"]
    assert e["ranks"] == 1 and e["h2d_GBs_min"] > 0 and e["h2d_GBs_min"] <= e["h2d_GBs_max"]
    assert e["frames_with_points_last_step"] == 256
    assert isinstance(res["cpu_binding"], list) and len(res["cpu_binding"]) == 1
    assert res["scratch_GiB"] <= 32.0


def test_bench_cluttered_workload_small_run():
    res = _bench("--gpus", "1", "--workload", "c3_cluttered", "--batch", "8", "--steps", "3", "--warmup", "1",
                 "--prime", "2", "--no-cpu-baseline", "--no-end-to-end")
    assert res["config"]["background"] == "clutter" and res["config"]["frames_with_full_grid_last_step"] == 8


def test_cpu_baseline_harness_reports_efficiency():
    import bench
    fr = synth.board_batch(4, 640, 480, 10, 0).numpy()
    b = bench.cpu_baseline(fr, 3, leg_seconds=0.3)
    assert b["kind"] == "port" and b["value"] >= b["tall_frames_s"] > 0
    # (level 3 of a 640x480 frame is too coarse for all 100 corners: the oracle finds 60-odd candidates there)
    want = np.mean([len(oracle.find_corners(f, 3)) for f in fr])
    assert 0 < b["parallel_efficiency"] <= 3.0 and abs(b["candidates_per_frame"] - want) < 1.0
    if oracle.have_reference_build():
        assert b["upstream_chess_level0_frames_s_tall"] > 0


def test_mixed_stream_bench_pulls_units_off_the_shared_counter():
    """tools/mixed_stream_bench.py with the dynamic balance (the default): every frame of the stream is processed
    exactly once and every board found, with the counter in a (one-rank) process group's store."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for balance in ("queue", "lpt"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mixed_stream_bench.py"), "--frames", "24",
                            "--repeat", "1", "--unit", "4", "--balance", balance], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert res["balance"] == balance and res["records_on_rank0"] == 24 and res["not_found"] == 0, res
        assert sum(res["found_at_level"]) == 24
