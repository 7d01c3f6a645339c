"""Multi-GPU at the C boundary (include/mrgingham_amd.h, "several GPUs"): mrgingham_amd_chain_multi -- shards over
several contexts, one gather into the first context's device --, the thread -> device mapping of the reference-symbol
wrappers, page-locked host frames, and the command-line tool's --gpus.  On a one-GPU box the contexts of a multi call
share the device (same code path up to the kind of copy); the genuinely two-device cases skip themselves."""
import os
import subprocess
import threading

import numpy as np
import pytest
import torch

import mrgingham_amd
from mrgingham_amd import api, synth
from oracle import oracle

pytestmark = pytest.mark.gpu
NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0


def _same(a, b):
    pa, la, na = [t.cpu().numpy() for t in a]
    pb, lb, nb = [t.cpu().numpy() for t in b]
    assert np.array_equal(na, nb)
    for f in range(len(na)):
        n = int(na[f])
        assert np.array_equal(la[f, :n], lb[f, :n]) and np.array_equal(pa[f, :n], pb[f, :n]), f


@pytest.mark.parametrize("nctx,devices", [(2, "same"), (3, "same"), (2, "two")])
def test_chain_multi_gathers_the_shards_in_frame_order(nctx, devices):
    if devices == "two" and NDEV < 2:
        pytest.skip("needs two GPUs")
    devs = [k % 2 if devices == "two" else 0 for k in range(nctx)]
    frames = synth.board_batch(7, 1024, 768, 10, 20, device="cuda:0")
    frames[5] = synth.noise_frame(1024, 768, 3, smooth=1, device="cuda:0")       # a frame without a board
    one = mrgingham_amd.Detector(0)
    dets = [mrgingham_amd.Detector(d) for d in devs]
    try:
        want = one.chain(frames, 3, 512)
        ranges = [api.shard_range(7, k, nctx) for k in range(nctx)]
        shards = [frames[a:a + c].to(f"cuda:{d}").contiguous() for (a, c), d in zip(ranges, devs)]
        got = api.chain_multi(dets, shards, 3, 512)
        assert got[0].device == torch.device("cuda", 0)
        _same(want, got)
        wp, wl = oracle.chain(frames[6].cpu().numpy(), 3)                          # and the oracle on the last shard's last frame
        n = int(got[2][6])
        assert n == len(wp) and np.array_equal(got[0][6, :n].cpu().numpy(), wp) and np.array_equal(got[1][6, :n].cpu().numpy(), wl)
        # an empty shard, calls back to back without a sync in between (each with its own outputs), a device-side wait
        shards2 = [frames, frames[:0]] + [frames[:0]] * (nctx - 2)
        shards2 = [s.to(f"cuda:{d}").contiguous() for s, d in zip(shards2, devs)]
        _same(want, api.chain_multi(dets, shards2, 3, 512))      # (synchronous first: the noise frame makes the tables of
        #                                                           the context that has not seen it grow -- one retry)
        outs = [api.chain_multi(dets, shards if i % 2 == 0 else shards2, 3, 512, sync=False) for i in range(4)]
        ctxs = (api.ctypes.c_void_p * nctx)(*[d.ctx for d in dets])
        st = torch.cuda.current_stream(torch.device("cuda", 0))
        assert one.L.mrgingham_amd_stream_wait_multi(ctxs, nctx, st.cuda_stream) == 0
        tail = outs[-1][2].clone()                                                 # on torch's stream: behind the gathers
        assert one.L.mrgingham_amd_sync_multi(ctxs, nctx) == 0
        for o in outs:
            _same(want, o)
        assert torch.equal(tail, want[2])
    finally:
        one.close()
        for d in dets:
            d.close()


def test_chain_multi_queues_eight_shards_in_the_time_of_one():
    """Eight contexts (on the one device there is), eight shards: every shard is queued by its own context's submit
    thread, so the CALL costs about what queueing one chain costs (~70 us of host time each: 0.56 ms from one thread);
    the lists equal one chain over all the frames."""
    import time
    nctx, B, P = 8, 64, 256
    frames = synth.board_batch(B, 640, 480, 10, 100, device="cuda:0")
    one = mrgingham_amd.Detector(0)
    dets = [mrgingham_amd.Detector(0) for _ in range(nctx)]
    try:
        want = one.chain(frames, 3, P)
        ranges = [api.shard_range(B, k, nctx) for k in range(nctx)]
        shards = [frames[a:a + c].contiguous() for a, c in ranges]
        got = api.chain_multi(dets, shards, 3, P)                  # (also: first call, scratch allocation)
        _same(want, got)
        L = one.L
        frs = (mrgingham_amd._lib.Frames * nctx)()
        for k, (d, fr) in enumerate(zip(dets, shards)):
            frs[k] = d._frames(fr)[0]
        ctxs = (api.ctypes.c_void_p * nctx)(*[d.ctx for d in dets])
        outs = [(torch.empty((B, P, 2), dtype=torch.float64, device="cuda:0"), torch.empty((B, P), dtype=torch.int8, device="cuda:0"),
                 torch.empty((B,), dtype=torch.int32, device="cuda:0")) for _ in range(3)]
        torch.cuda.synchronize()
        issue = []
        for i in range(30):
            o = outs[i % 3]
            t0 = time.perf_counter()
            assert L.mrgingham_amd_chain_multi(ctxs, nctx, frs, 3, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), P) == 0
            issue.append(time.perf_counter() - t0)
            if i % 3 == 2:
                assert L.mrgingham_amd_sync_multi(ctxs, nctx) == 0
        assert L.mrgingham_amd_sync_multi(ctxs, nctx) == 0
        for o in outs:
            _same(want, o)
        t_one = []
        o = outs[0]
        for i in range(30):                                        # the same frames as ONE chain on one context, for scale
            t0 = time.perf_counter()
            one.chain(frames, 3, P, out=o, sync=False)
            t_one.append(time.perf_counter() - t0)
            if i % 3 == 2:
                one.sync()
        one.sync()
        med, med1 = sorted(issue[5:])[len(issue[5:]) // 2], sorted(t_one[5:])[len(t_one[5:]) // 2]
        print(f"chain_multi over 8 contexts: {med * 1e6:.0f} us per call (one chain_batch through the Python mirror: {med1 * 1e6:.0f} us)")
        assert med < 0.30e-3, (med, med1)                         # (0.10-0.15 ms measured; eight sequential submissions: ~0.6 ms)
    finally:
        one.close()
        for d in dets:
            d.close()


def test_chain_multi_argument_errors():
    det = mrgingham_amd.Detector(0)
    try:
        fr = synth.board_batch(2, 640, 480, 10, 0, device="cuda:0")
        with pytest.raises(RuntimeError):
            api.chain_multi([det, det], [fr[:1], fr[1:]], 3, 256)                 # one context per shard
        with pytest.raises(RuntimeError):
            api.chain_multi([det], [fr], 11, 256)                                  # level out of range
    finally:
        det.close()


def test_calling_threads_spread_over_the_devices():
    """The k-th thread that calls a reference symbol gets device k % devices (MRGINGHAM_AMD_DEVICE unset), a thread
    can choose, and every thread's find_points gives the same candidates."""
    assert "MRGINGHAM_AMD_DEVICE" not in os.environ
    img = synth.board_frame(640, 480, 10, 0).numpy()
    want = mrgingham_amd.find_points(img, 0)
    res = {}

    current = {}

    def work(k, choose):
        if choose is not None:
            api.set_thread_device(choose)
        torch.cuda.set_device(0)                                 # the caller's own current device ...
        res[k] = (api.thread_device(), mrgingham_amd.find_points(img, 0))
        current[k] = torch.cuda.current_device()                 # ... is what it was when the wrappers return

    ths = [threading.Thread(target=work, args=(k, None)) for k in range(4)]
    for t in ths:
        t.start(); t.join()                                                        # (one after the other: creation order = k)
    devs = [res[k][0] for k in range(4)]
    assert all(0 <= d < NDEV for d in devs)
    assert [(d - devs[0]) % NDEV for d in devs] == [k % NDEV for k in range(4)], devs   # consecutive threads, consecutive devices
    assert all(np.array_equal(res[k][1], want) for k in range(4))
    assert all(current[k] == 0 for k in range(4)), current      # (the contexts of threads 1, 3 live on device 1 when there is one)
    t = threading.Thread(target=work, args=(9, NDEV - 1)); t.start(); t.join()
    assert res[9][0] == NDEV - 1 and np.array_equal(res[9][1], want)
    with pytest.raises(ValueError):
        api.set_thread_device(NDEV)


def test_device_variable_still_wins():
    code = ("import mrgingham_amd, numpy as np\nfrom mrgingham_amd import api, synth\nimport threading\nout = []\n"
            "def w():\n    out.append(api.thread_device())\n"
            "for k in range(3):\n    t = threading.Thread(target=w); t.start(); t.join()\nprint(out)\n")
    env = dict(os.environ, MRGINGHAM_AMD_DEVICE="0")
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-1500:]
    assert r.stdout.strip().endswith("[0, 0, 0]")


def test_pinned_host_frames_give_the_same_results():
    img = synth.board_frame(1280, 960, 10, 2).numpy()
    pin = api.PinnedArray(img.shape, np.uint8)
    try:
        pin.array[...] = img
        assert np.array_equal(mrgingham_amd.find_points(pin.array, 0), mrgingham_amd.find_points(img, 0))
        assert np.array_equal(mrgingham_amd.find_board(pin.array), mrgingham_amd.find_board(img))
        assert np.array_equal(mrgingham_amd.ChESS_response_5(pin.array), mrgingham_amd.ChESS_response_5(img))
        L = mrgingham_amd._lib.lib()                                                # registering memory the caller owns
        buf = np.ascontiguousarray(img)
        assert L.mrgingham_amd_host_register(buf.ctypes.data, buf.nbytes) == 0
        assert np.array_equal(mrgingham_amd.find_points(buf, 1), mrgingham_amd.find_points(img, 1))
        assert L.mrgingham_amd_host_unregister(buf.ctypes.data) == 0
    finally:
        pin.close()


def _write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0])); f.write(img.tobytes())


@pytest.mark.parametrize("gpus", ["1", "all", "2"])
def test_cli_gpus_option(tmp_path, gpus):
    """--gpus N|all: worker k on device k % N; more than the node has is cut down with a warning; the vnlog does
    not depend on it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "mrgingham_amd", "bin", "mrgingham-amd-from-image")
    for i in range(5):
        _write_pgm(str(tmp_path / f"f{i}.pgm"), synth.board_frame(640, 480, 10, i).numpy())
    base = subprocess.run([cli, "--jobs", "1", str(tmp_path / "f*.pgm")], capture_output=True, text=True, timeout=300)
    assert base.returncode == 0, base.stderr[-1000:]
    r = subprocess.run([cli, "--jobs", "3", "--gpus", gpus, str(tmp_path / "f*.pgm")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    rec = lambda out: sorted(l for l in out.splitlines() if not l.startswith("#"))
    assert rec(r.stdout) == rec(base.stdout) and len(rec(r.stdout)) == 500
    if gpus == "2" and NDEV < 2:
        assert "using 1" in r.stderr
    bad = subprocess.run([cli, "--gpus", "0", str(tmp_path / "f0.pgm")], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "--gpus" in bad.stderr


@pytest.mark.gpu
def test_wait_policy_is_accepted_or_refused_cleanly_and_changes_no_result():
    """mrgingham_amd_set_wait_policy inside a process whose HIP runtime is already running (PyTorch's): 0, or
    MRGINGHAM_AMD_ERR_DEVICE when the runtime refuses -- never a crash, and the detector answers as before."""
    import mrgingham_amd
    from mrgingham_amd import synth
    img = synth.board_frame(640, 480, 10, 1).numpy()
    before = mrgingham_amd.find_board(img, gridn=10)
    for policy in (3, 2, 0):
        assert api.set_wait_policy(policy) in (0, -2), policy
    after = mrgingham_amd.find_board(img, gridn=10)
    assert before is not None and np.array_equal(before, after)
    assert api.set_wait_policy(7) == -1
