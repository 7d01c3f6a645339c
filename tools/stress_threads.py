"""Stress of what round 4 added, for races and leaks: several threads, each with its own Detector, run pipelined
find_boards / chains concurrently (ctypes releases the GIL); contexts are created and destroyed with batches in flight;
the single-image wrappers are hammered from many threads (contexts spread over the devices).  Every result is compared
with the one computed up front.  python tools/stress_threads.py [seconds]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import mrgingham_amd
from mrgingham_amd import synth

T = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
ref = mrgingham_amd.Detector(0)
ref.set_option("find_boards_pipeline", 0)
shapes = [(640, 480, 6), (1280, 960, 4), (1920, 1080, 3), (4096, 3072, 2)]
batches = [synth.board_batch(B, W, H, 10, 11 * i, device="cuda") for i, (W, H, B) in enumerate(shapes)]
want = [ref.find_boards(b, gridn=10) for b in batches]
chains = [tuple(t.cpu().numpy() for t in ref.chain(b, 3, 256)) for b in batches]
imgs = [b[0].cpu().numpy() for b in batches]
single = [mrgingham_amd.find_board(im) for im in imgs]
bad, done = [], [0] * 8
stop = time.time() + T


def worker(k):
    try:
        it = 0
        while time.time() < stop:
            det = mrgingham_amd.Detector(0)
            det.set_option("sparse_refine", (0, 1, 2)[(k + it) % 3])
            jobs = []
            for r in range(6):
                i = (k + it + r) % len(batches)
                jobs.append((i, det.find_boards_submit(batches[i], gridn=10, nthreads=2)))
                if len(jobs) > (k % 3):
                    j, job = jobs.pop(0)
                    gb, gf = det.find_boards_collect(job)
                    if not (np.array_equal(gf, want[j][1]) and np.array_equal(gb, want[j][0])):
                        bad.append(("boards", k, it, j))
                if r % 3 == 2:
                    p, l, n = [t.cpu().numpy() for t in det.chain(batches[i], 3, 256)]
                    cp, cl, cn = chains[i]
                    if not np.array_equal(n, cn) or any(not np.array_equal(p[f, :cn[f]], cp[f, :cn[f]]) for f in range(len(cn))):
                        bad.append(("chain", k, it, i))
            if it % 2 == 0:
                for _, job in jobs:
                    det.find_boards_collect(job)
            det.close()                                   # (odd iterations: destroyed with batches in flight)
            got = mrgingham_amd.find_board(imgs[(k + it) % len(imgs)])
            if not np.array_equal(got, single[(k + it) % len(imgs)]):
                bad.append(("single", k, it))
            it += 1
            done[k] = it
    except Exception as e:                                # noqa
        bad.append(("exception", k, repr(e)))


ths = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
for t in ths: t.start()
for t in ths: t.join()
free, total = torch.cuda.mem_get_info()
print(f"{sum(done)} iterations on 6 threads in {T:.0f} s, {len(bad)} bad results {bad[:5]}; device memory in use at the end: {(total - free) / 2**30:.1f} GiB")
sys.exit(1 if bad else 0)
