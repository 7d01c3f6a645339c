import sys, os, time
sys.path.insert(0, os.getcwd()); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import mrgingham_amd
from mrgingham_amd import synth
det = mrgingham_amd.Detector(0)
B = 16
fr = synth.board_batch(B, 4096, 3072, 10, 0, device="cuda")
pre = det.preprocess(fr, clahe=True, blur_radius=1)
torch.cuda.synchronize()
for name, f in (("raw", fr), ("clahe+blur", pre)):
    for L in (3, 2, 1, 0):
        r = det.chess_response(f[:2], L, clamp=True)
        hot = (r > 15).sum(dim=(1, 2)).tolist()
        print(name, "level", L, "hot pixels per frame", hot, flush=True)
    for _ in range(2):
        b, fd = det.find_boards(f, gridn=10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): b, fd = det.find_boards(f, gridn=10)
    dt = (time.perf_counter() - t0) / n
    print(name, f"find_boards {B} frames: {dt*1e3:.2f} ms -> {B/dt:.0f} frames/s, found levels {np.bincount(fd[fd>=0], minlength=4).tolist()} none {(fd<0).sum()}", flush=True)
    jobs = []
    for _ in range(3): jobs.append(det.find_boards_submit(f, gridn=10))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        jobs.append(det.find_boards_submit(f, gridn=10)); det.find_boards_collect(jobs.pop(0))
    dt = (time.perf_counter() - t0) / n
    while jobs: det.find_boards_collect(jobs.pop(0))
    print(name, f"pipelined depth 3: {dt*1e3:.2f} ms per batch -> {B/dt:.0f} frames/s", flush=True)
    t0 = time.perf_counter()
    for _ in range(5): det.chain(f, 3, 1024)
    print(name, f"chain (default options): {(time.perf_counter()-t0)/5*1e3:.2f} ms per batch", flush=True)
