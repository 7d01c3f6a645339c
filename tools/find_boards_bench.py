"""Throughput of the full detector over a device-resident batch (mrgingham_amd_find_boards_batch:
per-frame adaptive pyramid depth, device candidates + refinement, host grid-finder threads)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mrgingham_amd
from mrgingham_amd import synth
det = mrgingham_amd.Detector(0)
for (W, H, B) in [(4096, 3072, 64), (1920, 1080, 64), (640, 480, 64)]:
    frames = synth.board_batch(B, W, H, 10, 0, device='cuda')
    for nthreads in (0, 16, 4, 1):
        boards, found = det.find_boards(frames, gridn=10, nthreads=nthreads)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            boards, found = det.find_boards(frames, gridn=10, nthreads=nthreads)
        dt = (time.perf_counter() - t0) / n
        print(f"{W}x{H} x{B}  grid-finder threads {(str(nthreads) if nthreads else 'auto'):>4s}: {dt*1e3:8.2f} ms per batch -> {B/dt:8.0f} frames/s; "
              f"found at levels {np.bincount(found[found >= 0], minlength=4).tolist()}, not found {int((found < 0).sum())}")
    del frames
