"""Blob path timing: python tools/blob_time.py  -- find_points(blobs=True) on circle-grid and noise frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mrgingham_amd
from mrgingham_amd import synth
cases = [("dots 4096x3072", synth.dots_frame(4096, 3072, 10, 1, device="cuda")),
         ("dots 1920x1080", synth.dots_frame(1920, 1080, 10, 2, device="cuda")),
         ("dots 640x480", synth.dots_frame(640, 480, 10, 3, device="cuda")),
         ("noise 2048x1536 smooth 2", synth.noise_frame(2048, 1536, 3, smooth=2, device="cuda")),
         ("board 4096x3072", synth.board_frame(4096, 3072, 10, 4, device="cuda"))]
for name, fr in cases:
    img = fr.cpu().numpy()
    pts = mrgingham_amd.find_points(img, 0, blobs=True)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        pts = mrgingham_amd.find_points(img, 0, blobs=True)
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: {len(pts)} blobs, {min(ts):.2f} ms (best of 5), median {sorted(ts)[2]:.2f}", flush=True)
