"""Timing ablations of the LDS detect kernel (results are wrong by construction): python tools/cc_ablate.py MODE
under rocprofv3 --kernel-trace; MODE = value of the "cc_lds" option."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
W, H, B = 4096, 3072, 64
mode = int(sys.argv[1])
frames = synth.board_batch(8, W, H, 10, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
det = mrgingham_amd.Detector(0)
det.set_option("cc_lds", mode)
for level in (3, 0):
    for rep in range(10):
        det.detect(frames, level, capacity=512)
