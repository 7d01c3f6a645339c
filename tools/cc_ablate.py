"""Timing ablations of the LDS detect kernel (results are wrong by construction): python tools/cc_ablate.py MODE [gridn]
under rocprofv3 --kernel-trace; MODE = value of the "cc_lds" option (1 = everything; | 2 no variance test, | 4 no
fills, | 8 stop after band planning + load + labelling of the first band)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
W, H, B = 4096, 3072, 64
mode = int(sys.argv[1])
gridn = int(sys.argv[2]) if len(sys.argv) > 2 else 10
frames = synth.board_batch(8, W, H, gridn, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
det = mrgingham_amd.Detector(0)
det.set_option("cc_lds", mode)
for level in (3, 1, 0):
    for rep in range(10):
        det.detect(frames, level, capacity=1024)
