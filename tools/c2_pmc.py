"""Target of the rocprofv3 counter passes for BASELINE config 2: a few level-0 detect steps (chess_v1_kernel<hot> + the
component search) and a few plain responses (chess_v16_kernel) on 64 frames of 1920x1080.  python tools/c2_pmc.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
det = mrgingham_amd.Detector(0)
fr = synth.board_batch(4, 1920, 1080, 10, 0, device="cuda").repeat(16, 1, 1).contiguous()
out = torch.empty((64, 1080, 1920), dtype=torch.int16, device="cuda")
for _ in range(4):
    det.detect(fr, 0, capacity=256)
    det.chess_response(fr, 0, clamp=False, out=out)
torch.cuda.synchronize()
print("done")
