"""chess_v16_kernel / chess_v1_kernel alone for the counter passes: python tools/chess16_pmc.py [0|16]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
det = mrgingham_amd.Detector(0)
W, H, B = 4096, 3072, 32
frames = synth.board_batch(4, W, H, 10, 0, device='cuda').repeat(B // 4, 1, 1).contiguous()
out = torch.empty((B, H, W), dtype=torch.int16, device='cuda')
det.set_option("chess_variant", (int(sys.argv[1]) or 1) if len(sys.argv) > 1 else 16)
for _ in range(3): det.chess_response(frames, 0, clamp=False, out=out)
torch.cuda.synchronize()
