"""Level-0 ChESS response kernel alone (32 frames 4096x3072, clamp, no hot list): the target of the SQ counter passes."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
det = mrgingham_amd.Detector(0)
W,H,B = 4096,3072,32
frames = synth.board_batch(4, W, H, 10, 0, device='cuda').repeat(B//4,1,1).contiguous()
out = torch.empty((B,H,W), dtype=torch.int16, device='cuda')
if len(sys.argv)>1: det.set_option("chess_v0", int(sys.argv[1]))
for _ in range(3): det.chess_response(frames, 0, clamp=True, out=out)
torch.cuda.synchronize()
