"""The plain ChESS pass alone -- mrgingham_amd_chess_response_batch(level 0, clamp 0), the output of ChESS.c:56-106 -- on
64 frames of 4096x3072 (or W H B from the command line), 120 launches back to back with nothing else on the device:
the target of `rocprofv3 --kernel-trace --stats` for profiles/rNN_chess_alone_kernel_trace.txt, and the same figure
bench.py prints as `chess_pass_alone` (hipEvents)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import mrgingham_amd
from mrgingham_amd import synth

W, H, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 3072, 64)
det = mrgingham_amd.Detector(0)
frames = synth.board_batch(B, W, H, 10, 0, device="cuda")
print(json.dumps(dict(bench.chess_pass_alone_leg(det, frames), width=W, height=H, frames=B)))
