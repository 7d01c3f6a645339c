"""Chain loop with one ChESS launch per level (option multi_level_launch 0) or merged (1): for kernel traces.
python tools/levels_trace.py [multi 0/1] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import mrgingham_amd
from mrgingham_amd import synth
multi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
frames = synth.board_batch(8, 4096, 3072, 10, 0, device="cuda").repeat(8, 1, 1).contiguous()
det = mrgingham_amd.Detector(0)
det.set_option("multi_level_launch", multi)
out = det.chain(frames, 3, 256)
outs = [out] + [tuple(torch.empty_like(o) for o in out) for _ in range(2)]
for i in range(steps):
    det.chain(frames, 3, 256, out=outs[i % 3], sync=False)
det.sync()
