"""Timing of the preprocessing kernels (normalize + CLAHE + blur) on 64 frames of 4096x3072.
python tools/preprocess_bench.py [calls]   (a small number of calls for the rocprofv3 counter passes: every dispatch is serialised there)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import mrgingham_amd
from mrgingham_amd import synth
det = mrgingham_amd.Detector(0)
W,H,B = 4096,3072,64
frames = synth.board_batch(4, W, H, 10, 0, device='cuda').repeat(B//4,1,1).contiguous()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for clahe, blur in [(True,1),(True,0),(False,1)]:
    for _ in range(30 if N >= 40 else 1): out = det.preprocess(frames, clahe=clahe, blur_radius=blur)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): out = det.preprocess(frames, clahe=clahe, blur_radius=blur)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/N
    print(f"clahe={clahe} blur={blur}: {ms:.3f} ms per {B} frames -> {B/ms*1e3:.0f} frames/s, {B*W*H/ms/1e6:.0f} Mpx/ms-ish")
