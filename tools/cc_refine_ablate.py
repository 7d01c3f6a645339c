"""Timing ablations of the LDS refine kernel (results are wrong by construction): python tools/cc_refine_ablate.py [gridn]
Host-timed refine calls alone on the GPU, per level, for cc_lds = 1 (everything), 3 (no variance test), 5 (no fills),
9 (band planning + load + labelling of the first band only); the pixel kernels of the level are in every figure."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
gridn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
W, H, B, P = 4096, 3072, 64, 1024
frames = synth.board_batch(8, W, H, gridn, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
ref = mrgingham_amd.Detector(0)
xy, counts = ref.detect(frames, 3, capacity=P, sync=True)
pts0 = (xy.to(torch.float64) / 1000.0).contiguous()
# inputs of every level from an unablated chain
inputs = {}
p = pts0.clone(); lv = torch.full((B, P), 3, dtype=torch.int8, device='cuda')
for L in (2, 1, 0):
    inputs[L] = (p.clone(), lv.clone())
    ref.refine(frames, L, p, lv, counts, sync=True)
for mode in (1, 3, 5, 9, 0):
    det = mrgingham_amd.Detector(0)
    det.set_option("cc_lds", mode)
    out = []
    for L in (2, 1, 0):
        best = 1e9
        for rep in range(5):
            p, lv = inputs[L][0].clone(), inputs[L][1].clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); det.refine(frames, L, p, lv, counts, sync=True); best = min(best, time.perf_counter() - t0)
        out.append(best * 1e3)
    print(f"cc_lds {mode}: refine L2 / L1 / L0 = {out[0]:.3f} / {out[1]:.3f} / {out[2]:.3f} ms", flush=True)
    det.close()
