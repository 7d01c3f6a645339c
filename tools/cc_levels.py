"""Component-search kernels level by level, alone on the GPU (host-timed, one call at a time):
python tools/cc_levels.py [gridn] [cc_lds]   -- prints hot pixels, points, and ms per detect / refine call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mrgingham_amd
from mrgingham_amd import synth
gridn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
W, H, B, P = 4096, 3072, 64, 1024
frames = synth.board_batch(8, W, H, gridn, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
det = mrgingham_amd.Detector(0)
if len(sys.argv) > 2:
    det.set_option("cc_lds", int(sys.argv[2]))
from scipy import ndimage
for L in range(4):
    r = det.chess_response(frames[:1], L, clamp=True)[0].cpu().numpy()
    hot = r > 15
    lab, n = ndimage.label(hot)
    sizes = np.bincount(lab.ravel())[1:]
    print(f"level {L}: hot {hot.sum()}, components {n}, largest {np.sort(sizes)[-5:].tolist()}, >=2 px: {(sizes >= 2).sum()}", flush=True)
xy, counts = det.detect(frames, 3, capacity=P, sync=True)
print("level-3 candidates per frame:", counts[:8].tolist())
pts = (xy.to(torch.float64) / 1000.0).contiguous()
for rep in range(3):
    t0 = time.perf_counter(); det.detect(frames, 3, capacity=P, sync=True); t_det = time.perf_counter() - t0
    p = pts.clone(); lv = torch.full((B, P), 3, dtype=torch.int8, device='cuda'); n = counts.clone()
    ts = []
    for L in (2, 1, 0):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); det.refine(frames, L, p, lv, n, sync=True); ts.append(time.perf_counter() - t0)
        paths = det.debug_paths(L, 8).tolist()
        if rep == 2:
            print(f"  refine level {L}: paths {paths}, refined to this level: {(lv[:8] == L).sum(1).tolist()}")
    print(f"rep {rep}: detect L3 {t_det*1e3:.3f} ms (includes the level-3 pixel kernels); refine L2/L1/L0 "
          f"{ts[0]*1e3:.3f} / {ts[1]*1e3:.3f} / {ts[2]*1e3:.3f} ms (each includes that level's pixel kernels)", flush=True)
