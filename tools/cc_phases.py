"""Phase clock of the LDS refine kernel (option cc_lds = 1 | 512, mrgingham_amd_debug_refine_clock):
python tools/cc_phases.py [gridn] [W H]   -- refinement level by level, alone on the GPU; microseconds of the
first band of the first frame."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
gridn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 3072)
B, P = 64, 1024
frames = synth.board_batch(8, W, H, gridn, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
det = mrgingham_amd.Detector(0)
det.set_option("cc_lds", 1 | 512)
xy, counts = det.detect(frames, 3, capacity=P, sync=True)
p = (xy.to(torch.float64) / 1000.0).contiguous()
lv = torch.full((B, P), 3, dtype=torch.int8, device='cuda')
names = ["bands planned", "load + label", "R1 seeds", "R2 groups", "R3 demand + neighbour table", "R4 fills", "rest"]
for L in (2, 1, 0):
    for rep in range(2):
        pp, ll = p.clone(), lv.clone()
        det.refine(frames, L, pp, ll, counts, sync=True)
    t = det.debug_refine_clock()
    print(f"refine level {t[11]}: {t[8]} hot pixels, {t[9]} points, {t[10]} band(s): " +
          ", ".join(f"{n} {(t[k + 1] - t[k]) / 100:.1f}" for k, n in enumerate(names)) + f"  (us; total {(t[7] - t[0]) / 100:.1f})")
    det.refine(frames, L, p, lv, counts, sync=True)
