"""Phase clock of the refinement kernel inside a sparse chain (experiment build: cc_lds 1 | 512): the level-0 kernel
of the last call, first frame, microseconds.  python tools/sparse_phases.py [sparse 0/1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = synth.board_batch(8, 4096, 3072, 10, 0, device="cuda").repeat(8, 1, 1).contiguous()
det = mrgingham_amd.Detector(0)
det.set_option("cc_lds", 1 | 512)
det.set_option("sparse_refine", 2 * mode)
names = ["plan (sparse: masks -> list, cells)", "load + label", "R1 seeds", "R2 groups", "R3 demand + neighbour table", "R4 fills", "rest"]
for rep in range(3):
    det.chain(frames, 3, 1024)
    t = det.debug_refine_clock()
    print(f"sparse={mode} level {t[11]}: {t[8]} hot pixels, {t[9]} points, {t[10]} band(s): " +
          ", ".join(f"{n} {(t[k + 1] - t[k]) / 100:.1f}" for k, n in enumerate(names)) + f"  (us; total {(t[7] - t[0]) / 100:.1f})")
