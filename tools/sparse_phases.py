"""Phase clock of the LDS refinement kernel inside SPARSE chains (experiment build: make -C mrgingham_amd/csrc EXPERIMENT=1,
MRGINGHAM_AMD_LIB=.../libmrgingham_amd_experiment.so): call after call synchronised ("alone": the latency of one call) and
pipelined; microseconds of the first band of the first frame at level 0.  python tools/sparse_phases.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import mrgingham_amd
from mrgingham_amd import synth
W, H, B, P = 4096, 3072, 64, 256
gridn = int(sys.argv[1]) if len(sys.argv) > 1 else 10   # (a 5x5 board: what one of four workgroups per frame would see of a 10x10)
frames = synth.board_batch(B, W, H, gridn, 0, device='cuda')
det = mrgingham_amd.Detector(0)
det.set_option("cc_lds", 1 | 512)
det.set_option("sparse_refine", 2)
names = ["bands planned", "load + label", "R1 seeds", "R2 groups", "R3 demand + neighbour table", "R4 fills", "rest"]
outs = [None] * 3
for mode in ("alone", "pipeline"):
    for rep in range(3):
        n = 60
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            outs[i % 3] = det.chain(frames, 3, P, sync=(mode == "alone"))
        det.sync(); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        t = det.debug_refine_clock()
        print(f"{mode}: {dt*1e3:.3f} ms/step; refine level {t[11]}: {t[8]} hot pixels, {t[9]} points, {t[10]} band(s): " +
              ", ".join(f"{nm} {(t[k + 1] - t[k]) / 100:.1f}" for k, nm in enumerate(names)) + f"  (us; total {(t[7] - t[0]) / 100:.1f})")
