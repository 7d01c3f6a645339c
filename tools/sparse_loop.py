"""chain() loop in one mode (for profilers).  usage: python tools/sparse_loop.py sparse W H B gridn clutter [steps]"""
import sys, time
import torch
sys.path.insert(0, ".")
from mrgingham_amd import Detector, synth
mode, W, H, B, gridn, clutter = [int(x) for x in sys.argv[1:7]]
steps = int(sys.argv[7]) if len(sys.argv) > 7 else 30
sets = int(sys.argv[8]) if len(sys.argv) > 8 else 0
serial = int(sys.argv[9]) if len(sys.argv) > 9 else 0      # 1: wait for every step (kernel times without overlap)
dev = torch.device("cuda:0")
frames = (synth.cluttered_board_batch if clutter else synth.board_batch)(B, W, H, gridn, 0, device=dev)
det = Detector(0)
if mode:
    det.set_option("sparse_refine", 2 * mode)   # 2: always, whatever the size of the call
if sets:
    det.set_option("scratch_sets", sets)
out = det.chain(frames, 3, 1024)
outs = [out] + [tuple(torch.empty_like(o) for o in out) for _ in range(3)]
for i in range(4):
    det.chain(frames, 3, 1024, out=outs[i % 4], sync=False)
det.sync()
t0 = time.perf_counter()
for i in range(steps):
    det.chain(frames, 3, 1024, out=outs[i % 4], sync=bool(serial))
det.sync()
same = all(torch.equal(out[2], o[2]) and torch.equal(out[0][:, :64], o[0][:, :64]) for o in outs[1:])
print("outputs of the steps in flight equal the first call's:", same)
print(f"sparse={mode} {W}x{H} B={B} gridn={gridn} clutter={clutter} sets={sets}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
