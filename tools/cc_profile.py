"""Hot-pixel statistics of the bench workload and the component kernels alone vs in the pipeline
(run under rocprofv3 --kernel-trace --stats for the per-kernel durations)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import mrgingham_amd
from mrgingham_amd import synth
W, H, B, P = 4096, 3072, 64, 256
frames = synth.board_batch(8, W, H, 10, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
det = mrgingham_amd.Detector(0)
for L in range(4):
    r = det.chess_response(frames[:8], L, clamp=True)
    hot = (r > 15)
    n = hot.flatten(1).sum(1).float()
    # hot pixels with at least one hot 4-neighbour
    nb = torch.zeros_like(hot)
    nb[:, 1:] |= hot[:, :-1]; nb[:, :-1] |= hot[:, 1:]; nb[:, :, 1:] |= hot[:, :, :-1]; nb[:, :, :-1] |= hot[:, :, 1:]
    m = (hot & nb).flatten(1).sum(1).float()
    print(f"level {L}: hot pixels per frame {n.mean():.0f} (min {n.min():.0f} max {n.max():.0f}), with a hot neighbour {m.mean():.0f}", flush=True)
outs = [(torch.empty((B, P, 2), dtype=torch.float64, device='cuda'), torch.empty((B, P), dtype=torch.int8, device='cuda'),
         torch.empty((B,), dtype=torch.int32, device='cuda')) for _ in range(3)]
mode = sys.argv[1] if len(sys.argv) > 1 else "pipe"
for i in range(10): det.chain(frames, 3, P, out=outs[i % 3], sync=False)
det.sync()
t0 = time.perf_counter()
for i in range(30):
    det.chain(frames, 3, P, out=outs[i % 3], sync=(mode == "alone"))
det.sync()
print(f"{mode}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step", flush=True)
