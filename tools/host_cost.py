import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, mrgingham_amd
from mrgingham_amd import synth
for (W, H) in ((640, 480), (4096, 3072)):
    frames = synth.board_batch(8, W, H, 10, 0, device='cuda').repeat(8, 1, 1).contiguous()
    for ns in (2, 3):
        det = mrgingham_amd.Detector(0)
        det.set_option("scratch_sets", ns)
        outs = [(torch.empty((64, 256, 2), dtype=torch.float64, device='cuda'), torch.empty((64, 256), dtype=torch.int8, device='cuda'),
                 torch.empty((64,), dtype=torch.int32, device='cuda')) for _ in range(4)]
        for i in range(50): det.chain(frames, 3, 256, out=outs[i % 4], sync=False)
        det.sync()
        t0 = time.perf_counter()
        for i in range(300): det.chain(frames, 3, 256, out=outs[i % 4], sync=False)
        t1 = time.perf_counter()
        det.sync()
        t2 = time.perf_counter()
        print(f"{W}x{H} sets {ns}: enqueue {1e6*(t1-t0)/300:.1f} us per call, total {1e6*(t2-t0)/300:.1f} us per step")
        det.close()
