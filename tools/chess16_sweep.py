"""v16 (segment heights) against v1 over frame sizes, plain response, 64 frames: median us per launch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
det = mrgingham_amd.Detector(0)
B = 64
for (W, H) in [(4096, 3072), (2560, 1920), (1920, 1080), (1280, 960), (640, 480), (2048, 1536), (1024, 768), (512, 384)]:
    frames = synth.board_batch(4, W, H, 10, 0, device="cuda").repeat(B // 4, 1, 1).contiguous()
    out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    row = {}
    for rnd in range(3):
        for v in [0, 64, 128, 256, 512, 1024]:
            det.set_option("chess_variant", 16 if v else 1)
            if v:
                det.set_option("chess16_seg", v)
            for _ in range(3):
                det.chess_response(frames, 0, clamp=False, out=out)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            ev[0].record()
            for i in range(20):
                det.chess_response(frames, 0, clamp=False, out=out)
                ev[i + 1].record()
            torch.cuda.synchronize()
            row.setdefault(v, []).extend(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(20))
    med = {v: sorted(t)[len(t) // 2] for v, t in row.items()}
    print(W, H, " ".join("%s=%.1f" % ("v1" if v == 0 else "seg%d" % v, m) for v, m in med.items()), "| frac3Bpx v1 %.3f best16 %.3f" % (
        B * W * H * 3 / med[0] / 8e6, B * W * H * 3 / min(m for v, m in med.items() if v) / 8e6), flush=True)
