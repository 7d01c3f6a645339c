"""chess_v16_kernel (sixteen pixels per lane, option chess_variant 16) against chess_v1_kernel: bit-exact equality of the plain and
the clamped response on random and board frames at several sizes, then interleaved timing of the plain response alone on
64 x 4096x3072 (hipEvents, per-launch medians)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth

det = mrgingham_amd.Detector(0)
bad = 0
for (W, H, B) in [(64, 48, 3), (256, 256, 2), (272, 300, 2), (640, 480, 4), (1008, 777, 2), (1920, 1080, 3), (4096, 3072, 2), (512, 16, 2), (16, 40, 2)]:
    fr = torch.stack([synth.noise_frame(W, H, seed=s, smooth=s % 3, device="cuda") for s in range(B - 1)] +
                     [synth.board_frame(W, H, 10, seed=1, device="cuda") if W >= 320 else synth.noise_frame(W, H, seed=9, device="cuda")])
    for clamp in (False, True):
        det.set_option("chess_variant", 0)
        a = det.chess_response(fr, 0, clamp=clamp)
        det.set_option("chess_variant", 16)
        b = det.chess_response(fr, 0, clamp=clamp)
        torch.cuda.synchronize()
        same = bool(torch.equal(a, b))
        if not same:
            bad += 1
            d = (a != b).nonzero()
            print("MISMATCH", W, H, clamp, d.shape[0], d[:5].tolist(), a[tuple(d[0])].item(), b[tuple(d[0])].item())
        else:
            print("same", W, H, B, "clamp" if clamp else "raw")
if "--no-time" in sys.argv:
    sys.exit(bad)
W, H, B = 4096, 3072, 64
frames = synth.board_batch(B, W, H, 10, 0, device="cuda")
out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
res = {0: [], 16: []}
for rnd in range(6):
    for v in (0, 16):
        det.set_option("chess_variant", v)
        for _ in range(5):
            det.chess_response(frames, 0, clamp=False, out=out)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
        ev[0].record()
        for i in range(30):
            det.chess_response(frames, 0, clamp=False, out=out)
            ev[i + 1].record()
        torch.cuda.synchronize()
        res[v] += [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(30)]
for v in (0, 16):
    t = sorted(res[v])
    print(json.dumps({"variant": v, "median_us": t[len(t) // 2], "p10_us": t[len(t) // 10], "p90_us": t[len(t) * 9 // 10],
                      "frac_3Bpx_median": B * W * H * 3.0 / (t[len(t) // 2] * 1e-6) / 8e12}))
sys.exit(bad)
