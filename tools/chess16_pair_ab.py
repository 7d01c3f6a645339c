"""Round 6, the half strip (widths with w % 256 = 128): chess_v16_kernel against chess_v16_pair_kernel (experiment build:
MRGINGHAM_AMD_LIB=.../libmrgingham_amd_experiment.so, option chess16_pair), whose last-strip workgroups take two row
segments at once.  Equality of the whole response with the unpaired kernel's (raw and clamped), then interleaved timing
over segment counts: python tools/chess16_pair_ab.py [WxH ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth

B = 64
sizes = [(1920, 1080), (640, 480), (1408, 800), (2432, 1368)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]


det = mrgingham_amd.Detector(0)
det.set_option("chess_variant", 16)
for (W, H) in sizes:
    frames = synth.board_batch(4, W, H, 10, 0, device="cuda").repeat(B // 4, 1, 1).contiguous()
    noise = torch.stack([synth.noise_frame(W, H, seed=11 + b, smooth=1, device="cuda") for b in range(2)])
    frames[1], frames[2] = noise[0], noise[1]              # every pixel matters somewhere
    out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    ks = [k for k in (2, 4, 6, 8, 10, 12, 16) if H // k >= 32]
    rows_of = lambda k: (H + k - 1) // k
    # equality first: raw and clamped, every k, against the unpaired automatic choice
    for clamp in (False, True):
        det.set_option("chess16_pair", 0); det.set_option("chess16_seg", 0)
        want = det.chess_response(frames, 0, clamp=clamp).clone()
        for k in ks:
            det.set_option("chess16_pair", 1); det.set_option("chess16_seg", rows_of(k))
            got = det.chess_response(frames, 0, clamp=clamp)
            assert torch.equal(got, want), ("paired differs", W, H, k, clamp, int((got != want).sum()))
    res = {}
    for rnd in range(3):
        for pair in (0, 1):
            for k in [0] + ks:
                det.set_option("chess16_pair", pair); det.set_option("chess16_seg", rows_of(k) if k else 0)
                res.setdefault((pair, k), []).extend(timed(lambda: det.chess_response(frames, 0, clamp=False, out=out)))
    det.set_option("chess16_pair", 0); det.set_option("chess16_seg", 0)
    med = {key: sorted(v)[len(v) // 2] for key, v in res.items()}
    px3 = B * W * H * 3.0
    best0 = min((k for (p, k) in med if p == 0), key=lambda k: med[(0, k)])
    best1 = min((k for (p, k) in med if p == 1), key=lambda k: med[(1, k)])
    print(json.dumps({"size": f"{W}x{H}", "identical": True,
                      "unpaired_us": {("auto" if k == 0 else str(k)): round(med[(0, k)], 1) for (p, k) in sorted(med) if p == 0},
                      "paired_us": {("auto+even" if k == 0 else str(k)): round(med[(1, k)], 1) for (p, k) in sorted(med) if p == 1},
                      "best_unpaired": [best0, round(med[(0, best0)], 1), round(px3 / med[(0, best0)] / 8e6, 4)],
                      "best_paired": [best1, round(med[(1, best1)], 1), round(px3 / med[(1, best1)] / 8e6, 4)],
                      "gain_pct": round(100 * (med[(0, best0)] / med[(1, best1)] - 1), 2)}), flush=True)
