"""chess_v16_kernel (sixteen pixels per lane, option chess_variant 16) against chess_v1_kernel: bit-exact equality of the plain and
the clamped response on random and board frames at several sizes, then interleaved timing of the plain response alone on
64 x 4096x3072 (hipEvents, per-launch medians)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth

if "--libs" in sys.argv:     # A/B of library builds: one child per library and round, interleaved
    import subprocess
    libs = sys.argv[sys.argv.index("--libs") + 1].split(",")
    acc = {l: [] for l in libs}
    for rnd in range(3):
        for l in libs:
            r = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, MRGINGHAM_AMD_LIB=os.path.abspath(l)),
                               capture_output=True, text=True)
            if r.returncode != 0 and not r.stdout.strip():
                print(l, "FAILED", r.stderr[-500:])
                continue
            acc[l].append(json.loads(r.stdout.strip().splitlines()[-1]))
    for l in libs:
        if acc[l]:
            print("%-40s v16 median %s us   v1 median %s us   equal=%s" % (os.path.basename(l), sorted(round(a["v16"], 1) for a in acc[l]),
                  sorted(round(a["v1"], 1) for a in acc[l]), all(a["equal"] for a in acc[l])))
    sys.exit(0)
det = mrgingham_amd.Detector(0)
if "--child" in sys.argv:
    W, H, B = 4096, 3072, 64
    frames = synth.board_batch(8, W, H, 10, 0, device="cuda").repeat(8, 1, 1).contiguous()
    out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    chk = {}
    res = {0: [], 16: []}
    for rnd in range(3):
        for v in (0, 16):
            det.set_option("chess_variant", v or 1)
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
            chk[v] = int(out[::5, ::7, ::3].to(torch.int64).sum().item())
    med = lambda t: sorted(t)[len(t) // 2]
    print(json.dumps({"v1": med(res[0]), "v16": med(res[16]), "equal": chk[0] == chk[16]}))
    sys.exit(0)
bad = 0
for (W, H, B) in [(64, 48, 3), (256, 256, 2), (272, 300, 2), (640, 480, 4), (1008, 777, 2), (1920, 1080, 3), (4096, 3072, 2), (512, 16, 2), (16, 40, 2)]:
    fr = torch.stack([synth.noise_frame(W, H, seed=s, smooth=s % 3, device="cuda") for s in range(B - 1)] +
                     [synth.board_frame(W, H, 10, seed=1, device="cuda") if W >= 320 else synth.noise_frame(W, H, seed=9, device="cuda")])
    for clamp in (False, True):
        det.set_option("chess_variant", 1)
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
segs = [int(a) for a in sys.argv[sys.argv.index("--segs") + 1].split(",")] if "--segs" in sys.argv else [0]
res = {0: []}
res.update({(16, sg): [] for sg in segs})
for rnd in range(6):
    for v in [0] + [(16, sg) for sg in segs]:
        det.set_option("chess_variant", 16 if v else 1)
        if v:
            det.set_option("chess16_seg", v[1])
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
for v in res:
    t = sorted(res[v])
    print(json.dumps({"variant": v, "median_us": t[len(t) // 2], "p10_us": t[len(t) // 10], "p90_us": t[len(t) * 9 // 10],
                      "frac_3Bpx_median": B * W * H * 3.0 / (t[len(t) // 2] * 1e-6) / 8e12}))
sys.exit(bad)
