"""Round 6, item 1: segment heights by WHOLE ROUNDS of resident workgroups.  For a list of frame sizes and every height
ceil(h / k / 16) * 16, k = 1..12 (plus the powers of two the pickers knew), median us per launch of
  v16     the plain response, chess_v16_kernel (option chess16_seg)
  v1      the plain response, chess_v1_kernel (option chess_seg)
  hot     level-0 detect's response launch, chess_v1_kernel<hot> (option chess_seg; events around the launch)
against the automatic choice (seg 0), and equality of every output with the automatic choice's.
python tools/seg_rounds_sweep.py [WxH ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth

B = 64
sizes = [(1920, 1080), (1280, 800), (2560, 1440), (4096, 2160), (1280, 960), (2048, 1536), (640, 480), (4096, 3072)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]


def heights(h):
    """0 (automatic) and the row counts that make k = 1..24 balanced segments (the launchers cut ceil(h / rows) of them)"""
    s = {0}
    for k in range(1, 25):
        rows = (h + k - 1) // k
        if rows >= 32:
            s.add(rows)
    return sorted(s)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]


out_rows = []
for (W, H) in sizes:
    frames = synth.board_batch(4, W, H, 10, 0, device="cuda").repeat(B // 4, 1, 1).contiguous()
    out = torch.empty((B, H, W), dtype=torch.int16, device="cuda")
    hs = heights(H)
    res = {"v16": {}, "v1": {}, "hot": {}}
    ref = None
    det = mrgingham_amd.Detector(0)
    for rnd in range(3):
        for seg in hs:
            if W % 16 == 0:
                det.set_option("chess_variant", 16)
                det.set_option("chess16_seg", seg)
                res["v16"].setdefault(seg, []).extend(timed(lambda: det.chess_response(frames, 0, clamp=False, out=out)))
                if ref is None:
                    ref = out.clone()
                assert torch.equal(out, ref), ("v16", W, H, seg)
                det.set_option("chess16_seg", 0)
            if True:
                det.set_option("chess_variant", 1)
                det.set_option("chess_seg", seg)
                res["v1"].setdefault(seg, []).extend(timed(lambda: det.chess_response(frames, 0, clamp=False, out=out)))
                if ref is None:
                    ref = out.clone()
                assert torch.equal(out, ref), ("v1", W, H, seg)
                det.set_option("chess_seg", 0)
            det.set_option("chess_variant", 0)
    det.close()
    # the hot kernel inside pipelined level-0 detect calls, the heights interleaved like the others
    want = None
    d2 = mrgingham_amd.Detector(0)
    d2.set_kernel_timing(True)
    for rnd in range(3):
        for seg in hs:
            d2.set_option("chess_seg", seg)
            for _ in range(4):
                r = d2.detect(frames, 0, capacity=256, sync=False)
            d2.sync()
            got = tuple(t.clone() for t in r[:2])
            if want is None:
                want = got
            assert all(torch.equal(a, b) for a, b in zip(got, want)), ("hot", W, H, seg)
            d2.chess_kernel_ms()
            for _ in range(20):
                d2.detect(frames, 0, capacity=256, sync=False)
            d2.sync()
            ms, n = d2.chess_kernel_ms()
            res["hot"].setdefault(seg, []).append(ms * 1e3)
    d2.close()
    med = {k: {s: sorted(t)[len(t) // 2] for s, t in v.items()} for k, v in res.items()}
    px = B * W * H
    print(f"== {W}x{H} x{B}  ({px * 3 / 8e6:.1f} us at 8 TB/s on 3 B/px)", flush=True)
    for k in ("v16", "v1", "hot"):
        if not med[k]:
            continue
        auto = med[k][0]
        best = min((s for s in med[k] if s), key=lambda s: med[k][s])
        print(f"  {k:4s} auto {auto:7.1f} us (frac {px * 3 / auto / 8e6:.3f}) | best seg {best} {med[k][best]:7.1f} us "
              f"(frac {px * 3 / med[k][best] / 8e6:.3f}, {100 * (auto / med[k][best] - 1):+.1f} %)", flush=True)
        print("       " + " ".join(f"{s}={m:.1f}" for s, m in med[k].items() if s), flush=True)
    out_rows.append({"w": W, "h": H, "frames": B, "median_us": med})
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out_rows, open("gpurun_out/seg_rounds_sweep.json", "w"), indent=1)
