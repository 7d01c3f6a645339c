"""Level-0 detect (response + clamp + hot list -> components -> candidates; BASELINE config 2's call) with the hot-list
response on chess_v1_kernel<hot> (option chess_variant_hot 0) and on chess_v16_hot_kernel (16), sizes of configs 2 and 5:
candidate lists must be identical; interleaved timing of the pipelined step and of the response launch inside it.
Needs the experiment build (make EXPERIMENT=1; MRGINGHAM_AMD_LIB=.../libmrgingham_amd_experiment.so): the shipped library does not carry the option."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth
B = 64
sizes = [(1920, 1080), (1280, 800), (2560, 1440), (4096, 2160), (640, 480), (4096, 3072)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
det = mrgingham_amd.Detector(0)
for (W, H) in sizes:
    frames = synth.board_batch(B, W, H, 10, 0, device="cuda")
    res = {0: [], 16: []}
    want = None
    for rnd in range(3):
        for v in (0, 16):
            det.set_option("chess_variant_hot", v)
            xy, cnt = det.detect(frames, 0, capacity=256)
            if want is None:
                want = (xy.clone(), cnt.clone())
            else:
                n = want[1].tolist()
                assert torch.equal(want[1], cnt) and all(torch.equal(want[0][f, :n[f]], xy[f, :n[f]]) for f in range(B)), "lists differ"
            for _ in range(40):
                det.detect(frames, 0, capacity=256, sync=False)
            det.sync()
            det.set_kernel_timing(True); det.chess_kernel_ms(); det.sclk_mhz()
            t0 = time.perf_counter()
            for _ in range(150):
                det.detect(frames, 0, capacity=256, sync=False)
            det.sync()
            dt = (time.perf_counter() - t0) / 150
            ms, nl = det.chess_kernel_ms(); clk = det.sclk_mhz(); det.set_kernel_timing(False)
            res[v].append((ms * 1e3, dt * 1e3, clk))
    px3 = B * W * H * 3.0
    for v in (0, 16):
        r = sorted(res[v])
        med = r[len(r) // 2]
        print(json.dumps({"size": f"{W}x{H}", "chess_variant_hot": v, "launch_us": [round(x[0], 1) for x in r], "step_ms": [round(x[1], 4) for x in r],
                          "sclk_mhz": [round(x[2]) for x in r], "frac_3Bpx_median": round(px3 / (med[0] * 1e-6) / 8e12, 4),
                          "min_candidates": int(want[1].min())}), flush=True)
    del frames
print("lists identical")
