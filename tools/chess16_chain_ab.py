"""The dense chain (bench workload) with the levels' response kernels on chess_v1 (option chess_variant_hot 0) and on the
sixteen-pixels-per-lane kernels (experiment build: MRGINGHAM_AMD_LIB=.../libmrgingham_amd_experiment.so; 32 = level 0 +
level images on chess_v16_pyr_kernel, 16 = levels 3..1 on chess_v16_multi_kernel, 48 = both): outputs must be identical;
interleaved timing of the pipelined step and of the level-0 launch inside it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import mrgingham_amd
from mrgingham_amd import synth
W, H, B, P = 4096, 3072, 64, 256
gridn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
frames = synth.board_batch(B, W, H, gridn, 0, device="cuda")
det = mrgingham_amd.Detector(0)
det.set_option("sparse_refine", 0)
outs = [tuple(torch.empty(s, dtype=d, device="cuda") for s, d in (((B, P, 2), torch.float64), ((B, P), torch.int8), ((B,), torch.int32))) for _ in range(3)]
want = None
variants = (0, 32, 48)
res = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        det.set_option("chess_variant_hot", v)
        got = det.chain(frames, 3, P)
        if want is None:
            want = got
        else:
            n = want[2].tolist()
            same = bool(torch.equal(want[2], got[2])) and all(bool(torch.equal(want[0][f, :n[f]], got[0][f, :n[f]]) and torch.equal(want[1][f, :n[f]], got[1][f, :n[f]])) for f in range(B))
            assert same, "outputs differ"
        for i in range(30):
            det.chain(frames, 3, P, out=outs[i % 3], sync=False)
        det.sync()
        det.set_kernel_timing(True); det.chess_kernel_ms()
        t0 = time.perf_counter()
        for i in range(200):
            det.chain(frames, 3, P, out=outs[i % 3], sync=False)
        det.sync()
        dt = (time.perf_counter() - t0) / 200
        ms, nl = det.chess_kernel_ms(); det.set_kernel_timing(False)
        res[v].append((dt * 1e3, ms * 1e3))
for v in variants:
    r = sorted(res[v])
    print(json.dumps({"chess_variant_hot": v, "step_ms_median": r[len(r) // 2][0], "step_ms_all": [round(x[0], 4) for x in r], "l0_us": [round(x[1], 1) for x in r]}))
print("outputs identical")
