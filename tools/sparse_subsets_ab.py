"""Option sparse_subsets (workgroups per frame of the sparse refinement) 1 / 2 / 4, interleaved on one box: pipelined sparse
chain on 64 x 4096x3072 (10x10 and 14x14 boards) and the pipelined full detector; outputs compared with the dense chain's."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import mrgingham_amd
from mrgingham_amd import synth
W, H, B, P = 4096, 3072, 64, 256
for gridn in (10, 14):
    frames = synth.board_batch(B, W, H, gridn, 0, device="cuda")
    det = mrgingham_amd.Detector(0)
    det.set_option("sparse_refine", 0)
    want = det.chain(frames, 3, P)
    det.set_option("sparse_refine", 2)
    outs = [tuple(torch.empty_like(t) for t in want) for _ in range(3)]
    res = {1: [], 2: [], 4: []}
    for rnd in range(3):
        for k in (1, 2, 4):
            det.set_option("sparse_subsets", k)
            got = det.chain(frames, 3, P)
            n = want[2].tolist()
            assert torch.equal(want[2], got[2]) and all(torch.equal(want[0][f, :n[f]], got[0][f, :n[f]]) and torch.equal(want[1][f, :n[f]], got[1][f, :n[f]]) for f in range(B)), k
            for i in range(20):
                det.chain(frames, 3, P, out=outs[i % 3], sync=False)
            det.sync()
            t0 = time.perf_counter()
            for i in range(150):
                det.chain(frames, 3, P, out=outs[i % 3], sync=False)
            det.sync()
            res[k].append((time.perf_counter() - t0) / 150 * 1e3)
    print(json.dumps({"gridn": gridn, "sparse_step_ms": {k: [round(x, 4) for x in sorted(v)] for k, v in res.items()}, "fallbacks": det.sparse_fallbacks()}), flush=True)
    det.close()
