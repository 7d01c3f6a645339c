"""Sparse step against the dynamic-LDS pad of the gentle pyramid pass (experiment build, option pyramid_lds_pad) and the
workgroups per frame of the refinement: python tools/pyr_pad_ab.py  (MRGINGHAM_AMD_LIB = the experiment library)"""
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
    det.set_option("sparse_refine", 2)
    outs = [tuple(torch.empty(s, dtype=d, device="cuda") for s, d in (((B, P, 2), torch.float64), ((B, P), torch.int8), ((B,), torch.int32))) for _ in range(3)]
    res = {}
    for rnd in range(2):
        for pad in (80000, 60000, 56000, 40000):
            for k in (1, 2, 4):
                det.set_option("pyramid_lds_pad", pad)
                det.set_option("sparse_subsets", k)
                for i in range(20):
                    det.chain(frames, 3, P, out=outs[i % 3], sync=False)
                det.sync()
                t0 = time.perf_counter()
                for i in range(150):
                    det.chain(frames, 3, P, out=outs[i % 3], sync=False)
                det.sync()
                res.setdefault((pad, k), []).append((time.perf_counter() - t0) / 150 * 1e3)
    print("gridn", gridn, {f"pad{p}_k{k}": round(min(v), 4) for (p, k), v in res.items()}, flush=True)
    det.close()
