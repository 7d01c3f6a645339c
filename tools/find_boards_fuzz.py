"""Randomised check of the pipelined full detector (mrgingham_amd_find_boards_submit / _collect): random frame sizes
(odd ones included), boards, noise overlays, textured backgrounds, frames without a board, gridn, requested level, batch
sizes and pipeline depth; every board and found level must equal the synchronous dense schedule's (option
find_boards_pipeline 0, sparse_refine 0) -- double for double.  python tools/find_boards_fuzz.py [iterations] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import mrgingham_amd
from mrgingham_amd import synth


def make_batch(rng, dev):
    W = rng.choice([640, 801, 1024, 1283, 1920, 2048, 3000, 4096]) + rng.choice([0, 0, 1, 7, 16])
    H = max(240, int(W * rng.choice([0.5625, 0.75, 1.0])) + rng.choice([0, 0, 3, 8]))
    B = rng.choice([1, 2, 3, 5, 8])
    gridn = rng.choice([6, 8, 10, 10, 12])
    seed = rng.randrange(1 << 20)
    frames = []
    for b in range(B):
        kind = rng.choice(["clean", "clean", "clean", "clutter", "noise", "noise_smooth", "none", "small"])
        if kind == "clutter":
            f = synth.cluttered_board_frame(W, H, gridn, seed + b, smooth=rng.choice([1, 2, 3]), amp=rng.choice([64, 128]), device=dev)
        elif kind == "none":
            f = synth.noise_frame(W, H, seed + b, smooth=rng.choice([0, 1, 2]), device=dev)
        elif kind == "small":                       # a board that needs a finer level than its neighbours
            f = torch.full((H, W), 200, dtype=torch.uint8, device=dev)
            f[:H // 2, :W // 2] = synth.board_frame(W // 2, H // 2, gridn, seed + b, device=dev)
        else:
            f = synth.board_frame(W, H, gridn, seed + b, device=dev)
            if kind in ("noise", "noise_smooth"):
                nz = synth.noise_frame(W, H, seed=seed + 7 + b, smooth=0 if kind == "noise" else 1, device=dev).to(torch.int64)
                f = (f.to(torch.int64) + (nz - 128) * rng.choice([20, 40, 80]) // 255).clamp(0, 255).to(torch.uint8)
        frames.append(f)
    return torch.stack(frames), gridn, (W, H, B, gridn)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    ref, det = mrgingham_amd.Detector(0), mrgingham_amd.Detector(0)
    ref.set_option("find_boards_pipeline", 0)
    ref.set_option("sparse_refine", 0)
    det.set_option("sparse_refine", rng.choice([1, 2]))
    jobs, bad, frames, found = [], 0, 0, 0

    def collect():
        nonlocal bad, frames, found
        job, want, desc, lvl = jobs.pop(0)
        gb, gf = det.find_boards_collect(job)
        wb, wf = want
        ok = np.array_equal(wf, gf)
        for f in range(len(wf)):
            if ok and wf[f] >= 0:
                ok = np.array_equal(wb[f], gb[f])
        frames += len(wf)
        found += int((wf >= 0).sum())
        if not ok:
            bad += 1
            print("MISMATCH", desc, "level", lvl, wf.tolist(), gf.tolist(), flush=True)

    for it in range(iters):
        batch, gridn, desc = make_batch(rng, dev)
        lvl = rng.choice([-1, -1, -1, 0, 1, 2, 3])
        want = ref.find_boards(batch, gridn=gridn, image_pyramid_level=lvl, nthreads=4)
        jobs.append((det.find_boards_submit(batch, gridn=gridn, image_pyramid_level=lvl, nthreads=4), want, desc, lvl))
        while len(jobs) >= rng.choice([1, 2, 3, 4]):
            collect()
    while jobs:
        collect()
    print(f"{iters} batches, {frames} frames ({found} with a board), {det.sparse_fallbacks()} frames repeated densely by the library, {bad} mismatching")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
