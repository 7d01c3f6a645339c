"""Randomised parity against the oracle: random frame sizes (odd ones included), boards, white / smoothed noise
overlays, textured backgrounds, start levels and batch sizes; every frame's chain output (doubles, levels, order)
and a random level's detection list must be the oracle's, and the sparse schedule must either agree or report.
Short by default; MRG_FUZZ_ITERS=2000 python -m pytest tests/test_gpu_fuzz.py -m gpu for the long form."""
import os
import random

import numpy as np
import pytest
import torch

import mrgingham_amd
from mrgingham_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


def _frames(rng, dev):
    W = rng.choice([320, 640, 801, 1024, 1283, 1920, 2048, 2600]) + rng.choice([0, 0, 1, 7, 16])
    H = max(200, int(W * rng.choice([0.5625, 0.75, 1.0])) + rng.choice([0, 0, 3, 8]))
    B = rng.choice([1, 2, 3])
    gridn = rng.choice([6, 8, 10, 12, 14])
    kind = rng.choice(["clean", "clean", "clutter", "noise", "noise_smooth", "pure_noise"])
    seed = rng.randrange(1 << 20)
    if kind == "clutter":
        fr = synth.cluttered_board_batch(B, W, H, gridn, seed, device=dev, smooth=rng.choice([1, 2, 3]), amp=rng.choice([64, 128, 200]))
    elif kind == "pure_noise":
        fr = torch.stack([synth.noise_frame(W, H, seed=seed + b, smooth=rng.choice([0, 1, 2]), device=dev) for b in range(B)])
    else:
        fr = synth.board_batch(B, W, H, gridn, seed, device=dev)
        if kind != "clean":
            sm = 0 if kind == "noise" else rng.choice([1, 2])
            nz = torch.stack([synth.noise_frame(W, H, seed=seed + 7 + b, smooth=sm, device=dev) for b in range(B)]).to(torch.int64)
            fr = (fr.to(torch.int64) + (nz - 128) * rng.choice([20, 40, 80, 120]) // 255).clamp(0, 255).to(torch.uint8)
    return fr, (W, H, B, gridn, kind, seed)


def test_random_frames_against_the_oracle():
    iters = int(os.environ.get("MRG_FUZZ_ITERS", "40"))
    rng = random.Random(int(os.environ.get("MRG_FUZZ_SEED", "5")))
    dev = torch.device("cuda:0")
    dense, sparse = mrgingham_amd.Detector(0), mrgingham_amd.Detector(0)
    dense.set_option("sparse_refine", 0)
    sparse.set_option("sparse_refine", 2)
    compared = reported = 0
    try:
        for it in range(iters):
            fr, desc = _frames(rng, dev)
            start = rng.choice([0, 1, 2, 3, 3, 3, 4])
            P = 4096
            pts, lv, n = [t.cpu().numpy() for t in dense.chain(fr, start, P)]
            host = fr.cpu().numpy()
            for f in range(fr.shape[0]):
                wp, wl = oracle.chain(host[f], start)
                k = int(n[f])
                assert k == min(len(wp), P), (desc, start, f, k, len(wp))       # (more than the pitch: the first P, in order)
                assert np.array_equal(lv[f, :k], wl[:k]), (desc, start, f)
                assert np.array_equal(pts[f, :k], wp[:k]), (desc, start, f)     # identical doubles, identical order
                compared += 1
            level = rng.choice([0, 1, 2, 3])
            xy, counts = dense.detect(fr, level, capacity=1 << 16)
            want = oracle.find_corners(host[0], level)
            assert int(counts[0]) == len(want), (desc, level)
            assert np.array_equal(xy[0, :len(want)].cpu().numpy(), want), (desc, level)
            if start >= 1:
                sp = [t.cpu().numpy() for t in sparse.chain(fr, start, P)]
                assert np.array_equal(sp[2], n), (desc, start)
                for f in range(fr.shape[0]):
                    k = min(int(n[f]), P)
                    assert np.array_equal(sp[0][f, :k], pts[f, :k]) and np.array_equal(sp[1][f, :k], lv[f, :k]), (desc, start, f)
                reported += sparse.sparse_fallbacks()
        print(f"{iters} calls, {compared} frames identical to the oracle, {reported} frames repeated densely inside a sparse call")
    finally:
        dense.close(); sparse.close()


def test_random_images_through_the_preprocessing():
    """normalize + CLAHE(8) + box blur, 8 and 16 bit, on random sizes (tiles that do not divide evenly, tiny frames),
    random contents (boards, noise, flat, two-valued, narrow ranges) and random blur radii, against the oracle."""
    iters = int(os.environ.get("MRG_FUZZ_ITERS", "40"))
    rng = random.Random(int(os.environ.get("MRG_FUZZ_SEED", "5")) + 1000)
    nrng = np.random.default_rng(rng.randrange(1 << 30))
    for it in range(iters):
        W, H = rng.randrange(9, 900), rng.randrange(9, 700)
        kind = rng.choice(["board", "noise", "flat", "two", "narrow", "ramp"])
        if kind == "board" and W >= 64 and H >= 64:
            img = synth.board_frame(W, H, rng.choice([4, 6, 10]), rng.randrange(100)).numpy()
        elif kind == "noise":
            img = nrng.integers(0, 256, (H, W), dtype=np.uint8)
        elif kind == "flat":
            img = np.full((H, W), rng.randrange(256), np.uint8)
        elif kind == "two":
            img = (nrng.integers(0, 2, (H, W)) * rng.randrange(1, 256)).astype(np.uint8)
        elif kind == "narrow":
            lo = rng.randrange(0, 250)
            img = nrng.integers(lo, lo + 6, (H, W)).astype(np.uint8)
        else:
            img = ((np.arange(W)[None, :] * 255 // max(W - 1, 1)) + np.zeros((H, 1), np.int64)).astype(np.uint8)
        clahe, blur = rng.random() < 0.7, rng.choice([0, 1, 1, 2, 3])
        got = mrgingham_amd.preprocess(img, clahe=clahe, blur_radius=blur)
        want = oracle.preprocess(img, clahe=clahe, blur_radius=blur)
        assert np.array_equal(got, want), ("8 bit", it, W, H, kind, clahe, blur)
        img16 = (img.astype(np.uint16) * rng.choice([1, 17, 120, 257]) + rng.randrange(0, 200)).astype(np.uint16)
        if rng.random() < 0.3:
            img16 = nrng.integers(0, 65536, (H, W)).astype(np.uint16)
        got16 = mrgingham_amd.api.preprocess16(img16, clahe=clahe, blur_radius=blur)
        want16 = oracle.preprocess16(img16, clahe=clahe, blur_radius=blur)
        assert np.array_equal(got16, want16), ("16 bit", it, W, H, kind, clahe, blur)
