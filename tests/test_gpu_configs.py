"""BASELINE.json's configurations AS STATED, under -m gpu:
  C3  64 x 4096x3072 frames with a 14x14 board, chain 3 -> 0 (checked bit-exactly against the oracle on
      a sample of the frames, by properties on all of them);
  C4  one GPU's shard of the 2048-frame job: 256 frames of 4096x3072 in ONE call, both scratch sets
      resident (properties: batch-size independence, determinism, sample vs the oracle);
  C5  in miniature lives in tests/test_gpu_parallel.py.
"""
import numpy as np
import pytest
import torch

import mrgingham_amd
from mrgingham_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu

W, H = 4096, 3072


@pytest.fixture(scope="module")
def det():
    d = mrgingham_amd.Detector(0)
    yield d
    d.close()


def _check_against_oracle(frame_u8, pts, lv, n, start_level=3):
    wp, wl = oracle.chain(frame_u8, start_level)
    assert n == len(wp)
    assert np.array_equal(lv[:n], wl)
    assert np.abs(pts[:n] - wp).max(initial=0.0) <= 1e-4      # north star tolerance ...
    assert np.array_equal(pts[:n], wp)                        # ... and in fact identical doubles


def test_c3_as_stated_14x14_board_chain(det):
    B = 64
    frames = synth.board_batch(B, W, H, gridn=14, seed0=100, device="cuda")
    pts, lv, npts = det.chain(frames, start_level=3, max_points=512)
    pts, lv, npts = pts.cpu().numpy(), lv.cpu().numpy(), npts.cpu().numpy()
    for f in (0, 17, 63):                                      # bit-exact sample
        _check_against_oracle(frames[f].cpu().numpy(), pts[f], lv[f], int(npts[f]))
    for f in range(B):
        n = int(npts[f])
        assert 196 <= n <= 512, (f, n)
        assert int((lv[f, :n] == 0).sum()) >= 196, f           # the 14x14 grid refines down to level 0
    # every level of the stated 4-level pyramid finds the 196 corners on its own as well
    for level in (0, 1, 2):
        xy, counts = det.detect(frames[:4], level, capacity=1024)
        assert (counts.cpu().numpy() >= 196).all(), level
        want = oracle.find_corners(frames[1].cpu().numpy(), level)
        assert int(counts[1]) == len(want) and np.array_equal(xy[1, :len(want)].cpu().numpy(), want)


def test_c4_shard_256_frames_in_one_call(det):
    B = 256
    base = synth.board_batch(32, W, H, gridn=10, seed0=500, device="cuda")
    frames = torch.empty((B, H, W), dtype=torch.uint8, device="cuda")
    for k in range(B // 32):                                   # 32 distinct frames, 8 rotations of the order
        frames[k * 32:(k + 1) * 32] = torch.roll(base, shifts=k, dims=0)
    del base
    out = det.chain(frames, start_level=3, max_points=256)
    pts, lv, npts = [t.cpu().numpy() for t in out]
    assert (npts >= 100).all()
    # sample vs the oracle
    for f in (0, 101, 255):
        _check_against_oracle(frames[f].cpu().numpy(), pts[f], lv[f], int(npts[f]))
    # batch-size independence: the same frames as four calls of 64 give the same lists
    for k in range(4):
        p2, l2, n2 = det.chain(frames[k * 64:(k + 1) * 64], start_level=3, max_points=256)
        assert np.array_equal(n2.cpu().numpy(), npts[k * 64:(k + 1) * 64])
        for f in range(64):
            n = int(npts[k * 64 + f])
            assert np.array_equal(p2[f, :n].cpu().numpy(), pts[k * 64 + f, :n])
            assert np.array_equal(l2[f, :n].cpu().numpy(), lv[k * 64 + f, :n])
    # the same frame at different positions of the batch gives the same list (frame 0 == frame 33 == ...)
    for k in range(1, B // 32):
        a, b = 0, k * 32 + k
        n = int(npts[a])
        assert int(npts[b]) == n and np.array_equal(pts[a, :n], pts[b, :n]) and np.array_equal(lv[a, :n], lv[b, :n])
    # both scratch sets of the 256-frame shard are resident, and well below what a dense
    # int32-per-pixel index alone used to take for them (4 B/px * 1.33 * 256 frames * 2 sets = 34 GB)
    gib = det.scratch_bytes() / 2**30
    assert gib <= 32, gib                                      # (round 2: 78.9 GiB)
    print(f"scratch for 256 frames of {W}x{H}, both sets: {gib:.1f} GiB")


def test_scratch_for_64_frames():
    d = mrgingham_amd.Detector(0)
    try:
        frames = synth.board_batch(2, W, H, gridn=10, seed0=1, device="cuda").repeat(32, 1, 1)
        d.set_option("sparse_refine", 0)                       # the dense schedule (bench.py's): two scratch sets
        d.chain(frames, start_level=3, max_points=256)
        gib = d.scratch_bytes() / 2**30
        print(f"scratch for 64 frames of {W}x{H}, both sets: {gib:.2f} GiB")
        # round 1 held 4 B/px of dense index per level and set: 64 * 12.58 MB * 4 * 1.33 * 2 = 8.6 GB for
        # that table alone, ~31 GiB in total; the bound below fails if anything of that size comes back
        assert gib <= 8, gib                                   # (round 2: 19.7 GiB)
        d.set_option("sparse_refine", 1)                       # the library's default: a call of this size goes sparse,
        d.chain(frames, start_level=3, max_points=256)         # and a context that has done that keeps a third set
        assert d.chain_info()[1] == -1
        assert d.scratch_bytes() / 2**30 <= 12
    finally:
        d.close()


def test_c3_cluttered_frames_match_the_oracle_on_both_search_paths(det):
    """The board over a textured background (bench workload c3_cluttered): tens of thousands of hot pixels per
    frame at level 0 -- more than the LDS tables hold and no empty rows to cut bands at -- so the frames go
    through the global-memory kernels at the fine levels and through LDS at the coarse ones; the lists equal
    the oracle's bit for bit either way."""
    B = 3
    frames = synth.cluttered_board_batch(B, W, H, gridn=10, seed0=7, device="cuda")
    host = frames.cpu().numpy()
    pts, lv, npts = det.chain(frames, start_level=3, max_points=512)
    paths = {L: det.debug_paths(L, B).tolist() for L in range(4)}
    pts, lv, npts = pts.cpu().numpy(), lv.cpu().numpy(), npts.cpu().numpy()
    for f in range(B):
        _check_against_oracle(host[f], pts[f], lv[f], int(npts[f]))
        assert int((lv[f, :int(npts[f])] == 0).sum()) >= 100, f
    # refinement at level 0 (~7e4 hot pixels per frame): out of LDS all the same, on the hot pixels in the cells
    # around the points only (cc.hip, WinSel); level 3 (~1.6e3) fits as it is
    assert paths[0] == [1] * B and paths[3] == [1] * B, paths
    hot0 = int((oracle.clamped_response(host[0], 0)[0] > 15).sum())
    assert 2e4 < hot0 < 2e5, hot0
    for level in (0, 1, 2, 3):                                 # every level on its own as well (detect)
        xy, counts = det.detect(frames, level, capacity=1024)
        if level == 0:
            assert det.debug_paths(0, B).tolist() == [0] * B   # detect needs every component: global memory
        for f in range(B):
            want = oracle.find_corners(host[f], level)
            assert int(counts[f]) == len(want) and np.array_equal(xy[f, :len(want)].cpu().numpy(), want), (level, f)


def test_c2_as_stated_64_frames_1920x1080_level0_detect(det):
    """BASELINE config 2 as stated: one batch of 64 frames of 1920x1080 with a 10x10 board through the ChESS path
    at level 0 (detect).  Three frames bit-exact against the oracle (lists in the reference's order), the dense
    response of one frame against the oracle, properties on all 64."""
    B, w, h = 64, 1920, 1080
    frames = synth.board_batch(B, w, h, gridn=10, seed0=300, device="cuda")
    xy, counts = det.detect(frames, 0, capacity=1024)
    xy, counts = xy.cpu().numpy(), counts.cpu().numpy()
    for f in (0, 31, 63):
        want = oracle.find_corners(frames[f].cpu().numpy(), 0)
        assert int(counts[f]) == len(want) and np.array_equal(xy[f, :len(want)], want), f
    resp = det.chess_response(frames[5:6], 0, clamp=True)[0].cpu().numpy()
    assert np.array_equal(resp, oracle.clamped_response(frames[5].cpu().numpy(), 0)[0])
    for f in range(B):
        n = int(counts[f])
        assert 100 <= n <= 1024, (f, n)
        lat = synth.board_lattice(w, h, 10, 300 + f).reshape(-1, 2)
        c = xy[f, :n].astype(np.float64) / 1000.0
        d = np.sqrt(((c[None, :, :] - lat[:, None, :]) ** 2).sum(-1)).min(1)   # every rendered corner has a candidate
        assert d.max() < 1.0, (f, d.max())                                      # within a pixel of where it was drawn
    # batch-size independence: the same frames one at a time
    for f in (7, 40):
        x1, c1 = det.detect(frames[f:f + 1], 0, capacity=1024)
        assert int(c1[0]) == int(counts[f]) and np.array_equal(x1[0, :int(c1[0])].cpu().numpy(), xy[f, :int(counts[f])])
