"""Option "sparse_refine": chain() with the response of the levels below the start level computed only in the
cells around the points (include/mrgingham_amd.h).  The bar is the same as for the dense schedule: identical
doubles, identical levels, identical order -- against the dense schedule on every frame, against the oracle on a
sample -- and a frame the sparse kernels cannot take is repeated densely BY THE LIBRARY inside the same call (on the
device, no error, no second call): the outputs never differ, `sparse_fallbacks()` says how many frames took that way."""
import numpy as np
import pytest
import torch

import mrgingham_amd
from mrgingham_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


def _pair():
    dense, sparse = mrgingham_amd.Detector(0), mrgingham_amd.Detector(0)
    dense.set_option("sparse_refine", 0)       # (the library's default is 1: sparse where it pays)
    sparse.set_option("sparse_refine", 2)      # 2: always (1 would leave the small test frames to the dense schedule)
    return dense, sparse


def _same(a, b):
    pa, la, na = [t.cpu().numpy() for t in a]
    pb, lb, nb = [t.cpu().numpy() for t in b]
    assert np.array_equal(na, nb)
    for f in range(len(na)):
        n = int(na[f])
        assert np.array_equal(la[f, :n], lb[f, :n]), f
        assert np.array_equal(pa[f, :n], pb[f, :n]), f        # identical doubles


@pytest.mark.parametrize("W,H,B,gridn,start", [
    (640, 480, 8, 10, 3), (1024, 768, 8, 10, 3), (1024, 768, 4, 10, 2), (1024, 768, 4, 10, 1),
    (1001, 777, 4, 8, 3), (1920, 1080, 4, 10, 3), (4096, 3072, 4, 10, 3), (4096, 3072, 4, 14, 3),
    (2048, 1536, 2, 10, 4),
    (4608, 4608, 2, 10, 3),      # the box around the points has > 40 960 cells of 16 px: cells of 32 (4 micro-tiles each)
    (8192, 8192, 1, 12, 3),      # ... and of 64 (16 micro-tiles each)
])
def test_sparse_chain_is_the_dense_chain(W, H, B, gridn, start):
    dense, sparse = _pair()
    try:
        frames = synth.board_batch(B, W, H, gridn=gridn, seed0=300, device="cuda")
        want = dense.chain(frames, start, 1024)
        got = sparse.chain(frames, start, 1024)
        assert sparse.sparse_fallbacks() == 0                  # (no frame was repeated: the sparse kernels themselves)
        _same(want, got)
        assert int(want[2].min()) >= gridn * gridn // 2
        f = B - 1
        wp, wl = oracle.chain(frames[f].cpu().numpy(), start)  # and the oracle on one frame
        n = int(got[2][f])
        assert n == len(wp) and np.array_equal(got[1][f, :n].cpu().numpy(), wl)
        assert np.array_equal(got[0][f, :n].cpu().numpy(), wp)
    finally:
        dense.close(); sparse.close()


@pytest.mark.parametrize("W,H,gridn,start", [(4096, 3072, 14, 3), (2048, 1536, 14, 3), (3000, 3000, 12, 2), (4096, 3072, 16, 3)])
def test_several_workgroups_per_frame_give_the_same_chain(W, H, gridn, start):
    """A frame with at least 128 points to refine is cut into up to `sparse_subsets` subsets of points that are far enough
    apart, one workgroup of the refinement kernel each (cc.hip, "Several workgroups"): 1, 2 and 4 give the dense chain's
    doubles, levels and order; frames mixed (a 10x10 board's frame in the same batch stays with one workgroup), calls
    pipelined without a sync, no frame repeated densely that one workgroup would have taken."""
    dense, sparse = _pair()
    try:
        frames = synth.board_batch(3, W, H, gridn, 11, device="cuda")
        frames[1] = synth.board_frame(W, H, 10, 5, device="cuda")
        want = dense.chain(frames, start, 1024)
        assert int(want[2].max()) >= 128
        repeated = {}
        for k in (1, 2, 4, 3):
            sparse.set_option("sparse_subsets", k)
            sparse.sparse_fallbacks()
            _same(want, sparse.chain(frames, start, 1024))
            outs = [sparse.chain(frames, start, 1024, sync=False) for _ in range(4)]
            sparse.sync()
            for o in outs:
                _same(want, o)
            repeated[k] = sparse.sparse_fallbacks()
        # the cut gives no frame up that one workgroup takes (a 16x16 board at level 2 is given up either way)
        assert all(repeated[k] <= repeated[1] for k in repeated), repeated
        assert gridn > 14 or repeated[1] == 0, repeated
        with pytest.raises(ValueError):
            sparse.set_option("sparse_subsets", 5)
    finally:
        dense.close(); sparse.close()


def test_sparse_chain_on_textured_frames():
    dense, sparse = _pair()
    try:
        frames = synth.cluttered_board_batch(3, 4096, 3072, 10, 7, device="cuda")
        want = dense.chain(frames, 3, 1024)
        got = sparse.chain(frames, 3, 1024)                    # (whatever happens inside, the same answer)
        _same(want, got)
    finally:
        dense.close(); sparse.close()


def test_sparse_chain_pipelined_steps_stay_identical():
    """Calls in flight share the cell lists and masks of a scratch set: 12 steps without a sync in between, two
    batches alternating, every step's output compared."""
    dense, sparse = _pair()
    try:
        fa = synth.board_batch(8, 1024, 768, 10, 0, device="cuda")
        fb = synth.board_batch(8, 1024, 768, 10, 50, device="cuda")
        wa, wb = dense.chain(fa, 3, 512), dense.chain(fb, 3, 512)
        outs = []
        for i in range(12):
            out = tuple(torch.empty_like(t) for t in wa)
            sparse.chain(fa if i % 2 == 0 else fb, 3, 512, out=out, sync=False)
            outs.append(out)
        sparse.sync()
        for i, out in enumerate(outs):
            _same(wa if i % 2 == 0 else wb, out)
    finally:
        dense.close(); sparse.close()


def test_a_frame_the_sparse_kernels_cannot_take_is_repeated_densely_inside_the_call():
    """White noise over the board: at level 0 the blobs of the corners run into the noise around them and out of
    the cells that were computed.  The library repeats those frames with the dense kernels before the call
    completes -- one call, no error, the dense answer; a clean frame in the same batch stays with the sparse kernels."""
    dense, sparse = _pair()
    try:
        b = synth.board_batch(3, 1024, 768, 10, 5, device="cuda").to(torch.int64)
        nz = torch.stack([synth.noise_frame(1024, 768, seed=9 + s, device="cuda") for s in range(3)]).to(torch.int64)
        nz[1] = 128                                            # frame 1 stays clean
        frames = (b + (nz - 128) * 80 // 255).clamp(0, 255).to(torch.uint8)
        want = dense.chain(frames, 3, 2048)
        assert int(want[2].min()) >= 100
        got = sparse.chain(frames, 3, 2048)                    # (the noisy frames make the level-0 tables grow: retried once)
        _same(want, got)
        sparse.sparse_fallbacks()
        got = sparse.chain(frames, 3, 2048, retry=False)       # the tables have grown: ONE call, no error
        nrep = sparse.sparse_fallbacks()
        assert 1 <= nrep <= 2                                  # (the clean frame never; the noisy ones as the noise falls)
        _same(want, got)
        # pipelined, never synchronised in between: every call in flight repeats its own frames
        outs = []
        for i in range(6):
            out = tuple(torch.empty_like(t) for t in want)
            sparse.chain(frames, 3, 2048, out=out, sync=False)
            outs.append(out)
        sparse.sync()
        assert sparse.sparse_fallbacks() == 6 * nrep
        for out in outs:
            _same(want, out)
        good = synth.board_batch(2, 1024, 768, 10, 9, device="cuda")
        _same(dense.chain(good, 3, 512), sparse.chain(good, 3, 512))   # and the context still works, sparse
        assert sparse.sparse_fallbacks() == 0
    finally:
        dense.close(); sparse.close()


def test_more_points_than_the_sparse_kernels_take():
    """576 corners (the LDS refinement takes 512): repeated densely inside the call."""
    dense, sparse = _pair()
    try:
        frames = synth.board_batch(2, 2048, 1536, 24, 3, device="cuda")
        want = dense.chain(frames, 1, 2048)
        assert int(want[2].min()) >= 576
        _same(want, sparse.chain(frames, 1, 2048))
        assert sparse.sparse_fallbacks() >= 2                  # (more when a table had to grow and the call was repeated)
        ok = synth.board_batch(2, 2048, 1536, 22, 3, device="cuda")    # 484: taken
        _same(dense.chain(ok, 1, 2048), sparse.chain(ok, 1, 2048))
        assert sparse.sparse_fallbacks() == 0
    finally:
        dense.close(); sparse.close()


def test_sparse_chain_frames_without_points_and_start_level_zero():
    dense, sparse = _pair()
    try:
        frames = torch.full((3, 480, 640), 128, dtype=torch.uint8, device="cuda")
        frames[2] = synth.board_frame(640, 480, 10, 2, device="cuda")
        _same(dense.chain(frames, 3, 256), sparse.chain(frames, 3, 256))
        _same(dense.chain(frames, 0, 256), sparse.chain(frames, 0, 256))   # nothing below level 0: the dense schedule
        assert sparse.sparse_fallbacks() == 0
    finally:
        dense.close(); sparse.close()


def test_sparse_refine_1_is_the_default_and_leaves_small_calls_to_the_dense_schedule():
    """Option value 1 (the default): sparse only where it pays (>= 96 Mi frame pixels per call); chain_info says which
    ran.  0 = never (what bench.py's `value` is measured with)."""
    det = mrgingham_amd.Detector(0)
    try:
        small = synth.board_batch(4, 1024, 768, 10, 0, device="cuda")
        det.chain(small, 3, 256)
        assert det.chain_info()[1] >= 0                        # dense
        big = synth.board_batch(8, 4096, 3072, 10, 0, device="cuda")
        got = det.chain(big, 3, 256)
        assert det.chain_info()[1] == -1                       # sparse
        det.set_option("sparse_refine", 0)
        want = det.chain(big, 3, 256)
        assert det.chain_info()[1] >= 0                        # dense
        _same(want, got)
        with pytest.raises(ValueError):
            det.set_option("sparse_refine", 3)
    finally:
        det.close()


def test_sparse_fuzz_never_differs():
    """tools/sparse_fuzz.py, short form: random sizes / boards / noise / textures / start levels; the sparse schedule
    (with the library's dense repeat of what it cannot take) equals the dense output on every frame."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sparse_fuzz.py"), "120", "11"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 mismatching" in r.stdout
