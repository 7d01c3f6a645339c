"""GPU tests of the full detector, find_chessboard_from_image_array_C / find_board: level loop
(mrgingham.cc:116-139), host grid finder, refinement to level 0 (mrgingham.cc:81-99).  Expected
values: the CPU oracle for the detector and the refinement, composed with the same grid finder."""
import numpy as np
import pytest

import mrgingham_amd
from mrgingham_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


def _lattice_order(cand, lattice):
    """The candidates in the order of the lattice the frame was rendered from (synth.board_lattice), or None
    when some lattice point does not have exactly one candidate within 0.4 pitches."""
    c = cand.astype(np.float64) / 1000.0
    r2 = (0.4 * np.linalg.norm(lattice[0, 1] - lattice[0, 0])) ** 2
    out = []
    for p in lattice.reshape(-1, 2):
        near = np.nonzero(((c - p) ** 2).sum(1) <= r2)[0]
        if len(near) != 1:
            return None
        out.append(c[near[0]])
    return np.array(out)


def _expected_board(img, gridn, level, lattice=None):
    """Expected corners: the ORACLE's candidates and refinement.  With `lattice` (synthetic frames) the board
    -- which candidates, in which order -- comes from the geometry the frame was rendered from, not from
    the product's grid finder; the finder (CPU entry point) is then only asked WHETHER it accepts a level
    (its refusals at coarse levels are pinned geometrically in tests/test_grid.py)."""
    levels = [level] if level >= 0 else [3, 2, 1, 0]
    for L in levels:
        cand = oracle.find_corners(img, L)
        if cand is None or len(cand) < gridn * gridn:
            continue
        grid = mrgingham_amd.find_grid_from_points(cand, gridn)
        if grid is None:
            continue
        if lattice is not None:
            want = _lattice_order(cand, lattice)
            assert want is not None and np.array_equal(grid, want), L   # the accepted board IS the rendered lattice
            grid = want
        pts, lv = grid.copy(), np.full(gridn * gridn, L, np.int8)
        for l in range(L - 1, -1, -1):
            pts, lv, n = oracle.refine_corners(pts, lv, img, l)
            if n <= 0:
                break
        return pts, L
    return None, None


@pytest.mark.parametrize("case", [(640, 480, 10, 0), (640, 480, 10, 5), (800, 600, 14, 1), (1920, 1080, 10, 3),
                                  (4096, 3072, 10, 11)])
def test_find_board_matches_composed_oracle(case):
    w, h, gridn, seed = case
    img = synth.board_frame(w, h, gridn, seed).numpy()
    lattice = synth.board_lattice(w, h, gridn, seed)
    for level in (-1, 0, 1, 2):
        want, found_at = _expected_board(img, gridn, level, lattice)
        got = mrgingham_amd.find_board(img, image_pyramid_level=level, gridn=gridn)
        if want is None:
            assert got is None, (case, level)
            continue
        assert got is not None and got.shape == (gridn * gridn, 2), (case, level)
        assert np.abs(got - want).max() <= 1e-4 and np.array_equal(got, want), (case, level, found_at)
    # the default level search starts at level 3 (mrgingham.cc:127) and the refined corners sit on the
    # level-0 lattice of the board: rows top to bottom, columns left to right
    got = mrgingham_amd.find_chessboard(img, gridn=gridn)            # alias, mrgingham_pywrap.c:366
    assert got is not None
    c, s = np.cos(0.1), np.sin(0.1)
    U = (got[:, 0] * c + got[:, 1] * s).reshape(gridn, gridn)
    V = (-got[:, 0] * s + got[:, 1] * c).reshape(gridn, gridn)
    assert (np.diff(U, axis=1) > 0).all() and (np.diff(V, axis=0) > 0).all()


def test_find_board_negative_and_argument_paths():
    noise = synth.noise_frame(640, 480, 1, smooth=1).numpy()
    assert mrgingham_amd.find_board(noise) is None                        # no board: None (:326-330)
    assert mrgingham_amd.find_board(np.zeros((64, 64), np.uint8)) is None
    img = synth.board_frame(640, 480, 10, 0).numpy()
    assert mrgingham_amd.find_board(img, gridn=12) is None                # wrong board size
    assert mrgingham_amd.find_board(img, blobs=True, image_pyramid_level=0) is None   # a chessboard is not a circle grid
    assert mrgingham_amd.find_board(img, image_pyramid_level=11) is None
    with pytest.raises(RuntimeError, match="gridn"):
        mrgingham_amd.find_board(img, gridn=1)
    with pytest.raises(RuntimeError, match="image_pyramid_level == 0"):
        mrgingham_amd.find_board(img, blobs=True)
    with pytest.raises(RuntimeError, match="INTEGER,INTEGER"):
        mrgingham_amd.find_board(img, debug_sequence="1;2")
    assert mrgingham_amd.find_board(img, debug_sequence="3,4") is not None


def test_find_boards_batch_adaptive_levels():
    """Per-frame adaptive pyramid depth: frames whose grid is found at level 3 stop there, the others
    go on to levels 2, 1, 0; every board equals the single-frame detector's."""
    det = mrgingham_amd.Detector(0)
    import torch
    imgs = [synth.board_frame(1280, 960, 10, s).numpy() for s in (0, 1, 2)]       # found at level 3
    small = np.full((960, 1280), 200, np.uint8)                                     # a small board in a big frame:
    small[:480, :640] = synth.board_frame(640, 480, 10, 4).numpy()                  # level 3 is too coarse for it
    imgs.append(small)
    imgs.append(synth.noise_frame(1280, 960, 3, smooth=1).numpy())                  # no board at all
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    boards, found = det.find_boards(frames, gridn=10, nthreads=3)
    single = [mrgingham_amd.find_board(im, gridn=10) for im in imgs]
    assert found[4] == -1 and single[4] is None
    levels = [_expected_board(im, 10, -1)[1] for im in imgs[:4]]
    assert found[:4].tolist() == levels and 3 in levels and min(levels) < 3
    for f in range(4):
        assert single[f] is not None and np.array_equal(boards[f], single[f]), f
    assert np.isnan(boards[4]).all()
    b1, f1 = det.find_boards(frames, gridn=10, image_pyramid_level=1, nthreads=1)
    for f in range(5):
        s = mrgingham_amd.find_board(imgs[f], image_pyramid_level=1, gridn=10)
        assert (f1[f] == 1) == (s is not None)
        if s is not None:
            assert np.array_equal(b1[f], s)
    det.close()


def _mixed_batch(seed0, W=1280, H=960):
    """Boards found at level 3, one that needs a finer level, one with noise over it, one frame without a board."""
    import torch
    imgs = [synth.board_frame(W, H, 10, seed0 + s).numpy() for s in (0, 1)]
    small = np.full((H, W), 200, np.uint8)
    small[:H // 2, :W // 2] = synth.board_frame(W // 2, H // 2, 10, seed0 + 4).numpy()
    imgs.append(small)
    noisy = imgs[0].astype(np.int64) + (synth.noise_frame(W, H, seed0 + 9).numpy().astype(np.int64) - 128) * 60 // 255
    imgs.append(np.clip(noisy, 0, 255).astype(np.uint8))
    imgs.append(synth.noise_frame(W, H, seed0 + 3, smooth=1).numpy())
    return torch.from_numpy(np.stack(imgs)).cuda()


@pytest.mark.parametrize("sparse", [0, 2])
def test_find_boards_pipelined_equals_the_synchronous_dense_schedule(sparse):
    """submit / collect with several batches in flight (device passes of batch n+1 under the grid finder of batch n,
    refinement on its own stream, with option sparse_refine 2: out of the cells around the corners, frames the sparse
    kernels cannot take repeated densely on the device): boards and levels are those of the synchronous dense
    schedule (option find_boards_pipeline 0), double for double, whatever the interleaving."""
    ref, det = mrgingham_amd.Detector(0), mrgingham_amd.Detector(0)
    try:
        ref.set_option("find_boards_pipeline", 0)
        ref.set_option("sparse_refine", 0)
        det.set_option("sparse_refine", sparse)
        batches = [_mixed_batch(10 * i) for i in range(5)]
        want = [ref.find_boards(b, gridn=10, nthreads=4) for b in batches]
        assert any((w[1] == 3).any() for w in want) and any(((w[1] >= 0) & (w[1] < 3)).any() for w in want)
        assert all(w[1][4] == -1 for w in want)
        # depth 1 (submit + collect), depth 2, depth 3 (more than the sets: the oldest is completed by the submit)
        for depth in (1, 2, 3, 4):
            jobs, got = [], []
            for b in batches:
                jobs.append(det.find_boards_submit(b, gridn=10, nthreads=4))
                if len(jobs) >= depth:
                    got.append(det.find_boards_collect(jobs.pop(0)))
            while jobs:
                got.append(det.find_boards_collect(jobs.pop(0)))
            for i, ((wb, wf), (gb, gf)) in enumerate(zip(want, got)):
                assert np.array_equal(wf, gf), (depth, i, wf, gf)
                for f in range(len(wf)):
                    if wf[f] >= 0:
                        assert np.array_equal(wb[f], gb[f]), (depth, i, f)
        # collected out of order, and a single level asked for
        j0 = det.find_boards_submit(batches[0], gridn=10, image_pyramid_level=2)
        j1 = det.find_boards_submit(batches[1], gridn=10, image_pyramid_level=1)
        (b1, f1), (b0, f0) = det.find_boards_collect(j1), det.find_boards_collect(j0)
        for (bb, ff, batch, lvl) in ((b0, f0, batches[0], 2), (b1, f1, batches[1], 1)):
            wb, wf = ref.find_boards(batch, gridn=10, image_pyramid_level=lvl)
            assert np.array_equal(wf, ff) and set(ff.tolist()) <= {-1, lvl}
            for f in range(len(wf)):
                if wf[f] >= 0:
                    assert np.array_equal(wb[f], bb[f]), (lvl, f)
    finally:
        ref.close(); det.close()


def test_find_boards_pipelined_at_4096x3072_against_the_single_frame_detector():
    """The bench shape of tools/find_boards_bench.py: 12 MP frames (found at level 2, refined sparsely to level 0)
    in a pipeline of depth 2, against find_board() on every frame."""
    import torch
    det = mrgingham_amd.Detector(0)
    try:
        frames = synth.board_batch(6, 4096, 3072, 10, 40, device="cuda")
        batches = [frames[:3].contiguous(), frames[3:].contiguous(), frames[:3].contiguous(), frames.repeat(3, 1, 1)]
        jobs = [det.find_boards_submit(b, gridn=10) for b in batches[:2]]
        out = [det.find_boards_collect(jobs[0])]
        jobs.append(det.find_boards_submit(batches[2], gridn=10))
        out.append(det.find_boards_collect(jobs[1]))
        jobs.append(det.find_boards_submit(batches[3], gridn=10))         # another batch size: the rotation is re-planned
        out.append(det.find_boards_collect(jobs[2]))
        out.append(det.find_boards_collect(jobs[3]))
        single = [mrgingham_amd.find_board(frames[f].cpu().numpy(), gridn=10) for f in range(6)]
        assert all(s is not None for s in single)
        for (bb, ff), idx in zip(out, ([0, 1, 2], [3, 4, 5], [0, 1, 2], list(range(6)) * 3)):
            assert (ff >= 0).all() and (ff < 3).all(), ff
            for k, f in enumerate(idx):
                assert np.array_equal(bb[k], single[f]), (k, f)
    finally:
        det.close()


def test_find_boards_pipelined_with_mixed_resolutions_in_flight():
    """BASELINE config 5's stream: consecutive jobs of DIFFERENT frame sizes in flight at once (every job keeps its
    level sizes with its own scratch set), growing and shrinking; boards equal the synchronous dense schedule's."""
    import torch
    ref, det = mrgingham_amd.Detector(0), mrgingham_amd.Detector(0)
    try:
        ref.set_option("find_boards_pipeline", 0)
        ref.set_option("sparse_refine", 0)
        det.set_option("sparse_refine", 2)
        shapes = [(640, 480, 5), (1920, 1080, 3), (1280, 800, 4), (2560, 1440, 2), (640, 480, 7), (4096, 2160, 2), (1280, 800, 1)]
        batches = [synth.board_batch(B, W, H, 10, 7 * i, device="cuda") for i, (W, H, B) in enumerate(shapes)]
        want = [ref.find_boards(b, gridn=10) for b in batches]
        for depth in (2, 3):
            jobs, got = [], []
            for b in batches + batches[::-1]:
                jobs.append(det.find_boards_submit(b, gridn=10))
                if len(jobs) >= depth:
                    got.append(det.find_boards_collect(jobs.pop(0)))
            while jobs:
                got.append(det.find_boards_collect(jobs.pop(0)))
            for i, (gb, gf) in enumerate(got):
                wb, wf = (want + want[::-1])[i]
                assert np.array_equal(wf, gf) and (wf >= 0).all(), (depth, i, wf, gf)
                assert np.array_equal(wb, gb), (depth, i)
    finally:
        ref.close(); det.close()


def test_find_boards_fuzz_never_differs_from_the_synchronous_dense_schedule():
    """tools/find_boards_fuzz.py, short form: random sizes / boards / noise / textures / frames without a board /
    requested levels / batch sizes / pipeline depths through submit / collect; boards and levels equal the synchronous
    dense schedule's on every frame."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "find_boards_fuzz.py"), "40", "3"], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 mismatching" in r.stdout


def test_other_calls_between_submit_and_collect_complete_the_batches_in_flight():
    """A chain / detect / set_option on the same context while find_boards batches are in flight: the library completes
    those batches first (they stay collectable, with the right boards), and a context destroyed with batches in flight
    goes down cleanly."""
    det, ref = mrgingham_amd.Detector(0), mrgingham_amd.Detector(0)
    try:
        ref.set_option("find_boards_pipeline", 0)
        a, b = _mixed_batch(50), _mixed_batch(60, 1920, 1080)
        wa, wb = ref.find_boards(a, gridn=10), ref.find_boards(b, gridn=10)
        ja = det.find_boards_submit(a, gridn=10)
        jb = det.find_boards_submit(b, gridn=10)
        pts, lv, npts = det.chain(b, 3, 512)                    # rotates through the scratch sets: ja and jb are completed first
        xy, cnt = det.detect(a, 2, capacity=1024)
        det.set_option("scratch_sets", 2)
        for job, (wbd, wf) in ((jb, wb), (ja, wa)):
            gb, gf = det.find_boards_collect(job)
            assert np.array_equal(gf, wf)
            for f in range(len(wf)):
                if wf[f] >= 0:
                    assert np.array_equal(gb[f], wbd[f])
        assert int(npts[0]) >= 100 and int(cnt[0]) >= 100
        with pytest.raises(RuntimeError):
            det.find_boards_collect(ja)                         # a ticket is collected once
        det.find_boards_submit(a, gridn=10)                     # ... and left in flight
        det.find_boards_submit(b, gridn=10)
    finally:
        det.close(); ref.close()
