"""CPU tests of the host-side grid finder (mrgingham_amd/csrc/grid.cpp, the restatement of
find_grid.cc).  The reference ships no test for it and needs boost to build, so these are
geometric ground-truth tests: known boards in, the same corners in board order out."""
import numpy as np
import pytest

import mrgingham_amd
from mrgingham_amd import synth
from oracle import oracle


def _board(gridn, H, jitter=0.0, seed=0, outliers=0, shuffle=True, origin=(400.0, 300.0), pitch=60.0):
    """gridn x gridn corners through a homography (mild perspective + rotation), as x1000 ints."""
    rng = np.random.RandomState(seed)
    ii, jj = np.meshgrid(np.arange(gridn), np.arange(gridn), indexing="ij")       # ii = row, jj = column
    p = np.stack([jj.ravel() * pitch, ii.ravel() * pitch, np.ones(gridn * gridn)])
    q = H @ p
    xy = (q[:2] / q[2]).T + np.array(origin)
    xy = xy + rng.normal(0, jitter, xy.shape)
    truth = np.round(xy * 1000).astype(np.int64)
    pts = truth.copy()
    if outliers:
        lo, hi = truth.min(0) - 200000, truth.max(0) + 200000
        pts = np.concatenate([pts, np.stack([rng.randint(lo[0], hi[0], outliers), rng.randint(lo[1], hi[1], outliers)], 1)])
    if shuffle:
        pts = pts[rng.permutation(len(pts))]
    return pts.astype(np.int32), truth


def _rot(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


@pytest.mark.parametrize("gridn", [3, 6, 10, 14])
@pytest.mark.parametrize("deg", [0, 7, -12, 20])
def test_recovers_board_in_row_major_order(gridn, deg):
    Hm = _rot(deg)
    Hm[2, 0], Hm[2, 1] = 2e-4, -1e-4                       # perspective
    pts, truth = _board(gridn, Hm, jitter=0.15, seed=gridn * 100 + deg + 50)
    got = mrgingham_amd.find_grid_from_points(pts, gridn)
    assert got is not None, (gridn, deg)
    assert got.shape == (gridn * gridn, 2)
    assert np.array_equal(np.round(got * 1000).astype(np.int64), truth), (gridn, deg)


def test_outliers_around_the_board_are_ignored():
    Hm = _rot(5)
    Hm[2, 0] = 1.5e-4
    pts, truth = _board(10, Hm, jitter=0.2, seed=3, outliers=25)
    # outliers inside the board break its lattice; keep only those outside the hull of the grid
    keep = []
    lo, hi = truth.min(0), truth.max(0)
    for p in pts:
        inside = lo[0] - 30000 < p[0] < hi[0] + 30000 and lo[1] - 30000 < p[1] < hi[1] + 30000
        if not inside or (truth == p).all(1).any():
            keep.append(p)
    got = mrgingham_amd.find_grid_from_points(np.array(keep, np.int32), 10)
    assert got is not None and np.array_equal(np.round(got * 1000).astype(np.int64), truth)


def test_no_grid_cases():
    Hm = _rot(3)
    pts, truth = _board(10, Hm, seed=1)
    assert mrgingham_amd.find_grid_from_points(pts, 12) is None                  # wrong size
    missing = np.array([p for p in pts if not (p == truth[44]).all()], np.int32)
    assert mrgingham_amd.find_grid_from_points(missing, 10) is None              # a hole in the lattice
    assert mrgingham_amd.find_grid_from_points(pts[:50], 10) is None             # too few points
    rng = np.random.RandomState(0)
    assert mrgingham_amd.find_grid_from_points(rng.randint(0, 2000000, (150, 2)).astype(np.int32), 10) is None
    assert mrgingham_amd.find_grid_from_points(np.zeros((0, 2), np.int32), 10) is None
    line = np.stack([np.arange(120) * 10000, np.arange(120) * 5000], 1).astype(np.int32)
    assert mrgingham_amd.find_grid_from_points(line, 10) is None                 # all collinear
    dup = np.concatenate([pts, pts[:7]])                                         # duplicated candidates
    got = mrgingham_amd.find_grid_from_points(dup, 10)
    assert got is not None and np.array_equal(np.round(got * 1000).astype(np.int64), truth)


def test_grid_from_detector_candidates_of_synthetic_frames():
    """Candidates of the (oracle) detector on rendered boards: 100 / 196 corners in board order."""
    for (w, h, gridn, seed, level) in [(640, 480, 10, 0, 0), (640, 480, 10, 5, 1), (800, 600, 14, 1, 0),
                                      (1280, 960, 10, 2, 2)]:
        img = synth.board_frame(w, h, gridn, seed).numpy()
        cand = oracle.find_corners(img, level)
        got = mrgingham_amd.find_grid_from_points(cand, gridn)
        assert got is not None, (w, h, gridn, level)
        # every output point is one of the candidates, all distinct
        as_int = np.round(got * 1000).astype(np.int64)
        cset = {tuple(c) for c in cand.tolist()}
        assert all(tuple(p) in cset for p in as_int.tolist()) and len({tuple(p) for p in as_int.tolist()}) == gridn * gridn
        # board order: the synthetic board is rotated by +0.1 rad; rows advance along its v axis, columns along u
        c, s = np.cos(0.1), np.sin(0.1)
        u = got[:, 0] * c + got[:, 1] * s
        v = -got[:, 0] * s + got[:, 1] * c
        U, V = u.reshape(gridn, gridn), v.reshape(gridn, gridn)
        assert (np.diff(U, axis=1) > 0).all() and (np.diff(V, axis=0) > 0).all()
        assert np.abs(np.diff(U, axis=0)).max() < 0.2 * np.diff(U, axis=1).min()
