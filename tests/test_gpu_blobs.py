"""Rows a-19 / (f)-4: the blob path, find_blobs_from_image_array (find_blobs.cc:14-46) = cv::SimpleBlobDetector
with mrgingham's parameters, through the reference's own entry points (find_points(blobs=True),
find_board(blobs=True), the tool's --blobs).  Expected values: oracle/blobs_oracle.c, the sequential
restatement of OpenCV's published algorithm (parity unpinned: OpenCV is not available here).  Everything
is compared exactly -- the keypoints are (x, y) * 1000 integers, in the detector's output order."""
import subprocess

import numpy as np
import pytest

import mrgingham_amd
from mrgingham_amd import synth
from oracle import oracle

from test_cli import CLI, _parse, _write_pgm

pytestmark = pytest.mark.gpu


def _blobs(img):
    return np.round(mrgingham_amd.find_points(img, image_pyramid_level=0, blobs=True) * 1000).astype(np.int64)


@pytest.mark.parametrize("case", [(640, 480, 10, 0), (800, 600, 7, 2), (1280, 960, 10, 5), (333, 251, 4, 1)])
def test_circle_grids(case):
    w, h, gridn, seed = case
    img = synth.dots_frame(w, h, gridn, seed).numpy()
    want = oracle.find_blobs(img)
    got = _blobs(img)
    assert np.array_equal(got, want.astype(np.int64)), case
    assert len(want) >= gridn * gridn
    # the discs sit on the lattice the frame was rendered from
    lat = synth.board_lattice(w, h, gridn, seed).reshape(-1, 2)
    d = np.sqrt(((lat[:, None, :] - got[None, :, :] / 1000.0) ** 2).sum(-1))
    assert d.min(1).max() < 0.25


def test_find_board_with_blobs_orders_the_grid():
    w, h, gridn, seed = 800, 600, 10, 3
    img = synth.dots_frame(w, h, gridn, seed).numpy()
    board = mrgingham_amd.find_board(img, image_pyramid_level=0, gridn=gridn, blobs=True)
    assert board is not None and board.shape == (gridn * gridn, 2)
    want = mrgingham_amd.find_grid_from_points(oracle.find_blobs(img), gridn)      # bridge.cc:104-113: no refinement
    assert np.array_equal(board, want)
    lat = synth.board_lattice(w, h, gridn, seed).reshape(-1, 2)
    assert np.abs(board - lat).max() < 0.25                                          # and in the lattice's order
    assert mrgingham_amd.find_board(img, image_pyramid_level=0, gridn=gridn + 1, blobs=True) is None


def _shapes(h, w):
    """Hand-made shapes that exercise the filters: discs of several sizes, a ring (its outer border has a dark
    centre), a bar (inertia), an L (convexity), blobs that touch the frame, tiny specks (area), a big square
    (area >= 80000 at some thresholds), a soft-edged disc (different contours per threshold)."""
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.full((h, w), 200, np.float64)

    def disc(cx, cy, r, v=30.0, soft=0.0):
        d = np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2)
        if soft > 0:
            img[:] = np.minimum(img, v + (200 - v) * np.clip((d - r) / soft, 0, 1))
        else:
            img[d <= r] = v
    disc(60, 60, 12); disc(120, 60, 6); disc(160, 60, 2.2); disc(200, 60, 4.5); disc(260, 60, 25)
    disc(70, 150, 18, soft=12); disc(150, 150, 10, v=120); disc(210, 150, 10, v=180)
    d = np.sqrt((xx - 300) ** 2 + (yy - 150) ** 2)
    img[(d <= 22) & (d >= 12)] = 40                                   # ring
    img[220:228, 40:160] = 20                                         # bar, 8 x 120: inertia ratio below 0.1
    img[250:300, 200:212] = 25; img[288:300, 200:260] = 25            # L shape: convexity below 0.95
    disc(0, 240, 14); disc(w - 1, 100, 9); disc(180, h - 1, 11)       # cut by the frame
    img[10:12, 300:303] = 10                                          # speck
    if h > 330 and w > 420:
        img[h - 320:h - 10, w - 310:w - 20] = 60                      # 310 x 290 = 89900 px: above maxArea
        disc(w - 160, h - 160, 30, v=230)                             # a light disc inside it: a hole of a dark blob
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(320, 360), (400, 640), (697, 811)])
def test_filters_on_hand_made_shapes(shape):
    h, w = shape
    img = _shapes(h, w)
    want = oracle.find_blobs(img)
    assert np.array_equal(_blobs(img), want.astype(np.int64)), shape
    assert 4 <= len(want) <= 16


@pytest.mark.parametrize("seed,smooth", [(0, 1), (1, 2), (2, 3), (3, 4)])
def test_noise_frames(seed, smooth):
    """Smoothed noise: thousands of ragged borders per threshold, nearly all rejected -- the same few survive."""
    img = synth.noise_frame(320, 240, seed, smooth=smooth).numpy()
    assert np.array_equal(_blobs(img), oracle.find_blobs(img).astype(np.int64)), (seed, smooth)


def test_degenerate_frames():
    for img in (np.zeros((64, 64), np.uint8), np.full((64, 64), 255, np.uint8), np.zeros((1, 1), np.uint8),
                (np.indices((40, 50)).sum(0) % 2 * 255).astype(np.uint8)):
        got = mrgingham_amd.find_points(img, image_pyramid_level=0, blobs=True)
        want = oracle.find_blobs(img)
        assert got.shape == (len(want), 2) and np.array_equal(np.round(got * 1000).astype(np.int64), want.astype(np.int64))
    with pytest.raises(RuntimeError):
        mrgingham_amd.find_points(np.zeros((64, 64), np.uint8), image_pyramid_level=1, blobs=True)


def test_cli_blobs(tmp_path):
    """--blobs: preprocessing (CLAHE + blur) -> blobs -> grid finder; level column 0, no refinement
    (mrgingham-from-image.cc:153-160, :174-183)."""
    w, h, gridn, seed = 800, 600, 10, 1
    img = synth.dots_frame(w, h, gridn, seed).numpy()
    p = str(tmp_path / "dots.pgm")
    _write_pgm(p, img)
    r = subprocess.run([CLI, "--blobs", "--gridn", str(gridn), p], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows = _parse(r.stdout)[p]
    pre = oracle.preprocess(img, clahe=True, blur_radius=1)
    want = mrgingham_amd.find_grid_from_points(oracle.find_blobs(pre), gridn)
    assert want is not None and len(rows) == gridn * gridn
    g = np.array([(x, y) for x, y, _ in rows])
    assert np.abs(g - want).max() < 1e-6 and all(lv == 0 for _, _, lv in rows)


def _random_scene(rng, h, w):
    """Random strokes and shapes on a random background: discs, rings, rectangles, one-pixel-wide horizontal /
    vertical / diagonal lines (borders whose outer-type and hole-type starts coincide), shapes cut by every edge of
    the frame (the last column included), grey levels on both sides of several thresholds, optional noise."""
    img = np.full((h, w), rng.choice([40, 120, 200, 250]), np.float64)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(rng.randrange(5, 40)):
        kind = rng.choice(["disc", "ring", "rect", "hline", "vline", "dline", "dot"])
        v = rng.choice([0, 30, 55, 95, 125, 160, 205, 255])
        cx, cy = rng.randrange(-5, w + 5), rng.randrange(-5, h + 5)
        if kind == "disc":
            img[(xx - cx) ** 2 + (yy - cy) ** 2 <= rng.uniform(1.5, 40) ** 2] = v
        elif kind == "ring":
            r = rng.uniform(6, 40)
            d2 = (xx - cx) ** 2 + (yy - cy) ** 2
            img[(d2 <= r * r) & (d2 >= (r * rng.uniform(0.3, 0.9)) ** 2)] = v
        elif kind == "rect":
            x1, y1 = cx + rng.randrange(1, 120), cy + rng.randrange(1, 120)
            img[max(cy, 0):max(y1, 0), max(cx, 0):max(x1, 0)] = v
        elif kind == "hline":
            if 0 <= cy < h:
                img[cy, max(cx, 0):max(cx + rng.randrange(2, 200), 0)] = v
        elif kind == "vline":
            if 0 <= cx < w:
                img[max(cy, 0):max(cy + rng.randrange(2, 200), 0), cx] = v
        elif kind == "dline":
            n, sgn = rng.randrange(2, 150), rng.choice([-1, 1])
            for i in range(n):
                x, y = cx + i, cy + sgn * i
                if 0 <= x < w and 0 <= y < h:
                    img[y, x] = v
        else:
            if 0 <= cx < w and 0 <= cy < h:
                img[cy, cx] = v
    if rng.random() < 0.5:
        nz = synth.noise_frame(w, h, rng.randrange(1 << 16), smooth=rng.choice([0, 1, 2])).numpy().astype(np.float64)
        img += (nz - 128) * rng.choice([0.05, 0.2, 0.6])
    return np.clip(img, 0, 255).astype(np.uint8)


def test_random_scenes_against_the_oracle():
    """Short by default; MRG_FUZZ_ITERS=1500 for the long form (run on the GPU box: 0 mismatches)."""
    import os
    import random
    rng = random.Random(int(os.environ.get("MRG_FUZZ_SEED", "3")))
    for it in range(int(os.environ.get("MRG_FUZZ_ITERS", "40"))):
        w, h = rng.randrange(8, 700), rng.randrange(8, 500)
        if rng.random() < 0.3:
            w = rng.choice([31, 32, 33, 63, 64, 65, 96, 256, 257])       # word boundaries of the bit planes
        img = _random_scene(rng, h, w)
        want = oracle.find_blobs(img)
        got = _blobs(img)
        assert got.shape == (len(want), 2) and np.array_equal(got, want.astype(np.int64)), (it, w, h)


def test_product_against_the_committed_vectors(golden_dir):
    """The HIP path against tests/golden/blobs_golden.npz directly (no oracle in the loop): blob keypoints and the
    8 / 16 bit preprocessing."""
    import hashlib
    import os
    from test_oracle import _blob_scenes
    z = np.load(os.path.join(golden_dir, "blobs_golden.npz"))
    for name, img in _blob_scenes().items():
        assert hashlib.sha256(img.tobytes()).digest() == z["sha_" + name].tobytes(), name
        assert np.array_equal(_blobs(img), z["blobs_" + name]), name
    pre = synth.board_frame(333, 251, 6, 3).numpy()
    assert np.array_equal(mrgingham_amd.preprocess(pre, clahe=True, blur_radius=1), z["pre8_board_333x251"])
    pre16 = (pre.astype(np.uint16) * 120 + 9000).astype(np.uint16)
    assert np.array_equal(mrgingham_amd.api.preprocess16(pre16, clahe=True, blur_radius=1), z["pre16_board_333x251"])


@pytest.mark.parametrize("kind", ["dots_4096x3072", "board_4096x3072", "noise_2048x1536", "random_3000x2000"])
def test_large_frames(kind):
    """Arcs that cross many words and rows, the frame's edges at full length, ~10^6 candidates per call."""
    import random
    if kind == "dots_4096x3072":
        img = synth.dots_frame(4096, 3072, 10, 1).numpy()
    elif kind == "board_4096x3072":
        img = synth.board_frame(4096, 3072, 10, 4).numpy()
    elif kind == "noise_2048x1536":
        img = synth.noise_frame(2048, 1536, 3, smooth=2).numpy()
    else:
        img = _random_scene(random.Random(77), 2000, 3000)
    want = oracle.find_blobs(img)
    got = _blobs(img)
    assert got.shape == (len(want), 2) and np.array_equal(got, want.astype(np.int64)), kind
