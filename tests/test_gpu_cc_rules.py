"""The HIP component-search kernels (cc_label / cc_detect / cc_refine, csrc/cc.hip) driven with
HAND-BUILT and random sparse responses through mrgingham_amd_cc_on_response_batch: every rule of
find_chessboard_corners.cc:159-267 / :284-397 is met on purpose here, not only when a natural image
happens to produce it.  Expectations: the hand-derived values of tests/cc_cases.py AND the oracle on
the same buffers, bit-exact (integers and order; refined doubles compared for equality)."""
import numpy as np
import pytest
import torch

import mrgingham_amd
from oracle import oracle

import cc_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["lds", "global"])
def det(request):
    """Both implementations of the search: out of LDS (frames with <= 2048 hot pixels; larger ones fall
    through to the other) and the global-memory kernels alone (option cc_lds = 0)."""
    d = mrgingham_amd.Detector(0)
    d.set_option("cc_lds", 1 if request.param == "lds" else 0)
    yield d
    d.close()


def _detect(det, ds, imgs, level=0, capacity=4096):
    xy, cnt = det.cc_detect_on_response(torch.from_numpy(np.stack(ds)).cuda(), torch.from_numpy(np.stack(imgs)).cuda(),
                                        level=level, capacity=capacity)
    xy, cnt = xy.cpu().numpy(), cnt.cpu().numpy()
    return [xy[f, :cnt[f]] for f in range(len(ds))]


def test_handbuilt_detect_rules_one_frame_at_a_time(det):
    for name, d, img, expected in cc_cases.detect_cases():
        got = _detect(det, [d], [img])[0]
        want = oracle.cc_detect_on_response(d, img)
        assert np.array_equal(got, want), name
        if expected is not None:
            assert got.tolist() == expected, name


def test_handbuilt_detect_rules_as_one_batch(det):
    cases = cc_cases.detect_cases()
    got = _detect(det, [c[1] for c in cases], [c[2] for c in cases])
    for (name, d, img, expected), g in zip(cases, got):
        assert np.array_equal(g, oracle.cc_detect_on_response(d, img)), name


def test_detect_does_not_modify_the_callers_response_and_clamps(det):
    name, d, img, expected = cc_cases.detect_cases()[1]
    d = d.copy()
    d[5, 5] = 900          # outside [7, w-7) x [7, h-7): structurally zero in the reference (:506)
    d[20, 22] = -300       # negative: clamped (:527-529); would otherwise never matter
    dd = torch.from_numpy(d[None]).cuda()
    before = dd.clone()
    xy, cnt = det.cc_detect_on_response(dd, torch.from_numpy(img[None]).cuda())
    assert torch.equal(dd, before)
    assert xy[0, :int(cnt[0])].cpu().tolist() == expected


@pytest.mark.parametrize("level", [0, 1, 3])
def test_detect_level_scaling(det, level):
    name, d, img, _ = cc_cases.detect_cases()[1]
    got = _detect(det, [d], [img], level=level)[0]
    assert np.array_equal(got, oracle.cc_detect_on_response(d, img, level=level))


@pytest.mark.parametrize("shape,nblobs,noise", [((48, 64), 12, 0.0), ((96, 131), 60, 0.002), ((97, 129), 80, 0.01),
                                                ((240, 333), 300, 0.004), ((64, 1001), 200, 0.003),
                                                ((600, 77), 150, 0.003), ((15, 15), 3, 0.1), ((16, 40), 6, 0.05)])
def test_random_sparse_responses(det, shape, nblobs, noise):
    h, w = shape
    rng = np.random.RandomState(h * 1009 + w)
    B = 6
    ds = [cc_cases.random_sparse_response(rng, h, w, nblobs, noise) for _ in range(B)]
    imgs = [cc_cases.flat_img(h, w) if f % 3 else (rng.randint(0, 256, size=(h, w)).astype(np.uint8)) for f in range(B)]
    imgs[1] = (rng.randint(0, 40, size=(h, w)) + 100).astype(np.uint8)     # low variance: sigma ~ 11.5 < 20
    got = _detect(det, ds, imgs, capacity=8192)
    total = 0
    for f in range(B):
        want = oracle.cc_detect_on_response(ds[f], imgs[f])
        assert np.array_equal(got[f], want), (shape, f, len(got[f]), len(want))
        total += len(want)
    if h >= 48:
        assert total > 0


def test_dense_texture_response(det):
    """Every other pixel hot: one huge super-component per frame (tables at one entry per pixel)."""
    h, w = 64, 96
    rng = np.random.RandomState(5)
    d = rng.randint(16, 400, size=(h, w)).astype(np.int16)
    img = cc_cases.flat_img(h, w)
    det.set_option("hot_capacity_shift", 0)
    try:
        got = _detect(det, [d], [img])[0]
    finally:
        det.set_option("hot_capacity_shift", 7)
    assert np.array_equal(got, oracle.cc_detect_on_response(d, img))


def test_capacity_overflow_is_reported(det):
    h, w = 64, 96
    d = np.full((h, w), 100, np.int16)
    img = cc_cases.flat_img(h, w)
    d2 = mrgingham_amd.Detector(0)          # a fresh context: tables at their default size
    try:
        with pytest.raises(RuntimeError, match="overflowed"):
            d2.cc_detect_on_response(torch.from_numpy(d[None]).cuda(), torch.from_numpy(img[None]).cuda(), retry=False)
        # the tables grew: asked again (the Python method does that itself), the frame goes through
        assert np.array_equal(_detect(d2, [d], [img])[0], oracle.cc_detect_on_response(d, img))
    finally:
        d2.close()
    # and the context recovers
    name, d, img, expected = cc_cases.detect_cases()[1]
    assert _detect(det, [d], [img])[0].tolist() == expected


def _refine(det, pts, lv, d, img, level):
    P = max(1, len(lv))
    tp = torch.zeros((1, P, 2), dtype=torch.float64)
    tl = torch.zeros((1, P), dtype=torch.int8)
    tp[0, :len(lv)] = torch.from_numpy(np.asarray(pts, np.float64).reshape(-1, 2))
    tl[0, :len(lv)] = torch.from_numpy(np.asarray(lv, np.int8))
    tp, tl = tp.cuda(), tl.cuda()
    n = torch.tensor([len(lv)], dtype=torch.int32).cuda()
    nref = det.cc_refine_on_response(torch.from_numpy(d[None]).cuda(), torch.from_numpy(img[None]).cuda(), level,
                                     tp, tl, n)
    return tp[0, :len(lv)].cpu().numpy(), tl[0, :len(lv)].cpu().numpy(), int(nref[0])


def test_handbuilt_refine_rules(det):
    for name, pts, lv, d, img, level, expected in cc_cases.refine_cases():
        gp, gl, gn = _refine(det, pts, lv, d, img, level)
        wp, wl, wn = oracle.cc_refine_on_response(pts, lv, d, img, level)
        assert gn == wn and np.array_equal(gl, wl) and np.array_equal(gp, wp), name
        if expected is not None:
            ep, el, en = expected
            assert gn == en and gl.tolist() == el and np.array_equal(gp, ep), name


@pytest.mark.parametrize("shape,level", [((96, 131), 0), ((240, 333), 1), ((97, 129), 2)])
def test_random_refine_against_oracle(det, shape, level):
    h, w = shape
    rng = np.random.RandomState(h + 31 * w + level)
    for trial in range(4):
        d = cc_cases.random_sparse_response(rng, h, w, 60 + 40 * trial, 0.003)
        img = cc_cases.flat_img(h, w) if trial % 2 else rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        # points: near hot pixels (full-resolution coordinates), duplicates, far-away, out-of-image
        ys, xs = np.nonzero(d > 15)
        k = min(len(xs), 120)
        sel = rng.choice(len(xs), size=k, replace=True)
        scale = float(1 << level)
        pts = np.stack([(xs[sel] + rng.uniform(-1.6, 1.6, k) + 0.5) * scale - 0.5,
                        (ys[sel] + rng.uniform(-1.6, 1.6, k) + 0.5) * scale - 0.5], axis=1)
        pts = np.concatenate([pts, pts[:10], [[-5.0, -5.0], [w * scale + 3, h * scale + 3], [0.0, 0.0]]])
        lv = np.full(len(pts), level + 1, np.int8)
        lv[::7] = level + 2                              # not refinable at this level
        lv[3::11] = level                                # already there
        gp, gl, gn = _refine(det, pts, lv, d, img, level)
        wp, wl, wn = oracle.cc_refine_on_response(pts, lv, d, img, level)
        assert gn == wn and np.array_equal(gl, wl) and np.array_equal(gp, wp), (shape, level, trial)
        assert wn > 0


# ----------------------------------------------------------------------------- which implementation ran

def _blob_field(h, w, nblobs, size, rng, row_step=3, y_first=12):
    """`nblobs` separate blobs of `size` hot pixels each (a row of pixels), on a grid, all accepted-looking.
    row_step 2 leaves one row without hot pixels between blob rows (the band planner needs two), 3 two, 5 four."""
    d = np.zeros((h, w), np.int16)
    per_row = (w - 40) // (size + 3)
    for k in range(nblobs):
        y = y_first + row_step * (k // per_row)
        x = 12 + (size + 3) * (k % per_row)
        assert y < h - 12
        d[y, x:x + size] = rng.randint(130, 400, size)
    return d


def _slanted_field(h, w, nrows, size, slope, rng, y_first=12, row_step=14):
    """Rows of blobs like _blob_field, but every row runs along y = y_row + slope * x: between such rows no image
    row is free of hot pixels (for |slope| * w > row_step), a sheared row is."""
    d = np.zeros((h, w), np.int16)
    per_row = (w - 40) // (size + 3)
    for r in range(nrows):
        for c in range(per_row):
            x = 12 + (size + 3) * c
            y = y_first + row_step * r + int(round(slope * (x if slope >= 0 else x - w)))
            assert 12 <= y < h - 12, (r, c, y)
            d[y, x:x + size] = rng.randint(130, 400, size)
            d[y + 1, x:x + size // 2] = rng.randint(130, 400, size // 2)   # two rows tall: no pair of empty image rows in between
    return d


def _points_near_hot(d, npts, rng, replace=False):
    ys, xs = np.nonzero(d > 15)
    sel = rng.choice(len(xs), size=npts, replace=replace)
    return np.stack([xs[sel] + rng.uniform(-1, 1, npts), ys[sel] + rng.uniform(-1, 1, npts)], axis=1)


def test_every_fallback_reason_of_the_lds_path():
    """Frames that fit are handled out of LDS in one pass, frames with more hot pixels band by band when their
    rows offer separators; each reason for leaving a frame to the global-memory kernels is met on purpose -- hot
    pixels that cannot be banded, too many of them, multi-pixel components, LIFO demand (in the first band
    and in a later one), points -- in ONE batch with frames that fit, and every frame equals the oracle."""
    rng = np.random.RandomState(3)
    h, w = 400, 600
    late = _blob_field(h, w, 180, 8, rng, row_step=5)                   # 1440 hot in rows 12..27, then a solid block
    late[200:244, 100:144] = rng.randint(200, 300, (44, 44))            # 1936 hot, degree sum 7568: the second band gives up
    frames = {
        "fits": _blob_field(h, w, 100, 6, rng),                          # 600 hot pixels, 100 components
        "2400 hot pixels, bands": _blob_field(h, w, 300, 8, rng, row_step=5),
        "2400 hot pixels, no separators": _blob_field(h, w, 300, 8, rng, row_step=2),   # one empty row between blob rows
        "513+ components": _blob_field(h, w, 600, 3, rng),               # 1800 hot, 600 components
        "LIFO demand": np.zeros((h, w), np.int16),                       # one solid 44 x 44 block
        "fits too": _blob_field(h, w, 40, 12, rng),
        "9600 hot pixels, bands": _blob_field(h, w, 1200, 8, rng, row_step=5),
        "16385+ hot pixels": _blob_field(h, w, 2100, 8, rng, row_step=5),
        "LIFO demand in the second band": late,
        "3600 hot pixels in slanted rows": _slanted_field(h, w, 6, 8, 0.25, rng),
        "4800 hot pixels in rows slanted the other way": _slanted_field(h, w, 8, 8, -0.4, rng),
    }
    frames["LIFO demand"][100:144, 100:144] = rng.randint(200, 300, (44, 44))
    names = list(frames)
    ds = [frames[n] for n in names]
    img = cc_cases.flat_img(h, w)
    wants = [oracle.cc_detect_on_response(d, img) for d in ds]
    assert len(wants[6]) == 1200 and len(wants[1]) == 300

    def run(det, expect_paths):
        got = _detect(det, ds, [img] * len(ds), capacity=4096)
        paths = det.debug_paths(0, len(ds))
        assert paths.tolist() == expect_paths, dict(zip(names, paths.tolist()))
        for n, g, want in zip(names, got, wants):
            assert np.array_equal(g, want), n

    for mode, expect in [(1, [1, 1, 0, 0, 0, 1, 1, 0, 0, 1, 1]), (1 | 256, [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0])]:
        det = mrgingham_amd.Detector(0)
        try:
            det.set_option("cc_lds", mode)
            det.set_option("hot_capacity_shift", 1)     # room in the global-memory tables for the 16800-pixel frame
            run(det, expect)
            run(det, expect)
        finally:
            det.close()

    def refine(det, d, pts):
        lv = np.ones(len(pts), np.int8)
        tp = torch.from_numpy(pts[None].copy()).cuda()
        tl = torch.from_numpy(lv[None].copy()).cuda()
        n = torch.tensor([len(pts)], dtype=torch.int32).cuda()
        nref = det.cc_refine_on_response(torch.from_numpy(d[None]).cuda(), torch.from_numpy(img[None]).cuda(), 0, tp, tl, n)
        path = det.debug_paths(0, 1).tolist()
        wp, wl, wn = oracle.cc_refine_on_response(pts, lv, d, img, 0)
        assert int(nref[0]) == wn, (int(nref[0]), wn, path)
        assert np.array_equal(tl[0].cpu().numpy(), wl) and np.array_equal(tp[0].cpu().numpy(), wp), path
        return path

    det = mrgingham_amd.Detector(0)
    try:
        # refine: more than 512 points per frame goes to the global-memory kernels, 512 stay in LDS
        for npts, want_path in [(512, 1), (513, 0)]:
            assert refine(det, frames["fits"], _points_near_hot(frames["fits"], npts, rng, replace=True)) == [want_path]
        # refine band by band: points in every band, several per component (replayed in index order), and
        # points with nothing hot around them
        for name in ("2400 hot pixels, bands", "9600 hot pixels, bands", "3600 hot pixels in slanted rows",
                     "4800 hot pixels in rows slanted the other way"):
            d = frames[name]
            pts = np.concatenate([_points_near_hot(d, 300, rng), _points_near_hot(d, 100, rng, replace=True),
                                  [[300.0, 390.0], [5.0, 5.0]]])
            assert refine(det, d, pts[rng.permutation(len(pts))]) == [1], name
        assert refine(det, frames["2400 hot pixels, no separators"],
                      _points_near_hot(frames["2400 hot pixels, no separators"], 300, rng)) == [0]
        # the first band is refined out of LDS, the second gives up: the global-memory kernel finishes the frame.
        # Two points on one blob of the FIRST band: the second of them finds its component consumed, there and
        # again in the kernel that finishes the frame
        d = frames["LIFO demand in the second band"]
        ys, xs = np.nonzero(d[:100] > 15)
        twin = np.array([[xs[5] + 0.2, ys[5] - 0.3], [xs[5] + 0.4, ys[5] + 0.1]])
        pts = np.concatenate([_points_near_hot(d, 200, rng), twin, [[120.0, 220.0], [130.0, 230.0]]])
        assert refine(det, d, pts) == [2]
    finally:
        det.close()


@pytest.mark.parametrize("angle", [12.0, -27.0])
def test_rotated_dense_board_is_searched_band_by_band(angle):
    """A 14x14 board rotated in the image: ~2600 hot pixels at level 0, corner rows on slanted lines, so no pair
    of image rows between them is free of hot pixels.  The band planner follows the slant (sheared rows), the
    whole chain stays in LDS, and corners and levels equal the oracle's."""
    from scipy import ndimage
    from mrgingham_amd import synth
    base = synth.board_frame(2560, 1920, 14, 4).numpy()
    frame = ndimage.rotate(base, angle, reshape=False, order=1, mode="nearest").astype(np.uint8)
    frames = np.stack([frame, base])
    det = mrgingham_amd.Detector(0)
    try:
        pts, lv, npts = det.chain(torch.from_numpy(frames).cuda(), start_level=2, max_points=2048)
        paths = det.debug_paths(0, 2).tolist()
        for f in range(2):
            wp, wl = oracle.chain(frames[f], 2)
            n = int(npts[f])
            assert n == len(wp) and n >= 196, (f, n, len(wp))
            assert np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl), f
        hot0 = int((oracle.clamped_response(frame, 0)[0] > 15).sum())
        assert hot0 > 2048, hot0                      # otherwise this test does not exercise the bands
        assert paths == [1, 1], paths
    finally:
        det.close()
