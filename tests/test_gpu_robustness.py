"""GPU tests of the boundary's edge cases: ragged sizes, strided device frames, table
overflow through the batch API, deep pyramid levels, tiny frames, concurrent host threads
(the reference CLI calls the detector from --jobs worker threads,
mrgingham-from-image.cc:374-379)."""
import threading

import numpy as np
import pytest
import torch

import mrgingham_amd
from mrgingham_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det():
    d = mrgingham_amd.Detector(0)
    yield d
    d.close()


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("shape", [(333, 251), (1001, 999), (650, 490), (647, 483)])
def test_ragged_sizes_every_level(det, shape):
    """Sizes that are not multiples of 8/16: the generic staging, decimation and store paths."""
    w, h = shape
    frames = np.stack([synth.board_frame(w, h, 10, 3).numpy(), synth.noise_frame(w, h, 4, smooth=1).numpy()])
    d = _cuda(frames)
    for level in range(4):
        xy, counts = det.detect(d, level, capacity=8192)
        for f in range(2):
            want = oracle.find_corners(frames[f], level)
            assert int(counts[f]) == len(want), (shape, level, f)
            assert np.array_equal(xy[f, :len(want)].cpu().numpy(), want), (shape, level, f)
    pts, lv, npts = det.chain(d, 3, 4096)
    for f in range(2):
        wp, wl = oracle.chain(frames[f], 3)
        n = int(npts[f])
        assert n == len(wp) and np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)


def test_strided_device_frames_detect_and_chain(det):
    big = np.zeros((3, 500, 700), np.uint8)
    inner = np.stack([synth.board_frame(640, 480, 10, s).numpy() for s in range(3)])
    big[:, 10:490, 32:672] = inner
    view = _cuda(big)[:, 10:490, 32:672]       # row stride 700, frame pitch 350000: nothing dense
    assert not view.is_contiguous()
    for level in (0, 2):
        xy, counts = det.detect(view, level, capacity=1024)
        for f in range(3):
            want = oracle.find_corners(inner[f], level)
            assert int(counts[f]) == len(want) and np.array_equal(xy[f, :len(want)].cpu().numpy(), want)
    pts, lv, npts = det.chain(view, 3, 512)
    for f in range(3):
        wp, wl = oracle.chain(inner[f], 3)
        n = int(npts[f])
        assert n == len(wp) and np.array_equal(pts[f, :n].cpu().numpy(), wp)


def test_table_overflow_is_reported_and_recoverable(det):
    yy, xx = np.arange(240).reshape(-1, 1), np.arange(320).reshape(1, -1)
    dense = (((yy // 3 + xx // 3) & 1) * 255).astype(np.uint8)          # 42 % hot pixels
    frames = _cuda(np.stack([synth.board_frame(320, 240, 10, 0).numpy(), dense]))
    imgs = [synth.board_frame(320, 240, 10, 0).numpy(), dense]

    def check(xy, counts):
        for f, img in enumerate(imgs):
            want = oracle.find_corners(img, 0)
            assert int(counts[f]) == len(want) and np.array_equal(xy[f, :len(want)].cpu().numpy(), want)

    d2 = mrgingham_amd.Detector(0)
    try:
        # default tables (1/128 of the pixels, at least 4096 entries): the dense frame does not fit and says so ...
        with pytest.raises(RuntimeError, match="overflow") as ei:
            d2.detect(frames, 0, capacity=65536, retry=False)
        assert "make the call again" in str(ei.value)
        # ... the tables of that level have grown meanwhile: the same call, made again, succeeds (a second overflow
        # of another table -- candidates, LIFO words -- may take one more round)
        for attempt in range(3):
            try:
                xy, counts = d2.detect(frames, 0, capacity=65536, retry=False)
                break
            except RuntimeError as e:
                assert "overflow" in str(e) and attempt < 2
        check(xy, counts)
        gib = d2.scratch_bytes() / 2**30
        assert gib < 0.5                                       # grown for 320x240 frames, not for the world
        # the Python methods make that second call themselves
        d3 = mrgingham_amd.Detector(0)
        try:
            check(*d3.detect(frames, 0, capacity=65536))
            pts, lv, npts = d3.chain(frames, start_level=1, max_points=8192)
            for f, img in enumerate(imgs):
                wp, wl = oracle.chain(img, 1)
                n = int(npts[f])
                assert n == len(wp) and np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)
        finally:
            d3.close()
    finally:
        d2.close()
    # an explicit capacity still works (one entry per pixel), and resets what has grown
    det.set_option("hot_capacity_shift", 0)
    try:
        check(*det.detect(frames, 0, capacity=65536, retry=False))
    finally:
        det.set_option("hot_capacity_shift", 7)
    # and the context keeps working at the default setting afterwards
    xy, counts = det.detect(frames[:1], 0, capacity=1024)
    assert int(counts[0]) == len(oracle.find_corners(synth.board_frame(320, 240, 10, 0).numpy(), 0))


def test_deep_levels_and_tiny_frames(det):
    img = synth.board_frame(1920, 1080, 10, 2).numpy()
    d = _cuda(img[None])
    for level in (4, 5, 7, 10):                  # beyond the one-pass pyramid: single-level decimation
        assert np.array_equal(det.decimate(d, level)[0].cpu().numpy(), oracle.decimate(img, level))
        xy, counts = det.detect(d, level, capacity=256)
        want = oracle.find_corners(img, level)
        assert int(counts[0]) == len(want) and np.array_equal(xy[0, :len(want)].cpu().numpy(), want)
        got = mrgingham_amd.find_points(img, image_pyramid_level=level)
        assert len(got) == len(want)
    with pytest.raises(RuntimeError):
        det.detect(d, 11)
    for shape in [(1, 1), (1, 40), (14, 14), (15, 15), (16, 17), (22, 22), (23, 200)]:
        tiny = np.random.RandomState(sum(shape)).randint(0, 256, size=shape).astype(np.uint8)
        r = det.chess_response(_cuda(tiny[None]), 0)[0].cpu().numpy()
        assert np.array_equal(r, oracle.chess_response_5(tiny, fill=0)), shape
        for level in (0, 1):
            want = oracle.find_corners(tiny, level)
            got = mrgingham_amd.find_points(tiny, image_pyramid_level=level)
            assert len(got) == (0 if want is None else len(want)), (shape, level)
        pts, lv, npts = det.chain(_cuda(tiny[None]), 3, 16)
        assert int(npts[0]) == len(oracle.chain(tiny, 3)[0])


def test_zero_frames_and_zero_points(det):
    empty = torch.empty((0, 480, 640), dtype=torch.uint8, device="cuda")
    xy, counts = det.detect(empty, 0, capacity=8)
    assert counts.numel() == 0
    img = synth.board_frame(320, 240, 10, 0).numpy()
    p, l, n = mrgingham_amd.refine_points(np.zeros((0, 2)), np.zeros(0, np.int8), img, 0)
    assert n == 0 and p.shape == (0, 2)


def test_concurrent_host_threads_use_private_contexts():
    imgs = [synth.board_frame(640, 480, 10, s).numpy() for s in range(4)]
    want = [oracle.find_corners(im, 1) for im in imgs]
    errors = []

    def worker(k):
        try:
            for _ in range(3):
                got = mrgingham_amd.find_points(imgs[k], image_pyramid_level=1)
                if not np.array_equal(np.round(got * 1000).astype(np.int64), want[k].astype(np.int64)):
                    errors.append(k)
                r = mrgingham_amd.ChESS_response_5(imgs[k])
                if not np.array_equal(r, oracle.chess_response_5(imgs[k], fill=0)):
                    errors.append(("chess", k))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_chain_start_levels_and_pitch_truncation(det):
    frames = _cuda(np.stack([synth.board_frame(800, 600, 14, 1).numpy(), synth.noise_frame(800, 600, 2).numpy()]))
    host = frames.cpu().numpy()
    for start in (1, 4):
        pts, lv, npts = det.chain(frames, start, 4096)
        for f in range(2):
            wp, wl = oracle.chain(host[f], start)
            n = int(npts[f])
            assert n == len(wp) and np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)
    # fewer point slots than candidates: the first `pitch` candidates (reference order) are refined
    pts, lv, npts = det.chain(frames, 2, 50)
    for f in range(2):
        cand = oracle.find_corners(host[f], 2)
        m = min(50, len(cand))
        p0 = cand[:m].astype(np.float64) / 1000.0
        l0 = np.full(m, 2, np.int8)
        for L in (1, 0):
            p0, l0, _ = oracle.refine_corners(p0, l0, host[f], L)
        assert int(npts[f]) == m and np.array_equal(pts[f, :m].cpu().numpy(), p0) and np.array_equal(lv[f, :m].cpu().numpy(), l0)


def test_mixed_resolution_stream_plan_runs_per_resolution_batches(det):
    """BASELINE config 5 in miniature: an LPT plan over a mixed-resolution stream, one batch per
    resolution on this rank, results keyed back to the stream order and equal to the oracle."""
    from mrgingham_amd import parallel
    res = [(640, 400), (960, 540), (1280, 720)]
    sizes = [res[k % 3] for k in range(7)]
    frames = [synth.board_frame(w, h, 10, k).numpy() for k, (w, h) in enumerate(sizes)]
    results = {}
    for world in (1, 2):
        for rank in range(world):
            for (w, h), idx in parallel.plan_mixed_stream(sizes, world, rank).items():
                batch = _cuda(np.stack([frames[i] for i in idx]))
                pts, lv, npts = det.chain(batch, 3, 512)
                for j, i in enumerate(idx):
                    n = int(npts[j])
                    results[(world, i)] = (pts[j, :n].cpu().numpy(), lv[j, :n].cpu().numpy())
    for i, img in enumerate(frames):
        wp, wl = oracle.chain(img, 3)
        for world in (1, 2):
            gp, gl = results[(world, i)]
            assert np.array_equal(gp, wp) and np.array_equal(gl, wl), (world, i)


@pytest.mark.parametrize("shape", [(32752, 48), (48, 32752), (32767, 33), (31, 32767)])
def test_extreme_aspect_frames_near_the_int16_coordinate_limit(det, shape):
    """The reference keeps pixel coordinates in int16 (find_chessboard_corners.cc:50, :91, :332-333), so a
    side can be at most 32767: 128 strips of one segment, or one strip of 256 segments."""
    w, h = shape
    base = synth.noise_frame(w, h, 9, smooth=1).numpy()
    # paste corner-like 2x2 checker junctions so that there are candidates along the long side
    frame = base.copy()
    n = max(w, h) // 97
    for i in range(n):
        cx = (31 + 97 * i) if w > h else w // 2
        cy = h // 2 if w > h else (31 + 97 * i)
        x0, x1, y0, y1 = max(cx - 12, 0), min(cx + 12, w), max(cy - 12, 0), min(cy + 12, h)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        frame[y0:y1, x0:x1] = np.where((xx < cx) ^ (yy < cy), 30, 220).astype(np.uint8)
    frame = synth.box_blur3(torch.from_numpy(frame).to(torch.int64)).numpy().astype(np.uint8)
    d = _cuda(frame[None])
    resp = det.chess_response(d, 0, clamp=True).cpu().numpy()[0]
    want_resp, _ = oracle.clamped_response(frame, 0)
    assert np.array_equal(resp, want_resp)
    for level in (0, 1):
        xy, counts = det.detect(d, level, capacity=16384)
        want = oracle.find_corners(frame, level)
        assert int(counts[0]) == len(want) and (level > 0 or len(want) > 0), (shape, level, len(want))
        assert np.array_equal(xy[0, :len(want)].cpu().numpy(), want), (shape, level)


def test_randomised_size_sweep(det):
    """Seeded sweep over awkward frame sizes, strides and content mixes: response (raw and clamped),
    detection at levels 0..2 and the chain must equal the oracle for every case."""
    rng = np.random.RandomState(20240917)
    sizes = [(15, 15), (16, 16), (17, 31), (31, 17), (32, 47), (48, 48), (63, 65), (64, 64), (80, 33), (96, 100),
             (127, 129), (128, 128), (130, 70), (160, 120), (255, 257), (256, 256), (272, 64), (288, 31), (300, 300),
             (320, 240), (511, 40), (512, 64), (528, 48), (640, 39)]
    for (w, h) in sizes:
        kind = rng.randint(3)
        if kind == 0 or min(w, h) < 64:
            img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        elif kind == 1:
            img = synth.noise_frame(w, h, int(rng.randint(1000)), smooth=1).numpy()
        else:
            img = synth.board_frame(w, h, 6, int(rng.randint(1000))).numpy()
        pad = int(rng.choice([0, 1, 13, 16]))
        buf = np.zeros((h, w + pad), dtype=np.uint8)
        buf[:, :w] = img
        d = torch.from_numpy(buf).cuda()[None, :, :w]                     # row stride w + pad
        for clamp in (False, True):
            got = det.chess_response(d, 0, clamp=clamp).cpu().numpy()[0]
            want = oracle.chess_response_5(img, fill=0)
            if clamp:
                want = np.maximum(want, 0)
            assert np.array_equal(got, want), (w, h, pad, clamp)
        for level in (0, 1, 2):
            xy, counts = det.detect(d, level, capacity=8192)
            want = oracle.find_corners(img, level)
            n = 0 if want is None else len(want)
            assert int(counts[0]) == n, (w, h, pad, level)
            if n:
                assert np.array_equal(xy[0, :n].cpu().numpy(), want), (w, h, pad, level)
        pts, lv, npts = det.chain(d, 2, 4096)
        wp, wl = oracle.chain(img, 2)
        n = int(npts[0])
        assert n == len(wp) and np.array_equal(pts[0, :n].cpu().numpy(), wp) and np.array_equal(lv[0, :n].cpu().numpy(), wl), (w, h, pad)
